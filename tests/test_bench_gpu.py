"""bench.py end to end on the GPU box: the one-GPU line, and the multi-rank path (RCCL process group, gradient
buckets, collectives between graph replays) rehearsed with a single rank under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd, env=None):
    out = subprocess.run(cmd, cwd=ROOT, env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line(dev):
    r = _line([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    assert r["unit"] == "steps/s" and r["n_gpus"] == 1 and r["steps"] == 5 and r["value"] > 0
    assert r["dtype"] == "f32" and r["scaling"] == "weak" and r["config"]["launch"] == "GraphedTrainer"
    roof = r["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(r["ms_per_step"] * r["value"] - 1000.0) < 1.0  # value = steps / time at one GPU


def test_bench_multi_rank_path_with_one_rank(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2",
               "--no-cpu-baseline"], env={"EEGGAN_FORCE_DIST": "1"})
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["config"]["launch"] == "GraphedTrainer"
