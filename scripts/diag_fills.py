"""Which Python frames issue the fills / ATen elementwise kernels that end up inside the captured graphs of a
training step: the first train() of a graph trainer (warm-up runs and captures) under the CPU-side profiler."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eeggan_amd import layers  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
layers.set_drop_mode("device")
tr = bench.make_trainer(dev, 7, 1.0, 32, 0, 1, 0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.train()
torch.cuda.synchronize()
want = ("aten::zeros", "aten::zeros_like", "aten::fill_", "aten::zero_", "aten::ones_like", "aten::new_zeros", "aten::add",
        "aten::add_", "aten::cat", "aten::mul", "aten::sum", "aten::uniform_", "aten::ones", "aten::clone")
seen = {}
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if "eeg" in f or "bench" in f][:5]
        key = (ev.name, tuple(frames))
        seen[key] = seen.get(key, 0) + 1
for (name, frames), n in sorted(seen.items(), key=lambda kv: -kv[1])[:40]:
    print(n, name)
    for f in frames:
        print("      ", f)
