"""Do two independent kernel chains, captured as parallel branches of one HIP graph, overlap their launch gaps?
Times 2n launches in one chain against n + n in two forked chains, for a launch-bound kernel, the
chip-filling 64-channel conv and a mix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeggan_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def make(kind):
    if kind == "small":
        a = torch.randn(32, 5, 1920, device=dev)
        return lambda: ops.axpby_raw(a, 0.5)
    if kind == "conv":
        x = torch.randn(32, 64, 1920, device=dev)
        w = torch.randn(64, 64, 3, device=dev)
        b = torch.randn(64, device=dev)
        return lambda: ops.conv1d_raw(x, w, b, None, in_scale=0.1, act=1, act_param=0.2)
    if kind == "conv512":
        x = torch.randn(64, 64, 512, device=dev)
        w = torch.randn(64, 64, 3, device=dev)
        return lambda: ops.conv1d_raw(x, w, None, None, in_scale=0.1)
    if kind == "resample":
        from eeggan_amd.utils import resample_signal
        x = torch.randn(32, 64, 1920, device=dev)
        return lambda: resample_signal(x, 3, 2)
    raise ValueError(kind)


def timed(graph, reps=20):
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(kind_a, kind_b, n=40):
    fa, fb = make(kind_a), make(kind_b)
    for f in (fa, fb):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for _ in range(n):
            fa()
        for _ in range(n):
            fb()
    g2 = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(n):
                fb()
        for _ in range(n):
            fa()
        cur.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        for _ in range(n):
            fa()
    t1, t2, t3 = timed(g1), timed(g2), timed(g3)
    print(f"{kind_a:9s}+{kind_b:9s}: one chain of {2*n}: {t1:8.1f} us ({t1/(2*n):5.2f}/launch)   two chains of {n}: {t2:8.1f} us   "
          f"chain a alone: {t3:8.1f} us ({t3/n:5.2f}/launch)", flush=True)


for a, b in (("small", "small"), ("conv", "conv"), ("conv512", "conv512"), ("resample", "resample"), ("conv", "small"),
             ("conv", "resample"), ("conv512", "small")):
    run(a, b)
