"""Timing of the k = 32 critic tail kernels (forward, input gradient, weight gradient)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, n=30):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B in (32, 64):
    x = torch.randn(B, 256, 32, device=dev); w = torch.randn(256, 256, 32, device=dev)
    b = torch.randn(256, device=dev); r = torch.rand(B, 256, device=dev) + 0.5
    gc = torch.randn(B, 256, 1, device=dev)
    print(f"B={B:3d} tail fwd {timeit(lambda: ops._tail_fwd_raw(x, w, b, r, 0.1, 1, 0.2)):6.1f} us  "
          f"dgrad {timeit(lambda: ops._tail_dgrad_raw(gc, w, 0.1)):6.1f} us  "
          f"wgrad {timeit(lambda: ops._tail_wgrad_raw(x, gc, 0.1, True)):6.1f} us", flush=True)
