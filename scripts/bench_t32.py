"""Timing of the short wide layers (T = 32 and 128) per batch; EEGGAN_KSPLIT_BELOW=0 gives the unsplit tile."""
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
    for cin, cout, T in ((256, 256, 32), (257, 256, 32), (256, 257, 32), (256, 256, 128), (128, 256, 256), (128, 128, 256)):
        x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin, 3, device=dev)
        b = torch.randn(cout, device=dev); r = torch.rand(B, cout, device=dev) + 0.5
        t = timeit(lambda: ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2))
        fl = 2.0 * B * cin * cout * 3 * T
        print(f"B={B:3d} {cin:3d}->{cout:3d} T={T:4d}  {t:6.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)
