"""Fused conv+pixelnorm vs conv then pixelnorm (timing aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for T in (512, 768, 1024, 1280, 1920):
    B, C = 32, 64
    x = torch.randn(B, C, T, device=dev); w = torch.randn(C, C, 3, device=dev)
    b = torch.randn(C, device=dev); r = torch.rand(B, C, device=dev) + 0.5
    def sep():
        h = ops.conv1d_raw(x, w, b, None, in_scale=0.1, act=1, act_param=0.2)
        return ops.pixelnorm_drop_fwd_raw(h, r)
    fused = lambda: ops.conv1d_pixelnorm_raw(x, w, b, r, in_scale=0.1, slope=0.2)
    conv = lambda: ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2)
    print(f"T={T:5d} conv {timeit(conv):6.1f}us  conv+pn separate {timeit(sep):6.1f}us  fused {timeit(fused):6.1f}us")
