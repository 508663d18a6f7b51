"""Per-shape timing of the conv kernels (HIP events on the launch stream)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(64, 64, 1920, 3), (64, 64, 1280, 3), (64, 64, 768, 3), (64, 64, 512, 3), (128, 64, 512, 3), (64, 128, 512, 3),
          (128, 128, 256, 3), (256, 128, 256, 3), (128, 256, 256, 3), (256, 256, 128, 3), (256, 256, 32, 3), (257, 256, 32, 3),
          (64, 5, 1920, 1), (5, 64, 1920, 1)]

def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n

print(f"{'shape':>22s} {'fwd us':>8s} {'TF':>6s} {'dgrad us':>9s} {'TF':>6s} {'wgrad us':>9s} {'TF':>6s}")
marker = torch.zeros(8, device=dev)
for cin, cout, T, K in shapes:
    ops.pow_base_raw(marker, 2.0)  # run separator in kernel traces
    x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin, K, device=dev)
    b = torch.randn(cout, device=dev); r = torch.rand(B, cout, device=dev) + 0.5
    gy = torch.randn(B, cout, T, device=dev)
    fl = 2.0 * B * cin * cout * K * T
    tf = timeit(lambda: ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2))
    td = timeit(lambda: ops.conv1d_raw(gy, w, None, None, in_scale=0.1, transposed=True))
    tw = timeit(lambda: ops.conv1d_wgrad_raw(x, gy, K, in_scale=0.1))
    print(f"{cin:4d}->{cout:3d} T={T:4d} K={K} {tf*1e6:8.1f} {fl/tf/1e12:6.1f} {td*1e6:9.1f} {fl/td/1e12:6.1f} {tw*1e6:9.1f} {fl/tw/1e12:6.1f}")
