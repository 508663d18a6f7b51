"""Weight-stationary conv, fp32 MFMA against the three-piece bf16 form: error against fp64 and time.
Run twice: plainly and with EEGGAN_WS_SPLIT=1."""
import sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops
from kernel_roofline import timeit
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
for B, T in ((32, 1920), (32, 1024), (64, 1920), (512, 1920)):
    x = torch.randn(B, 64, T, generator=g); w = torch.randn(64, 64, 3, generator=g); b = torch.randn(64, generator=g)
    r = torch.rand(B, 64, generator=g) + 0.5
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), r.to(dev)
    y = ops.conv1d_raw(xd, wd, bd, rd, in_scale=0.1, act=1, act_param=0.2)
    n = 4 if B <= 64 else 1
    ref = F.leaky_relu(F.conv1d(x[:n].double() * 0.1, w.double(), b.double(), padding=1), 0.2) * r[:n].double()[:, :, None]
    err = float((y[:n].cpu().double() - ref).norm() / ref.norm())
    emax = float((y[:n].cpu().double() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: ops.conv1d_raw(xd, wd, bd, rd, in_scale=0.1, act=1, act_param=0.2), 30 if B <= 64 else 8)
    fl = 2.0 * B * 64 * 64 * 3 * T
    print(f"B={B:3d} T={T:4d}: rel err {err:.2e} (max {emax:.2e})  {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TFLOP/s", flush=True)
