import sys, os
import torch, torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eeggan_amd import ops
dev = torch.device("cuda", 0)
def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())
torch.manual_seed(1)
for (B, cin, cout, T, K) in [(4, 256, 256, 128, 3), (4, 128, 256, 256, 3), (4, 256, 128, 256, 3), (4, 128, 128, 256, 3), (4, 64, 128, 512, 3), (4, 64, 64, 512, 3)]:
    x = torch.randn(B, cin, T); w = torch.randn(cout, cin, K); gy = torch.randn(B, cout, T)
    ymask = torch.randn(B, cout, T); md = torch.rand(B, cout) + 0.5
    mask = torch.where(ymask > 0, 1.0, 0.2).double() * md.double()[:, :, None]
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    lin = F.conv1d(x64 * 0.37, w64, None, padding=1)
    gx_ref, gw_ref = torch.autograd.grad(lin, [x64, w64], gy.double() * mask)
    xd, wd, gyd, ymd, mdd = (t.to(dev) for t in (x, w, gy, ymask, md))
    cache = ops.FlipCache(wd)
    gx = ops.conv1d_raw(gyd, wd, None, None, in_scale=0.37, transposed=True, mask_y=ymd, mask_drop=mdd, mask_slope=0.2, flip=cache)
    gx2 = ops.conv1d_raw(gyd, wd, None, None, in_scale=0.37, transposed=True, mask_y=ymd, mask_drop=mdd, mask_slope=0.2, flip=cache)
    gw, _ = ops.conv1d_wgrad_raw(xd, gyd, K, in_scale=0.37, need_bias=True, mask_y=ymd, mask_drop=mdd, mask_slope=0.2)
    print((B, cin, cout, T), "dgrad", rel(gx, gx_ref), "cached", rel(gx2, gx_ref), "wgrad", rel(gw, gw_ref))
