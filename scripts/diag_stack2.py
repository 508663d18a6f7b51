"""One critic block, batch 16 vs two batches of 8 (debugging aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from eeggan_amd import config, layers, ops
from eeggan_amd.layers import GDropLayer
dev = torch.device("cuda", 0)
g, d = config.build_models(5)
d = d.to(dev)
for m in d.modules():
    if isinstance(m, GDropLayer):
        m.strength = 0
blk = d.blocks[6]
print(blk)
gen = torch.Generator().manual_seed(3)
h = torch.randn(16, 256, 128, generator=gen).to(dev)
gy = torch.randn(16, 256, 128, generator=gen).to(dev)
def run(hh, gg):
    d.zero_grad()
    hh = hh.clone().requires_grad_(True)
    out = blk(hh)
    out.backward(gg)
    return out.detach(), hh.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}
o, gx, gw = run(h, gy)
o1, gx1, gw1 = run(h[:8], gy[:8]); o2, gx2, gw2 = run(h[8:], gy[8:])
print("out", float((o - torch.cat([o1, o2])).abs().max()), "gx", float((gx - torch.cat([gx1, gx2])).abs().max()), float(gx.abs().max()))
for k in gw:
    e = float((gw[k] - gw1[k] - gw2[k]).norm()) / float(gw[k].norm())
    print(k, f"{e:.2e}")
