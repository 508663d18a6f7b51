"""Does the first evaluation in a fresh process differ from later ones? (debugging aid)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from eeggan_amd import config, layers
from eeggan_amd.layers import GDropLayer
dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
depth, B = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 8
T = config.seq_len(depth)
gen = torch.Generator().manual_seed(4)
real = torch.randn(B, 5, T, generator=gen).to(dev)
z = torch.randn(B, 256, generator=gen).to(dev)
mix = torch.rand(B, 1, 1, generator=gen).to(dev)
g, d = config.build_models(9)
g, d = g.to(dev), d.to(dev)
g.depth = d.depth = depth
for m in list(g.modules()) + list(d.modules()):
    if isinstance(m, GDropLayer):
        m.strength = 0
d_loss_fn, g_loss_fn = config.loss_closures()
outs = []
for it in range(3):
    loss = d_loss_fn(d, g, {"x": real}, {"z": z}, mixing=mix)
    loss.backward()
    torch.cuda.synchronize()
    outs.append((float(loss), {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None}))
print("losses", [o[0] for o in outs])
for k in outs[0][1]:
    e01 = float((outs[0][1][k] - outs[1][1][k]).abs().max()); e12 = float((outs[1][1][k] - outs[2][1][k]).abs().max())
    if e01 or e12:
        print(f"{k:40s} run0-run1 {e01:.3e} run1-run2 {e12:.3e} scale {float(outs[1][1][k].abs().max()):.3e}")
outs = []
for it in range(3):
    loss = g_loss_fn(d, g, {"x": real}, {"z": z})
    loss.backward()
    torch.cuda.synchronize()
    outs.append((float(loss), {k: p.grad.clone() for k, p in g.named_parameters() if p.grad is not None}))
print("g losses", [o[0] for o in outs])
for k in outs[0][1]:
    e01 = float((outs[0][1][k] - outs[1][1][k]).abs().max()); e12 = float((outs[1][1][k] - outs[2][1][k]).abs().max())
    if e01 or e12:
        print(f"{k:40s} run0-run1 {e01:.3e} run1-run2 {e12:.3e} scale {float(outs[1][1][k].abs().max()):.3e}")
