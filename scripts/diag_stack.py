"""Stacked [real; fake] critic pass vs two sequential passes, first-order terms only (debugging aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from eeggan_amd import config, layers, ops
from eeggan_amd.layers import GDropLayer
dev = torch.device("cuda", 0)
depth, B = 3, 8
T = config.seq_len(depth)
gen = torch.Generator().manual_seed(11)
real = torch.randn(B, 5, T, generator=gen).to(dev)
fake = torch.randn(B, 5, T, generator=gen).to(dev)
g, d = config.build_models(5)
d = d.to(dev)
d.depth = depth
for m in d.modules():
    if isinstance(m, GDropLayer):
        m.strength = 0
def grads():
    out = {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None}
    d.zero_grad()
    return out
sr = d({"x": real})[0]; sf = d({"x": fake})[0]
(ops.mean_scale(sf, 1.0) + ops.mean_scale(sr, -1.0)).backward()
seq = grads()
ops.mean_scale(d({"x": real})[0], -1.0).backward()
rterm = grads()
with layers.batch_segments(2):
    sc = d({"x": torch.cat([real, fake])})[0]
print("scores real", float((sc[:B] - sr).abs().max()), "fake", float((sc[B:] - sf).abs().max()))
(ops.mean_scale(sc[B:], 1.0) + ops.mean_scale(sc[:B], -1.0)).backward()
st = grads()
for k in seq:
    e = float((seq[k] - st[k]).norm()); n = float(seq[k].norm()); r = float(rterm[k].norm())
    print(f"{k:40s} diff {e:.3e} total {n:.3e} real-term {r:.3e} rel-to-term {e / r:.2e}")

print("---- gradients at module outputs")
import collections
store = collections.defaultdict(list)
hooks = []
for name, m in d.named_modules():
    if type(m).__name__ in ("GeneralConv", "MinibatchStddev", "DBlock"):
        def fh(mod, inp, out, name=name):
            if torch.is_tensor(out) and out.requires_grad:
                out.register_hook(lambda gr, name=name: store[name].append(gr.detach().clone()))
        hooks.append(m.register_forward_hook(fh))
sr = d({"x": real})[0]; sf = d({"x": fake})[0]
(ops.mean_scale(sf, 1.0) + ops.mean_scale(sr, -1.0)).backward()
seqg = {k: v for k, v in store.items()}
store = collections.defaultdict(list)
with layers.batch_segments(2):
    sc = d({"x": torch.cat([real, fake])})[0]
(ops.mean_scale(sc[B:], 1.0) + ops.mean_scale(sc[:B], -1.0)).backward()
for k, v in store.items():
    a = v[0]
    cands = seqg[k]
    # sequential hooks fire once per pass (order unknown): match either order
    e = min(float((a - torch.cat([cands[0], cands[1]])).norm()), float((a - torch.cat([cands[1], cands[0]])).norm())) / float(a.norm())
    print(f"{k:30s} {tuple(a.shape)} rel diff {e:.2e}")
