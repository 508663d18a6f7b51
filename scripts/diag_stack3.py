"""Log every conv kernel call of the backward in the two flows (debugging aid)."""
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
d = d.to(dev); d.depth = depth
for m in d.modules():
    if isinstance(m, GDropLayer): m.strength = 0
log = []
orig_c, orig_w = ops.conv1d_raw, ops.conv1d_wgrad_raw
def cs(t):
    return None if t is None else t.detach().double()
def conv_logged(x, w, bias=None, drop=None, **kw):
    y = orig_c(x, w, bias, drop, **kw)
    log.append(("conv", tuple(x.shape), tuple(w.shape), kw.get("transposed", False), kw.get("act", 0), kw.get("mask_y") is not None,
                kw.get("mask_drop") is not None, kw.get("mask_slope", 0.0), kw.get("in_scale"), cs(x), cs(kw.get("mask_y")), cs(y)))
    return y
def wgrad_logged(x, gy, K, **kw):
    gw, gb = orig_w(x, gy, K, **kw)
    log.append(("wgrad", tuple(x.shape), tuple(gy.shape), False, 0, kw.get("mask_y") is not None, kw.get("mask_drop") is not None,
                kw.get("mask_slope", 0.0), kw.get("in_scale"), cs(x), cs(gy), cs(gw)))
    return gw, gb
ops.conv1d_raw, ops.conv1d_wgrad_raw = conv_logged, wgrad_logged
sr = d({"x": real})[0]; sf = d({"x": fake})[0]
log.clear()
(ops.mean_scale(sf, 1.0) + ops.mean_scale(sr, -1.0)).backward()
seq = list(log)
d.zero_grad()
with layers.batch_segments(2):
    sc = d({"x": torch.cat([real, fake])})[0]
log.clear()
(ops.mean_scale(sc[B:], 1.0) + ops.mean_scale(sc[:B], -1.0)).backward()
st = list(log)
print(len(seq), len(st))
for e in st:
    kind, xs, ws = e[0], e[1], e[2]
    half = (xs[0] // 2,) + xs[1:]
    cands = [s for s in seq if s[0] == kind and s[1] == half and s[2][1:] == ws[1:] and s[3] == e[3] and s[4] == e[4]] if kind == "conv" else \
            [s for s in seq if s[0] == kind and s[1] == half and s[2][1:] == ws[1:]]
    def best(field, combine):
        # match the stacked tensor against (cat or sum) of two sequential candidates
        b = 1e9
        for i in range(len(cands)):
            for j in range(len(cands)):
                if i == j: continue
                ref = combine(cands[i][field], cands[j][field])
                if ref is None or e[field] is None: continue
                b = min(b, float((e[field] - ref).norm() / e[field].norm()))
        return b
    cat = lambda a, b: None if a is None or b is None else torch.cat([a, b])
    add = lambda a, b: a + b
    print(kind, xs, ws, "T" if e[3] else "", "act", e[4], "mask", e[5], e[6], e[7], "scale %.4f" % e[8], "ncand", len(cands),
          "in %.1e" % best(9, cat), "aux %.1e" % best(10, cat), "out %.1e" % best(11, cat if kind == "conv" else add))
