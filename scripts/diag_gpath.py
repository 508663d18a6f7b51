import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import load_fixture, rel, oracle_params
from eeggan_amd import config, layers
from oracle import ref_cpu as O

fx = load_fixture("e2e_d7_a1")
dev = torch.device("cuda", 0)
depth, alpha, seed = 7, 1.0, 104
layers.set_drop_mode("numpy")
g, d = config.build_models(int(fx["model_seed"]))
cfg = O.NetConfig()
real = torch.from_numpy(fx["real"]); z = torch.from_numpy(fx["z"])
res = {}
for dt in (torch.float64, torch.float32):
    dp, ds = oracle_params(d, dt); gp, gs = oracle_params(g, dt)
    x = real.to(dt).clone().requires_grad_(True)
    np.random.seed(1); src = O.DropSource("numpy")
    s, _ = O.discriminator_forward(dp, ds, x, depth, alpha, cfg, src)
    (gx,) = torch.autograd.grad(s.sum(), x)
    # G backward for a fixed upstream gradient
    for p in gp.values(): p.requires_grad_(True)
    np.random.seed(2); src = O.DropSource("numpy")
    fake = O.generator_forward(gp, gs, z.to(dt), depth, alpha, cfg, src)
    up = torch.from_numpy(np.random.default_rng(0).standard_normal(fake.shape)).to(dt)
    gg = torch.autograd.grad(fake, list(gp.values()), up, allow_unused=True)
    res[dt] = (s.detach(), gx, fake.detach(), dict(zip(gp.keys(), gg)))
gd, dd = g.to(dev), d.to(dev)
gd.depth = dd.depth = depth
x = real.to(dev).requires_grad_(True)
np.random.seed(1)
s = dd(x)[0]
(gx,) = torch.autograd.grad(s.sum(), x)
print("D score  e64=%.2e  e32=%.2e  o32-vs-o64=%.2e" % (rel(s, res[torch.float64][0]), rel(s, res[torch.float32][0]), rel(res[torch.float32][0], res[torch.float64][0])))
print("D dx     e64=%.2e  e32=%.2e  o32-vs-o64=%.2e" % (rel(gx, res[torch.float64][1]), rel(gx, res[torch.float32][1]), rel(res[torch.float32][1], res[torch.float64][1])))
np.random.seed(2)
fake = gd(z.to(dev))[0]["x"]
up = torch.from_numpy(np.random.default_rng(0).standard_normal(fake.shape)).float().to(dev)
gg = torch.autograd.grad(fake, [p for p in gd.parameters()], up, allow_unused=True)
print("G fake   e64=%.2e  e32=%.2e  o32-vs-o64=%.2e" % (rel(fake, res[torch.float64][2]), rel(fake, res[torch.float32][2]), rel(res[torch.float32][2], res[torch.float64][2])))
for (k, p), gr in zip(gd.named_parameters(), gg):
    if gr is None: continue
    a, b = res[torch.float64][3][k], res[torch.float32][3][k]
    print(f"G {k:42s} e64={rel(gr, a):.2e} e32={rel(gr, b):.2e} o32-vs-o64={rel(b, a):.2e}")
