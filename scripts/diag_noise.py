"""Print, per parameter tensor, the HIP path's gradient error against the reference's fp64 and
fp32 runs next to the reference's own fp32-vs-fp64 error (from a golden fixture)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import load_fixture, rel, subsample
from eeggan_amd import config, layers, losses

name = sys.argv[1] if len(sys.argv) > 1 else "e2e_d7_a1"
fx = load_fixture(name)
dev = torch.device("cuda", 0)
depth, alpha, seed = int(fx["depth"]), float(fx["alpha"]), int(fx["seed"])
layers.set_drop_mode("numpy")
g, d = config.build_models(int(fx["model_seed"]))
g, d = g.to(dev), d.to(dev)
g.depth = d.depth = depth; g.alpha = d.alpha = alpha
real, z, mix = (torch.from_numpy(fx[k]).to(dev) for k in ("real", "z", "mix"))
np.random.seed(seed + 1)
dl = losses.discriminator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", 1e-3, 10.0, 1.0, mixing=mix); dl.backward()
np.random.seed(seed + 2)
gl = losses.generator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", False, 0.0); gl.backward()
print("d_loss", float(dl), float(fx["ref64_d_loss"]), "g_loss", float(gl), float(fx["ref64_g_loss"]))
for key, m in (("d_grads", d), ("g_grads", g)):
    for k, p in m.named_parameters():
        if p.grad is None: continue
        s = subsample(p.grad)
        print(f"{key}.{k:45s} e64={rel(s, fx[f'{key}_sub.{k}']):.2e} e32={rel(s, fx[f'{key}_sub32.{k}']):.2e} refnoise={float(fx[f'{key}_noise.{k}']):.2e}")
np.random.seed(seed + 2)
fake = g(z)[0]["x"]
(up,) = torch.autograd.grad(-d(fake)[0].mean(), fake)
s = subsample(up, 2048)
print("fake e64=%.2e" % rel(fake, fx["ref64_fake"]) if False else "", end="")
print(f"g_upstream e64={rel(s, fx['g_upstream_sub']):.2e} e32={rel(s, fx['g_upstream_sub32']):.2e} refnoise={float(fx['g_upstream_noise']):.2e}")
# same, but feeding the reference's own fp64 fake (isolates the discriminator's first-order path)
