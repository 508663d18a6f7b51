#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference; nothing of the reference is copied -- only numeric inputs/outputs are stored.
Two modules the reference imports but the hot path never touches are absent from the image
(`pytorch_pretrained_bert`, `imageio`); they are replaced by empty in-process stubs.

For every fixture the script also
  * builds this repository's modules from the same seed and checks that their initial weights
    and equalized-lr scales are bit-identical to the reference's, and
  * runs the CPU restatement (oracle/ref_cpu.py) on the same inputs and checks it against the
    reference (fp64: round-off), so the restatement is pinned before the fixtures are written.

Usage: python tools/make_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def import_reference():
    bert = types.ModuleType("pytorch_pretrained_bert")
    bert.modeling = types.ModuleType("pytorch_pretrained_bert.modeling")
    bert.modeling.BertLayer = object
    bert.BertAdam = object
    sys.modules["pytorch_pretrained_bert"] = bert
    sys.modules["pytorch_pretrained_bert.modeling"] = bert.modeling
    imageio = types.ModuleType("imageio")
    imageio.imwrite = lambda *a, **k: None
    sys.modules["imageio"] = imageio
    sys.path.insert(0, REF)
    import layers as ref_layers  # noqa
    import losses as ref_losses  # noqa
    import network as ref_network  # noqa
    import utils as ref_utils  # noqa
    sys.path.remove(REF)
    return ref_layers, ref_losses, ref_network, ref_utils


ref_layers, ref_losses, ref_network, ref_utils = import_reference()
# the reference's top-level module names shadow nothing of ours: ours live in a package
sys.path.insert(0, REPO)
import eeggan_amd.config as cfgmod  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402


class DAdapter(torch.nn.Module):
    """The reference's losses pass dicts while its networks take tensors (losses.py:76 vs
    network.py:528); this is the minimal adapter under which the losses run unmodified."""

    def __init__(self, d):
        super().__init__()
        self.d = d

    def forward(self, x):
        return self.d(x["x"])


class GAdapter(torch.nn.Module):
    def __init__(self, g):
        super().__init__()
        self.g = g

    def forward(self, z):
        return self.g(z["z"])


def scales_of(module):
    # (key: the layer's path, i.e. the module's name without the ".conv" a plain layer ends in; the two halves of a
    # separable layer, "... .conv.net.0" / "... .conv.net.1", keep their full names)
    return {(name[:-len(".conv")] if name.endswith(".conv") else name): m.scale for name, m in module.named_modules()
            if type(m).__name__ == "EqualizedConv1d"}


def to_oracle(module, dtype):
    params = {k: v.detach().clone().to(dtype) for k, v in module.state_dict().items()}
    return params, scales_of(module)


def checksums(module):
    out = {}
    for k, v in module.state_dict().items():
        v = v.double()
        out[k] = np.array([float(v.sum()), float((v * v).sum())])
    return out


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def subsample(t, n=512):
    f = t.detach().flatten()
    stride = max(1, f.numel() // n)
    return f[::stride][:n].double().numpy()


def build_pair(seed):
    g_ref, d_ref = cfgmod.build_models(seed, ref_network.Generator, ref_network.Discriminator)
    g_my, d_my = cfgmod.build_models(seed)
    for a, b, nm in ((g_ref, g_my, "G"), (d_ref, d_my, "D")):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()), f"{nm}: state-dict keys differ"
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"{nm}.{k}: initial weights differ"
        ra, rb = scales_of(a), scales_of(b)
        assert ra == rb, f"{nm}: equalized scales differ"
    return g_ref, d_ref


# ------------------------------------------------------------------------------------------ per-op
def per_op_fixture(out_dir):
    torch.manual_seed(7)
    np.random.seed(7)
    fx = {}
    # equalized conv, the four shapes of the path (small channel counts)
    for tag, (cin, cout, k, pad, T) in {
        "k3": (6, 10, 3, 1, 48), "k1": (5, 7, 1, 0, 40), "head": (12, 6, 32, 31, 1), "tail": (6, 9, 32, 0, 32),
    }.items():
        conv = ref_layers.EqualizedConv1d(cin, cout, k, pad, False, True, "kaiming_normal", 0.2, True, 1, 1).double()
        conv.conv.bias.data.normal_()
        x = torch.randn(3, cin, T, dtype=torch.float64, requires_grad=True)
        y = conv(x)
        gy = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad(y, [x, conv.conv.weight, conv.conv.bias], gy)
        yo = O.equalized_conv1d(x, conv.conv.weight, conv.conv.bias, conv.scale, pad)
        assert rel(yo, y) < 1e-14
        for n, v in dict(x=x, w=conv.conv.weight, b=conv.conv.bias, y=y, gy=gy, gx=gx, gw=gw, gb=gb).items():
            fx[f"conv_{tag}_{n}"] = v.detach().numpy()
        fx[f"conv_{tag}_scale"] = np.array(conv.scale)
        fx[f"conv_{tag}_pad"] = np.array(pad)
    # pixel norm
    h = torch.randn(3, 9, 20, dtype=torch.float64, requires_grad=True)
    y = ref_utils.pixel_norm(h)
    gy = torch.randn_like(y)
    (gh,) = torch.autograd.grad(y, h, gy)
    assert rel(O.pixel_norm(h), y) < 1e-14
    fx.update(pn_h=h.detach().numpy(), pn_y=y.detach().numpy(), pn_gy=gy.numpy(), pn_gh=gh.numpy())
    # resampling: every ratio of the schedule in both directions at its real length, + repeat
    up, dn = cfgmod.PROGRESSION_SCALE_UP, cfgmod.PROGRESSION_SCALE_DOWN
    T = 32
    for i in range(len(up)):
        x = torch.randn(1, 2, T, dtype=torch.float64)
        y = ref_utils.resample_signal(x, dn[i], up[i])
        assert rel(O.resample(x, dn[i], up[i]), y) < 1e-6, (i, rel(O.resample(x, dn[i], up[i]), y))
        xb = torch.randn(1, 2, y.shape[2], dtype=torch.float64)
        yb = ref_utils.resample_signal(xb, up[i], dn[i])
        assert rel(O.resample(xb, up[i], dn[i]), yb) < 1e-6
        fx[f"rs_up{i}_x"], fx[f"rs_up{i}_y"] = x.numpy(), y.numpy()
        fx[f"rs_dn{i}_x"], fx[f"rs_dn{i}_y"] = xb.numpy(), yb.numpy()
        T = y.shape[2]
    x = torch.randn(2, 3, 1, dtype=torch.float64)
    fx["rs_rep_x"], fx["rs_rep_y"] = x.numpy(), ref_utils.resample_signal(x, 1, 4).numpy()
    # minibatch stddev: forward, gradient, gradient of (gradient . v)
    mb = ref_layers.MinibatchStddev(4, 1, 32)
    x = torch.randn(8, 6, 32, dtype=torch.float64, requires_grad=True)
    y = mb(x)
    assert rel(O.minibatch_stddev(x, 4, 32), y) < 1e-14
    gy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    v = torch.randn_like(gx)
    gy_leaf = gy.clone().requires_grad_(True)
    (gx2,) = torch.autograd.grad(mb(x), x, gy_leaf, create_graph=True)
    ggy, gxx = torch.autograd.grad(gx2, [gy_leaf, x], v)
    fx.update(mb_x=x.detach().numpy(), mb_y=y.detach().numpy(), mb_gy=gy.numpy(), mb_gx=gx.detach().numpy(),
              mb_v=v.numpy(), mb_ggy=ggy.numpy(), mb_gxx=gxx.numpy())
    # scaled tanh
    c = torch.randn(2, 5, 16, dtype=torch.float64)
    fx.update(tanh_c=c.numpy(), tanh_y=ref_layers.ScaledTanh()(c).numpy())
    np.savez_compressed(os.path.join(out_dir, "per_op.npz"), **fx)
    print("per_op.npz:", len(fx), "arrays")


# ------------------------------------------------------------------------------------------ end to end
def run_reference(g, d, depth, alpha, real, z, mix, seed, dtype):
    g, d = g.to(dtype), d.to(dtype)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    real, z, mix = real.to(dtype), z.to(dtype), mix.to(dtype)
    ga, da = GAdapter(g), DAdapter(d)
    out = {}
    np.random.seed(seed)
    with torch.no_grad():
        out["fake"] = g(z)[0]["x"]
        out["score_real"] = d(real)[0]
    # discriminator loss: force the recorded mixing factors
    np.random.seed(seed + 1)
    ref_losses.mixing_factors = None
    ref_losses.one = None
    orig = ref_losses.get_mixing_factor
    ref_losses.get_mixing_factor = lambda b: mix
    try:
        d_loss = ref_losses.discriminator_loss(da, ga, {"x": real}, {"z": z}, "wgan_gp", 1e-3, 10.0, 1.0)
    finally:
        ref_losses.get_mixing_factor = orig
    d_loss.backward()
    out["d_loss"] = d_loss.detach()
    out["d_grads"] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in d.named_parameters()}
    np.random.seed(seed + 2)
    g_loss = ref_losses.generator_loss(da, ga, {"x": real}, {"z": z}, "wgan_gp", False, 0.0)
    g_loss.backward()
    out["g_loss"] = g_loss.detach()
    out["g_grads"] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in g.named_parameters()}
    # the three terms of the discriminator loss differentiated separately (same call order, hence
    # same noise draws, as losses.discriminator_loss): their sum cancels heavily at initialisation
    # (d mean(D(fake)) and d mean(D(real)) nearly coincide), which makes the TOTAL gradient
    # ill-conditioned while each term is not.
    np.random.seed(seed + 1)
    d.zero_grad()
    d_real = d(real)[0]
    with torch.no_grad():
        fake_ng = g(z)[0]["x"]
    d_fake = d(fake_ng)[0]
    x_hat = (mix * real + (1.0 - mix) * fake_ng).requires_grad_(True)
    pred = d(x_hat)[0]
    gg = ref_losses.calc_grad(x_hat, pred).view(real.size(0), -1)
    terms = {"real": -d_real.mean() + (d_real ** 2).mean() * 1e-3, "fake": d_fake.mean(),
             "gp": ((gg.norm(p=2, dim=1) - 1.0) ** 2).mean() * 10.0}
    params = dict(d.named_parameters())
    out["d_terms"] = {}
    total = {k: None for k in params}
    for name, val in terms.items():
        grads = torch.autograd.grad(val, list(params.values()), retain_graph=True, allow_unused=True)
        out["d_terms"][name] = dict(zip(params.keys(), grads))
        for k, gr in zip(params.keys(), grads):
            if gr is not None:
                total[k] = gr if total[k] is None else total[k] + gr
    for k, v in out["d_grads"].items():
        if v is not None:
            assert rel(total[k], v) < (1e-9 if dtype == torch.float64 else 1e-3), (k, rel(total[k], v))
    # the gradient the generator receives from the discriminator in that same iteration
    np.random.seed(seed + 2)
    fake = g(z)[0]["x"]
    (out["g_upstream"],) = torch.autograd.grad(-d(fake)[0].mean(), fake)
    return out


def run_oracle(g, d, depth, alpha, real, z, mix, seed, dtype):
    cfg = O.NetConfig()
    gp, gs = to_oracle(g, dtype)
    dp, ds = to_oracle(d, dtype)
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(True)
    real, z, mix = real.to(dtype), z.to(dtype), mix.to(dtype)
    out = {}
    np.random.seed(seed)
    src = O.DropSource("numpy")
    with torch.no_grad():
        out["fake"] = O.generator_forward(gp, gs, z, depth, alpha, cfg, src)
        out["score_real"] = O.discriminator_forward(dp, ds, real, depth, alpha, cfg, src)[0]
    np.random.seed(seed + 1)
    src = O.DropSource("numpy")
    loss, _ = O.d_loss_wgan_gp(dp, ds, gp, gs, real, z, depth, alpha, cfg, mix, src, src)
    grads = torch.autograd.grad(loss, list(dp.values()), allow_unused=True)
    out["d_loss"] = loss.detach()
    out["d_grads"] = dict(zip(dp.keys(), grads))
    np.random.seed(seed + 2)
    src = O.DropSource("numpy")
    loss, _ = O.g_loss_wgan(dp, ds, gp, gs, z, depth, alpha, cfg, src, src)
    grads = torch.autograd.grad(loss, list(gp.values()), allow_unused=True)
    out["g_loss"] = loss.detach()
    out["g_grads"] = dict(zip(gp.keys(), grads))
    return out


def compare(a, b, tol, what):
    worst = 0.0
    for k in ("fake", "score_real", "d_loss", "g_loss"):
        e = rel(a[k], b[k])
        worst = max(worst, e)
        assert e < tol, f"{what}: {k} differs by {e:.3e}"
    for gk in ("d_grads", "g_grads"):
        for k, v in b[gk].items():
            w = a[gk][k]
            assert (v is None) == (w is None), f"{what}: {gk}.{k} None-ness differs"
            if v is not None:
                e = rel(w, v)
                worst = max(worst, e)
                assert e < tol, f"{what}: {gk}.{k} differs by {e:.3e}"
    return worst


def e2e_fixture(out_dir, name, depth, alpha, batch, seed):
    g, d = build_pair(cfgmod.RANDOM_SEED)
    weights = {**{f"wsum_G.{k}": v for k, v in checksums(g).items()},
               **{f"wsum_D.{k}": v for k, v in checksums(d).items()}}
    T = cfgmod.seq_len(depth)
    rng = np.random.default_rng(seed)
    real = torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng))
    torch.manual_seed(seed)
    z = torch.randn(batch, cfgmod.LATENT_SIZE)
    mix = torch.rand(batch, 1, 1)
    r64 = run_reference(g, d, depth, alpha, real, z, mix, seed, torch.float64)
    o64 = run_oracle(g, d, depth, alpha, real, z, mix, seed, torch.float64)
    w = compare(o64, r64, 1e-9, f"{name} oracle fp64 vs reference fp64")
    g, d = build_pair(cfgmod.RANDOM_SEED)  # fresh fp32 copies
    r32 = run_reference(g, d, depth, alpha, real, z, mix, seed, torch.float32)
    o32 = run_oracle(g, d, depth, alpha, real, z, mix, seed, torch.float32)
    fx = dict(weights)
    fx.update(depth=np.array(depth), alpha=np.array(alpha), batch=np.array(batch), seed=np.array(seed),
              model_seed=np.array(cfgmod.RANDOM_SEED), real=real.numpy(), z=z.numpy(), mix=mix.numpy())
    for k in ("fake", "score_real", "d_loss", "g_loss"):
        fx[f"ref64_{k}"] = r64[k].numpy()
        fx[f"ref32_{k}"] = r32[k].numpy()
        fx[f"noise_{k}"] = np.array(rel(r32[k], r64[k]))
    fx["g_upstream_noise"] = np.array(rel(r32["g_upstream"], r64["g_upstream"]))
    fx["g_upstream_sub"] = subsample(r64["g_upstream"], 2048)
    fx["g_upstream_sub32"] = subsample(r32["g_upstream"], 2048)
    for term in ("real", "fake", "gp"):
        for k, v in r64["d_terms"][term].items():
            if v is None:
                continue
            fx[f"d_term_{term}_norm.{k}"] = np.array(float(v.norm()))
            fx[f"d_term_{term}_sub.{k}"] = subsample(v)
            fx[f"d_term_{term}_sub32.{k}"] = subsample(r32["d_terms"][term][k])
            fx[f"d_term_{term}_noise.{k}"] = np.array(rel(r32["d_terms"][term][k], v))
    worst_noise = 0.0
    for gk in ("d_grads", "g_grads"):
        for k, v in r64[gk].items():
            if v is None:
                fx[f"{gk}_none.{k}"] = np.array(1)
                continue
            fx[f"{gk}_norm.{k}"] = np.array(float(v.norm()))
            fx[f"{gk}_sub.{k}"] = subsample(v)
            fx[f"{gk}_sub32.{k}"] = subsample(r32[gk][k])
            n32 = rel(r32[gk][k], v)
            fx[f"{gk}_noise.{k}"] = np.array(n32)
            worst_noise = max(worst_noise, n32)
            fx[f"{gk}_oracle32.{k}"] = np.array(rel(o32[gk][k], v))
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **fx)
    print(f"{name}.npz: depth={depth} alpha={alpha} B={batch}  oracle64-vs-ref64 worst {w:.2e}; "
          f"ref32-vs-ref64 worst grad noise {worst_noise:.2e}; D input-gradient noise "
          f"{float(fx['g_upstream_noise']):.2e}")


def step_fixture(out_dir, seed=11):
    """Parameters after one full Trainer.train() (5 D + 1 G Adam updates) at depth 0, B=8, fp64 and fp32."""
    from torch.optim import Adam
    depth, batch = 0, 8
    T = cfgmod.seq_len(depth)
    rng = np.random.default_rng(seed)
    reals = [torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng)) for _ in range(5)]
    torch.manual_seed(seed)
    zs = [torch.randn(batch, cfgmod.LATENT_SIZE) for _ in range(6)]
    mixes = [torch.rand(batch, 1, 1) for _ in range(5)]
    lr = cfgmod.stage_lr(depth)
    results = {}
    for dtype, tag in ((torch.float64, "ref64"), (torch.float32, "ref32")):
        g, d = build_pair(cfgmod.RANDOM_SEED)
        g, d = g.to(dtype), d.to(dtype)
        g.depth = d.depth = depth
        ga, da = GAdapter(g), DAdapter(d)
        opt_g = Adam(g.parameters(), lr, betas=(0.0, 0.99), eps=1e-8)
        opt_d = Adam(d.parameters(), lr, betas=(0.0, 0.99), eps=1e-8)
        np.random.seed(seed)
        orig = ref_losses.get_mixing_factor
        losses = []
        try:
            for i in range(5):
                ref_losses.get_mixing_factor = lambda b, i=i: mixes[i].to(dtype)
                dl = ref_losses.discriminator_loss(da, ga, {"x": reals[i].to(dtype)}, {"z": zs[i].to(dtype)},
                                                   "wgan_gp", 1e-3, 10.0, 1.0)
                dl.backward()
                opt_d.step()
                losses.append(float(dl))
        finally:
            ref_losses.get_mixing_factor = orig
        gl = ref_losses.generator_loss(da, ga, {"x": reals[4].to(dtype)}, {"z": zs[5].to(dtype)}, "wgan_gp", False,
                                       0.0)
        gl.backward()
        opt_g.step()
        losses.append(float(gl))
        results[tag] = (g, d, losses)
    # oracle step, fp64
    g, d = build_pair(cfgmod.RANDOM_SEED)
    gp, gs = to_oracle(g, torch.float64)
    dp, ds = to_oracle(d, torch.float64)
    tr = O.OracleTrainer(dp, ds, gp, gs, O.NetConfig(), depth, 1.0, lr)
    np.random.seed(seed)
    src = O.DropSource("numpy")
    tr.step([r.double() for r in reals], [z.double() for z in zs], [m.double() for m in mixes], src, src)
    g64, d64, l64 = results["ref64"]
    for k, v in d64.state_dict().items():
        assert rel(dp[k], v) < 1e-8, f"oracle step D.{k}: {rel(dp[k], v):.2e}"
    for k, v in g64.state_dict().items():
        assert rel(gp[k], v) < 1e-8, f"oracle step G.{k}: {rel(gp[k], v):.2e}"
    fx = dict(seed=np.array(seed), depth=np.array(depth), batch=np.array(batch), lr=np.array(lr),
              model_seed=np.array(cfgmod.RANDOM_SEED), losses64=np.array(l64), losses32=np.array(results["ref32"][2]))
    for i in range(5):
        fx[f"real{i}"], fx[f"mix{i}"] = reals[i].numpy(), mixes[i].numpy()
    for i in range(6):
        fx[f"z{i}"] = zs[i].numpy()
    g0, d0 = build_pair(cfgmod.RANDOM_SEED)
    for net, m64, m32, m0 in (("G", g64, results["ref32"][0], g0), ("D", d64, results["ref32"][1], d0)):
        s0 = m0.state_dict()
        for k, v in m64.state_dict().items():
            delta = v.double() - s0[k].double()
            fx[f"{net}_delta_norm.{k}"] = np.array(float(delta.norm()))
            fx[f"{net}_delta_sub.{k}"] = subsample(delta)
            d32 = m32.state_dict()[k].double() - s0[k].double()
            fx[f"{net}_delta_noise.{k}"] = np.array(rel(d32, delta) if float(delta.norm()) > 0 else 0.0)
    np.savez_compressed(os.path.join(out_dir, "step_depth0.npz"), **fx)
    print("step_depth0.npz: losses", [f"{x:.5f}" for x in l64])


def dataset_fixture(out_dir):
    """The reference's EEGDataset.__getitem__ (dataset.py:161-187) on a synthetic recording, at every
    stage and at two fade-in factors; inputs are the windows its load_file cuts (same host RNG)."""
    sys.path.insert(0, REF)
    import dataset as ref_dataset  # noqa
    sys.path.remove(REF)
    rng = np.random.default_rng(7)
    stride = 2880
    recording = rng.standard_normal((5, stride * 3)).astype(np.float32)
    recording = np.cumsum(recording, axis=1) * 0.05  # smooth-ish, EEG-like drift
    recording = (recording / np.abs(recording).max()).astype(np.float32)
    ds = ref_dataset.EEGDataset([[(0, 0), (0, 1), (0, 2)], [recording]])
    assert ds.seq_len == 1920 and ds.stride == stride
    fx = {"seed": np.array(11)}
    np.random.seed(11)
    windows = np.stack([ds.load_file(i) for i in range(3)])
    fx["windows"] = windows
    for depth in range(0, 8):
        for alpha in (1.0, 0.3):
            if alpha != 1.0 and depth == 0:
                continue
            ds.model_depth, ds.alpha = depth, alpha
            np.random.seed(11)
            out = torch.stack([ds[i]["x"] for i in range(3)])
            fx[f"x_d{depth}_a{alpha:g}"] = out.numpy()
            # oracle (fp64) pinned against the reference's float32 arithmetic
            ref = O.alpha_fade_reals(O.datapoint_from_depth(torch.from_numpy(windows).double(), depth), depth, alpha)
            err = rel(ref, out)
            assert ref.shape == out.shape and err < 2e-4, (depth, alpha, err)
    np.savez_compressed(os.path.join(out_dir, "dataset_ops.npz"), **fx)
    print("dataset_ops.npz:", len(fx), "arrays")


# ------------------------------------------------------------------------------------------ blocks
BLOCK_CASES = (
    # name, network, module path, kwargs of forward, input shape (B, C, T)
    ("g_head", "g", "block0", dict(last=False), (8, 256, 1)),
    ("g_mid_rgb", "g", "blocks.3", dict(last=True), (4, 64, 768)),
    ("g_wide", "g", "blocks.1", dict(last=False), (4, 256, 256)),
    ("d_first", "d", "blocks.1", dict(first=True), (4, 5, 1280)),
    ("d_mid", "d", "blocks.4", dict(first=False), (4, 64, 512)),
    ("d_last", "d", "blocks.7", dict(first=False), (8, 256, 32)),
)


def block_fixture(out_dir, seed=21):
    """GBlock.forward / DBlock.forward of the reference (network.py:206-218, 441-449) on the seed-initialised
    default networks, with the GDrop draws of the numpy stream seeded per case: inputs, outputs and input
    gradients in fp64 and fp32."""
    import copy
    g_ref, d_ref = build_pair(cfgmod.RANDOM_SEED)
    fx = {"generator_seed": np.array(seed)}
    gen = torch.Generator().manual_seed(seed)
    cfg = O.NetConfig()
    for n, (name, net, path, kw, shape) in enumerate(BLOCK_CASES):
        model = g_ref if net == "g" else d_ref
        block = model.get_submodule(path)
        x = torch.randn(*shape, generator=gen)
        gy = None
        outs = {}
        for dtype, tag in ((torch.float64, "64"), (torch.float32, "32")):
            blk = copy.deepcopy(block).to(dtype)
            xi = x.detach().clone().to(dtype).requires_grad_(True)
            np.random.seed(seed + n)
            y = blk(xi, **kw)
            if gy is None:
                gy = torch.randn(y.shape, generator=gen)
            (gx,) = torch.autograd.grad(y, xi, gy.to(dtype))
            outs[tag] = (y.detach(), gx)
        # the oracle's block against the reference's, same draws
        params, scales = to_oracle(model, torch.float64)
        np.random.seed(seed + n)
        src = O.DropSource("numpy")
        if net == "g":
            yo = O.g_block(params, scales, path, x.double(), cfg, src, first=(path == "block0"), last=kw["last"])
        else:
            yo = O.d_block(params, scales, int(path.split(".")[1]), x.double(), cfg, src, first=kw["first"])
        err = rel(yo.detach(), outs["64"][0])
        assert err < 1e-12, (name, err)
        # inputs are re-drawn by the tests (same generator, same order); outputs are kept as 4096 samples
        fx[f"{name}_x_check"] = np.array([float(x.double().sum()), float((x.double() ** 2).sum())])
        fx[f"{name}_gy_check"] = np.array([float(gy.double().sum()), float((gy.double() ** 2).sum())])
        fx[f"{name}_yshape"] = np.array(outs["64"][0].shape)
        fx[f"{name}_y64"], fx[f"{name}_gx64"] = subsample(outs["64"][0], 4096), subsample(outs["64"][1], 4096)
        fx[f"{name}_y32"], fx[f"{name}_gx32"] = subsample(outs["32"][0], 4096), subsample(outs["32"][1], 4096)
        fx[f"{name}_ynorm"] = np.array(float(outs["64"][0].double().norm()))
        fx[f"{name}_gxnorm"] = np.array(float(outs["64"][1].double().norm()))
        fx[f"{name}_seed"] = np.array(seed + n)
        print(f"  {name}: y {tuple(outs['64'][0].shape)}, fp32 vs fp64 {rel(outs['32'][0], outs['64'][0]):.1e}")
    np.savez_compressed(os.path.join(out_dir, "blocks.npz"), **fx)
    print("blocks.npz:", len(fx), "arrays")


# ------------------------------------------------------------------------------------------ other loss types
def loss_types_fixture(out_dir):
    """The reference's critic and generator losses for every loss type (losses.py:40-120) at depth 1, B = 8, in
    fp64: loss values and all parameter gradients, with the oracle held to them at 1e-9.  ``wgan_theirs`` is
    recorded twice: once with its penalty gain above zero and once at zero (the reference then skips the
    interpolates' pass)."""
    depth, alpha, batch = 1, 1.0, 8
    T = cfgmod.seq_len(depth)
    fx = {"depth": np.array(depth), "alpha": np.array(alpha), "model_seed": np.array(cfgmod.RANDOM_SEED)}
    cases = [("hinge", 201, 10.0), ("rsgan", 202, 10.0), ("rasgan", 203, 10.0), ("rahinge", 204, 10.0),
             ("wgan_gp", 205, 10.0)]
    found = {}
    for seed in range(206, 260):  # wgan_theirs: one seed for each sign of mean D(real) - mean D(fake)
        cases_try = ("wgan_theirs", seed)
        g, d = build_pair(cfgmod.RANDOM_SEED)
        g, d = g.double(), d.double()
        g.depth = d.depth = depth
        rng = np.random.default_rng(seed)
        real = torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng)).double()
        torch.manual_seed(seed)
        z = torch.randn(batch, cfgmod.LATENT_SIZE).double()
        np.random.seed(seed + 1)
        with torch.no_grad():
            gain = float(torch.relu(d(real)[0].mean() - d(g(z)[0]["x"])[0].mean()))
        key = "on" if gain > 0 else "off"
        if key not in found:
            found[key] = cases_try
        if len(found) == 2:
            break
    cases += [("wgan_theirs", found["on"][1], 10.0), ("wgan_theirs", found["off"][1], 10.0)]
    # wgan_gp without its penalty (grad_lambda = 0: the reference then skips the interpolates' pass, losses.py:106)
    cases += [("wgan_gp", 261, 0.0)]
    names = []
    for loss_type, seed, grad_lambda in cases:
        rng = np.random.default_rng(seed)
        real32 = torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng))
        torch.manual_seed(seed)
        z32 = torch.randn(batch, cfgmod.LATENT_SIZE)
        mix32 = torch.rand(batch, 1, 1)

        def run_ref(dtype):
            g, d = build_pair(cfgmod.RANDOM_SEED)
            g, d = g.to(dtype), d.to(dtype)
            g.depth = d.depth = depth
            g.alpha = d.alpha = alpha
            ga, da = GAdapter(g), DAdapter(d)
            real, z, mix = real32.to(dtype), z32.to(dtype), mix32.to(dtype)
            np.random.seed(seed + 1)
            orig = ref_losses.get_mixing_factor
            ref_losses.get_mixing_factor = lambda b: mix
            # (the reference caches float32 targets for its BCE terms, losses.py:21-32; a run is in one dtype throughout)
            ref_losses.one, ref_losses.zero = torch.ones(batch).to(dtype), torch.zeros(batch).to(dtype)
            try:
                d_loss = ref_losses.discriminator_loss(da, ga, {"x": real}, {"z": z}, loss_type, 1e-3, grad_lambda, 1.0)
            finally:
                ref_losses.get_mixing_factor = orig
            d_loss.backward()
            d_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in d.named_parameters()}
            np.random.seed(seed + 2)
            g_loss = ref_losses.generator_loss(da, ga, {"x": real}, {"z": z}, loss_type, False, 0.0)
            g_loss.backward()
            g_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in g.named_parameters()}
            return g, d, real, z, mix, d_loss, d_grads, g_loss, g_grads

        _, _, _, _, _, d_loss32, d_grads32, g_loss32, g_grads32 = run_ref(torch.float32)
        g, d, real, z, mix, d_loss, d_grads, g_loss, g_grads = run_ref(torch.float64)
        # the oracle on the same draws
        gp, gs = to_oracle(g, torch.float64)
        dp, ds = to_oracle(d, torch.float64)
        for p in list(gp.values()) + list(dp.values()):
            p.requires_grad_(True)
        cfg = O.NetConfig()
        np.random.seed(seed + 1)
        src = O.DropSource("numpy")
        ol, info = O.d_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, mix, src, src,
                                   grad_lambda=grad_lambda)
        og = dict(zip(dp.keys(), torch.autograd.grad(ol, list(dp.values()), allow_unused=True)))
        assert rel(ol, d_loss) < 1e-9, (loss_type, "d_loss", rel(ol, d_loss))
        for k, v in d_grads.items():
            assert (v is None) == (og[k] is None) and (v is None or rel(og[k], v) < 1e-9), (loss_type, k)
        np.random.seed(seed + 2)
        src = O.DropSource("numpy")
        ol, _ = O.g_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, src, src)
        og = dict(zip(gp.keys(), torch.autograd.grad(ol, list(gp.values()), allow_unused=True)))
        assert rel(ol, g_loss) < 1e-9, (loss_type, "g_loss", rel(ol, g_loss))
        for k, v in g_grads.items():
            assert (v is None) == (og[k] is None) and (v is None or rel(og[k], v) < 1e-9), (loss_type, k)
        gain = info.get("gain")
        name = loss_type if loss_type != "wgan_theirs" else ("wgan_theirs_on" if float(gain) > 0 else "wgan_theirs_off")
        if grad_lambda == 0:
            name += "_nopenalty"
        names.append(name)
        fx[f"{name}.seed"] = np.array(seed)
        fx[f"{name}.grad_lambda"] = np.array(grad_lambda)
        fx[f"{name}.real"], fx[f"{name}.z"], fx[f"{name}.mix"] = real.float().numpy(), z.float().numpy(), mix.float().numpy()
        fx[f"{name}.d_loss"], fx[f"{name}.g_loss"] = d_loss.detach().numpy(), g_loss.detach().numpy()
        fx[f"{name}.d_loss32"], fx[f"{name}.g_loss32"] = d_loss32.detach().numpy(), g_loss32.detach().numpy()
        for tag, grads, grads32 in (("d", d_grads, d_grads32), ("g", g_grads, g_grads32)):
            for k, v in grads.items():
                if v is None:
                    fx[f"{name}.{tag}_none.{k}"] = np.array(1)
                else:
                    fx[f"{name}.{tag}_sub.{k}"] = subsample(v)
                    fx[f"{name}.{tag}_norm.{k}"] = np.array(float(v.norm()))
                    if grads32[k] is not None:  # (wgan_theirs at a gain near zero may skip the penalty in one precision)
                        fx[f"{name}.{tag}_sub32.{k}"] = subsample(grads32[k])
                        fx[f"{name}.{tag}_noise.{k}"] = np.array(rel(grads32[k], v))
        ref_losses.one = ref_losses.zero = None
        print(f"loss type {name}: seed {seed}, d_loss {float(d_loss):.6f}, g_loss {float(g_loss):.6f}, oracle matches")
    fx["names"] = np.array(names)
    np.savez_compressed(os.path.join(out_dir, "loss_types.npz"), **fx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    jobs = {
        "per_op": lambda: per_op_fixture(args.out),
        "e2e_d0": lambda: e2e_fixture(args.out, "e2e_d0_a1", 0, 1.0, 8, 101),
        "e2e_d1": lambda: e2e_fixture(args.out, "e2e_d1_a05", 1, 0.5, 8, 102),
        "e2e_d4": lambda: e2e_fixture(args.out, "e2e_d4_a1", 4, 1.0, 4, 103),
        "e2e_d7": lambda: e2e_fixture(args.out, "e2e_d7_a1", 7, 1.0, 4, 104),
        "e2e_d7f": lambda: e2e_fixture(args.out, "e2e_d7_a05", 7, 0.5, 4, 105),
        "step": lambda: step_fixture(args.out),
        "dataset": lambda: dataset_fixture(args.out),
        "blocks": lambda: block_fixture(args.out),
        "loss_types": lambda: loss_types_fixture(args.out),
    }
    for k, fn in jobs.items():
        if not args.only or k in args.only.split(","):
            fn()


if __name__ == "__main__":
    main()
