#!/usr/bin/env python3
"""Golden fixture of the class-conditional networks (tests/golden/conditional.npz), from the REFERENCE implementation
(see make_golden.py for how it is imported and what is stored: numbers only).

The reference's training step does not reach these paths (its losses hand the networks dicts their forward does not
take, losses.py:76-79, and the generator returns the ENCODED classes under 'y', network.py:352, which the critic's
embedding cannot take), so the networks are driven directly: ``Generator(z, y)`` with the classes entering every batch
norm through the shared embedding ('CBN': class-conditional; 'CSM': latent and classes; layers.py:93-134,
network.py:284-292, 355-361) and ``Discriminator(x, y)`` with its projection term (network.py:518-523, 555-557).
Per case, in fp64 and fp32: the fakes and a linear functional of them with all generator gradients; the scores and a
functional of them plus a penalty on the input gradient's norm (the second order the gradient penalty needs) with all
critic gradients.  This repository's modules are checked to initialise bit-identically and the CPU restatement to
reproduce the fp64 run.

Usage: python tools/make_golden_conditional.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as M  # noqa: E402  (imports the reference with its two stubs)

cfgmod, O, ref_network = M.cfgmod, M.O, M.ref_network

CASES = {
    "cbn": dict(shared=dict(num_classes=4), generator=dict(act_norm="batch"), depth=2, alpha=0.6),
    "csm": dict(shared=dict(num_classes=3), generator=dict(act_norm="batch", z_to_bn=True), depth=1, alpha=1.0),
    "cbn_raw_classes": dict(shared=dict(num_classes=5, equalized=False), generator=dict(act_norm="batch", shared_embedding_size=0),
                            discriminator=dict(spectral=True), depth=2, alpha=1.0),
    # class scores with a time axis (16 steps): the modulation of every batch norm is resampled to its layer's length
    # (layers.py:121-134).  The reference's critic takes per-sample classes only (its embedding is an nn.Linear on the
    # last axis, network.py:518-519, 555-557): it is evaluated without classes here
    "cbn_timed": dict(shared=dict(num_classes=4), generator=dict(act_norm="batch"), depth=2, alpha=0.6, y_steps=16,
                      critic_without_classes=True),
    "csm_timed": dict(shared=dict(num_classes=3), generator=dict(act_norm="batch", z_to_bn=True), depth=1, alpha=1.0,
                      y_steps=8, critic_without_classes=True),
}


def oracle_config(case):
    ge = case.get("generator", {})
    return O.NetConfig(g_act_norm=ge.get("act_norm", "pixel"), z_to_bn=ge.get("z_to_bn", False))


def build_pair(case):
    kw = dict(shared=case.get("shared"), generator=case.get("generator"), discriminator=case.get("discriminator"))
    g_ref, d_ref = cfgmod.build_models(cfgmod.RANDOM_SEED, ref_network.Generator, ref_network.Discriminator, **kw)
    g_my, d_my = cfgmod.build_models(cfgmod.RANDOM_SEED, **kw)
    for a, b, nm in ((g_ref, g_my, "G"), (d_ref, d_my, "D")):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()), f"{nm}: state-dict keys differ: {set(sa) ^ set(sb)}"
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"{nm}.{k}: initial weights differ"
        assert M.scales_of(a) == M.scales_of(b), f"{nm}: equalized scales differ"
    return g_ref, d_ref


def functionals(fake, r1, score, gx, r2):
    l_g = (fake * r1).sum()
    l_d = (score * r2).sum() + 10.0 * ((gx.reshape(gx.shape[0], -1).norm(dim=1) - 1.0) ** 2).mean()
    return l_g, l_d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(M.REPO, "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    batch = 8
    fx = {"model_seed": np.array(cfgmod.RANDOM_SEED), "names": np.array(list(CASES))}
    for idx, (name, case) in enumerate(CASES.items()):
        seed = 701 + idx
        depth, alpha = case["depth"], case["alpha"]
        classes = case["shared"]["num_classes"]
        T = cfgmod.seq_len(depth)
        rng = np.random.default_rng(seed)
        real32 = torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng))
        torch.manual_seed(seed)
        z32 = torch.randn(batch, cfgmod.LATENT_SIZE)
        y32 = torch.nn.functional.one_hot(torch.randint(classes, (batch,)), classes).float() + 0.1 * torch.randn(batch, classes)
        if "y_steps" in case:  # class scores per time step: [B, classes, steps]
            y32 = (y32.unsqueeze(2) + 0.5 * torch.randn(batch, classes, case["y_steps"])).contiguous()
        d_y = (lambda y: None) if case.get("critic_without_classes") else (lambda y: y)
        r1_32, r2_32 = torch.randn(batch, cfgmod.NUM_CHANNELS, T), torch.randn(batch)

        def run_ref(dtype):
            g, d = build_pair(case)
            g, d = g.to(dtype), d.to(dtype)
            start = tuple({k: v.detach().clone() for k, v in m.state_dict().items()} for m in (g, d))
            g.depth = d.depth = depth
            g.alpha = d.alpha = alpha
            z, y, real, r1, r2 = (t.to(dtype) for t in (z32, y32, real32, r1_32, r2_32))
            np.random.seed(seed)
            fake = g(z, y)[0]["x"]
            x_in = real.clone().requires_grad_(True)
            np.random.seed(seed + 1)
            score = d(x_in, d_y(y))[0]
            (gx,) = torch.autograd.grad(score.sum(), x_in, create_graph=True)
            l_g, l_d = functionals(fake, r1, score, gx, r2)
            g_grads = dict(zip([k for k, _ in g.named_parameters()],
                               torch.autograd.grad(l_g, list(g.parameters()), allow_unused=True)))
            d_grads = dict(zip([k for k, _ in d.named_parameters()],
                               torch.autograd.grad(l_d, list(d.parameters()), allow_unused=True)))
            return g, d, fake.detach(), score.detach(), l_g.detach(), l_d.detach(), g_grads, d_grads, start

        r32 = run_ref(torch.float32)
        g, d, fake, score, l_g, l_d, g_grads, d_grads, start = run_ref(torch.float64)
        cfg = oracle_config(case)
        gp, gs = {k: v.clone().to(torch.float64) for k, v in start[0].items()}, M.scales_of(g)
        dp, ds = {k: v.clone().to(torch.float64) for k, v in start[1].items()}, M.scales_of(d)
        for p in list(gp.values()) + list(dp.values()):
            p.requires_grad_(True)
        z, y, real, r1, r2 = (t.double() for t in (z32, y32, real32, r1_32, r2_32))
        np.random.seed(seed)
        ofake = O.generator_forward(gp, gs, z, depth, alpha, cfg, O.DropSource("numpy"), y=y)
        x_in = real.clone().requires_grad_(True)
        np.random.seed(seed + 1)
        oscore, _ = O.discriminator_forward(dp, ds, x_in, depth, alpha, cfg, O.DropSource("numpy"), y=d_y(y))
        (ogx,) = torch.autograd.grad(oscore.sum(), x_in, create_graph=True)
        ol_g, ol_d = functionals(ofake, r1, oscore, ogx, r2)
        assert M.rel(ofake, fake) < 1e-9 and M.rel(oscore, score) < 1e-9, (name, M.rel(ofake, fake), M.rel(oscore, score))
        assert M.rel(ol_g, l_g) < 1e-9 and M.rel(ol_d, l_d) < 1e-9, (name, "functionals")

        def close(a, b):
            return M.rel(a, b) < 1e-9 or float((a - b).norm()) < 1e-12
        og = dict(zip(gp.keys(), torch.autograd.grad(ol_g, list(gp.values()), allow_unused=True)))
        for k, v in g_grads.items():
            assert (v is None) == (og[k] is None) and (v is None or close(og[k], v)), (name, "G", k)
        od = dict(zip(dp.keys(), torch.autograd.grad(ol_d, list(dp.values()), allow_unused=True)))
        for k, v in d_grads.items():
            assert (v is None) == (od[k] is None) and (v is None or close(od[k], v)), (name, "D", k)
        fx[f"{name}.seed"], fx[f"{name}.depth"], fx[f"{name}.alpha"] = np.array(seed), np.array(depth), np.array(alpha)
        for key, t in (("real", real32), ("z", z32), ("y", y32), ("r1", r1_32), ("r2", r2_32)):
            fx[f"{name}.{key}"] = t.numpy()
        fx[f"{name}.fake"], fx[f"{name}.score"] = fake.numpy(), score.numpy()
        fx[f"{name}.l_g"], fx[f"{name}.l_d"] = l_g.numpy(), l_d.numpy()
        fx[f"{name}.l_g32"], fx[f"{name}.l_d32"] = r32[4].numpy(), r32[5].numpy()
        for tag, grads, grads32 in (("g", g_grads, r32[6]), ("d", d_grads, r32[7])):
            for k, v in grads.items():
                if v is None:
                    fx[f"{name}.{tag}_none.{k}"] = np.array(1)
                else:
                    fx[f"{name}.{tag}_sub.{k}"] = M.subsample(v)
                    fx[f"{name}.{tag}_norm.{k}"] = np.array(float(v.norm()))
                    if grads32[k] is not None:
                        fx[f"{name}.{tag}_noise.{k}"] = np.array(M.rel(grads32[k], v))
        print(f"{name}: depth {depth}, alpha {alpha}, l_g {float(l_g):.6f}, l_d {float(l_d):.6f}; weights identical, "
              f"oracle matches", flush=True)
    np.savez_compressed(os.path.join(args.out, "conditional.npz"), **fx)


if __name__ == "__main__":
    main()
