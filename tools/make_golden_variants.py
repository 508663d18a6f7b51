#!/usr/bin/env python3
"""Golden fixture of the network and loss options beside the defaults (tests/golden/variants.npz), from the
REFERENCE implementation (see make_golden.py for how it is imported and what is stored: numbers only).

Cases: residual blocks, the generator's 'residual' and 'mean' signal modes (mid fade-in and at alpha = 1), the
critic's input at every stage, plain layers (ReLU, no equalization, xavier init, no tanh, latents not normalised),
separable layers, deep blocks, self attention (wide and narrow stages), batch norm in the generator (plain and
modulated by the latent), the 'sngan' / 'sagan' / 'biggan' forms of to_rgb, spectral norm in either network, fully convolutional networks, batch norm and pixel norm in the critic, and the generator loss's random multiplier
and feature-matching term.  Every case records one critic and one
generator loss evaluation in fp64 and fp32 (values, fakes, scores, all parameter gradients), after this
repository's modules were checked to initialise bit-identically and the CPU restatement to reproduce the fp64 run.

Usage: python tools/make_golden_variants.py [--out tests/golden]
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as M  # noqa: E402  (imports the reference with its two stubs)

cfgmod, O, ref_network, ref_losses = M.cfgmod, M.O, M.ref_network, M.ref_losses

CASES = {
    "residual": dict(shared=dict(residual=True), depth=3, alpha=1.0),
    "residual_fade": dict(shared=dict(residual=True), depth=2, alpha=0.6),
    "rgb_residual": dict(generator=dict(rgb_generation_mode="residual"), depth=2, alpha=0.5),
    "rgb_mean": dict(generator=dict(rgb_generation_mode="mean"), depth=2, alpha=0.5),
    "rgb_mean_full": dict(generator=dict(rgb_generation_mode="mean"), depth=2, alpha=1.0),
    "all_layers": dict(discriminator=dict(input_to_all_layers=True), depth=2, alpha=0.7),
    "plain": dict(shared=dict(act_alpha=0.0, equalized=False, init="xavier_uniform"),
                  generator=dict(no_tanh=True, normalize_latents=False), discriminator=dict(group_size=-1),
                  depth=1, alpha=1.0),  # (ReLU feeds MinibatchStddev exact zeros, whose sqrt has no gradient: NaN in the reference)
    "separable": dict(shared=dict(separable=True), depth=2, alpha=1.0),
    "deep": dict(shared=dict(deep=True), depth=3, alpha=0.8),
    "g_loss_opts": dict(depth=1, alpha=1.0, random_multiply=True, feature_matching_lambda=1.0),
    "g_loss_opts_rel": dict(depth=1, alpha=1.0, random_multiply=True, feature_matching_lambda=0.5, loss_type="rasgan"),
    # self attention in both networks; gamma (zero at construction) is set so that the layer takes part
    "attention": dict(shared=dict(self_attention_layers=[0, 1]), depth=3, alpha=1.0, gamma=0.4),
    "attention_narrow": dict(shared=dict(self_attention_layers=[3]), depth=5, alpha=1.0, gamma=-0.8),
    # batch norm in the generator's layers: plain (affine), and modulated by the latent
    "batch_norm": dict(generator=dict(act_norm="batch"), depth=3, alpha=0.6),
    "batch_norm_z": dict(generator=dict(act_norm="batch", z_to_bn=True), depth=2, alpha=1.0),
    "split_z": dict(generator=dict(act_norm="batch", z_to_bn=True, split_z=True), depth=2, alpha=0.5),
    # the other forms of the layer that turns features into the signal
    "to_rgb_sngan": dict(generator=dict(to_rgb_mode="sngan"), depth=2, alpha=0.5),
    "to_rgb_sagan": dict(generator=dict(to_rgb_mode="sagan"), depth=3, alpha=1.0),
    "to_rgb_biggan": dict(generator=dict(to_rgb_mode="biggan"), depth=2, alpha=0.7),
    "to_rgb_biggan_plain": dict(generator=dict(to_rgb_mode="biggan", no_tanh=True), depth=1, alpha=1.0),
    # spectral norm of every conv weight: one power iteration per forward, so the runs below start from the
    # vectors drawn at construction and make their passes in the same order
    # (not equalized: a normalised weight times the equalized layers' input scale leaves no signal to compare)
    "spectral_d": dict(shared=dict(equalized=False), discriminator=dict(spectral=True), depth=2, alpha=0.6),
    "spectral_g": dict(shared=dict(equalized=False), generator=dict(spectral=True), depth=2, alpha=1.0),
    "spectral_both": dict(shared=dict(equalized=False, self_attention_layers=[1]), generator=dict(spectral=True),
                          discriminator=dict(spectral=True), depth=3, alpha=1.0, gamma=0.3),
    # fully convolutional networks: the critic alone (its score the mean over time), and both (latents [B, 256, 32])
    "conv_only_d": dict(shared=dict(residual=True), discriminator=dict(conv_only=True), depth=2, alpha=0.6),
    "conv_only": dict(generator=dict(conv_only=True), discriminator=dict(conv_only=True), depth=2, alpha=1.0, z_len=32),
    # batch norm in the critic's layers: differentiated twice under the gradient penalty
    "d_batch_norm": dict(discriminator=dict(act_norm="batch"), depth=2, alpha=0.6),
    # (depth 2 at alpha 1 was tried first: one pre-activation of the critic lies within 5e-7 of zero there, and the
    # fp32 rounding of the fakes flips its LeakyReLU -- 5e-3 on that sample's input gradient in pure fp64 arithmetic)
    "d_pixel_norm": dict(discriminator=dict(act_norm="pixel"), depth=3, alpha=0.7),
    # MinibatchStddev over several temporal windows (layers.py:72, 81-89): two windows of 16 steps, and 32 // 5 = 6
    # steps per window, i.e. five windows of 6 and the remainder of 2
    # per-channel noise on every generator layer's conv output (network.py:162-172, 203-208): the learned weights
    # start at zero (which zeroes the layer), so they are set away from it; the draws come from torch's host generator
    "per_channel_noise": dict(generator=dict(per_channel_noise=True), depth=2, alpha=0.6, noise_weights=True),
    "per_channel_noise_deep": dict(shared=dict(deep=True), generator=dict(per_channel_noise=True, act_norm="batch"),
                                   depth=1, alpha=1.0, noise_weights=True),
    # a fully convolutional generator whose batch norms are modulated by the latents [B, 256, 32]: conditioning with a
    # time axis, resampled to every layer's length (layers.py:121-134)
    "conv_only_z_to_bn": dict(generator=dict(conv_only=True, act_norm="batch", z_to_bn=True),
                              discriminator=dict(conv_only=True), depth=2, alpha=0.7, z_len=32),
    "temporal_groups": dict(discriminator=dict(temporal_groups_per_window=2), depth=2, alpha=1.0),
    "temporal_groups_ragged": dict(discriminator=dict(temporal_groups_per_window=5, group_size=2), depth=1, alpha=0.5),
}


def oracle_config(case):
    kw = {}
    sh, ge, di = case.get("shared", {}), case.get("generator", {}), case.get("discriminator", {})
    if "residual" in sh:
        kw["residual"] = sh["residual"]
    # ("equalized" and "spectral" need no oracle option: the scales and the state dict carry them)
    if "act_alpha" in sh:
        kw["act_alpha"] = sh["act_alpha"]
    if "deep" in sh:
        kw["deep"] = sh["deep"]
    if "self_attention_layers" in sh:
        kw["self_attention_layers"] = tuple(sh["self_attention_layers"])
    for k in ("rgb_generation_mode", "no_tanh", "normalize_latents", "z_to_bn", "split_z", "to_rgb_mode"):
        if k in ge:
            kw[k] = ge[k]
    if "act_norm" in ge:
        kw["g_act_norm"] = ge["act_norm"]
    kw["g_conv_only"], kw["d_conv_only"] = bool(ge.get("conv_only")), bool(di.get("conv_only"))
    if "act_norm" in di:
        kw["d_act_norm"] = di["act_norm"]
    for k in ("input_to_all_layers", "group_size", "temporal_groups_per_window"):
        if k in di:
            kw[k] = di[k]
    return O.NetConfig(**kw)


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
    if "gamma" in case:
        set_gamma(g_ref, d_ref, case["gamma"])
    if case.get("noise_weights"):
        set_noise_weights(g_ref)
    return g_ref, d_ref


def set_noise_weights(g):
    """The per-channel noise weights (zero at construction): 0.6 .. 1.4 along the channels, the sign alternating from
    tensor to tensor."""
    with torch.no_grad():
        j = 0
        for k, p in g.named_parameters():
            if ".noises." in k:
                p.copy_(torch.linspace(0.6, 1.4, p.shape[1]).view(1, -1, 1) * (1.0 if j % 2 == 0 else -1.0))
                j += 1


def set_gamma(g, d, value):
    """Every SelfAttention layer's scalar, G's and D's apart so that no two layers share a value."""
    with torch.no_grad():
        for net, sign in ((g, 1.0), (d, -1.0)):
            for j, layer in enumerate(getattr(net, "dummy", [])):
                layer.gamma.fill_(sign * value * (1.0 + 0.25 * j))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(M.REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma-separated cases to regenerate; the others are kept from the file")
    args = ap.parse_args()
    torch.set_num_threads(8)
    batch = 8
    path = os.path.join(args.out, "variants.npz")
    only = [n for n in args.only.split(",") if n]
    fx = {}
    if only:
        fx = {k: v for k, v in np.load(path).items() if not any(k.startswith(n + ".") for n in only)}
    fx.update({"model_seed": np.array(cfgmod.RANDOM_SEED), "names": np.array(list(CASES))})
    for idx, (name, case) in enumerate(CASES.items()):
        seed = 301 + idx
        if only and name not in only:
            continue

        def close(a, b):  # (a gradient that cancels exactly in the mathematics is round-off noise in both runs)
            return M.rel(a, b) < 1e-9 or float((a - b).norm()) < 1e-12
        depth, alpha = case["depth"], case["alpha"]
        loss_type = case.get("loss_type", "wgan_gp")
        rm, fm = case.get("random_multiply", False), case.get("feature_matching_lambda", 0.0)
        T = cfgmod.seq_len(depth)
        rng = np.random.default_rng(seed)
        real32 = torch.from_numpy(O.synthetic_eeg(batch, cfgmod.NUM_CHANNELS, T, rng))
        torch.manual_seed(seed)
        z32 = torch.randn(batch, cfgmod.LATENT_SIZE, *([case["z_len"]] if "z_len" in case else []))
        mix32 = torch.rand(batch, 1, 1)

        def run_ref(dtype):
            g, d = build_pair(case)
            g, d = g.to(dtype), d.to(dtype)
            # (the state the passes below start from: the vectors of a spectral norm move at every forward)
            start = tuple({k: v.detach().clone() for k, v in m.state_dict().items()} for m in (g, d))
            g.depth = d.depth = depth
            g.alpha = d.alpha = alpha
            ga, da = M.GAdapter(g), M.DAdapter(d)
            real, z, mix = real32.to(dtype), z32.to(dtype), mix32.to(dtype)
            np.random.seed(seed)
            torch.manual_seed(seed + 10)  # (host draws inside the networks: per-channel noise)
            with torch.no_grad():
                fake = g(z)[0]["x"]
                score = d(real)[0]
            np.random.seed(seed + 1)
            torch.manual_seed(seed + 11)
            orig = ref_losses.get_mixing_factor
            ref_losses.get_mixing_factor = lambda b: mix
            ref_losses.one, ref_losses.zero = torch.ones(batch).to(dtype), torch.zeros(batch).to(dtype)
            try:
                d_loss = ref_losses.discriminator_loss(da, ga, {"x": real}, {"z": z}, loss_type, 1e-3, 10.0, 1.0)
            finally:
                ref_losses.get_mixing_factor = orig
            d_loss.backward()
            d_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in d.named_parameters()}
            np.random.seed(seed + 2)
            random.seed(seed + 3)
            torch.manual_seed(seed + 12)
            g_loss = ref_losses.generator_loss(da, ga, {"x": real}, {"z": z}, loss_type, rm, fm)
            g_loss.backward()
            g_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in g.named_parameters()}
            ref_losses.one = ref_losses.zero = None
            return g, d, real, z, mix, fake, score, d_loss, d_grads, g_loss, g_grads, start

        r32 = run_ref(torch.float32)
        g, d, real, z, mix, fake, score, d_loss, d_grads, g_loss, g_grads, start = run_ref(torch.float64)
        # the CPU restatement on the same draws, from the same starting state
        cfg = oracle_config(case)
        gp, gs = {k: v.clone().to(torch.float64) for k, v in start[0].items()}, M.scales_of(g)
        dp, ds = {k: v.clone().to(torch.float64) for k, v in start[1].items()}, M.scales_of(d)
        for p in list(gp.values()) + list(dp.values()):
            p.requires_grad_(True)
        np.random.seed(seed)
        torch.manual_seed(seed + 10)
        src = O.DropSource("numpy")
        with torch.no_grad():
            ofake = O.generator_forward(gp, gs, z, depth, alpha, cfg, src)
            oscore, _ = O.discriminator_forward(dp, ds, real, depth, alpha, cfg, src)
        assert M.rel(ofake, fake) < 1e-9 and M.rel(oscore, score) < 1e-9, (name, M.rel(ofake, fake), M.rel(oscore, score))
        np.random.seed(seed + 1)
        torch.manual_seed(seed + 11)
        src = O.DropSource("numpy")
        ol, _ = O.d_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, mix, src, src)
        og = dict(zip(dp.keys(), torch.autograd.grad(ol, list(dp.values()), allow_unused=True)))
        assert M.rel(ol, d_loss) < 1e-9, (name, "d_loss", M.rel(ol, d_loss))
        for k, v in d_grads.items():
            assert (v is None) == (og[k] is None) and (v is None or close(og[k], v)), (name, k)
        np.random.seed(seed + 2)
        random.seed(seed + 3)
        torch.manual_seed(seed + 12)
        src = O.DropSource("numpy")
        ol, _ = O.g_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, src, src, random_multiply=rm,
                             feature_matching_lambda=fm)
        og = dict(zip(gp.keys(), torch.autograd.grad(ol, list(gp.values()), allow_unused=True)))
        assert M.rel(ol, g_loss) < 1e-9, (name, "g_loss", M.rel(ol, g_loss))
        for k, v in g_grads.items():
            assert (v is None) == (og[k] is None) and (v is None or close(og[k], v)), (name, k)
        fx[f"{name}.seed"], fx[f"{name}.depth"], fx[f"{name}.alpha"] = np.array(seed), np.array(depth), np.array(alpha)
        fx[f"{name}.real"], fx[f"{name}.z"], fx[f"{name}.mix"] = real32.numpy(), z32.numpy(), mix32.numpy()
        fx[f"{name}.fake"], fx[f"{name}.score"] = fake.numpy(), score.numpy()
        fx[f"{name}.d_loss"], fx[f"{name}.g_loss"] = d_loss.detach().numpy(), g_loss.detach().numpy()
        fx[f"{name}.d_loss32"], fx[f"{name}.g_loss32"] = r32[7].detach().numpy(), r32[9].detach().numpy()
        for tag, grads, grads32 in (("d", d_grads, r32[8]), ("g", g_grads, r32[10])):
            for k, v in grads.items():
                if v is None:
                    fx[f"{name}.{tag}_none.{k}"] = np.array(1)
                else:
                    fx[f"{name}.{tag}_sub.{k}"] = M.subsample(v)
                    fx[f"{name}.{tag}_norm.{k}"] = np.array(float(v.norm()))
                    if grads32[k] is not None:
                        fx[f"{name}.{tag}_sub32.{k}"] = M.subsample(grads32[k])
                        fx[f"{name}.{tag}_noise.{k}"] = np.array(M.rel(grads32[k], v))
        print(f"{name}: depth {depth}, alpha {alpha}, d_loss {float(d_loss):.6f}, g_loss {float(g_loss):.6f}; "
              f"weights identical, oracle matches", flush=True)
    np.savez_compressed(path, **fx)


if __name__ == "__main__":
    main()
