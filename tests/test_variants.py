"""Network and loss options beside the defaults: residual blocks (network.py:195-199, 217-218, 435-438, 447-448), the
generator's 'residual' / 'mean' signal modes (332-349), the critic's input at every stage (533-549), plain layers
(ReLU, no equalization, xavier init, to_rgb without tanh, latents not normalised, no MinibatchStddev), separable
layers, deep blocks, self attention (layers.py:41-64), batch norm in the generator plain and modulated by the latent
(layers.py:93-134), the 'sngan' / 'sagan' / 'biggan' forms of to_rgb (network.py:173-187), spectral norm (layers.py:170-171), fully convolutional networks (network.py:364-365, 416-418) and the generator loss's
random multiplier and feature-matching term (losses.py:46, 62-70).

Fixture ``variants.npz`` (``tools/make_golden_variants.py``): per case one critic and one generator loss evaluation
of the imported reference, 8 samples, in fp64 and fp32."""
import random

import numpy as np
import pytest
import torch

from helpers import load_fixture, oracle_params, rel, subsample

NOISE_FACTOR, COND_EPS, COND_TRIALS, COND_CAP = 2.0, 1e-5, 4, 5e-3

CASES = {
    "residual": dict(shared=dict(residual=True)),
    "residual_fade": dict(shared=dict(residual=True)),
    "rgb_residual": dict(generator=dict(rgb_generation_mode="residual")),
    "rgb_mean": dict(generator=dict(rgb_generation_mode="mean")),
    "rgb_mean_full": dict(generator=dict(rgb_generation_mode="mean")),
    "all_layers": dict(discriminator=dict(input_to_all_layers=True)),
    "plain": dict(shared=dict(act_alpha=0.0, equalized=False, init="xavier_uniform"),
                  generator=dict(no_tanh=True, normalize_latents=False), discriminator=dict(group_size=-1)),
    "separable": dict(shared=dict(separable=True)),
    "deep": dict(shared=dict(deep=True)),
    "g_loss_opts": dict(random_multiply=True, feature_matching_lambda=1.0),
    "g_loss_opts_rel": dict(random_multiply=True, feature_matching_lambda=0.5, loss_type="rasgan"),
    "attention": dict(shared=dict(self_attention_layers=[0, 1]), gamma=0.4),
    "attention_narrow": dict(shared=dict(self_attention_layers=[3]), gamma=-0.8),
    "batch_norm": dict(generator=dict(act_norm="batch")),
    "batch_norm_z": dict(generator=dict(act_norm="batch", z_to_bn=True)),
    "split_z": dict(generator=dict(act_norm="batch", z_to_bn=True, split_z=True)),
    "to_rgb_sngan": dict(generator=dict(to_rgb_mode="sngan")),
    "to_rgb_sagan": dict(generator=dict(to_rgb_mode="sagan")),
    "to_rgb_biggan": dict(generator=dict(to_rgb_mode="biggan")),
    "to_rgb_biggan_plain": dict(generator=dict(to_rgb_mode="biggan", no_tanh=True)),
    "spectral_d": dict(shared=dict(equalized=False), discriminator=dict(spectral=True)),
    "spectral_g": dict(shared=dict(equalized=False), generator=dict(spectral=True)),
    "spectral_both": dict(shared=dict(equalized=False, self_attention_layers=[1]), generator=dict(spectral=True),
                          discriminator=dict(spectral=True), gamma=0.3),
    "conv_only_d": dict(shared=dict(residual=True), discriminator=dict(conv_only=True)),
    "conv_only": dict(generator=dict(conv_only=True), discriminator=dict(conv_only=True)),
    "d_batch_norm": dict(discriminator=dict(act_norm="batch")),
    "d_pixel_norm": dict(discriminator=dict(act_norm="pixel")),
    "per_channel_noise": dict(generator=dict(per_channel_noise=True), noise_weights=True),
    "per_channel_noise_deep": dict(shared=dict(deep=True), generator=dict(per_channel_noise=True, act_norm="batch"),
                                   noise_weights=True),
    "conv_only_z_to_bn": dict(generator=dict(conv_only=True, act_norm="batch", z_to_bn=True),
                              discriminator=dict(conv_only=True)),
    "temporal_groups": dict(discriminator=dict(temporal_groups_per_window=2)),
    "temporal_groups_ragged": dict(discriminator=dict(temporal_groups_per_window=5, group_size=2)),
}


def _case(fx, name):
    return {k[len(name) + 1:]: fx[k] for k in fx if k.startswith(name + ".")}


def _models(case):
    from eeggan_amd import config
    g, d = config.build_models(config.RANDOM_SEED, shared=case.get("shared"), generator=case.get("generator"),
                               discriminator=case.get("discriminator"))
    if "gamma" in case:  # the scalars of the attention layers, zero at construction (the fixture script's values)
        with torch.no_grad():
            for net, sign in ((g, 1.0), (d, -1.0)):
                for j, layer in enumerate(net.dummy):
                    layer.gamma.fill_(sign * case["gamma"] * (1.0 + 0.25 * j))
    if case.get("noise_weights"):  # the per-channel noise weights, zero at construction (the fixture script's values)
        with torch.no_grad():
            noises = [p for k, p in g.named_parameters() if ".noises." in k]
            for j, p in enumerate(noises):
                p.copy_(torch.linspace(0.6, 1.4, p.shape[1]).view(1, -1, 1) * (1.0 if j % 2 == 0 else -1.0))
    return g, d


def _oracle_config(case):
    from oracle import ref_cpu as O
    kw = {}
    for src, keys in ((case.get("shared", {}), ("residual", "act_alpha", "deep", "self_attention_layers")),
                      (case.get("generator", {}), ("rgb_generation_mode", "no_tanh", "normalize_latents", "z_to_bn", "split_z",
                                                   "to_rgb_mode")),
                      (case.get("discriminator", {}), ("input_to_all_layers", "group_size", "temporal_groups_per_window"))):
        kw.update({k: src[k] for k in keys if k in src})
    if "act_norm" in case.get("generator", {}):
        kw["g_act_norm"] = case["generator"]["act_norm"]
    kw["g_conv_only"] = bool(case.get("generator", {}).get("conv_only"))
    kw["d_conv_only"] = bool(case.get("discriminator", {}).get("conv_only"))
    if "act_norm" in case.get("discriminator", {}):
        kw["d_act_norm"] = case["discriminator"]["act_norm"]
    return O.NetConfig(**kw)


def _close(a, b):
    """1e-9 relative, or both round-off noise (a gradient that cancels exactly in the mathematics)."""
    return rel(a, b) < 1e-9 or float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))) < 1e-12


def test_fixture_holds_every_case():
    assert list(load_fixture("variants")["names"]) == list(CASES)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference(name):
    from oracle import ref_cpu as O
    fx = load_fixture("variants")
    c, case = _case(fx, name), CASES[name]
    depth, alpha, seed = int(c["depth"]), float(c["alpha"]), int(c["seed"])
    loss_type = case.get("loss_type", "wgan_gp")
    g, d = _models(case)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(True)
    real, z, mix = (torch.from_numpy(c[k]).double() for k in ("real", "z", "mix"))
    cfg = _oracle_config(case)
    np.random.seed(seed)
    torch.manual_seed(seed + 10)  # (host draws inside the networks: per-channel noise)
    src = O.DropSource("numpy")
    with torch.no_grad():
        assert rel(O.generator_forward(gp, gs, z, depth, alpha, cfg, src), c["fake"]) < 1e-9
        assert rel(O.discriminator_forward(dp, ds, real, depth, alpha, cfg, src)[0], c["score"]) < 1e-9
    np.random.seed(seed + 1)
    torch.manual_seed(seed + 11)
    src = O.DropSource("numpy")
    loss, _ = O.d_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, mix, src, src)
    assert abs(float(loss) - float(c["d_loss"])) < 1e-9 * max(1.0, abs(float(c["d_loss"])))
    for k, gr in zip(dp.keys(), torch.autograd.grad(loss, list(dp.values()), allow_unused=True)):
        if gr is None and k not in dict(d.named_parameters()):
            continue  # (a buffer of the module: running statistics)
        assert (gr is None) == (f"d_none.{k}" in c) and (gr is None or _close(subsample(gr), c[f"d_sub.{k}"])), k
    np.random.seed(seed + 2)
    random.seed(seed + 3)
    torch.manual_seed(seed + 12)
    src = O.DropSource("numpy")
    loss, _ = O.g_loss_any(loss_type, dp, ds, gp, gs, real, z, depth, alpha, cfg, src, src,
                           random_multiply=case.get("random_multiply", False),
                           feature_matching_lambda=case.get("feature_matching_lambda", 0.0))
    assert abs(float(loss) - float(c["g_loss"])) < 1e-9 * max(1.0, abs(float(c["g_loss"])))
    for k, gr in zip(gp.keys(), torch.autograd.grad(loss, list(gp.values()), allow_unused=True)):
        if gr is None and k not in dict(g.named_parameters()):
            continue
        assert (gr is None) == (f"g_none.{k}" in c) and (gr is None or _close(subsample(gr), c[f"g_sub.{k}"])), k


def test_options_outside_the_supported_set_still_raise():
    from eeggan_amd import config
    for kw in (dict(discriminator=dict(sinc=True)),
               dict(discriminator=dict(act_norm="layer")), dict(shared=dict(num_classes=10))):  # (classes need batch norms)
        with pytest.raises(NotImplementedError):
            config.build_models(1, **kw)
    with pytest.raises(ValueError):
        config.build_models(1, generator=dict(rgb_generation_mode="median"))
    with pytest.raises(ValueError):
        config.build_models(1, generator=dict(to_rgb_mode="stylegan"))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 64, 256, 3, 1), (3, 256, 1, 32, 31), (5, 256, 32, 32, 0), (2, 37, 50, 5, 0)])
def test_depthwise_conv_to_second_order(dev, shape):
    """The depthwise kernels against torch's grouped conv in fp64: forward, input / weight / bias gradients, and the
    gradient of a function of the input gradient (what the penalty's double backward asks for)."""
    import torch.nn.functional as F
    from eeggan_amd import ops
    B, C, T, K, pad = shape
    gen = torch.Generator().manual_seed(K)
    x, w, b = torch.randn(B, C, T, generator=gen), torch.randn(C, 1, K, generator=gen), torch.randn(C, generator=gen)
    s = 0.3
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv1d(xd * s, wd, bd, padding=pad, groups=C)
    gy = torch.randn(yd.shape, generator=gen)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    yg = ops.dwconv1d(xg, wg, bg, s, pad)
    assert rel(yg, yd) < 2e-6
    (gxd,) = torch.autograd.grad(yd, xd, gy.double(), create_graph=True)
    (gxg,) = torch.autograd.grad(yg, xg, gy.to(dev), create_graph=True)
    assert rel(gxg, gxd) < 2e-6
    want = torch.autograd.grad((gxd ** 2).sum() + (yd * gy.double()).sum(), [xd, wd, bd], allow_unused=True)
    got = torch.autograd.grad((gxg ** 2).sum() + (yg * gy.to(dev)).sum(), [xg, wg, bg], allow_unused=True)
    for a, e, what in zip(got, want, ("input", "weight", "bias")):
        assert rel(a, e) < 1e-5, (what, rel(a, e))


@pytest.mark.gpu
def test_attention_pieces_to_second_order(dev):
    """Batched products with every transpose combination, the softmax over dim 1, max pooling and the product with a
    scalar parameter, each against torch in fp64: values, first gradients, and the gradients of a function of the
    first gradients (the penalty's double backward)."""
    import torch.nn.functional as F
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(21)

    def both(*shapes):
        ts = [torch.randn(tuple(s), generator=gen) for s in shapes]
        return [t.double().requires_grad_(True) for t in ts], [t.to(dev).requires_grad_(True) for t in ts]

    def compare(fd, fg, ins_d, ins_g, tol=2e-5):
        yd, yg = fd(*ins_d), fg(*ins_g)
        assert rel(yg, yd) < tol, rel(yg, yd)
        cot = torch.randn(yd.shape, generator=gen)
        g1d = torch.autograd.grad(yd, ins_d, cot.double(), create_graph=True)
        g1g = torch.autograd.grad(yg, ins_g, cot.to(dev), create_graph=True)
        for a, e in zip(g1g, g1d):
            assert rel(a, e) < tol, rel(a, e)
        fun = lambda gs, y, c: sum((g * g).sum() for g in gs) + (y * c).sum()
        g2d = torch.autograd.grad(fun(g1d, yd, cot.double()), ins_d, allow_unused=True)
        g2g = torch.autograd.grad(fun(g1g, yg, cot.to(dev)), ins_g, allow_unused=True)
        for a, e in zip(g2g, g2d):
            assert rel(a, e) < 5 * tol, rel(a, e)

    for ta, tb, (M, N, K) in ((False, False, (32, 480, 120)), (True, False, (120, 480, 8)), (False, True, (33, 70, 481)),
                              (True, True, (65, 17, 5))):
        sa, sb = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
        ins_d, ins_g = both((3,) + sa, (3,) + sb)
        op = lambda t, flag: t.transpose(1, 2) if flag else t
        compare(lambda a, b: 0.7 * torch.bmm(op(a, ta), op(b, tb)), lambda a, b: ops.bmm(a, b, ta, tb, 0.7), ins_d, ins_g)
    ins_d, ins_g = both((3, 120, 480))
    compare(lambda s: torch.softmax(2.0 * s, dim=1), lambda s: ops.softmax_dim1(ops.axpby(s, 2.0, None, 0.0)), ins_d, ins_g)
    ins_d, ins_g = both((3, 16, 480))
    compare(lambda x: F.max_pool1d(x, 4), lambda x: ops.maxpool(x, 4), ins_d, ins_g, tol=1e-6)
    ins_d, ins_g = both((3, 16, 100), ())
    compare(lambda f, s: f * s, lambda f, s: ops.mul_param(f, s), ins_d, ins_g)


@pytest.mark.gpu
def test_attention_layer_against_torch(dev):
    """The SelfAttention module on a 64-channel and a 256-channel stage against the same arithmetic in torch fp64."""
    import torch.nn.functional as F
    from eeggan_amd.layers import SelfAttention
    for C, T, B in ((64, 512, 3), (256, 32, 4)):
        torch.manual_seed(C)
        layer = SelfAttention(C, False, "kaiming_normal")
        with torch.no_grad():
            layer.gamma.fill_(0.9)
        x = torch.randn(B, C, T)
        w = {k: getattr(layer, k).conv.conv.weight.detach().double() for k in ("query_conv", "key_conv", "value_conv", "final_conv")}
        xd = x.double().requires_grad_(True)
        q, k, v = F.conv1d(xd, w["query_conv"]), F.max_pool1d(F.conv1d(xd, w["key_conv"]), 4), F.max_pool1d(F.conv1d(xd, w["value_conv"]), 4)
        p = torch.softmax(torch.bmm(k.transpose(1, 2), q), dim=1)
        want = 0.9 * F.conv1d(torch.bmm(v, p), w["final_conv"]) + xd
        layer = layer.to(dev)
        xg = x.to(dev).requires_grad_(True)
        got, amap = layer(xg)
        assert rel(got, want) < 1e-5 and rel(amap, p) < 1e-5
        cot = torch.randn(B, C, T)
        (gd,) = torch.autograd.grad(want, xd, cot.double())
        (gg,) = torch.autograd.grad(got, xg, cot.to(dev))
        assert rel(gg, gd) < 1e-5, rel(gg, gd)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 64, 480), (5, 37, 50), (4, 256, 32)])
def test_batch_norm_layer_against_torch(dev, shape):
    """ConditionalBatchNorm in 'BN' mode against nn.BatchNorm1d in fp64: training-mode values, gradients of input and
    affine parameters, running statistics after two calls, and evaluation mode; then the fused epilogue
    (modulation, LeakyReLU, multipliers) against the same arithmetic written out."""
    import torch.nn.functional as F
    from eeggan_amd import ops
    from eeggan_amd.layers import ConditionalBatchNorm
    B, C, T = shape
    gen = torch.Generator().manual_seed(C)
    x1, x2 = (torch.randn(B, C, T, generator=gen) * 1.7 + 0.4 for _ in range(2))
    cot = torch.randn(B, C, T, generator=gen)
    ref = torch.nn.BatchNorm1d(C).double()
    mine = ConditionalBatchNorm(C, 0, 0, False).to(dev)
    with torch.no_grad():
        w, b = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
        ref.weight.copy_(w), ref.bias.copy_(b)
        mine.normalizer.weight.copy_(w), mine.normalizer.bias.copy_(b)
    xd, xg = x1.double().requires_grad_(True), x1.to(dev).requires_grad_(True)
    yd, yg = ref(xd), mine(xg)
    assert rel(yg, yd) < 1e-5
    want = torch.autograd.grad(yd, [xd, ref.weight, ref.bias], cot.double())
    got = torch.autograd.grad(yg, [xg, mine.normalizer.weight, mine.normalizer.bias], cot.to(dev))
    for a, e, what in zip(got, want, ("input", "weight", "bias")):
        assert rel(a, e) < 2e-5, (what, rel(a, e))
    ref(x2.double()), mine(x2.to(dev))
    assert rel(mine.normalizer.running_mean, ref.running_mean) < 1e-5 and rel(mine.normalizer.running_var, ref.running_var) < 1e-5
    assert int(mine.normalizer.num_batches_tracked) == 2
    ref.eval(), mine.eval()
    xd, xg = x1.double().requires_grad_(True), x1.to(dev).requires_grad_(True)
    yd, yg = ref(xd), mine(xg)
    assert rel(yg, yd) < 1e-5
    assert rel(torch.autograd.grad(yg, xg, cot.to(dev))[0], torch.autograd.grad(yd, xd, cot.double())[0]) < 1e-5
    # the fused epilogue with per-sample modulation
    gm, bm, dr = torch.randn(B, C, generator=gen) * 0.3, torch.randn(B, C, generator=gen), torch.rand(B, C, generator=gen) + 0.5
    ins_d = [t.double().requires_grad_(True) for t in (x1, gm, bm)]
    n = F.batch_norm(ins_d[0], None, None, None, None, training=True, eps=1e-5)
    o = n + ins_d[1][:, :, None] * n + ins_d[2][:, :, None]
    yd = F.leaky_relu(o, 0.2) * dr.double()[:, :, None]
    ins_g = [t.to(dev).requires_grad_(True) for t in (x1, gm, bm)]
    yg = ops.bn_act(ins_g[0], dr.to(dev), None, None, ins_g[1], ins_g[2], None, None, 0.2, 1e-5, 0.1, True)
    assert rel(yg, yd) < 1e-5
    for a, e, what in zip(torch.autograd.grad(yg, ins_g, cot.to(dev)), torch.autograd.grad(yd, ins_d, cot.double()),
                          ("input", "scale modulation", "shift modulation")):
        assert rel(a, e) < 2e-5, (what, rel(a, e))


@pytest.mark.gpu
def test_feature_match_kernels(dev):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(4)
    for B, F in ((8, 256), (5, 37), (32, 1000)):
        r, f = torch.randn(B, F, 1, generator=gen), torch.randn(B, F, 1, generator=gen)
        fd = f.double().requires_grad_(True)
        diff = r.double().mean(dim=0) - fd.mean(dim=0)
        want = (diff * diff).mean()
        (wg,) = torch.autograd.grad(want, fd)
        x = f.to(dev).requires_grad_(True)
        got = ops.feature_match(r.to(dev), x)
        (gg,) = torch.autograd.grad(got, x, torch.tensor(0.6, device=dev))
        assert abs(float(got) - float(want)) < 1e-5 * float(want) and rel(gg, 0.6 * wg) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_against_reference_fixture(dev, name):
    """Forward (fakes, scores) and loss values 1e-4; gradients per tensor within the gate of the other end-to-end
    fixtures: max(1e-4, 2 x the reference's own fp32-vs-fp64 distance, min(2 x conditioning, 5e-3))."""
    from eeggan_amd import layers, losses
    fx = load_fixture("variants")
    c, case = _case(fx, name), CASES[name]
    depth, alpha, seed = int(c["depth"]), float(c["alpha"]), int(c["seed"])
    loss_type = case.get("loss_type", "wgan_gp")
    rm, fm = case.get("random_multiply", False), case.get("feature_matching_lambda", 0.0)
    layers.set_drop_mode("numpy")
    g, d = _models(case)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    real, z, mix = (torch.from_numpy(c[k]).to(dev) for k in ("real", "z", "mix"))
    np.random.seed(seed)
    torch.manual_seed(seed + 10)
    with torch.no_grad():
        assert rel(g({"z": z})[0]["x"], c["fake"]) < 1e-4
        assert rel(d({"x": real})[0], c["score"]) < 1e-4

    # (state that moves with every forward -- the vectors of a spectral norm -- starts every evaluation from here)
    moving = [(b, b.clone()) for net in (g, d) for b in net.buffers()]

    def evaluate(jitter=lambda t: t):
        with torch.no_grad():
            for b, saved in moving:
                b.copy_(saved)
        np.random.seed(seed + 1)
        torch.manual_seed(seed + 11)
        d_loss = losses.discriminator_loss(d, g, {"x": jitter(real)}, {"z": jitter(z)}, loss_type, 1e-3, 10.0, 1.0,
                                           mixing=mix)
        d_loss.backward()
        d_grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in d.named_parameters()}
        np.random.seed(seed + 2)
        random.seed(seed + 3)
        torch.manual_seed(seed + 12)
        g_loss = losses.generator_loss(d, g, {"x": jitter(real)}, {"z": jitter(z)}, loss_type, rm, fm)
        g_loss.backward()
        g_grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in g.named_parameters()}
        return d_loss, g_loss, {"d": d_grads, "g": g_grads}

    d_loss, g_loss, grads = evaluate()
    assert abs(float(d_loss) - float(c["d_loss"])) < 1e-4 * max(1.0, abs(float(c["d_loss"])))
    assert abs(float(g_loss) - float(c["g_loss"])) < 1e-4 * max(1.0, abs(float(c["g_loss"])))
    cond = {"d": {}, "g": {}}
    gen = torch.Generator(device=dev).manual_seed(seed)
    for _ in range(COND_TRIALS):
        _, _, pert = evaluate(lambda t: t * (1.0 + COND_EPS * torch.randn(t.shape, device=dev, generator=gen)))
        for tag in ("d", "g"):
            for k, a in grads[tag].items():
                b = pert[tag][k]
                if a is not None and b is not None and float(a.norm()) > 0.0:
                    cond[tag][k] = max(cond[tag].get(k, 0.0), float((a - b).norm()) / float(a.norm()))
    worst, failures = 0.0, []
    for tag in ("d", "g"):
        for k, got in grads[tag].items():
            if f"{tag}_none.{k}" in c:
                assert got is None, f"{tag}.{k}: the reference leaves this gradient None"
                continue
            assert got is not None, f"{tag}.{k}: missing gradient"
            if f"{tag}_norm.{k}" in c and float(c[f"{tag}_norm.{k}"]) < 1e-12:
                # (cancels exactly in the mathematics -- the output layer's bias under a relativistic loss:
                # round-off in every implementation)
                assert float(got.norm()) < 1e-6, f"{tag}.{k}: {float(got.norm()):.1e} where the gradient cancels"
                continue
            noise = float(c[f"{tag}_noise.{k}"]) if f"{tag}_noise.{k}" in c else 0.0
            tol = max(1e-4, NOISE_FACTOR * noise, min(NOISE_FACTOR * cond[tag].get(k, 0.0), COND_CAP))
            e = rel(subsample(got), c[f"{tag}_sub.{k}"])
            if f"{tag}_sub32.{k}" in c:
                e = min(e, rel(subsample(got), c[f"{tag}_sub32.{k}"]))
            worst = max(worst, e / tol)
            if not e < tol:
                failures.append(f"{tag}.{k}: {e:.3e} (tolerance {tol:.3e}, reference noise {noise:.1e}, "
                                f"conditioning {cond[tag].get(k, 0.0):.1e})")
    print(f"{name}: worst deviation = {worst:.2f} of its gate")
    assert not failures, f"{name}: " + "; ".join(failures)


@pytest.mark.gpu
def test_graph_trainer_with_residual_blocks(dev):
    """The graph trainer (stacked pass, two chains, grouped weight gradients) on networks with residual blocks and
    the summed-signal generator follows the eager trainer."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(shared=dict(residual=True), generator=dict(rgb_generation_mode="residual")))
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        # (the graph trainer evaluates reals and fakes in one stacked pass: other tile shapes, other last bits; with the
        # skips' additions the one-element bias of the output layer comes out at 1.2e-5 of its scale)
        E._same_trajectory(base, graphed, "graph replay, residual blocks", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
def test_graph_trainer_with_attention(dev):
    """The graph trainer on networks with a self-attention layer (its scalar set away from zero, so that the layer
    acts) follows the eager trainer: the layer's kernels under capture, its scalar read from device memory by the
    replayed kernels as Adam moves it."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer

    def prepare(g, d):
        with torch.no_grad():
            g.dummy[0].gamma.fill_(0.5)
            d.dummy[0].gamma.fill_(-0.5)

    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(shared=dict(self_attention_layers=[0])), prepare=prepare)
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        E._same_trajectory(base, graphed, "graph replay, self attention", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
def test_graph_trainer_with_batch_norm(dev):
    """A generator with batch norm layers modulated by the latent under the graph trainer: the fakes are made per
    critic iteration (batch statistics couple the samples of a pass, so one pass over all iterations' latents would
    change them), running statistics move with the replays, and the trajectory is the eager trainer's."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(generator=dict(act_norm="batch", z_to_bn=True)))
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        E._same_trajectory(base, graphed, "graph replay, batch norm", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sagan", "biggan"])
def test_graph_trainer_with_other_to_rgb(dev, mode):
    """The 'sagan' (two layers) and 'biggan' (batch norm, ReLU, k = 3 conv) forms of to_rgb under the graph trainer:
    the trajectory is the eager trainer's (with a batch norm the fakes are made per critic iteration)."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(generator=dict(to_rgb_mode=mode)))
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        E._same_trajectory(base, graphed, f"graph replay, to_rgb {mode}", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 257, 3), (64, 5, 1), (1, 256, 1), (256, 256, 32), (64, 1, 3), (5, 64, 3)])
def test_spectral_norm_against_torch(dev, shape):
    """The spectral normalisation of a weight (one power iteration in place, sigma, the normalised weight and its
    gradient with the vectors as constants) against float64 torch, 2e-5; evaluation mode leaves the vectors alone."""
    import torch.nn.functional as F
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(shape, generator=gen)
    rows, cols = shape[0], shape[1] * shape[2]
    u = F.normalize(torch.randn(rows, generator=gen), dim=0)
    v = F.normalize(torch.randn(cols, generator=gen), dim=0)
    g = torch.randn(shape, generator=gen)
    w64 = w.double().requires_grad_(True)
    wm = w64.reshape(rows, cols)
    with torch.no_grad():
        v1 = F.normalize(wm.t() @ u.double(), dim=0, eps=1e-12)
        u1 = F.normalize(wm @ v1, dim=0, eps=1e-12)
    sigma = torch.dot(u1, wm @ v1)
    ref = w64 / sigma
    (gref,) = torch.autograd.grad(ref, w64, g.double())
    wd, ud, vd = w.to(dev).requires_grad_(True), u.to(dev), v.to(dev)
    w_sn, sg = ops.spectral_norm(wd, ud, vd, 1e-12, True)
    assert rel(w_sn, ref) < 2e-5 and rel(sg, sigma.reshape(1)) < 2e-5
    assert rel(ud, u1) < 2e-5 and rel(vd, v1) < 2e-5
    (gw,) = torch.autograd.grad(w_sn, wd, g.to(dev))
    assert rel(gw, gref) < 2e-5
    # evaluation mode: sigma from the stored vectors
    u2, v2 = ud.clone(), vd.clone()
    w_ev, sg_ev = ops.spectral_norm(wd, ud, vd, 1e-12, False)
    assert torch.equal(ud, u2) and torch.equal(vd, v2)
    assert rel(sg_ev, torch.dot(u1, wm.detach() @ v1).reshape(1)) < 2e-5 and rel(w_ev, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["discriminator", "generator"])
def test_graph_trainer_with_spectral_norm(dev, where):
    """Spectral norm under the graph trainer: the power-iteration vectors move inside the captured graphs exactly as
    in the eager trainer (the critic iteration stays one chain -- two passes side by side would race on the vectors
    -- and a spectral generator makes its fakes per critic iteration)."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models={"shared": dict(equalized=False), where: dict(spectral=True)})
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        E._same_trajectory(base, graphed, f"graph replay, spectral {where}", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
def test_graph_trainer_with_conv_only_critic(dev):
    """A fully convolutional critic (k-tap last layer with its skip, score = mean over time) under the graph trainer:
    the trajectory is the eager trainer's."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(shared=dict(residual=True), discriminator=dict(conv_only=True)))
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        E._same_trajectory(base, graphed, "graph replay, conv_only critic", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
@pytest.mark.parametrize("train", [True, False])
def test_batch_norm_to_second_order(dev, train):
    """The critic's batch norm (a chain of multilinear pieces) with its LeakyReLU and multipliers against float64
    torch: values, first derivatives, and the derivatives of a penalty on the input gradient (the second order the
    gradient penalty asks for); running statistics as nn.BatchNorm1d keeps them."""
    import torch.nn.functional as F
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(5 + train)
    B, C, T = 6, 20, 48
    c = torch.randn(B, C, T, generator=gen) * 1.7 + 0.3
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    drop = torch.rand(B, C, generator=gen) + 0.5
    rm, rv = torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5
    q = torch.randn(B, C, T, generator=gen)

    def run(dt, device, fn):
        cc = c.to(device, dt).requires_grad_(True)
        ga, be = gamma.to(device, dt).requires_grad_(True), beta.to(device, dt).requires_grad_(True)
        m, v = rm.to(device, dt).clone(), rv.to(device, dt).clone()
        y = fn(cc, ga, be, m, v, drop.to(device, dt))
        (gc,) = torch.autograd.grad(y, cc, q.to(device, dt), create_graph=True)
        pen = ((gc * gc).sum(dim=(1, 2)).sqrt() - 1.0).pow(2).sum() + (y * y).mean()
        grads = torch.autograd.grad(pen, [cc, ga, be])
        return [y, gc, *grads, m, v]

    ref = run(torch.float64, "cpu", lambda cc, ga, be, m, v, dr: F.leaky_relu(
        F.batch_norm(cc, m, v, ga, be, training=train, momentum=0.1, eps=1e-5), 0.2) * dr[:, :, None])
    got = run(torch.float32, dev, lambda cc, ga, be, m, v, dr: ops.lrelu_drop(
        ops.batch_norm_twice(cc, ga, be, m, v, 1e-5, 0.1, train), dr, 0.2))
    for name, a, b in zip(("y", "input gradient", "d pen / d c", "d pen / d gamma", "d pen / d beta", "running mean",
                           "running variance"), got, ref):
        assert rel(a, b) < 5e-5, (name, rel(a, b))


@pytest.mark.gpu
def test_graph_trainer_with_batch_norm_critic(dev):
    """Batch norm in the critic's layers under the graph trainer: reals, fakes and interpolates stay batches of their
    own (one chain, no stacked pass), and the trajectory is the eager trainer's."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        kw = dict(models=dict(discriminator=dict(act_norm="batch")))
        seen = {}
        base = E._two_steps(dev, Trainer, lr=0.0, **kw), E._two_steps(dev, Trainer, **kw)
        graphed = E._two_steps(dev, GraphedTrainer, lr=0.0, seen=seen, **kw), E._two_steps(dev, GraphedTrainer, **kw)
        assert seen["critic"] != "_SplitCriticIteration"
        E._same_trajectory(base, graphed, "graph replay, batch norm critic", moment_tol=5e-5)
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
def test_pixel_norm_to_second_order(dev):
    """The critic's pixel norm (multilinear pieces) against float64 torch: values, first derivatives and the
    derivatives of a penalty on the input gradient."""
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(11)
    B, C, T = 5, 24, 40
    a = torch.randn(B, C, T, generator=gen)
    q = torch.randn(B, C, T, generator=gen)

    def run(dt, device, fn):
        aa = a.to(device, dt).requires_grad_(True)
        y = fn(aa)
        (ga,) = torch.autograd.grad(y, aa, q.to(device, dt), create_graph=True)
        pen = ((ga * ga).sum(dim=(1, 2)).sqrt() - 1.0).pow(2).sum() + (y * y * y).mean()
        (gp,) = torch.autograd.grad(pen, aa)
        return y, ga, gp

    ref = run(torch.float64, "cpu", lambda h: h * torch.rsqrt((h * h).mean(dim=1, keepdim=True) + 1e-8))
    got = run(torch.float32, dev, ops.pixel_norm_twice)
    for name, x, r in zip(("y", "input gradient", "d pen / d a"), got, ref):
        assert rel(x, r) < 5e-5, (name, rel(x, r))
