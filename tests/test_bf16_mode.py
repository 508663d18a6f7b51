"""The bf16 compute mode (``ops.set_precision('bf16')``): the matrix products of the predicate-free k = 3 conv
layers take their operands rounded to bf16 (one bf16 matrix instruction instead of six on three pieces), with fp32
accumulation and fp32 tensors in memory.

Two kinds of checks: (1) the mode does exactly what it says -- a conv in this mode equals, to fp32 summation
accuracy, the fp64 conv of the operands rounded to bf16; (2) the bounds of the mode end to end on the reference
fixtures, against the reference's fp64 run: forward (fakes, scores, losses) within 2e-2; the critic's summed
gradient within 1.5e-1 on the scale of its three terms (which cancel at initialisation); single critic terms and
generator gradients within 4e-1 norm-wise.  The worst tensor of every group is printed.

Why tenths and not a few times bf16's 2^-9: critic and generator are piecewise linear.  A pre-activation that the
bf16 products move across zero changes its LeakyReLU branch, and the gradient that passes through it by a factor of
five; with pre-activations off by ~0.4 % about 0.3 % of them do, which moves the gradient norm-wise by
sqrt(0.003 x 0.64) = 4 % per layer, summed in quadrature over the layers the gradient has passed: 16 for the
critic's first conv, 32 for the generator's (through the critic and back up the generator), and twice over for the
penalty term's double backward.  Measured at the final stage: critic terms 7e-2 .. 1.3e-1 (3.4e-1 for a bias that
the penalty reaches only through the minibatch statistic), summed critic gradient 5e-2 .. 1e-1, generator
1.6e-1 .. 2.6e-1; layers next to the loss stay below 1e-2.  This is a property of bf16 products on this network,
measured here with fp32 storage: rounding the stored activations too can only add to it.  The default mode
(fp32 products) is unaffected and keeps its 1e-4 bars."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import check_weights, load_fixture, rel, subsample

pytestmark = pytest.mark.gpu

FWD_BOUND, SUMMED_BOUND, GRAD_BOUND = 2e-2, 1.5e-1, 4e-1


@pytest.fixture()
def bf16_mode():
    from eeggan_amd import ops
    previous = ops.set_precision("bf16")
    try:
        yield
    finally:
        ops.set_precision(previous)


def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("shape", [(8, 64, 64, 1920), (64, 64, 64, 480), (4, 256, 256, 128), (16, 128, 256, 256)])
def test_products_are_those_of_bf16_operands(dev, bf16_mode, shape):
    from eeggan_amd import ops
    B, Cin, Cout, T = shape
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, T, generator=gen)
    w = torch.randn(Cout, Cin, 3, generator=gen)
    b = torch.randn(Cout, generator=gen)
    gy = torch.randn(B, Cout, T, generator=gen)
    s = 0.07
    xd, wd, bd, gyd = (t.to(dev) for t in (x, w, b, gy))
    assert ops.get_precision() == "bf16"
    # forward: weights are scaled before they are rounded
    y = ops.conv1d_raw(xd, wd, bd, None, in_scale=s)
    ref = F.conv1d(_bf(x), _bf(w * s), b.double(), padding=1)
    assert rel(y, ref) < 2e-6, rel(y, ref)
    exact = F.conv1d(x.double(), (w * s).double(), b.double(), padding=1)
    assert 1e-4 < rel(y, exact) < 1e-2  # the mode is in force (fp32 products would be at 1e-7), at bf16's accuracy
    # input gradient (transposed conv on flipped weights)
    gx = ops.conv1d_raw(gyd, wd, None, None, in_scale=s, transposed=True)
    ref = F.conv_transpose1d(_bf(gy), _bf(w * s), padding=1)
    assert rel(gx, ref) < 2e-6, rel(gx, ref)
    # weight gradient: both operands rounded, the scale applied to the sum
    gw, gb = ops.conv1d_wgrad_raw(xd, gyd, 3, in_scale=s)
    xp = F.pad(_bf(x), (1, 1))
    ref = s * torch.stack([torch.einsum("bot,bit->oi", _bf(gy), xp[:, :, j:j + T]) for j in range(3)], dim=2)
    assert rel(gw, ref) < 2e-6, rel(gw, ref)
    assert rel(gb, gy.double().sum(dim=(0, 2))) < 1e-5  # the bias gradient is summed in fp32 from the fp32 values


@pytest.mark.parametrize("shape", [(16, 257, 256, 32), (600, 257, 256, 32), (16, 256, 257, 32), (600, 256, 257, 32)])
def test_ragged_channel_layers_in_bf16_mode(dev, bf16_mode, shape):
    """The 256 + 1 channel layer and its input gradient in the bf16 mode: the 32 n channels' products are those of
    bf16 operands, the left-over channel's terms are added in float32 (unrounded)."""
    from eeggan_amd import ops
    B, Cin, Cout, T = shape
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T, generator=gen)
    w = torch.randn(Cout, Cin, 3, generator=gen)
    b = torch.randn(Cout, generator=gen)
    s = 0.07
    y = ops.conv1d_raw(x.to(dev), w.to(dev), b.to(dev), None, in_scale=s)
    main = Cin - Cin % 32
    ref = F.conv1d(_bf(x[:, :main]), _bf(w[:, :main] * s), b.double(), padding=1)
    if main < Cin:
        ref = ref + F.conv1d(x[:, main:].double(), (w[:, main:] * s).double(), None, padding=1)
    sel = sorted({0, 1, B // 2, B - 1})
    assert rel(y[sel], ref[sel]) < 2e-6, rel(y[sel], ref[sel])
    exact = F.conv1d(x.double(), (w * s).double(), b.double(), padding=1)
    assert 1e-4 < rel(y[sel], exact[sel]) < 1e-2  # (the matrix-core kernel with bf16 products ran, not the fp32 generic one)
    if Cin % 64:  # the weight gradient of such a layer: both operands rounded, the scale applied to the sum
        gy = torch.randn(B, Cout, T, generator=gen)
        # (through the grouped entry point, the one the training step submits its layers to)
        gw, gb = ops.conv1d_wgrad_group_raw([dict(x=x.to(dev), gy=gy.to(dev), K=3, in_scale=s)])[0]
        xp = F.pad(_bf(x), (1, 1))
        ref = s * torch.stack([torch.einsum("bot,bit->oi", _bf(gy), xp[:, :, j:j + T]) for j in range(3)], dim=2)
        assert rel(gw, ref) < 2e-6, rel(gw, ref)
        assert rel(gb, gy.double().sum(dim=(0, 2))) < 1e-5


def test_default_mode_is_untouched(dev):
    from eeggan_amd import ops
    assert ops.get_precision() == "fp32"
    gen = torch.Generator().manual_seed(3)
    x, w = torch.randn(8, 64, 960, generator=gen), torch.randn(64, 64, 3, generator=gen)
    y = ops.conv1d_raw(x.to(dev), w.to(dev), None, None, in_scale=0.07)
    assert rel(y, F.conv1d(x.double(), (w * 0.07).double(), padding=1)) < 1e-6
    with pytest.raises(ValueError):
        ops.set_precision("fp8")


@pytest.mark.parametrize("name", ["e2e_d4_a1", "e2e_d7_a1", "e2e_d7_a05"])
def test_stated_bounds_against_reference_fixture(dev, bf16_mode, name):
    run_bounds(dev, name, "bf16 mode", FWD_BOUND, SUMMED_BOUND, GRAD_BOUND)


BRANCH_BOUND, BRANCH_BIAS_BOUND = 2e-2, 1.5e-1


@pytest.mark.parametrize("depth,alpha,B", [(7, 1.0, 8), (7, 0.5, 8), (4, 1.0, 16)])
def test_gradients_under_reference_branches(dev, bf16_mode, depth, alpha, B):
    """What the mode costs when no LeakyReLU branch can flip: one critic iteration with every branch of the critic
    taken from the fp64 oracle (``test_e2e_gpu.critic_terms_under_reference_branches``).  Every term's gradient of
    every tensor within 2e-2 of its own norm -- the accumulated rounding of bf16 operands alone, twenty times tighter
    than the free-running bound above, and a bound a systematic kernel error would break.  One family of tensors gets
    1.5e-1: the penalty term's gradient of the BIASES below MinibatchStddev, which the penalty reaches only through
    the minibatch statistic -- a sum over all samples and positions of a second-order signal that cancels to a few
    per cent of its terms (measured 2e-2 .. 8.5e-2; the same tensors' weights stay below 2e-2)."""
    run_branch_bound(dev, depth, alpha, B, "bf16 mode", BRANCH_BOUND, BRANCH_BIAS_BOUND)


def run_branch_bound(dev, depth, alpha, B, label, bound, bias_bound):
    import test_e2e_gpu as E
    loss_err, gp_err, errs = E.critic_terms_under_reference_branches(dev, depth, alpha, B, 1373 + depth)
    cancelling = lambda k: k.startswith("gp.") and k.endswith(".bias")
    groups = {"penalty-term biases": {k: e for k, e in errs.items() if cancelling(k)},
              "all other term gradients": {k: e for k, e in errs.items() if not cancelling(k)}}
    print(f"{label}, depth {depth}, alpha {alpha}, under the oracle's branches: loss {loss_err:.1e}, penalty input "
          f"gradient {gp_err:.1e}")
    for name, group in groups.items():
        worst = max(group.items(), key=lambda kv: kv[1])
        print(f"    {name}: worst of {len(group)}: {worst[1]:.2e} ({worst[0]})")
    assert loss_err < bound and gp_err < bound
    for k, e in errs.items():
        limit = bias_bound if cancelling(k) else bound
        assert e < limit, f"{label}: critic term gradient {k} differs by {e:.3e} under the oracle's branches (bound {limit:.0e})"


def run_bounds(dev, name, label, FWD_BOUND, SUMMED_BOUND, GRAD_BOUND):
    """One critic and one generator iteration on a reference fixture in the mode that is in force, held to the
    given bounds against the reference's fp64 run (also used for the bf16 storage mode)."""
    from eeggan_amd import config, layers, losses, ops
    import test_e2e_gpu as E
    fx = load_fixture(name)
    depth, alpha, seed = int(fx["depth"]), float(fx["alpha"]), int(fx["seed"])
    layers.set_drop_mode("numpy")
    g, d = config.build_models(int(fx["model_seed"]))
    check_weights(fx, g, d)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    real, z, mix = (torch.from_numpy(fx[k]).to(dev) for k in ("real", "z", "mix"))
    np.random.seed(seed)
    with torch.no_grad():
        fake = g({"z": z})[0]["x"]
        score = d({"x": real})[0]
    worst = {"fwd": max(rel(fake, fx["ref64_fake"]), rel(score, fx["ref64_score_real"]))}
    assert worst["fwd"] < FWD_BOUND, worst

    # the three critic terms, each norm-wise; the summed gradient on the scale of its terms
    terms = E._term_grads(d, g, real, z, mix, seed)
    names = [k for k, _ in d.named_parameters()]
    worst["terms"] = 0.0
    worst_of = {}
    for term, grads in terms.items():
        for k, gr in zip(names, grads):
            key = f"d_term_{term}_sub.{k}"
            if key not in fx or gr is None or float(np.abs(fx[key]).max()) == 0.0:
                continue
            e = rel(subsample(gr), fx[key])
            worst["terms"] = max(worst["terms"], e)
            worst_of[f"critic term '{term}'"] = max(worst_of.get(f"critic term '{term}'", (0.0, "")), (e, k))
    np.random.seed(seed + 1)
    d_loss = losses.discriminator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", 1e-3, 10.0, 1.0, mixing=mix)
    d_loss.backward()
    assert abs(float(d_loss) - float(fx["ref64_d_loss"])) < FWD_BOUND * max(1.0, abs(float(fx["ref64_d_loss"])))
    worst["total"] = 0.0
    for k, p in d.named_parameters():
        if f"d_grads_none.{k}" in fx:
            continue
        present = [t for t in ("real", "fake", "gp") if f"d_term_{t}_norm.{k}" in fx]
        scale = sum(float(fx[f"d_term_{t}_norm.{k}"]) for t in present)
        ref = torch.from_numpy(fx[f"d_grads_sub.{k}"])
        frac = float(ref.norm()) / max(float(fx[f"d_grads_norm.{k}"]), 1e-300)
        e = float((torch.from_numpy(subsample(p.grad)) - ref).norm()) / (frac * scale + 1e-300)
        worst["total"] = max(worst["total"], e)
        worst_of["critic, summed"] = max(worst_of.get("critic, summed", (0.0, "")), (e, k))
    np.random.seed(seed + 2)
    g_loss = losses.generator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", False, 0.0)
    g_loss.backward()
    assert abs(float(g_loss) - float(fx["ref64_g_loss"])) < FWD_BOUND * max(1.0, abs(float(fx["ref64_g_loss"])))
    worst["g"] = 0.0
    for k, p in g.named_parameters():
        if f"g_grads_none.{k}" in fx:
            continue
        e = rel(subsample(p.grad), fx[f"g_grads_sub.{k}"])
        worst["g"] = max(worst["g"], e)
        worst_of["generator"] = max(worst_of.get("generator", (0.0, "")), (e, k))
    print(f"{name} {label}, worst deviations from the reference's fp64 run: forward {worst['fwd']:.2e}; " +
          "; ".join(f"{grp} {e:.2e} ({k})" for grp, (e, k) in worst_of.items()))
    for grp, (e, k) in worst_of.items():
        assert e < (SUMMED_BOUND if grp == "critic, summed" else GRAD_BOUND), f"{grp}, {k}: {e:.3e}"
