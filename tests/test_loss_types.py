"""The reference's other loss types (losses.py:40-60, 82-101: hinge, rsgan, rasgan, rahinge, wgan_theirs), the
one-sided gradient penalty and the running average of the generator's weights (plugins.py:305-307).

Fixture ``loss_types.npz`` (``tools/make_golden.py``): the reference's critic and generator losses and all
parameter gradients at depth 1, 8 samples, in fp64 and fp32, per loss type; ``wgan_theirs`` once with its penalty
gain above zero and once at zero; ``wgan_gp`` also with ``grad_lambda = 0`` (no penalty, losses.py:106)."""
import numpy as np
import pytest
import torch

from helpers import load_fixture, oracle_params, rel, subsample

NOISE_FACTOR = 2.0
TYPES = {"hinge": "hinge", "rsgan": "rsgan", "rasgan": "rasgan", "rahinge": "rahinge", "wgan_gp": "wgan_gp",
         "wgan_theirs_on": "wgan_theirs", "wgan_theirs_off": "wgan_theirs", "wgan_gp_nopenalty": "wgan_gp"}


def _case(fx, name):
    return {k[len(name) + 1:]: fx[k] for k in fx if k.startswith(name + ".")}


@pytest.mark.parametrize("name", list(TYPES))
def test_oracle_matches_reference(name):
    from eeggan_amd import config
    from oracle import ref_cpu as O
    fx = load_fixture("loss_types")
    c = _case(fx, name)
    depth, alpha, seed = int(fx["depth"]), float(fx["alpha"]), int(c["seed"])
    g, d = config.build_models(int(fx["model_seed"]))
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(True)
    real, z, mix = (torch.from_numpy(c[k]).double() for k in ("real", "z", "mix"))
    cfg = O.NetConfig()
    np.random.seed(seed + 1)
    src = O.DropSource("numpy")
    loss, _ = O.d_loss_any(TYPES[name], dp, ds, gp, gs, real, z, depth, alpha, cfg, mix, src, src,
                           grad_lambda=float(c["grad_lambda"]))
    assert abs(float(loss) - float(c["d_loss"])) < 1e-9 * max(1.0, abs(float(c["d_loss"])))
    for k, gr in zip(dp.keys(), torch.autograd.grad(loss, list(dp.values()), allow_unused=True)):
        if f"d_none.{k}" in c:
            assert gr is None
        else:
            assert rel(subsample(gr), c[f"d_sub.{k}"]) < 1e-9, k
    np.random.seed(seed + 2)
    src = O.DropSource("numpy")
    loss, _ = O.g_loss_any(TYPES[name], dp, ds, gp, gs, real, z, depth, alpha, cfg, src, src)
    assert abs(float(loss) - float(c["g_loss"])) < 1e-9 * max(1.0, abs(float(c["g_loss"])))
    for k, gr in zip(gp.keys(), torch.autograd.grad(loss, list(gp.values()), allow_unused=True)):
        if f"g_none.{k}" in c:
            assert gr is None
        else:
            assert rel(subsample(gr), c[f"g_sub.{k}"]) < 1e-9, k


def test_invalid_loss_type_is_rejected():
    from eeggan_amd import losses
    with pytest.raises(ValueError):
        losses.discriminator_loss(None, None, None, None, "lsgan", 1e-3, 10.0, 1.0)
    with pytest.raises(ValueError):
        losses.generator_loss(None, None, None, None, "lsgan", False)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [5, 37, 700])
def test_score_loss_kernels(dev, B):
    from eeggan_amd import ops
    from oracle import ref_cpu as O
    gen = torch.Generator().manual_seed(B)
    for shift in (0.4, -0.4):
        a = (torch.randn(B, generator=gen) * 1.5 + shift)
        b = torch.randn(B, generator=gen) * 1.5
        for kind in ("hinge", "rsgan", "rasgan", "rahinge", "wgan_theirs"):
            ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
            want, gain = O.critic_score_loss(kind, ad, bd, 1e-3)
            wa, wb = torch.autograd.grad(want, [ad, bd])
            x, y = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
            got = ops.score_loss(kind, x, y, 1e-3)
            up = torch.tensor(0.7, device=dev)
            ga, gb = torch.autograd.grad(got, [x, y], up)
            assert abs(float(got) - float(want)) < 1e-5 * max(1.0, abs(float(want))), kind
            assert rel(ga, 0.7 * wa) < 1e-5 and rel(gb, 0.7 * wb) < 1e-5, kind
            if gain is not None:
                gd = ops.score_loss("gain", x, y)
                assert abs(float(gd) - float(gain)) < 1e-6
                if float(gain) > 0:
                    ha, hb = torch.autograd.grad(gain, [ad, bd])
                    ka, kb = torch.autograd.grad(gd, [x, y])
                    assert rel(ka, ha) < 1e-5 and rel(kb, hb) < 1e-5
                else:
                    ka, kb = torch.autograd.grad(gd, [x, y])
                    assert float(ka.abs().max()) == 0.0 and float(kb.abs().max()) == 0.0
    with pytest.raises(ValueError):
        ops.score_loss("lsgan", a.to(dev), b.to(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("one_sided", [False, True])
def test_grad_penalty_kernels(dev, one_sided):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(9)
    g = torch.randn(12, 5 * 96, generator=gen) * torch.linspace(0.01, 0.1, 12)[:, None]  # norms on both sides of 1
    gd = g.double().requires_grad_(True)
    pen = gd.norm(p=2, dim=1) - 1.0
    if one_sided:
        pen = torch.relu(pen)
    want = (pen ** 2).mean() * 10.0
    (wg,) = torch.autograd.grad(want, gd)
    x = g.to(dev).requires_grad_(True)
    got = ops.grad_penalty(x, 10.0, 1.0, one_sided)
    (gg,) = torch.autograd.grad(got, x)
    assert abs(float(got) - float(want)) < 1e-5 * float(want)
    assert rel(gg, wg) < 1e-5


COND_EPS, COND_TRIALS, COND_CAP = 1e-5, 4, 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TYPES))
def test_losses_against_reference_fixture(dev, name):
    """Loss values 1e-4; gradients per tensor within max(1e-4, 2 x the reference's own fp32-vs-fp64 distance,
    min(2 x conditioning, 5e-3)) of the reference's fp64 or fp32 run -- the gate of the end-to-end fixtures
    (test_e2e_gpu.py): the conditioning is this tensor's largest response to re-evaluations with reals and
    latents perturbed by 1e-5 relative noise, i.e. the size of the jumps single LeakyReLU mask flips cause in it
    (8 samples at depth 1: one flip moves a bias gradient by ~1e-3), which the reference's one fp32 draw may or
    may not have met."""
    from eeggan_amd import config, layers, losses
    fx = load_fixture("loss_types")
    c = _case(fx, name)
    depth, alpha, seed = int(fx["depth"]), float(fx["alpha"]), int(c["seed"])
    layers.set_drop_mode("numpy")
    g, d = config.build_models(int(fx["model_seed"]))
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    real, z, mix = (torch.from_numpy(c[k]).to(dev) for k in ("real", "z", "mix"))

    def evaluate(jitter=lambda t: t):
        np.random.seed(seed + 1)
        d_loss = losses.discriminator_loss(d, g, {"x": jitter(real)}, {"z": jitter(z)}, TYPES[name], 1e-3,
                                           float(c["grad_lambda"]), 1.0, mixing=mix)
        d_loss.backward()
        d_grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in d.named_parameters()}
        np.random.seed(seed + 2)
        g_loss = losses.generator_loss(d, g, {"x": jitter(real)}, {"z": jitter(z)}, TYPES[name], False, 0.0)
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
    worst = 0.0
    for tag in ("d", "g"):
        for k, got in grads[tag].items():
            if f"{tag}_none.{k}" in c:
                assert got is None, f"{tag}.{k}: the reference leaves this gradient None"
                continue
            assert got is not None, f"{tag}.{k}: missing gradient"
            if f"{tag}_norm.{k}" in c and float(c[f"{tag}_norm.{k}"]) < 1e-12:
                # a gradient that cancels exactly in the mathematics (the output layer's bias under a relativistic
                # loss): round-off in every implementation -- 5e-17 in the reference's fp64, 0 in its fp32
                assert float(got.norm()) < 1e-6, f"{name} {tag}.{k}: {float(got.norm()):.1e} where the gradient cancels"
                continue
            noise = float(c[f"{tag}_noise.{k}"]) if f"{tag}_noise.{k}" in c else 0.0
            tol = max(1e-4, NOISE_FACTOR * noise, min(NOISE_FACTOR * cond[tag].get(k, 0.0), COND_CAP))
            e = rel(subsample(got), c[f"{tag}_sub.{k}"])
            if f"{tag}_sub32.{k}" in c:
                e = min(e, rel(subsample(got), c[f"{tag}_sub32.{k}"]))
            worst = max(worst, e / tol)
            assert e < tol, (f"{name} {tag}.{k}: {e:.3e} (tolerance {tol:.3e}, reference noise {noise:.1e}, "
                             f"conditioning {cond[tag].get(k, 0.0):.1e})")
    print(f"{name}: worst deviation = {worst:.2f} of its gate")


@pytest.mark.gpu
def test_graph_trainer_with_other_losses(dev):
    """hinge splits into its two chains like wgan_gp; wgan_theirs does not (its penalty is scaled by a function of
    the scores) and is captured as one graph per iteration.  Both follow the eager trainer."""
    import test_e2e_gpu as E
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        for loss_type, split in (("hinge", True), ("wgan_theirs", False)):
            base = E._two_steps(dev, Trainer, lr=0.0, loss_type=loss_type), E._two_steps(dev, Trainer, loss_type=loss_type)
            seen = {}
            graphed = (E._two_steps(dev, GraphedTrainer, lr=0.0, loss_type=loss_type, seen=seen),
                       E._two_steps(dev, GraphedTrainer, loss_type=loss_type))
            assert (seen["critic"] == "_SplitCriticIteration") == split, seen
            E._same_trajectory(base, graphed, f"graph replay, {loss_type}")
    finally:
        layers.set_drop_mode("numpy")


@pytest.mark.gpu
def test_generator_weight_average(dev):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(2)
    params = [torch.randn(n, generator=gen).to(dev) for n in (5, 4096, 70001)]
    params.append(torch.randn(3, 7, 5, generator=gen).to(dev))
    avgs = [torch.randn(p.shape, generator=gen).to(dev) for p in params]
    want = [0.999 * a.double() + (1 - 0.999) * p.double() for a, p in zip(avgs, params)]
    ops.ema_update(avgs, params, 0.999)
    for a, w in zip(avgs, want):
        assert rel(a, w) < 1e-6


@pytest.mark.gpu
def test_output_generator_plugin(dev, tmp_path):
    """The averaged generator after three ticks equals the recurrence of plugins.py:305-306; the snapshot tick
    generates with the averaged weights and leaves the live weights untouched."""
    from eeggan_amd import config, layers
    from eeggan_amd.plugins import OutputGenerator
    layers.set_drop_mode("device")
    try:
        g, _ = config.build_models(5)
        g = g.to(dev)
        g.depth = 1
        for m in g.modules():
            if isinstance(m, layers.GDropLayer):
                m.strength = 0

        class _Trainer:
            generator, cur_nimg = g, 12000

        def sample_fn(n):
            gen = torch.Generator().manual_seed(1)
            while True:
                yield {"z": torch.randn(n, 256, generator=gen)}

        plug = OutputGenerator(sample_fn, str(tmp_path), config.seq_len(1), 30.0, samples_count=4,
                               output_snapshot_ticks=3, old_weight=0.59)
        plug.register(_Trainer)
        want = [p.detach().double().clone() for p in g.parameters()]
        gen = torch.Generator().manual_seed(3)
        for tick in (1, 2, 3):
            with torch.no_grad():
                for p in g.parameters():
                    p.add_(0.1 * torch.randn(p.shape, generator=gen).to(dev))
            live = [p.detach().clone() for p in g.parameters()]
            plug.epoch(tick)
            want = [0.59 * w + 0.41 * p.double() for w, p in zip(want, live)]
            for p, q in zip(g.parameters(), live):
                assert torch.equal(p, q)  # live weights restored
        for a, w in zip(plug.my_g_clone, want):
            assert rel(a, w) < 1e-6
        assert plug.last_samples.shape == (4, 5, config.seq_len(1))
        OutputGenerator.load_params(plug.my_g_clone, g)
        z = next(sample_fn(4))["z"].to(dev)
        with torch.no_grad():
            assert rel(g({"z": z})[0]["x"], plug.last_samples) < 1e-6
        assert (tmp_path / "network-snapshot-smooth_generator-000012.dat").exists()
    finally:
        layers.set_drop_mode("numpy")
