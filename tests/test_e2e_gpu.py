"""End-to-end parity of the HIP path (networks, losses, gradient penalty, Adam) with the
reference, on the golden fixtures (`tools/make_golden.py`) and against the CPU oracle.

Bars (fp32): outputs, scores and loss values within 1e-4 relative of the reference's fp64 run.
Parameter gradients: the reference's own fp32 run differs from its fp64 run by up to ~2e-3 at the
final stage (mostly the float32 interpolation weights of its resampler, plus LeakyReLU mask flips),
and that distance is stored per tensor in the fixture as `noise`.  A gradient passes when it lies
within max(1e-4, NOISE_FACTOR x noise) of the reference's fp64 evaluation or of its fp32
evaluation, i.e. within a few times the reference's own uncertainty of one of its two answers.
Generator gradients are linear in the gradient the discriminator hands back (d loss / d fake),
whose own reference uncertainty (`g_upstream_noise`, up to 3e-3 at the final stage) is a floor for
them: sums such as the to_rgb bias gradient cancel heavily and amplify it.  (Where no LeakyReLU
mask flips between the two fp32 runs the hand-back gradient agrees with the reference's fp32 run
to ~1e-6; a single differing flip moves it by ~5e-4.)"""
import numpy as np
import pytest
import torch

from helpers import check_weights, load_fixture, oracle_params, rel, subsample

pytestmark = pytest.mark.gpu

# The reference's fp32 run is ONE draw of the rounding noise of an ill-conditioned gradient; another
# fp32 implementation (different summation order) is another draw.  A tensor passes when its
# deviation stays within NOISE_FACTOR x the reference's own fp32-vs-fp64 deviation (floor 1e-4).
# Largest ratio measured on the five fixtures: 0.8 (per term), see the printed "worst deviation" lines.
NOISE_FACTOR = 2.0

# Book-keeping of the gates below: every tensor they see is recorded as (name, deviation, tolerance without the
# conditioning term = max(1e-4, NOISE_FACTOR x noise), full tolerance).  ``_gate_report`` prints the table of the
# tensors that pass ONLY through the conditioning term and bounds how many and how far they may be: a systematic
# kernel error would show as many such tensors, or as one far above its noise-only tolerance.
COND_ONLY_FAR = 3.0             # "far": a deviation above this many times the tensor's noise-only tolerance ...
COND_ONLY_MAX_FAR = 0.1         # ... which at most this share of a test's tensors may show (two at least)
COND_ONLY_MAX_RATIO = 25.0      # and no deviation may exceed this many times its noise-only tolerance (COND_CAP / 2e-4)


class _GateLog:
    """``family``: tensors whose deviations have ONE common cause count once in the bound on far outliers, with
    the family's median ratio (the generator's gradients are all linear images of one tensor, the gradient the
    critic hands back: a LeakyReLU branch that differs in the critic moves every one of them together)."""

    def __init__(self):
        self.rows, self.family = [], {}

    def add(self, name, e, tol_noise, tol, family=None):
        self.rows.append((name, e, tol_noise, tol))
        if family is not None:
            self.family[name] = family

    def report(self, what):
        cond_only = [r for r in self.rows if r[1] >= r[2]]
        print(f"{what}: {len(self.rows)} tensors, {len(cond_only)} pass only through the conditioning term")
        for name, e, tol_noise, tol in sorted(cond_only, key=lambda r: -r[1] / r[2]):
            print(f"    {name:48s} deviation {e:.2e} = {e / tol_noise:5.1f} x its noise-only tolerance {tol_noise:.1e} "
                  f"(gate {tol:.1e})")
        worst = max((r[1] / r[3] for r in self.rows), default=0.0)
        print(f"    worst deviation / gate over all tensors: {worst:.2f}")
        ratios, families = [], {}
        for name, e, tol_noise, _ in self.rows:
            if name in self.family:
                families.setdefault(self.family[name], []).append(e / tol_noise)
            else:
                ratios.append(e / tol_noise)
        for fam, values in families.items():
            ratios.append(sorted(values)[len(values) // 2])
            print(f"    family {fam}: {len(values)} tensors with one upstream cause, median deviation "
                  f"{ratios[-1]:.1f} x the noise-only tolerance")
        far = [r for r in ratios if r >= COND_ONLY_FAR]
        assert len(far) <= max(2, COND_ONLY_MAX_FAR * len(ratios)), \
            f"{what}: {len(far)} of {len(self.rows)} tensors lie more than {COND_ONLY_FAR:g} x above their noise-only tolerance"
        for name, e, tol_noise, _ in cond_only:
            assert e < COND_ONLY_MAX_RATIO * tol_noise, f"{what}, {name}: {e:.2e} is {e / tol_noise:.0f} x its noise-only tolerance"


E2E = ["e2e_d0_a1", "e2e_d1_a05", "e2e_d4_a1", "e2e_d7_a1", "e2e_d7_a05"]


def _build(dev, depth, alpha, seed):
    from eeggan_amd import config, layers
    layers.set_drop_mode("numpy")
    g, d = config.build_models(seed)
    return g, d


def _check_grads(fx, key, module, floor=1e-4):
    worst = 0.0
    if key == "g_grads":
        floor = max(floor, NOISE_FACTOR * float(fx["g_upstream_noise"]))
    for k, p in module.named_parameters():
        if f"{key}_none.{k}" in fx:
            assert p.grad is None, f"{key}.{k}: the reference leaves this gradient None at this depth"
            continue
        assert p.grad is not None, f"{key}.{k}: missing gradient"
        tol = max(floor, NOISE_FACTOR * float(fx[f"{key}_noise.{k}"]))
        got_sub = subsample(p.grad)
        e64 = rel(got_sub, fx[f"{key}_sub.{k}"])
        e32 = rel(got_sub, fx[f"{key}_sub32.{k}"])
        e = min(e64, e32)
        worst = max(worst, e / tol)
        assert e < tol, f"{key}.{k}: differs by {e64:.3e} (fp64 ref) / {e32:.3e} (fp32 ref), tolerance {tol:.3e}"
        n = float(p.grad.double().norm())
        assert abs(n - float(fx[f"{key}_norm.{k}"])) <= 2 * tol * float(fx[f"{key}_norm.{k}"]) + 1e-12, f"{key}.{k}: norm"
    return worst


# LeakyReLU's derivative jumps at zero, so these gradients are not a Lipschitz function of the
# activations at rounding scale: among 10^5..10^7 pre-activations a few sit within an ulp-sized
# distance of zero, and a kernel that sums in another order flips their masks.  One flip moves a
# weight gradient by ~1/sqrt(#activations) of its norm (2e-3 at the first stage), far above 1e-4,
# and whether the reference's fp32 run hit one is chance.  The tests therefore MEASURE the local
# conditioning: the critic-term gradients are re-evaluated COND_TRIALS times with reals and fakes
# perturbed by COND_EPS relative noise, and the largest response per tensor joins the tolerance.  A
# systematic kernel error is not excused by this: on a well-conditioned tensor the response to 1e-5
# noise stays near 1e-5.
COND_EPS, COND_TRIALS = 1e-5, 4
COND_CAP = 5e-3  # the measured conditioning never excuses more than this


def _term_grads(d, g, real, z, mix, seed, jitter=lambda t: t):
    """Gradients of the three critic-loss terms; host RNG order as in the reference's loss
    (D(real), G(z), D(fake), D(x_hat)).  ``jitter`` perturbs the critic's inputs."""
    from eeggan_amd import losses, ops
    np.random.seed(seed + 1)
    d.zero_grad()
    real = jitter(real)
    d_real = d({"x": real})[0]
    with torch.no_grad():
        fake = jitter(g({"z": z})[0]["x"])
    d_fake = d({"x": fake})[0]
    x_hat = ops.mix_batch_raw(real, fake, mix).requires_grad_(True)
    pred = d({"x": x_hat})[0]
    gg = losses.calc_grad(x_hat, pred).view(real.size(0), -1)
    loss = ops.wgan_gp_d_loss(d_real, d_fake, gg, 1e-3, 10.0, 1.0)
    heads = torch.autograd.grad(loss, [d_real, d_fake, gg], retain_graph=True)
    params = list(d.parameters())
    out = {}
    for term, o, head in (("real", d_real, heads[0]), ("fake", d_fake, heads[1]), ("gp", gg, heads[2])):
        out[term] = torch.autograd.grad(o, params, head, retain_graph=True, allow_unused=True)
    return out


def _check_d_terms(fx, g, d, real, z, mix, seed):
    """The three terms of the critic loss, differentiated separately: -mean(D(real)) + drift,
    mean(D(fake)) and the gradient penalty.  Each must match the reference's fp64 (or fp32) gradient
    of that term within max(1e-4, NOISE_FACTOR x the reference's own fp32-vs-fp64 distance for that
    tensor, the measured conditioning of that tensor).  Returns the conditioning per tensor."""
    base = _term_grads(d, g, real, z, mix, seed)
    names = [k for k, _ in d.named_parameters()]
    log = _GateLog()
    cond = {k: 0.0 for k in names}
    gen = torch.Generator(device=real.device).manual_seed(seed)
    for _ in range(COND_TRIALS):
        jitter = lambda t: t * (1.0 + COND_EPS * torch.randn(t.shape, device=t.device, generator=gen))
        pert = _term_grads(d, g, real, z, mix, seed, jitter)
        for term in base:
            for k, a, b in zip(names, base[term], pert[term]):
                if a is not None and float(a.norm()) > 0.0:
                    cond[k] = min(COND_CAP, max(cond[k], float((a - b).norm()) / float(a.norm())))
    worst = (0.0, "")
    for term, grads in base.items():
        for k, gr in zip(names, grads):
            key = f"d_term_{term}_sub.{k}"
            if key not in fx:
                assert gr is None or float(gr.abs().max()) == 0.0, f"{term}.{k}: the reference has no gradient here"
                continue
            if gr is None:
                # no path in the graph (e.g. the penalty and the biases above MinibatchStddev: they only move
                # LeakyReLU inputs, whose mask has zero derivative): the reference must hold exact zeros there
                assert float(np.abs(np.asarray(fx[key])).max()) == 0.0, f"{term}.{k}: the reference has a gradient here"
                continue
            # noise scale of this tensor: the largest fp32-vs-fp64 deviation the reference shows on any
            # of its three terms (single-term figures are single draws and scatter by 10x)
            noise = max(float(fx[f"d_term_{t}_noise.{k}"]) for t in ("real", "fake", "gp")
                        if f"d_term_{t}_noise.{k}" in fx)
            tol_noise = max(1e-4, NOISE_FACTOR * noise)
            tol = max(tol_noise, cond[k])
            e64, e32 = rel(subsample(gr), fx[key]), rel(subsample(gr), fx[f"d_term_{term}_sub32.{k}"])
            assert min(e64, e32) < tol, (f"critic term '{term}', {k}: differs by {e64:.3e} (fp64 ref) / {e32:.3e} "
                                         f"(fp32 ref), tolerance {tol:.3e} (reference noise {noise:.1e}, "
                                         f"conditioning {cond[k]:.1e})")
            log.add(f"{term}.{k}", min(e64, e32), tol_noise, tol)
            worst = max(worst, (min(e64, e32) / max(noise, 2.5e-5), f"{term}.{k}"))
    print(f"critic terms: worst deviation = {worst[0]:.2f} x the reference's own fp32-vs-fp64 distance ({worst[1]})")
    log.report("critic terms")
    return cond


def _check_d_total(fx, d, cond):
    """The summed gradient cancels heavily between the real and fake terms at initialisation, so
    its error is measured against the sum of the term norms (backward error), not its own norm."""
    for k, p in d.named_parameters():
        if f"d_grads_none.{k}" in fx:
            assert p.grad is None, f"d_grads.{k}: the reference leaves this gradient None at this depth"
            continue
        terms = [t for t in ("real", "fake", "gp") if f"d_term_{t}_norm.{k}" in fx]
        scale = sum(float(fx[f"d_term_{t}_norm.{k}"]) for t in terms)
        noise = max(float(fx[f"d_term_{t}_noise.{k}"]) for t in terms)  # same per-tensor scale as the term checks
        ref = torch.from_numpy(fx[f"d_grads_sub.{k}"])
        got = torch.from_numpy(subsample(p.grad))
        frac = float(ref.norm()) / max(float(fx[f"d_grads_norm.{k}"]), 1e-300)  # share of the norm in the subsample
        err = float((got - ref).norm())
        bound = frac * scale * max(1e-4, NOISE_FACTOR * noise, cond[k]) + 1e-12
        assert err < bound, f"d_grads.{k}: error {err:.3e} above {bound:.3e}"


@pytest.mark.parametrize("name", E2E)
def test_against_reference_fixture(dev, name):
    from eeggan_amd import losses
    fx = load_fixture(name)
    depth, alpha, seed = int(fx["depth"]), float(fx["alpha"]), int(fx["seed"])
    g, d = _build(dev, depth, alpha, int(fx["model_seed"]))
    check_weights(fx, g, d)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    real = torch.from_numpy(fx["real"]).to(dev)
    z = torch.from_numpy(fx["z"]).to(dev)
    mix = torch.from_numpy(fx["mix"]).to(dev)

    np.random.seed(seed)
    with torch.no_grad():
        fake = g({"z": z})[0]["x"]
        score = d({"x": real})[0]
    assert fake.shape == fx["ref64_fake"].shape
    assert rel(fake, fx["ref64_fake"]) < 1e-4
    assert rel(score, fx["ref64_score_real"]) < 1e-4

    cond = _check_d_terms(fx, g, d, real, z, mix, seed)

    np.random.seed(seed + 1)
    d_loss = losses.discriminator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", 1e-3, 10.0, 1.0, mixing=mix)
    d_loss.backward()
    assert abs(float(d_loss) - float(fx["ref64_d_loss"])) < 1e-4 * max(1.0, abs(float(fx["ref64_d_loss"])))
    _check_d_total(fx, d, cond)

    np.random.seed(seed + 2)
    fake = g(z)[0]["x"]
    (up,) = torch.autograd.grad(ops_mean_neg(d(fake)[0]), fake)
    tol = max(1e-4, NOISE_FACTOR * float(fx["g_upstream_noise"]))
    e64, e32 = rel(subsample(up, 2048), fx["g_upstream_sub"]), rel(subsample(up, 2048), fx["g_upstream_sub32"])
    assert min(e64, e32) < tol, f"discriminator input gradient: {e64:.3e} (fp64 ref) / {e32:.3e} (fp32 ref)"

    np.random.seed(seed + 2)
    g_loss = losses.generator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", False, 0.0)
    g_loss.backward()
    assert abs(float(g_loss) - float(fx["ref64_g_loss"])) < 1e-4 * max(1.0, abs(float(fx["ref64_g_loss"])))
    _check_grads(fx, "g_grads", g)


def ops_mean_neg(v):
    from eeggan_amd import ops
    return ops.mean_scale(v, -1.0)


def _oracle_gate(name, got, ref64, ref32, worst, scale=None, cond=0.0, log=None, family=None):
    """Gate for a gradient checked against the oracle run on this machine in BOTH precisions: the fp32 oracle
    run is one draw of the rounding noise of this gradient (LeakyReLU mask flips, see above), so the tensor
    must lie within max(1e-4, NOISE_FACTOR x |ref32 - ref64|) of the fp64 run or of the fp32 run.  ``scale``:
    norm the deviations are measured against (default: the gradient's own; the critic's summed gradient
    uses the sum of its terms' norms, because the real and fake terms cancel in it -- completely for the
    output layer's bias, where only the drift term survives).  ``cond``: this tensor's measured response (on
    the same scale) to 1e-5 relative noise on the critic's inputs: the size of the jumps single LeakyReLU
    mask flips cause in it, which one fp32 oracle draw may or may not have met (taken NOISE_FACTOR times like
    the oracle's own distance, and capped at COND_CAP)."""
    def dev(a, b):
        a = torch.as_tensor(a).detach().double().cpu().flatten()
        b = torch.as_tensor(b).detach().double().cpu().flatten()
        return float((a - b).norm()) / (float(b.norm()) if scale is None else scale)

    noise = dev(ref32, ref64)
    tol_noise = max(1e-4, NOISE_FACTOR * noise)
    tol = max(tol_noise, min(NOISE_FACTOR * cond, COND_CAP))
    e64, e32 = dev(got, ref64), dev(got, ref32)
    e = min(e64, e32)
    assert e < tol, (f"{name}: differs by {e64:.3e} (fp64 oracle) / {e32:.3e} (fp32 oracle), tolerance {tol:.3e} "
                     f"(oracle fp32-vs-fp64 {noise:.1e}, conditioning {cond:.1e})")
    if log is not None:
        log.add(name, e, tol_noise, tol, family)
    return max(worst, (e / max(noise, 2.5e-5), name))


def _oracle_d_grads(O, dp, ds, gp, gs, real, z, mix, depth, alpha, dtype, src=None, term_scales=False):
    """(loss, gradients per critic parameter[, scale per parameter]) of one critic iteration of the oracle in
    ``dtype``.  The scale of a parameter is the sum of the norms of its summed gradient and of the gradients
    of the two score terms (-mean D(real), mean D(fake)) that cancel inside it."""
    cast = lambda prm: {k: v.detach().to(dtype).requires_grad_(True) for k, v in prm.items()}
    dpc, gpc = cast(dp), cast(gp)
    src = src or O.DropSource("off")
    loss, aux = O.d_loss_wgan_gp(dpc, ds, gpc, gs, real.to(dtype), z.to(dtype), depth, alpha, O.NetConfig(),
                                 mix.to(dtype), src, src)
    params = list(dpc.values())
    grads = torch.autograd.grad(loss, params, allow_unused=True, retain_graph=term_scales)
    if not term_scales:
        return loss.detach(), grads
    g_real = torch.autograd.grad(aux["d_real"].mean(), params, allow_unused=True, retain_graph=True)
    g_fake = torch.autograd.grad(aux["d_fake"].mean(), params, allow_unused=True)
    norm = lambda t: 0.0 if t is None else float(t.norm())
    scales = [norm(a) + norm(b) + norm(c) for a, b, c in zip(grads, g_real, g_fake)]
    return loss.detach(), grads, scales


def test_blocks_against_reference_fixture(dev):
    """One GBlock / DBlock at a time (reference network.py:206-218, 441-449) with the reference's GDrop
    draws: head block, a block with to_rgb, a wide block; first block with from_rgb, a middle block, the
    last block with MinibatchStddev and the full-length kernel.  Outputs to 1e-5 and input gradients to 1e-4
    of the reference (either precision; measured 2e-7 .. 5e-7 for both)."""
    from eeggan_amd import config
    from helpers import block_inputs
    fx = load_fixture("blocks")
    g, d = _build(dev, 0, 1.0, config.RANDOM_SEED)
    g, d = g.to(dev), d.to(dev)
    for (name, net, path, kw, _), x, gy in block_inputs(fx):
        block = (g if net == "g" else d).get_submodule(path)
        np.random.seed(int(fx[f"{name}_seed"]))
        xi = x.to(dev).requires_grad_(True)
        y = block(xi, **kw)
        (gx,) = torch.autograd.grad(y, xi, gy.to(dev))
        assert tuple(y.shape) == tuple(int(v) for v in fx[f"{name}_yshape"])
        ey = min(rel(subsample(y, 4096), fx[f"{name}_y64"]), rel(subsample(y, 4096), fx[f"{name}_y32"]))
        eg = min(rel(subsample(gx, 4096), fx[f"{name}_gx64"]), rel(subsample(gx, 4096), fx[f"{name}_gx32"]))
        print(f"block {name}: output {ey:.1e}, input gradient {eg:.1e}")
        assert ey < 1e-5, (name, ey)
        assert eg < 1e-4, (name, eg)


def test_against_oracle_other_config(dev):
    """A configuration no fixture holds (depth 3, alpha 0.7, B=8), against the fp64 oracle run here."""
    from eeggan_amd import config, losses
    from oracle import ref_cpu as O
    depth, alpha, B, seed = 3, 0.7, 8, 77
    g, d = _build(dev, depth, alpha, 5)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed)
    real = torch.from_numpy(O.synthetic_eeg(B, 5, T, rng))
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, 256, generator=gen)
    mix = torch.rand(B, 1, 1, generator=gen)
    np.random.seed(seed)
    loss_ref, grads_ref, scales = _oracle_d_grads(O, dp, ds, gp, gs, real, z, mix, depth, alpha, torch.float64,
                                                  O.DropSource("numpy"), term_scales=True)
    np.random.seed(seed)
    _, grads_ref32 = _oracle_d_grads(O, dp, ds, gp, gs, real, z, mix, depth, alpha, torch.float32,
                                     O.DropSource("numpy"))

    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    np.random.seed(seed)
    loss = losses.discriminator_loss(d, g, {"x": real.to(dev)}, {"z": z.to(dev)}, "wgan_gp", 1e-3, 10.0, 1.0,
                                     mixing=mix.to(dev))
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    worst = (0.0, "")
    for (k, p), gr, gr32, sc in zip(d.named_parameters(), grads_ref, grads_ref32, scales):
        if gr is None:
            assert p.grad is None, k
        else:
            worst = _oracle_gate(k, p.grad, gr, gr32, worst, sc)
    print(f"depth 3, alpha 0.7: worst deviation = {worst[0]:.2f} x the oracle's fp32-vs-fp64 distance ({worst[1]})")


def test_headline_configuration_against_oracle(dev):
    """The configuration the metric is quoted on (final stage, T = 1920, alpha 1, 32 samples): one critic
    iteration against the fp64 oracle run here, with the weight gradients taken the way the trainers take
    them (deposited directly, submitted in groups).  Every kernel runs in its production shape: one unit
    per wave in the weight-stationary conv, the reduction-split tile on the short wide layers, grouped
    weight gradients with ~16 splits per layer.  Loss to 1e-4; gradients per tensor within
    max(1e-4, NOISE_FACTOR x the oracle's own fp32-vs-fp64 distance for that tensor, the tensor's measured and
    capped conditioning) of either oracle run."""
    from eeggan_amd import config, losses, ops
    from oracle import ref_cpu as O
    depth, alpha, B, seed = 7, 1.0, 32, 1373
    g, d = _build(dev, depth, alpha, 5)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed)
    real = torch.from_numpy(O.synthetic_eeg(B, 5, T, rng))
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, 256, generator=gen)
    mix = torch.rand(B, 1, 1, generator=gen)
    np.random.seed(seed)
    loss_ref, grads_ref, scales = _oracle_d_grads(O, dp, ds, gp, gs, real, z, mix, depth, alpha, torch.float64,
                                                  O.DropSource("numpy"), term_scales=True)
    np.random.seed(seed)
    _, grads_ref32 = _oracle_d_grads(O, dp, ds, gp, gs, real, z, mix, depth, alpha, torch.float32,
                                     O.DropSource("numpy"))

    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha

    def run(real_d):
        np.random.seed(seed)
        loss = losses.discriminator_loss(d, g, {"x": real_d}, {"z": z.to(dev)}, "wgan_gp", 1e-3, 10.0, 1.0,
                                         mixing=mix.to(dev))
        with ops.direct_weight_grads():
            loss.backward()
        return loss.detach(), [None if p.grad is None else p.grad.clone() for p in d.parameters()]

    loss, grads = run(real.to(dev))
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    # conditioning: the response of every gradient to COND_EPS relative noise on the reals
    cond = [0.0] * len(grads)
    jit = torch.Generator(device=dev).manual_seed(seed)
    for _ in range(COND_TRIALS):
        rd = real.to(dev)
        _, pert = run(rd * (1.0 + COND_EPS * torch.randn(rd.shape, device=dev, generator=jit)))
        for i, (a, b) in enumerate(zip(grads, pert)):
            if a is not None:
                cond[i] = max(cond[i], float((a - b).norm()) / scales[i])
    worst, log = (0.0, ""), _GateLog()
    for (k, p), got, gr, gr32, sc, cd in zip(d.named_parameters(), grads, grads_ref, grads_ref32, scales, cond):
        if gr is None:
            assert got is None, k
        else:
            worst = _oracle_gate(k, got, gr, gr32, worst, sc, cd, log)
    print(f"headline configuration: worst deviation = {worst[0]:.2f} x the oracle's fp32-vs-fp64 distance ({worst[1]})")
    log.report("headline configuration")


def _taped_oracle_critic_iteration(O, dp, ds, gp, gs, real, z, mix, depth, alpha, src):
    """One critic iteration of the oracle in fp64 that also records, for every LeakyReLU of its three critic passes
    (reals, fakes, interpolates; reference losses.py:76-79, 113) in call order, which branch each output took.
    Returns (aux of the loss, parameter tensors, gradients per term {'real', 'fake', 'gp'}, masks)."""
    tape = []
    orig_forward, orig_lrelu = O.discriminator_forward, O.leaky_relu

    def taped_lrelu(x, slope):
        tape.append(torch.where(x.detach() > 0, 1.0, -1.0).float())
        return orig_lrelu(x, slope)

    def taped_forward(*a, **k):  # (the generator's activations are not recorded)
        O.leaky_relu = taped_lrelu
        try:
            return orig_forward(*a, **k)
        finally:
            O.leaky_relu = orig_lrelu

    cast = lambda prm: {k: v.detach().double().requires_grad_(True) for k, v in prm.items()}
    dpc, gpc = cast(dp), cast(gp)
    O.discriminator_forward = taped_forward
    try:
        loss, aux = O.d_loss_wgan_gp(dpc, ds, gpc, gs, real.double(), z.double(), depth, alpha, O.NetConfig(),
                                     mix.double(), src, src)
    finally:
        O.discriminator_forward = orig_forward
    params = list(dpc.values())
    t_real = -aux["d_real"].mean() + (aux["d_real"] ** 2).mean() * 1e-3
    t_fake = aux["d_fake"].mean()
    t_gp = ((aux["gp_grad"].norm(p=2, dim=1) - 1.0) ** 2).mean() * 10.0  # (not loss - the others: no cancellation residue)
    assert abs(float(t_real + t_fake + t_gp) - float(loss)) < 1e-12 * max(1.0, abs(float(loss)))
    terms = {name: torch.autograd.grad(t, params, allow_unused=True, retain_graph=True)
             for name, t in (("real", t_real), ("fake", t_fake), ("gp", t_gp))}
    return loss.detach(), aux, terms, tape


BRANCH_TOL = 1e-4


def critic_terms_under_reference_branches(dev, depth, alpha, B, seed):
    """One critic iteration on the device with every LeakyReLU of the critic taking the branch the fp64 oracle took,
    against that oracle run: (relative loss deviation, deviation of the penalty's input gradient, {term.tensor:
    deviation of the term's gradient relative to its own norm}).  Weight gradients are taken the trainers' way
    (deposited directly, in groups), GDrop multipliers follow the reference's host stream on both sides, and the
    critic sees the oracle's fakes (rounded to float32), so both sides evaluate the same piecewise-linear function
    on the same inputs."""
    from eeggan_amd import config, layers, losses, ops
    from oracle import ref_cpu as O
    g, d = _build(dev, depth, alpha, 5)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed)
    real = torch.from_numpy(O.synthetic_eeg(B, 5, T, rng))
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, 256, generator=gen)
    mix = torch.rand(B, 1, 1, generator=gen)
    np.random.seed(seed)
    loss_ref, aux, terms_ref, tape = _taped_oracle_critic_iteration(O, dp, ds, gp, gs, real, z, mix, depth, alpha,
                                                                    O.DropSource("numpy"))
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    # the critic's three passes in the reference's order; host RNG order as in the loss: D(real), G(z), D(fake), D(x_hat)
    np.random.seed(seed)
    reald = real.to(dev)
    with layers.forced_masks([m.to(dev) for m in tape]) as forced:
        d.zero_grad()
        d_real = d({"x": reald})[0]
        with torch.no_grad():
            g({"z": z.to(dev)})  # (consumes the generator's draws of the host stream)
        faked = aux["fake"].float().to(dev)
        d_fake = d({"x": faked})[0]
        x_hat = ops.mix_batch_raw(reald, faked, mix.to(dev)).requires_grad_(True)
        pred = d({"x": x_hat})[0]
        gg = losses.calc_grad(x_hat, pred).view(B, -1)
        assert next(forced.masks, None) is None, "the critic made fewer LeakyReLU calls than the oracle"
        loss = ops.wgan_gp_d_loss(d_real, d_fake, gg, 1e-3, 10.0, 1.0)
        heads = torch.autograd.grad(loss, [d_real, d_fake, gg], retain_graph=True)
        params = list(d.parameters())
        got = {}
        for term, o, head in (("real", d_real, heads[0]), ("fake", d_fake, heads[1]), ("gp", gg, heads[2])):
            d.zero_grad()
            with ops.direct_weight_grads():
                torch.autograd.backward(o, head, retain_graph=True)
            got[term] = [None if p.grad is None else p.grad.clone() for p in params]
    errs = {}
    for term in ("real", "fake", "gp"):
        for (k, _), mine, ref in zip(d.named_parameters(), got[term], terms_ref[term]):
            if ref is None or float(ref.norm()) == 0.0:
                assert mine is None or float(mine.abs().max()) == 0.0, f"{term}.{k}: the oracle has no gradient here"
                continue
            assert mine is not None, f"{term}.{k}: missing gradient"
            errs[f"{term}.{k}"] = rel(mine, ref)
    loss_err = abs(float(loss) - float(loss_ref)) / max(1.0, abs(float(loss_ref)))
    return loss_err, rel(gg, aux["gp_grad"]), errs


@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_headline_configuration_under_reference_branches(dev, alpha):
    """The configuration the metric is quoted on (final stage, T = 1920, 32 samples; alpha 1 and mid-transition), one
    critic iteration with every LeakyReLU of the critic taking the branch the fp64 oracle took (reference
    layers.py:196-200 inside losses.py:73-120).  The critic is then the oracle's piecewise-linear function on both
    sides -- no branch can flip under rounding -- and the gradient of EVERY term (reals, fakes, penalty: the double
    backward) of EVERY tensor must agree to 1e-4 of that term's own norm, with no noise or conditioning allowance:
    the test a systematic error of a kernel cannot pass."""
    loss_err, gp_err, errs = critic_terms_under_reference_branches(dev, 7, alpha, 32, 1373)
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"under the oracle's LeakyReLU branches (alpha {alpha}): loss {loss_err:.1e}, penalty input gradient "
          f"{gp_err:.1e}, worst of {len(errs)} term gradients {worst[1]:.2e} ({worst[0]})")
    assert loss_err < 1e-5 and gp_err < BRANCH_TOL
    for k, e in errs.items():
        assert e < BRANCH_TOL, f"critic term gradient {k}: differs by {e:.3e} under the oracle's branches"


@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_benchmarked_launch_path_against_oracle(dev, alpha):
    """The launch path ``bench.py`` times, at the configuration it times (final stage, T = 1920, 32 samples;
    alpha 1 and mid-transition): ``GraphedTrainer`` with device-side noise handling, i.e. ONE stacked critic
    pass over [reals; fakes] (64 samples) and the interpolates' pass as two chains on two streams (three
    graphs per critic iteration), ONE generator pass over the latents of all five critic iterations
    (160 samples), weight gradients deposited directly in groups, Adam inside the graphs, every iteration
    graph replays -- against the fp64 oracle's five critic iterations and generator iteration
    (reference losses.py:73-120, 40-48; trainer.py:79-96).  GDrop strength is 0 so that both sides see the
    same multipliers, the mixing factors are fixed, and lr = 0 keeps the weights in place, so with beta1 = 0
    Adam's first moment IS the last iteration's gradient and its second moment the decayed sum of squares of
    all five.  Losses to 1e-4; gradients per tensor within max(1e-4, NOISE_FACTOR x the oracle's own
    fp32-vs-fp64 distance, the tensor's measured and capped conditioning) of either oracle precision; second
    moments to 4x that."""
    from eeggan_amd import config, layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.layers import GDropLayer
    from eeggan_amd.optim import Adam
    from oracle import ref_cpu as O
    depth, B, seed, repeats = 7, 32, 1373, 5
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed)
    reals = [torch.from_numpy(O.synthetic_eeg(B, 5, T, rng)) for _ in range(repeats)]
    gen = torch.Generator().manual_seed(seed)
    zs = [torch.randn(B, 256, generator=gen) for _ in range(repeats + 1)]
    mix = torch.rand(B, 1, 1, generator=gen)
    g, d = config.build_models(5)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    layers.set_drop_mode("device")
    try:
        g, d = g.to(dev), d.to(dev)
        g.depth = d.depth = depth
        g.alpha = d.alpha = alpha
        for m in list(g.modules()) + list(d.modules()):
            if isinstance(m, GDropLayer):
                m.strength = 0
        opt_g = Adam(g.parameters(), 0.0, **config.ADAM)
        opt_d = Adam(d.parameters(), 0.0, **config.ADAM)
        d_loss_fn, g_loss_fn = config.loss_closures()
        mixd = mix.to(dev)

        class _DS:
            model_depth = depth

        tr = GraphedTrainer(d, g, lambda dis, ge, r, z, **kw: d_loss_fn(dis, ge, r, z, mixing=mixd, **kw), g_loss_fn,
                            _DS(), iter([{"z": z.to(dev)} for z in zs]), 0, opt_g, opt_d, d_training_repeats=repeats)
        tr.dataiter = iter([{"x": r.to(dev)} for r in reals])
        tr.train()
        torch.cuda.synchronize()
        assert tr._fakes_call is not None and tr._zall.shape[0] == repeats * B, "one generator pass per step"
        g_loss, d_loss = (float(v) for v in tr.last_losses)
        assert type(tr._d_parts).__name__ == "_SplitCriticIteration", "the critic iteration runs as two chains"
        moments = lambda net, opt: [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone())
                                    if p in opt.state else None for p in net.parameters()]
        d_mom, g_mom = moments(d, opt_d), moments(g, opt_g)
        # conditioning: replay the same step with COND_EPS relative noise on reals and latents (lr = 0 and
        # beta1 = 0: the first moments afterwards are the last gradients of the jittered step)
        jit = torch.Generator(device=dev).manual_seed(seed)
        noisy = lambda t: t.to(dev) * (1.0 + COND_EPS * torch.randn(t.shape, device=dev, generator=jit))
        d_cond, g_cond = [0.0] * len(d_mom), [0.0] * len(g_mom)
        for _ in range(COND_TRIALS):
            tr.dataiter = iter([{"x": noisy(r)} for r in reals])
            tr.random_latents_generator = iter([{"z": noisy(z)} for z in zs])
            tr.train()
            torch.cuda.synchronize()
            for cond, base, net, opt in ((d_cond, d_mom, d, opt_d), (g_cond, g_mom, g, opt_g)):
                for i, p in enumerate(net.parameters()):
                    if base[i] is not None:
                        cond[i] = max(cond[i], float((opt.state[p]["exp_avg"] - base[i][0]).norm()))
    finally:
        layers.set_drop_mode("numpy")

    # oracle: the five critic iterations in fp64 (weights never move: lr = 0), the last one in fp32 too
    per_iter = [_oracle_d_grads(O, dp, ds, gp, gs, reals[i], zs[i], mix, depth, alpha, torch.float64,
                                term_scales=(i == repeats - 1)) for i in range(repeats)]
    scales = per_iter[-1][2]
    _, last32 = _oracle_d_grads(O, dp, ds, gp, gs, reals[-1], zs[-1], mix, depth, alpha, torch.float32)
    loss_ref = float(per_iter[-1][0])
    assert abs(d_loss - loss_ref) < 1e-4 * max(1.0, abs(loss_ref)), (d_loss, loss_ref)
    worst, log = (0.0, ""), _GateLog()
    for i, (k, p) in enumerate(d.named_parameters()):
        gr = per_iter[-1][1][i]
        if gr is None:
            assert p not in opt_d.state or float(opt_d.state[p]["exp_avg"].abs().max()) == 0.0, k
            continue
        m, v = d_mom[i]
        assert float(opt_d.state[p]["step"]) == repeats * (1 + COND_TRIALS), k
        cd = d_cond[i] / scales[i]
        worst = _oracle_gate(f"D.{k}", m, gr, last32[i], worst, scales[i], cd, log)
        # second moment: squares of the five gradients; a deviation e of a gradient (on the scale above)
        # moves it by 2 e |g| scale, so it is held to 4x the gradient's tolerance on the squared scale
        v_ref = sum(0.01 * 0.99 ** (repeats - 1 - j) * per_iter[j][1][i] ** 2 for j in range(repeats))
        noise = float((last32[i].double() - gr).norm()) / scales[i]
        tol_v = 4 * max(1e-4, NOISE_FACTOR * noise, min(NOISE_FACTOR * cd, COND_CAP))
        v_scale = sum(0.01 * 0.99 ** (repeats - 1 - j) for j in range(repeats)) * scales[i] * float(gr.abs().max())
        err_v = float((v.double().cpu() - v_ref).norm())
        assert err_v < tol_v * v_scale, (k, err_v / v_scale, tol_v)

    def g_grads(dtype):
        cast = lambda prm, grad: {k: v.detach().to(dtype).requires_grad_(grad) for k, v in prm.items()}
        gpc, dpc = cast(gp, True), cast(dp, False)
        off = O.DropSource("off")
        loss, _ = O.g_loss_wgan(dpc, ds, gpc, gs, zs[-1].to(dtype), depth, alpha, O.NetConfig(), off, off)
        return loss.detach(), torch.autograd.grad(loss, list(gpc.values()), allow_unused=True)

    gl64, gg64 = g_grads(torch.float64)
    _, gg32 = g_grads(torch.float32)
    assert abs(g_loss - float(gl64)) < 1e-4 * max(1.0, abs(float(gl64))), (g_loss, float(gl64))
    for i, (k, p) in enumerate(g.named_parameters()):
        if gg64[i] is None:
            assert p not in opt_g.state or float(opt_g.state[p]["exp_avg"].abs().max()) == 0.0, k
            continue
        assert float(opt_g.state[p]["step"]) == 1.0 + COND_TRIALS, k
        worst = _oracle_gate(f"G.{k}", g_mom[i][0], gg64[i], gg32[i], worst, None,
                             g_cond[i] / max(float(gg64[i].norm()), 1e-300), log, family="generator")
    log.report(f"benchmarked path, alpha {alpha:g}")
    print(f"benchmarked path, alpha {alpha:g}: d_loss {d_loss:.6f} (oracle {loss_ref:.6f}), g_loss {g_loss:.6f} "
          f"(oracle {float(gl64):.6f}); worst gradient deviation = {worst[0]:.2f} x the oracle's fp32-vs-fp64 "
          f"distance ({worst[1]})")


def test_full_step_fixture(dev):
    """One Trainer.train() (5 D + 1 G iterations with Adam) at depth 0 against the reference's
    parameter updates."""
    from eeggan_amd import config, layers, losses
    from eeggan_amd.optim import Adam
    from eeggan_amd.trainer import Trainer
    fx = load_fixture("step_depth0")
    seed, depth, lr = int(fx["seed"]), int(fx["depth"]), float(fx["lr"])
    layers.set_drop_mode("numpy")
    g, d = config.build_models(int(fx["model_seed"]))
    g0 = {k: v.clone() for k, v in g.state_dict().items()}
    d0 = {k: v.clone() for k, v in d.state_dict().items()}
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    opt_g = Adam(g.parameters(), lr, betas=(0.0, 0.99), eps=1e-8)
    opt_d = Adam(d.parameters(), lr, betas=(0.0, 0.99), eps=1e-8)
    mixes = iter([torch.from_numpy(fx[f"mix{i}"]).to(dev) for i in range(5)])
    d_loss_fn, g_loss_fn = config.loss_closures()
    d_losses = []

    def d_loss(dis, gen, real, z):
        out = d_loss_fn(dis, gen, real, z, mixing=next(mixes))
        d_losses.append(out)
        return out

    class _DS:
        model_depth, alpha = depth, 1.0

    tr = Trainer(d, g, d_loss, g_loss_fn, _DS(), iter([{"z": torch.from_numpy(fx[f"z{i}"])} for i in range(6)]), 0,
                 opt_g, opt_d)
    tr.dataiter = iter([{"x": torch.from_numpy(fx[f"real{i}"])} for i in range(5)])
    np.random.seed(seed)
    tr.train()
    got = [float(x) for x in d_losses]
    np.testing.assert_allclose(got, fx["losses64"][:5], rtol=2e-3, atol=2e-4)
    for net, m, s0 in (("G", g, g0), ("D", d, d0)):
        for k, v in m.state_dict().items():
            ref_norm = float(fx[f"{net}_delta_norm.{k}"])
            delta = v.double().cpu() - s0[k].double()
            if ref_norm == 0.0:
                assert float(delta.norm()) == 0.0, f"{net}.{k} must not move at depth 0"
                continue
            tol = max(2e-3, 2.0 * float(fx[f"{net}_delta_noise.{k}"]))
            e = rel(subsample(delta), fx[f"{net}_delta_sub.{k}"])
            assert e < tol, f"{net}.{k}: update differs by {e:.3e} (tolerance {tol:.3e})"


def _two_steps(dev, cls, sync_factory=None, lr=0.002, alphas=(1.0, 1.0), parts=True, B=8, shard=None, order=None,
               jitter=0.0, loss_type="wgan_gp", seen=None, models=None, prepare=None):
    """Two training steps at depth 2 with the noise off and fixed mixing factors (so that every
    variant sees the same random inputs).  Returns the weights, Adam's moments and step counts.
    ``parts``: the critic loss can be asked for its independent parts (a graph trainer then runs them as two
    chains).  ``shard`` = (rank, world): this process takes its share of every batch of ``B`` samples.
    ``order``: the samples of every batch in another order.  ``jitter``: relative size of a random perturbation
    of the reals (a few ulps: what another rounding of the same arithmetic does to the inputs)."""
    from eeggan_amd import config
    from eeggan_amd.layers import GDropLayer
    from eeggan_amd.optim import Adam
    from eeggan_amd.parallel import shard_batch
    depth = 2
    T = config.seq_len(depth)
    gen = torch.Generator().manual_seed(4)
    if shard is not None:
        own = lambda t: shard_batch(t, shard[0], shard[1]).contiguous()  # noqa: E731
    elif order is not None:
        own = lambda t: t[torch.as_tensor(order)].contiguous()  # noqa: E731
    else:
        own = lambda t: t  # noqa: E731
    reals = [own(torch.randn(B, 5, T, generator=gen)).to(dev) for _ in range(10)]
    if jitter:
        jgen = torch.Generator().manual_seed(77)
        reals = [r * (1 + jitter * torch.randn(r.shape, generator=jgen).to(dev)) for r in reals]
    zs = [own(torch.randn(B, 256, generator=gen)).to(dev) for _ in range(12)]
    mix = own(torch.rand(B, 1, 1, generator=gen)).to(dev)
    g, d = config.build_models(9, **(models or {}))  # (``models``: constructor overrides, see config.build_models)
    if prepare is not None:  # (changes to the freshly built networks, e.g. parameters that start at zero)
        prepare(g, d)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    for m in list(g.modules()) + list(d.modules()):
        if isinstance(m, GDropLayer):
            m.strength = 0
    opt_g = Adam(g.parameters(), lr, **config.ADAM)
    opt_d = Adam(d.parameters(), lr, **config.ADAM)
    d_loss_fn, g_loss_fn = config.loss_closures(loss_type=loss_type)

    class _DS:
        model_depth, alpha = depth, 1.0

    if parts:
        critic_loss = lambda dis, ge, r, z, **kw: d_loss_fn(dis, ge, r, z, mixing=mix, **kw)  # noqa: E731
    else:
        critic_loss = lambda dis, ge, r, z: d_loss_fn(dis, ge, r, z, mixing=mix)  # noqa: E731
    tr = cls(d, g, critic_loss, g_loss_fn, _DS(), iter([{"z": z} for z in zs]), 0, opt_g, opt_d)
    tr.dataiter = iter([{"x": r} for r in reals])
    if sync_factory is not None:
        tr.grad_sync = sync_factory()
    for alpha in alphas:  # what the depth manager does between steps during a transition
        g.alpha = d.alpha = alpha
        tr.train()
    torch.cuda.synchronize()
    if seen is not None:  # which form the critic iteration was captured in
        seen["critic"] = type(getattr(tr, "_d_parts", None)).__name__
    out = {}
    for tag, net, opt in (("G", g, opt_g), ("D", d, opt_d)):
        for k, p in net.named_parameters():
            out[f"{tag}.{k}"] = p.detach().clone()
            if p in opt.state:
                st = opt.state[p]
                out[f"{tag}.{k}.m"], out[f"{tag}.{k}.v"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
                out[f"{tag}.{k}.step"] = st["step"].clone()
    return out


def _same_trajectory(base, other, what, moment_tol=1e-5):
    """The accumulation order of the three gradient contributions of a critic weight is up to the
    autograd engine (nodes made during the double backward are numbered by another thread), so two
    runs of the same eager code already differ by an ulp in some gradients, and Adam with beta1 = 0
    turns an ulp around zero into a full +-lr step.  Hence two comparisons:

    * with lr = 0 the weights stay put and every iteration is independent of that noise: Adam's
      moments (exp_avg = the last gradient, exp_avg_sq = the decayed sum of squares over all
      iterations) must agree to 1e-5 of the tensor's scale and the step counts exactly;
    * with lr > 0 the weight DISPLACEMENTS of the two runs (weights minus the lr = 0 run's, i.e. the
      initial weights) must agree to 25 % in norm.  With beta1 = 0 every update is +-lr per element;
      an element whose gradient changes sign between the runs contributes (2 lr)^2, so 0.5 % of such
      elements already make 8 % (measured), while a missing or doubled update makes 50-100 %."""
    frozen_a, live_a = base
    frozen_b, live_b = other
    assert frozen_a.keys() == frozen_b.keys() and live_a.keys() == live_b.keys()
    for k, a in frozen_a.items():
        b = frozen_b[k]
        if k.endswith(".step") or not k.endswith((".m", ".v")):
            assert torch.equal(a, b), f"{what}: {k} differs with lr = 0"
        else:
            err = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30)
            # The output layer's one-element bias: its gradient is -1 (reals' mean score) + 1 (fakes') + the small
            # drift term, i.e. ~1e-3 left over from sums of magnitude 1 -- fp32 leaves it ~1e-4 of relative accuracy,
            # and two launch paths that sum the batch in another order (one stacked pass against two passes) differ
            # by that much.
            tol = max(moment_tol, 1e-4) if k.startswith("D.linear.conv.conv.bias") else moment_tol
            assert err < tol, f"{what}: {k} differs by {err:.2e} of its scale with lr = 0"
    moved = 0
    for k, a in live_a.items():
        if k.endswith((".m", ".v", ".step")):
            continue
        da, db = a - frozen_a[k], live_b[k] - frozen_b[k]
        if float(da.norm()) == 0.0:
            assert float(db.norm()) == 0.0, f"{what}: {k} moved in one run only"
            continue
        err = float((da - db).norm()) / float(da.norm())
        assert err < 0.25, f"{what}: displacement of {k} differs by {100 * err:.1f}% after two steps"
        moved += 1
    assert moved > 0


def _autograd_trainer():
    """Trainer whose weight gradients take torch's own route (returned to autograd, accumulated by it)."""
    from eeggan_amd.trainer import Trainer

    class AutogradTrainer(Trainer):
        direct_grads = False

    return AutogradTrainer


def test_no_garbage_collection_during_capture(dev):
    """A collection that starts while a stream is capturing can release device objects of earlier work (graphs and
    tensors of a discarded trainer in a reference cycle), which the runtime forbids during a capture and answers by
    aborting the process: captures run with the collector switched off, after one collection outside them."""
    import gc
    from eeggan_amd.graphs import GraphedCall
    seen = []
    x = torch.ones(1024, device=dev)

    def fn():
        seen.append(gc.isenabled())
        return x * 2.0

    assert gc.isenabled()
    call = GraphedCall(fn)
    assert seen[:-1] == [True] * (len(seen) - 1) and seen[-1] is False  # warm-up runs, then the captured run
    assert gc.isenabled()
    assert float(call()[0]) == 2.0


def test_graph_replay_equals_eager(dev):
    """Two training steps follow the same trajectory (a) with weight gradients accumulated by autograd
    on one stream, (b) with the trainer's default side-stream accumulation into persistent buffers,
    (c) replayed from captured HIP graphs."""
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    try:
        plain = _autograd_trainer()
        base = _two_steps(dev, plain, lr=0.0), _two_steps(dev, plain)
        direct = _two_steps(dev, Trainer, lr=0.0), _two_steps(dev, Trainer)
        graphed = _two_steps(dev, GraphedTrainer, lr=0.0), _two_steps(dev, GraphedTrainer)
        single = _two_steps(dev, GraphedTrainer, lr=0.0, parts=False), _two_steps(dev, GraphedTrainer, parts=False)
    finally:
        layers.set_drop_mode("numpy")
    _same_trajectory(base, direct, "side-stream weight gradients")
    _same_trajectory(base, graphed, "graph replay, critic iteration as two chains")
    _same_trajectory(base, single, "graph replay, critic iteration as one graph")


def test_graph_replay_during_fade_in(dev):
    """With alpha moving from step to step (0.3 then 0.6) the captured graphs, which read alpha from
    device memory, follow the eager trainer, whose kernels get it as an argument."""
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    fade = (0.3, 0.6)
    try:
        base = _two_steps(dev, Trainer, lr=0.0, alphas=fade), _two_steps(dev, Trainer, alphas=fade)
        graphed = _two_steps(dev, GraphedTrainer, lr=0.0, alphas=fade), _two_steps(dev, GraphedTrainer, alphas=fade)
        constant = _two_steps(dev, GraphedTrainer, lr=0.0, alphas=(0.3, 0.3))
    finally:
        layers.set_drop_mode("numpy")
    _same_trajectory(base, graphed, "graph replay during fade-in")
    # the second step really used the new alpha: the last gradients differ from a run that kept 0.3
    k = next(k for k in graphed[0] if k.startswith("D.") and k.endswith(".m"))
    assert not torch.equal(graphed[0][k], constant[k])


def test_data_parallel_graphs_equal_eager(dev):
    """The data-parallel step (gradient bucket -> RCCL all-reduce -> Adam reading the bucket), eager
    and as two captured graphs around the collective, on a one-rank RCCL group: both must follow
    the plain single-process trainer."""
    import torch.distributed as dist
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.parallel import DataParallelStep
    from eeggan_amd.trainer import Trainer
    layers.set_drop_mode("device")
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device(dev))
    try:
        sync = lambda: DataParallelStep(always_bucket=True)
        base = _two_steps(dev, Trainer, lr=0.0), _two_steps(dev, Trainer)
        eager = _two_steps(dev, Trainer, sync, lr=0.0), _two_steps(dev, Trainer, sync)
        graphed = _two_steps(dev, GraphedTrainer, sync, lr=0.0), _two_steps(dev, GraphedTrainer, sync)
        single = (_two_steps(dev, GraphedTrainer, sync, lr=0.0, parts=False),
                  _two_steps(dev, GraphedTrainer, sync, parts=False))
    finally:
        layers.set_drop_mode("numpy")
        if own_group:
            dist.destroy_process_group()
    _same_trajectory(base, eager, "eager data-parallel step")
    _same_trajectory(base, graphed, "graphed data-parallel step, two chains and two buckets")
    _same_trajectory(base, single, "graphed data-parallel step, one graph and one bucket")


def _two_rank_worker(rank, world, port, out):
    """One rank of the data-parallel trainer.  Both ranks share the one GPU; the collective is gloo's, on
    host copies of the buckets (DataParallelStep stages them)."""
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import test_e2e_gpu as T
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.parallel import DataParallelStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    layers.set_drop_mode("device")
    res = []
    for lr in (0.0, 0.002):
        r = T._two_steps(dev, GraphedTrainer, DataParallelStep, lr=lr, B=16, shard=(rank, world))
        res.append({k: v.cpu().numpy() for k, v in r.items()})
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_follow_the_global_batch(dev):
    """Two processes, each with half of every 16-sample batch (whole minibatch-stddev groups, shard_batch),
    run the graph trainer with DataParallelStep (buckets packed on the device, summed over the ranks, Adam
    reading the buckets).  Their trajectory must be that of ONE process training on the 16-sample batches: the
    same weight displacements with lr > 0, and with lr = 0 the same moments up to what another rounding of the
    same arithmetic does to them.  That is more than an ulp: kernels pick their tiles by batch size, so
    activations differ in the last bits, a pre-activation within that of zero changes its LeakyReLU branch,
    and the critic's terms cancel at initialisation, so a gradient is small against its summands (the
    one-process trainer's own variants differ by 1.7e-3 of a tensor's scale here when the fakes of a step come
    from one 80-sample generator pass instead of five 16-sample passes, which changes them by 3e-7).  The level
    is measured, not assumed: the one-process run is repeated with the samples of every batch in another order
    that keeps the minibatch-stddev groups (their columns rotated) and with the reals perturbed by 1e-6 of
    their size; the two-rank run may differ from the first by NOISE_FACTOR times the worst tensor's difference
    among those, and by no more than COND_CAP.  A missing rank, a missing 1/world or a broken shard is an error
    of 0.5-1."""
    import torch.multiprocessing as mp
    import socket
    from eeggan_amd import layers
    from eeggan_amd.graphs import GraphedTrainer
    rotated = [k * 4 + (c + 1) % 4 for k in range(4) for c in range(4)]
    layers.set_drop_mode("device")
    try:
        base = _two_steps(dev, GraphedTrainer, lr=0.0, B=16), _two_steps(dev, GraphedTrainer, B=16)
        again = [_two_steps(dev, GraphedTrainer, lr=0.0, B=16, order=rotated),
                 _two_steps(dev, GraphedTrainer, lr=0.0, B=16, jitter=1e-6)]
    finally:
        layers.set_drop_mode("numpy")
    noise = max(float((a - other[k]).abs().max()) / max(float(a.abs().max()), 1e-30)
                for other in again for k, a in base[0].items() if k.endswith((".m", ".v")))
    gate = max(1e-5, min(NOISE_FACTOR * noise, COND_CAP))
    print(f"rounding noise of the moments (worst tensor): {noise:.2e} -> gate {gate:.2e}")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get() if _wait_for(procs, out, 240) else None
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert res is not None and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    ranks = tuple({k: torch.from_numpy(v).to(dev) for k, v in r.items()} for r in res)
    worst = max(float((a - ranks[0][k]).abs().max()) / max(float(a.abs().max()), 1e-30)
                for k, a in base[0].items() if k.endswith((".m", ".v")))
    print(f"two ranks against the global batch, moments (worst tensor): {worst:.2e}")
    _same_trajectory(base, ranks, "two ranks against the global batch", moment_tol=gate)


def _wait_for(procs, queue, seconds):
    """True once the queue holds a result; False when a worker has died or the time is up."""
    import time
    end = time.time() + seconds
    while time.time() < end:
        if not queue.empty():
            return True
        if any(p.exitcode not in (None, 0) for p in procs):
            return False
        time.sleep(0.2)
    return False


def test_stacked_critic_pass_equals_sequential(dev):
    """discriminator_loss with reals and fakes stacked into one critic pass (the device-noise default)
    gives the loss and the weight gradients of the reference's two sequential passes (noise off, so
    that both see the same multipliers).

    Scores and loss agree to 1e-5.  The gradients cannot be held to that: a stacked pass runs other
    tile shapes, so pre-activations differ in the last bit, and among ~10^6 of them a handful sit
    close enough to zero for their LeakyReLU mask to flip, each flip changing that element's
    gradient by 0.8 |g| (measured: 2e-3 of a gradient's norm, the same mechanism that separates the
    reference's own fp32 and fp64 runs).  So the bar is 2e-2 of the real term's norm plus the
    total's (the two terms nearly cancel in the total): loose for rounding, tight for a wrong
    segment, a dropped term or a double count, which are O(1)."""
    from eeggan_amd import config, ops
    from eeggan_amd.layers import GDropLayer
    depth, B = 3, 8
    T = config.seq_len(depth)
    gen = torch.Generator().manual_seed(11)
    real = torch.randn(B, 5, T, generator=gen).to(dev)
    z = torch.randn(B, 256, generator=gen).to(dev)
    mix = torch.rand(B, 1, 1, generator=gen).to(dev)
    g, d = config.build_models(5)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    for m in list(g.modules()) + list(d.modules()):
        if isinstance(m, GDropLayer):
            m.strength = 0
    d_loss_fn, _ = config.loss_closures()
    out = []
    # sequential passes; reals+fakes stacked, with weight gradients accumulated by autograd and deposited
    # directly (the trainers' mode)
    for stacked, direct in ((False, False), (True, False), (True, True)):
        loss = d_loss_fn(d, g, {"x": real}, {"z": z}, mixing=mix, stack_real_fake=stacked)
        with ops.direct_weight_grads(direct):
            loss.backward()
        out.append((float(loss.detach()), {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None}))
    (l0, g0) = out[0]
    d.zero_grad()
    ops.mean_scale(d({"x": real})[0], -1.0).backward()
    term = {k: float(p.grad.norm()) for k, p in d.named_parameters() if p.grad is not None}
    for what, (l1, g1) in zip(("stacked", "stacked, direct gradients"), out[1:]):
        assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), what
        assert g0.keys() == g1.keys() and len(g0) > 0, what
        for k in g0:
            e = float((g0[k] - g1[k]).norm()) / max(float(g0[k].norm()) + term[k], 1e-30)
            assert e < 2e-2, f"{k}: {what} pass differs by {e:.2e} of the real term's scale"


def _progression_run(dev, lr, shard=None, sync_factory=None, jitter=0.0):
    """The graph trainer driven by the reference's DepthManager (plugins.py:28-119) through three stable steps at
    depth 0, two steps of fade-in (alpha 1/3, 2/3) and two stable steps at depth 1, on global batches of 16 reals
    (noise off, fixed mixing factors, seeded data).  ``shard`` = (rank, world): this process takes its share of
    every batch and sums gradients through ``sync_factory()``.  Returns the final weights and moments (as
    ``_two_steps`` does) plus, per step, (depth, alpha, number of critic tensors in this step's gradient buckets)."""
    from eeggan_amd import config
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.layers import GDropLayer
    from eeggan_amd.optim import Adam
    from eeggan_amd.parallel import shard_batch
    from eeggan_amd.plugins import DepthManager, Plugin
    B = 16
    own = (lambda t: shard_batch(t, shard[0], shard[1]).contiguous()) if shard is not None else (lambda t: t)

    class SmallStages(DepthManager):
        minibatch_override = {0: B, 1: B}
        tick_kimg_override = training_kimg_override = transition_kimg_override = {}

    class Dataset:
        model_depth, alpha = 0, 1.0

    g, d = config.build_models(9)
    g, d = g.to(dev), d.to(dev)
    for m in list(g.modules()) + list(d.modules()):
        if isinstance(m, GDropLayer):
            m.strength = 0
    ds = Dataset()
    mix = own(torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(21))).to(dev)

    def loader(minibatch):  # (a new loader per stage, as the reference's create_dataloader_fun: seeded by the stage)
        depth = ds.model_depth
        gen, jgen = torch.Generator().manual_seed(100 + depth), torch.Generator().manual_seed(300 + depth)
        while True:
            r = torch.randn(minibatch, 5, config.seq_len(depth), generator=gen)
            if jitter:
                r = r * (1 + jitter * torch.randn(r.shape, generator=jgen))
            yield {"x": own(r).to(dev)}

    def latents(minibatch):
        gen = torch.Generator().manual_seed(200 + ds.model_depth)
        while True:
            yield {"z": own(torch.randn(minibatch, 256, generator=gen)).to(dev)}

    opts = []

    def optimizers(stage_lr):  # (the learning rate under test replaces the stage's; the reset per stage stays)
        opts[:] = [Adam(g.parameters(), lr, **config.ADAM), Adam(d.parameters(), lr, **config.ADAM)]
        return opts[0], opts[1], None, None

    class Watch(Plugin):
        def __init__(self):
            super().__init__([(1, "iteration")])
            self.seen = []

        def register(self, trainer):
            self.trainer = trainer

        def iteration(self, *args):  # (runs BEFORE the depth manager's hook of the same step: registered first)
            tr = self.trainer
            sync = tr.grad_sync
            tensors = None if sync is None else sum(len(b.params) for (mod, _), b in sync.buckets.items() if mod == id(d))
            self.seen.append((tr.generator.depth, round(float(tr.generator.alpha), 4), tensors))

    d_loss_fn, g_loss_fn = config.loss_closures()
    tr = GraphedTrainer(d, g, lambda dis, ge, r, z, **kw: d_loss_fn(dis, ge, r, z, mixing=mix, **kw), g_loss_fn, ds,
                        None, 0, None, None, tick_kimg_default=1.0)
    if sync_factory is not None:
        tr.grad_sync = sync_factory()
    watch = Watch()
    tr.register_plugin(watch)
    tr.register_plugin(SmallStages(loader, latents, 1, 1.0, optimizers, 0.001, lod_training_kimg=0.16,
                                   lod_transition_kimg=0.24))
    tr.run(total_kimg=0.56)
    torch.cuda.synchronize()
    out = {}
    for tag, net, opt in (("G", g, opts[0]), ("D", d, opts[1])):
        for k, p in net.named_parameters():
            out[f"{tag}.{k}"] = p.detach().clone()
            if p in opt.state:
                st = opt.state[p]
                out[f"{tag}.{k}.m"], out[f"{tag}.{k}.v"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
                out[f"{tag}.{k}.step"] = st["step"].clone()
    return out, watch.seen


def _progression_worker(rank, world, port, out):
    """One rank of ``_progression_run`` with DataParallelStep over gloo (both ranks share the one GPU; the buckets
    cross to host memory around the collective)."""
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import test_e2e_gpu as T
    from eeggan_amd import layers
    from eeggan_amd.parallel import DataParallelStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    layers.set_drop_mode("device")
    res = []
    for lr in (0.0, 0.002):
        state, seen = T._progression_run(dev, lr, shard=(rank, world), sync_factory=DataParallelStep)
        res.append(({k: v.cpu().numpy() for k, v in state.items()}, seen))
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_through_a_depth_change(dev):
    """Two ranks (DataParallelStep under the graph trainer) through the progressive schedule: a stable stage, the
    fade-in of the next one and that stage stable.  The set of critic tensors that receive gradients changes at
    both borders (8 -> 14 -> 12: the fade-in adds the lower stage's from_rgb, reference network.py:533-540), so the
    gradient buckets must be rebuilt twice; the optimizers are replaced at the depth change (plugins.py:106-108);
    the schedule must follow the GLOBAL count of reals on every rank.  The two-rank trajectory must be the
    one-process trajectory on the global batches: the depth manager's sequence exactly, Adam's moments of the last
    stage with lr = 0 up to the measured rounding response, the weight displacements with lr > 0 to 25 %."""
    import socket
    import torch.multiprocessing as mp
    from eeggan_amd import layers
    layers.set_drop_mode("device")
    try:
        (frozen, seen), (live, _) = _progression_run(dev, 0.0), _progression_run(dev, 0.002)
        again, _ = _progression_run(dev, 0.0, jitter=1e-6)
    finally:
        layers.set_drop_mode("numpy")
    want = [(0, 1.0), (0, 1.0), (0, 1.0), (1, 0.3333), (1, 0.6667), (1, 1.0), (1, 1.0)]
    assert [s[:2] for s in seen] == want, seen
    noise = max(float((a - again[k]).abs().max()) / max(float(a.abs().max()), 1e-30)
                for k, a in frozen.items() if k.endswith((".m", ".v")))
    gate = max(1e-5, min(NOISE_FACTOR * noise, COND_CAP))
    print(f"rounding noise of the last stage's moments (worst tensor): {noise:.2e} -> gate {gate:.2e}")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_progression_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get() if _wait_for(procs, out, 300) else None
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert res is not None and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    (frozen2, seen2), (live2, seen2_live) = res
    # every rank followed the one-process schedule (it counts the global batch) ...
    assert [s[:2] for s in seen2] == want and [s[:2] for s in seen2_live] == want, (seen2, seen2_live)
    # ... and its buckets were rebuilt at both borders: 8 critic tensors with gradients at depth 0, 14 in the
    # fade-in, 12 at depth 1
    assert [s[2] for s in seen2] == [8, 8, 8, 14, 14, 12, 12], [s[2] for s in seen2]
    ranks = tuple({k: torch.from_numpy(v).to(dev) for k, v in r.items()} for r in (frozen2, live2))
    worst = max(float((a - ranks[0][k]).abs().max()) / max(float(a.abs().max()), 1e-30)
                for k, a in frozen.items() if k.endswith((".m", ".v")))
    print(f"two ranks through the depth change against the global batch, moments (worst tensor): {worst:.2e}")
    _same_trajectory((frozen, live), ranks, "two ranks through a depth change", moment_tol=gate)


def test_progressive_schedule_with_graphs(dev):
    """DepthManager (the reference's plugins.py:78-119 schedule) driving the graph trainer through a
    stable stage, a fade-in and the next stable stage: depth/alpha/optimizers/minibatch change under
    the trainer's feet and it has to re-capture; alpha during the fade comes from device memory."""
    from eeggan_amd import config, layers
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.optim import Adam
    from eeggan_amd.plugins import DepthManager, Plugin

    class SmallStages(DepthManager):
        minibatch_override = {0: 8, 1: 8}
        tick_kimg_override = {}
        training_kimg_override = {}
        transition_kimg_override = {}

    class Dataset:
        model_depth, alpha = 0, 1.0

    class Watch(Plugin):
        def __init__(self):
            super().__init__([(1, "iteration")])
            self.seen = []

        def register(self, trainer):
            self.trainer = trainer

        def iteration(self, time, g_loss, d_loss):
            tr = self.trainer
            self.seen.append((tr.generator.depth, tr.generator.alpha, float(g_loss), float(d_loss), tr._key))

    layers.set_drop_mode("device")
    try:
        g, d = config.build_models(3)
        g, d = g.to(dev), d.to(dev)
        ds = Dataset()
        gen = torch.Generator(device=dev).manual_seed(1)

        def loader(minibatch):
            T = config.seq_len(ds.model_depth)
            while True:
                yield {"x": torch.randn(minibatch, 5, T, device=dev, generator=gen)}

        def latents(minibatch):
            while True:
                yield {"z": torch.randn(minibatch, 256, device=dev, generator=gen)}

        def optimizers(lr):
            return Adam(g.parameters(), lr, **config.ADAM), Adam(d.parameters(), lr, **config.ADAM), None, None

        d_loss, g_loss = config.loss_closures()
        tr = GraphedTrainer(d, g, d_loss, g_loss, ds, None, 0, None, None, tick_kimg_default=0.04)
        # 80 reals stable at depth 0, 160 reals of fade-in, then depth 1 stable (40 reals per step)
        tr.register_plugin(SmallStages(loader, latents, 1, 0.04, optimizers, 0.001, lod_training_kimg=0.08,
                                       lod_transition_kimg=0.16))
        watch = Watch()
        tr.register_plugin(watch)
        start = {k: v.clone() for k, v in g.state_dict().items()}
        tr.run(total_kimg=0.36)
    finally:
        layers.set_drop_mode("numpy")
    depths = [s[0] for s in watch.seen]
    alphas = [s[1] for s in watch.seen]
    assert depths == [0, 0] + [1] * 7
    # (the watcher runs after the depth manager, so it sees the settings of the NEXT step)
    assert alphas == [1.0, 1.0, 0.25, 0.5, 0.75, 1.0, 1.0, 1.0, 1.0]
    assert all(np.isfinite(s[2]) and np.isfinite(s[3]) for s in watch.seen)
    assert len({s[4] for s in watch.seen}) == 3, "one capture per (depth, fading?) phase"
    assert all(torch.isfinite(v).all() for v in g.state_dict().values())
    assert any(not torch.equal(v, start[k]) for k, v in g.state_dict().items())
    assert tr.cur_tick >= 8
