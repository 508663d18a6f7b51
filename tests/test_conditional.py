"""Class-conditional networks: ``Generator(z, y)`` with the classes entering every batch norm through the shared
embedding ('CBN' / 'CSM' modes of ConditionalBatchNorm, layers.py:93-134; network.py:284-292, 355-361) and
``Discriminator(x, y)`` with its projection on a linear embedding of the classes (network.py:518-523, 555-557).

Fixture ``conditional.npz`` (``tools/make_golden_conditional.py``): the imported reference's networks driven directly
(its training step does not reach these paths), 8 samples, fp64 and fp32: the fakes and a linear functional of them with
the generator's gradients; the scores and a functional of them plus a penalty on the norm of the input gradient with
the critic's gradients."""
import numpy as np
import pytest
import torch

from helpers import load_fixture, oracle_params, rel, subsample

NOISE_FACTOR, COND_EPS, COND_TRIALS, COND_CAP = 2.0, 1e-5, 4, 5e-3

CASES = {
    "cbn": dict(shared=dict(num_classes=4), generator=dict(act_norm="batch")),
    "csm": dict(shared=dict(num_classes=3), generator=dict(act_norm="batch", z_to_bn=True)),
    "cbn_raw_classes": dict(shared=dict(num_classes=5, equalized=False),
                            generator=dict(act_norm="batch", shared_embedding_size=0), discriminator=dict(spectral=True)),
    # class scores with a time axis (layers.py:121-134); the reference's critic takes per-sample classes only
    "cbn_timed": dict(shared=dict(num_classes=4), generator=dict(act_norm="batch"), critic_without_classes=True),
    "csm_timed": dict(shared=dict(num_classes=3), generator=dict(act_norm="batch", z_to_bn=True),
                      critic_without_classes=True),
}


def _case(fx, name):
    return {k[len(name) + 1:]: fx[k] for k in fx if k.startswith(name + ".")}


def _models(case):
    from eeggan_amd import config
    return config.build_models(config.RANDOM_SEED, shared=case.get("shared"), generator=case.get("generator"),
                               discriminator=case.get("discriminator"))


def _functionals(fake, r1, score, gx, r2):
    l_g = (fake * r1).sum()
    l_d = (score * r2).sum() + 10.0 * ((gx.reshape(gx.shape[0], -1).norm(dim=1) - 1.0) ** 2).mean()
    return l_g, l_d


def _close(a, b):
    return rel(a, b) < 1e-9 or float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))) < 1e-12


def test_fixture_holds_every_case():
    assert list(load_fixture("conditional")["names"]) == list(CASES)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference(name):
    from oracle import ref_cpu as O
    fx = load_fixture("conditional")
    c, case = _case(fx, name), CASES[name]
    depth, alpha, seed = int(c["depth"]), float(c["alpha"]), int(c["seed"])
    g, d = _models(case)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(True)
    z, y, real, r1, r2 = (torch.from_numpy(c[k]).double() for k in ("z", "y", "real", "r1", "r2"))
    ge = case.get("generator", {})
    cfg = O.NetConfig(g_act_norm=ge.get("act_norm", "pixel"), z_to_bn=ge.get("z_to_bn", False))
    np.random.seed(seed)
    fake = O.generator_forward(gp, gs, z, depth, alpha, cfg, O.DropSource("numpy"), y=y)
    x_in = real.clone().requires_grad_(True)
    np.random.seed(seed + 1)
    score, _ = O.discriminator_forward(dp, ds, x_in, depth, alpha, cfg, O.DropSource("numpy"),
                                       y=None if case.get("critic_without_classes") else y)
    (gx,) = torch.autograd.grad(score.sum(), x_in, create_graph=True)
    l_g, l_d = _functionals(fake, r1, score, gx, r2)
    assert rel(fake, c["fake"]) < 1e-9 and rel(score, c["score"]) < 1e-9
    assert abs(float(l_g) - float(c["l_g"])) < 1e-9 * max(1.0, abs(float(c["l_g"])))
    assert abs(float(l_d) - float(c["l_d"])) < 1e-9 * max(1.0, abs(float(c["l_d"])))
    for tag, loss, params, net in (("g", l_g, gp, g), ("d", l_d, dp, d)):
        names = dict(net.named_parameters())
        for k, gr in zip(params.keys(), torch.autograd.grad(loss, list(params.values()), allow_unused=True)):
            if k not in names:
                continue  # (a buffer: running statistics, the vectors of a spectral norm)
            assert (gr is None) == (f"{tag}_none.{k}" in c) and (gr is None or _close(subsample(gr), c[f"{tag}_sub.{k}"])), k


def test_conditioning_needs_its_inputs():
    """A class-conditional norm without the classes, and a critic asked about classes it was built without, say so."""
    from eeggan_amd import config
    from eeggan_amd.layers import ConditionalBatchNorm
    norm = ConditionalBatchNorm(8, 4, 0, False)
    assert norm.mode == "CBN" and ConditionalBatchNorm(8, 4, 16, False).mode == "CSM"
    with pytest.raises(ValueError):
        norm.modulation(None, None)
    g, d = config.build_models(1, shared=dict(num_classes=3), generator=dict(act_norm="batch"))
    assert g.y_encoder is not None and tuple(d.class_emb.weight.shape) == (256, 3)
    assert "class_emb.weight" in d.state_dict() and "y_encoder.conv.conv.weight" in g.state_dict()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_against_reference_fixture(dev, name):
    """Fakes, scores and the two functionals 1e-4; gradients per tensor within max(1e-4, 2 x the reference's own
    fp32-vs-fp64 distance, min(2 x measured conditioning, 5e-3)) -- the gate of the other end-to-end fixtures."""
    from eeggan_amd import layers
    fx = load_fixture("conditional")
    c, case = _case(fx, name), CASES[name]
    depth, alpha, seed = int(c["depth"]), float(c["alpha"]), int(c["seed"])
    layers.set_drop_mode("numpy")
    g, d = _models(case)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    z, y, real, r1, r2 = (torch.from_numpy(c[k]).to(dev) for k in ("z", "y", "real", "r1", "r2"))
    moving = [(b, b.clone()) for net in (g, d) for b in net.buffers()]

    def evaluate(jitter=lambda t: t):
        with torch.no_grad():
            for b, saved in moving:
                b.copy_(saved)
        np.random.seed(seed)
        fake = g(jitter(z), y)[0]["x"]
        x_in = jitter(real).clone().requires_grad_(True)
        np.random.seed(seed + 1)
        score = d(x_in, None if case.get("critic_without_classes") else y)[0]
        (gx,) = torch.autograd.grad(score.sum(), x_in, create_graph=True)
        l_g, l_d = _functionals(fake, r1, score, gx, r2)
        grads = {"g": dict(zip([k for k, _ in g.named_parameters()],
                               torch.autograd.grad(l_g, list(g.parameters()), allow_unused=True))),
                 "d": dict(zip([k for k, _ in d.named_parameters()],
                               torch.autograd.grad(l_d, list(d.parameters()), allow_unused=True)))}
        return fake.detach(), score.detach(), float(l_g), float(l_d), grads

    fake, score, l_g, l_d, grads = evaluate()
    assert rel(fake, c["fake"]) < 1e-4 and rel(score, c["score"]) < 1e-4
    assert abs(l_g - float(c["l_g"])) < 1e-4 * max(1.0, abs(float(c["l_g"])))
    assert abs(l_d - float(c["l_d"])) < 1e-4 * max(1.0, abs(float(c["l_d"])))
    cond = {"d": {}, "g": {}}
    gen = torch.Generator(device=dev).manual_seed(seed)
    for _ in range(COND_TRIALS):
        pert = evaluate(lambda t: t * (1.0 + COND_EPS * torch.randn(t.shape, device=dev, generator=gen)))[4]
        for tag in ("d", "g"):
            for k, a in grads[tag].items():
                b = pert[tag][k]
                if a is not None and b is not None and float(a.norm()) > 0.0:
                    cond[tag][k] = max(cond[tag].get(k, 0.0), float((a - b).norm()) / float(a.norm()))
    failures, worst = [], 0.0
    for tag in ("d", "g"):
        for k, got in grads[tag].items():
            if f"{tag}_none.{k}" in c:
                assert got is None, f"{tag}.{k}: the reference leaves this gradient None"
                continue
            assert got is not None, f"{tag}.{k}: missing gradient"
            want, norm = c[f"{tag}_sub.{k}"], float(c[f"{tag}_norm.{k}"])
            if norm < 1e-12:
                continue
            noise = float(c.get(f"{tag}_noise.{k}", 0.0))
            gate = max(1e-4, NOISE_FACTOR * noise, min(2.0 * cond[tag].get(k, 0.0), COND_CAP))
            err = float(np.linalg.norm(subsample(got) - want)) / max(float(np.linalg.norm(want)), 1e-30)
            worst = max(worst, err / gate)
            if err >= gate:
                failures.append(f"{tag}.{k}: {err:.2e} >= {gate:.2e} (noise {noise:.1e}, conditioning {cond[tag].get(k, 0.0):.1e})")
    print(f"{name}: worst gradient error / gate = {worst:.2f}")
    assert not failures, "\n".join(failures)
