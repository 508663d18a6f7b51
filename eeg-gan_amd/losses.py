"""``discriminator_loss`` / ``generator_loss`` / ``calc_grad`` with the signatures of the
reference's ``losses.py`` (35-37, 40-70, 73-120) for the default ``wgan_gp`` loss.

All reductions (means over the batch, the per-sample gradient norm and the penalty) run in
fused HIP kernels; the gradient penalty differentiates through the discriminator's input
gradient exactly as the reference does (``autograd.grad(..., create_graph=True)``).
"""
from __future__ import annotations

import contextlib
import random
from typing import Optional

import torch
from torch.autograd import grad

from . import layers, ops
from .utils import cudize

one = None
mixing_factors = None

def get_mixing_factor(batch_size):
    """A cached [B, 1, 1] buffer refilled with U[0, 1) at every call (reference losses.py:13-18)."""
    global mixing_factors
    if mixing_factors is None or batch_size != mixing_factors.size(0):
        mixing_factors = cudize(torch.empty(batch_size, 1, 1, dtype=torch.float32))
    mixing_factors.uniform_()
    return mixing_factors


def get_one(batch_size):
    global one
    if one is None or batch_size != one.size(0):
        one = cudize(torch.ones(batch_size))
    return one


zero = None


def get_zero(batch_size):
    global zero
    if zero is None or batch_size != zero.size(0):
        zero = cudize(torch.zeros(batch_size))
    return zero


def _stacked(real_x, fake_x):
    """[reals; fakes] as one batch: without a copy when the two already lie back to back in one buffer (a graph
    trainer keeps its static inputs that way), a concatenation otherwise."""
    if (real_x.shape == fake_x.shape and real_x.dtype == fake_x.dtype and real_x.is_contiguous()
            and fake_x.is_contiguous() and not real_x.requires_grad and not fake_x.requires_grad
            and real_x.untyped_storage().data_ptr() == fake_x.untyped_storage().data_ptr()
            and fake_x.storage_offset() == real_x.storage_offset() + real_x.numel()):
        return real_x.as_strided((2 * real_x.shape[0],) + tuple(real_x.shape[1:]), real_x.stride(),
                                 real_x.storage_offset())
    return torch.cat([real_x, fake_x])


def calc_grad(x_hat, pred_hat):
    """d(sum pred_hat)/d(x_hat) with a graph, so it can be differentiated again.  Weight
    gradients are not formed in this pass (only the input gradient is asked for)."""
    with ops.input_grad_only():
        return grad(outputs=pred_hat, inputs=x_hat, grad_outputs=get_one(pred_hat.size(0)), create_graph=True,
                    retain_graph=True, only_inputs=True)[0]


@contextlib.contextmanager
def _frozen(module: torch.nn.Module):
    """Inside: the module's parameters do not require grad, so no weight gradients are formed for them."""
    params = [p for p in module.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


LOSS_TYPES = ("hinge", "rsgan", "rasgan", "rahinge", "wgan_gp", "wgan_theirs")


def _batch_coupled(dis) -> bool:
    """Batch statistics couple the samples of a pass: such a critic is given reals, fakes and interpolates one after
    the other, as the reference does (losses.py:76-79, 107-111), never stacked."""
    return any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in dis.modules())


def _no_dropped_classes(dis):
    """The critic losses hand the critic signals only (``{'x': ...}`` for fakes, interpolates and the stacked
    pass), so a critic built with ``num_classes`` would silently lose its projection term.  The reference's losses
    cannot train such a critic either (they hand its forward whole dicts, losses.py:76-79, 108-113 against
    network.py:528); refuse instead of training it as an unconditional one."""
    if getattr(dis, "class_emb", None) is not None:
        raise NotImplementedError("discriminator_loss runs the critic without classes: a critic built with "
                                  "num_classes != 0 would be trained without its projection term")


class PartsUnavailable(ValueError):
    """The critic loss was asked for one of its independent parts but does not split for this configuration."""


def generator_loss(dis: torch.nn.Module, gen: torch.nn.Module, real, z, loss_type: str, random_multiply: bool,
                   feature_matching_lambda: float = 0.0, skip_dis_weight_grads: bool = True):
    """-mean(D(G(z))) for ``hinge`` and the ``wgan`` family; the relativistic losses (``rsgan``, ``rasgan``,
    ``rahinge``) compare with the critic's scores of the reals, evaluated without a graph after the fakes' pass
    (reference losses.py:40-60).  ``skip_dis_weight_grads`` (default on) leaves ``dis`` parameters' ``.grad``
    untouched: the reference accumulates them here only to zero them at the next discriminator
    step (losses.py:75), so they are never used."""
    if loss_type not in LOSS_TYPES and not loss_type.startswith("wgan"):
        raise ValueError("Invalid loss type")
    layers.drop_pool_reset()
    gen.zero_grad()
    g_, _ = gen(z)
    with (_frozen(dis) if skip_dis_weight_grads else contextlib.nullcontext()):
        d_fake, fake_features, _ = dis(g_)
    real_features = None
    # ``random_multiply``: the loss times a uniform draw from Python's generator (losses.py:46, 70), drawn here,
    # where the reference draws it
    scale = random.random() if random_multiply else 1.0
    if loss_type == "hinge" or loss_type.startswith("wgan"):
        g_loss = ops.mean_scale(d_fake, -1.0)
    else:
        with torch.no_grad():
            d_real, real_features, _ = dis(real)
        # the generator's relativistic losses are the critic's with the roles of the two score vectors swapped
        g_loss = ops.score_loss(loss_type, d_fake.reshape(-1), d_real.reshape(-1))
    if feature_matching_lambda != 0.0:
        # (like the reference, losses.py:62-69, the term enters with weight one whatever the non-zero lambda is)
        if real_features is None:
            with torch.no_grad():
                _, real_features, _ = dis(real)
        g_loss = g_loss + ops.feature_match(real_features, fake_features)
    return g_loss * scale if random_multiply else g_loss


def discriminator_loss(dis: torch.nn.Module, gen: torch.nn.Module, real, z, loss_type: str,
                       iwass_drift_epsilon: float, grad_lambda: float, iwass_target: float,
                       mixing: Optional[torch.Tensor] = None, stack_real_fake: Optional[bool] = None,
                       fake: Optional[torch.Tensor] = None, part: Optional[str] = None,
                       zero_grad: bool = True):
    """WGAN-GP critic loss.  ``mixing`` optionally supplies the per-sample interpolation factors
    [B, 1, 1] (otherwise drawn like the reference does); used by parity tests.  ``stack_real_fake``:
    evaluate reals and fakes in one stacked critic pass (default: when GDrop noise is drawn on the device).
    ``fake``: fakes generated beforehand from ``z`` with the current generator (a trainer may generate
    the fakes of all critic iterations of a step in one pass: the generator does not change between
    them, reference trainer.py:83-91); ``gen`` is then not called.  ``part`` ('scores' | 'penalty', stacked mode
    only): return only -mean D(real) + mean D(fake) + drift term, or only the gradient penalty; their sum is
    the loss ('real' and 'fake' split the score term once more, into a critic pass each).  ``zero_grad=False`` leaves the critic's gradients alone (the second part of a split
    iteration)."""
    if loss_type not in LOSS_TYPES:
        raise ValueError("Invalid loss type")
    if part not in (None, "scores", "penalty", "real", "fake"):
        raise ValueError(f"part={part!r}")
    if loss_type != "wgan_gp":
        return _other_discriminator_loss(dis, gen, real, z, loss_type, iwass_drift_epsilon, grad_lambda, iwass_target,
                                         mixing, stack_real_fake, fake, part, zero_grad)
    _no_dropped_classes(dis)
    layers.drop_pool_reset()
    if zero_grad:
        dis.zero_grad()
    real_x = real["x"] if isinstance(real, dict) else real
    if stack_real_fake is None:
        stack_real_fake = layers.get_drop_mode() == "device" and not _batch_coupled(dis)
    if part is not None and grad_lambda == 0:
        raise PartsUnavailable("without a penalty (grad_lambda == 0) the critic loss is one term")
    if part is not None and not stack_real_fake:
        if _batch_coupled(dis):
            raise PartsUnavailable("a critic with batch norm layers sees reals, fakes and interpolates as batches of "
                                   "their own")
        raise ValueError("part needs the stacked critic pass")
    if stack_real_fake:
        # one critic pass over [reals; fakes]: half the launches, twice the work per launch, and one
        # gradient contribution per weight instead of two.  The reference evaluates the two passes
        # one after the other (losses.py:76-79); the result differs only in which GDrop draw a
        # sample meets, so the stacked pass is used when the noise comes from the device generator
        # and the sequential order when it follows the reference's host RNG stream.
        if fake is None:
            with torch.no_grad():
                g_, _ = gen(z)
            fake_x = g_["x"]
        else:
            fake_x = fake
        n = real_x.size(0)
        # The loss is a sum of a term of the stacked pass alone and a term of the interpolates' pass alone
        # (reference losses.py:94-100 and 106-119), so the two passes and their backward passes are
        # independent: ``part`` asks for one of them (a graph trainer runs them side by side on two streams).
        if part in ("real", "fake"):  # the two halves of the score term, one critic pass each
            score, _, _ = dis({"x": real_x if part == "real" else fake_x})
            nothing, zeros = get_zero(n).view(n, 1), get_zero(n)
            if part == "real":
                return ops.wgan_gp_d_loss(score, zeros, nothing, iwass_drift_epsilon, 0.0, iwass_target)
            return ops.wgan_gp_d_loss(zeros, score, nothing, 0.0, 0.0, iwass_target)
        if part in (None, "scores"):
            with layers.batch_segments(2):
                scores, _, _ = dis({"x": _stacked(real_x, fake_x)})
            if part == "scores" or grad_lambda == 0:
                # (the reference skips the interpolates' pass without a penalty, losses.py:106)
                nothing = get_zero(n).view(n, 1)  # no penalty: lambda = 0 on an all-zero "gradient"
                return ops.wgan_gp_d_loss_stacked(scores, nothing, iwass_drift_epsilon, 0.0, iwass_target)
        alpha = get_mixing_factor(n) if mixing is None else mixing
        x_hat = {"x": ops.mix_batch_raw(real_x.detach(), fake_x.detach(), alpha).requires_grad_(True)}
        pred_hat, _, _ = dis(x_hat)
        g = calc_grad(x_hat["x"], pred_hat).view(n, -1)
        if part == "penalty":
            return ops.wgan_gp_d_loss(get_zero(n), get_zero(n), g, 0.0, grad_lambda, iwass_target)
        return ops.wgan_gp_d_loss_stacked(scores, g, iwass_drift_epsilon, grad_lambda, iwass_target)
    else:
        d_real, _, _ = dis(real)
        if fake is None:
            with torch.no_grad():
                g_, _ = gen(z)
            fake_x = g_["x"]
        else:
            fake_x = fake
        d_fake, _, _ = dis({"x": fake_x})
        if grad_lambda == 0:
            n = real_x.size(0)
            return ops.wgan_gp_d_loss(d_real, d_fake, get_zero(n).view(n, 1), iwass_drift_epsilon, 0.0, iwass_target)
    alpha = get_mixing_factor(real_x.size(0)) if mixing is None else mixing
    x_hat = {"x": ops.mix_batch_raw(real_x.detach(), fake_x.detach(), alpha).requires_grad_(True)}
    pred_hat, _, _ = dis(x_hat)
    g = calc_grad(x_hat["x"], pred_hat).view(real_x.size(0), -1)
    return ops.wgan_gp_d_loss(d_real, d_fake, g, iwass_drift_epsilon, grad_lambda, iwass_target)


def _other_discriminator_loss(dis, gen, real, z, loss_type, drift, grad_lambda, target, mixing, stack_real_fake, fake,
                              part, zero_grad):
    """The reference's other critic losses (losses.py:82-101): a term of the two score vectors, plus the gradient
    penalty on the interpolates when ``grad_lambda`` is not zero (two-sided like wgan_gp's; ``wgan_theirs`` takes
    the one-sided form and scales it by relu(mean D(real) - mean D(fake)), losses.py:96-98, 115-117).  Pass
    structure, stacking and ``part`` as in ``discriminator_loss``; ``wgan_theirs`` does not split (its penalty
    is scaled by a function of the scores) and neither does a loss without penalty."""
    if part in ("real", "fake") or (part is not None and (loss_type == "wgan_theirs" or grad_lambda == 0)):
        raise PartsUnavailable(f"loss_type={loss_type!r}, grad_lambda={grad_lambda}: part={part!r} is not available")
    _no_dropped_classes(dis)
    layers.drop_pool_reset()
    if zero_grad:
        dis.zero_grad()
    real_x = real["x"] if isinstance(real, dict) else real
    n = real_x.size(0)
    if stack_real_fake is None:
        stack_real_fake = layers.get_drop_mode() == "device" and not _batch_coupled(dis)
    if part is not None and not stack_real_fake:
        if _batch_coupled(dis):
            raise PartsUnavailable("a critic with batch norm layers sees reals, fakes and interpolates as batches of "
                                   "their own")
        raise ValueError("part needs the stacked critic pass")
    d_real = d_fake = None
    if stack_real_fake:
        if fake is None:
            with torch.no_grad():
                fake_x = gen(z)[0]["x"]
        else:
            fake_x = fake
        if part != "penalty":
            with layers.batch_segments(2):
                scores, _, _ = dis({"x": _stacked(real_x, fake_x)})
            scores = scores.reshape(-1)
            d_real, d_fake = scores[:n], scores[n:]
    else:  # the reference's order of passes (and of noise draws): D(real), G(z), D(fake), D(x_hat)
        d_real = dis(real)[0].reshape(-1)
        if fake is None:
            with torch.no_grad():
                fake_x = gen(z)[0]["x"]
        else:
            fake_x = fake
        d_fake = dis({"x": fake_x})[0].reshape(-1)
    loss = gain = None
    if part != "penalty":
        if loss_type == "wgan_theirs":
            loss = ops.score_loss("wgan_theirs", d_real, d_fake, drift)
            gain = ops.score_loss("gain", d_real, d_fake)
        else:
            loss = ops.score_loss(loss_type, d_real, d_fake)
        if part == "scores":
            return loss
    if grad_lambda == 0:
        return loss
    if gain is not None and layers.get_drop_mode() != "device" and float(gain) == 0.0:
        # the reference skips the interpolates' pass when the gain is zero (losses.py:103), and with it that
        # pass's noise draws from the host generator: follow it where the draws follow the reference's stream
        return loss
    alpha = get_mixing_factor(n) if mixing is None else mixing
    x_hat = {"x": ops.mix_batch_raw(real_x.detach(), fake_x.detach(), alpha).requires_grad_(True)}
    pred_hat, _, _ = dis(x_hat)
    g = calc_grad(x_hat["x"], pred_hat).view(n, -1)
    penalty = ops.grad_penalty(g, grad_lambda, target, one_sided=(loss_type == "wgan_theirs"))
    if part == "penalty":
        return penalty
    return loss + (penalty if gain is None else gain * penalty)
