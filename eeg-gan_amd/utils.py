"""Host-side helpers with the names and call conventions of the reference's ``utils.py``
for the functions the training step uses (reference utils.py:41-49,116-136,204-251)."""
from __future__ import annotations

from fractions import Fraction
from typing import TypeVar

import numpy as np
import torch

from . import ops
from .resample_tables import device_tables

EPSILON = 1e-8

T = TypeVar("T")


def cudize(thing: T) -> T:
    """Move a tensor / list / dict of tensors to the current HIP device (reference utils.py:116-126)."""
    if thing is None:
        return None
    if not torch.cuda.is_available():
        return thing
    if isinstance(thing, (list, tuple)):
        return [item.cuda() for item in thing]
    if isinstance(thing, dict):
        return {k: v.cuda() for k, v in thing.items()}
    return thing.cuda()


def trainable_params(model):
    return filter(lambda p: p.requires_grad, model.parameters())


def num_params(net):
    return sum(int(np.prod(p.size())) for p in trainable_params(net))


def random_latents(num_latents, latent_size, z_distribution="normal"):
    """reference utils.py:41-49; only the default distribution is on the accelerated path."""
    if z_distribution != "normal":
        raise NotImplementedError(f"z_distribution={z_distribution!r}: only 'normal' is supported")
    return torch.randn(num_latents, latent_size)


def pixel_norm(h):
    """h * rsqrt(mean over channels of h*h + 1e-8) (reference utils.py:133-136)."""
    return ops.pixel_norm(h)


def expand3d(signal):
    if signal.dim() == 2:
        return signal.unsqueeze(2)
    if signal.dim() == 1:
        return signal.unsqueeze(1).unsqueeze(2)
    return signal


def resample_signal(signal, signal_freq, desired_freq, pytorch=True):
    """Resample [B, C, T] by desired_freq/signal_freq: linear interpolation by the numerator,
    average pooling by the denominator, a length-1 signal is repeated (reference utils.py:221-251).
    Runs as one fused gather kernel; differentiable to any order (the op is linear)."""
    if not pytorch or isinstance(signal, np.ndarray):
        raise NotImplementedError("resample_signal: only device tensors are supported (pytorch=True)")
    if signal.dim() != 3:
        raise NotImplementedError("resample_signal: expected a [B, C, T] tensor")
    if isinstance(desired_freq, float) or isinstance(signal_freq, float):
        if desired_freq != int(desired_freq) or signal_freq != int(signal_freq):
            raise NotImplementedError("resample_signal: non-integer frequencies are not supported")
    ratio = Fraction(int(desired_freq), int(signal_freq))
    if ratio == 1:
        return signal
    fwd, bwd = device_tables(signal.shape[2], ratio.numerator, ratio.denominator, signal.device)
    return ops._Gather.apply(signal, fwd, bwd)


def upsample_signal(signal, upsample_factor):
    return resample_signal(signal, 1, upsample_factor)


def downsample_signal(signal, downsample_factor):
    return resample_signal(signal, downsample_factor, 1)
