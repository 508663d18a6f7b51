"""Sparse-row tables for the rational resampler.

The reference resamples by ``F.interpolate(scale_factor=p, mode='linear', align_corners=False)``
followed by ``F.avg_pool1d(q, q)`` (reference utils.py:204-251), materialising a p-times longer
signal in between.  Both steps are linear and fixed for a given (length, p, q), so their
composition is a sparse matrix with a handful of taps per output sample; the HIP kernel applies
that matrix (and, for the backward pass, its transpose) in one gather pass.
"""
from __future__ import annotations

from fractions import Fraction
from typing import Dict, Tuple

import numpy as np
import torch


def composite_rows(t_in: int, p: int, q: int):
    """List over output positions m of {input index: weight} for 'interpolate x p then pool / q'.

    The sampling position and the interpolation weight are computed in float32 exactly as the
    upsampling kernel of the reference framework does (scale = 1/p rounded to float32,
    src = scale * (j + 0.5) - 0.5 clamped below at 0, upper neighbour clamped to the last sample).
    """
    if t_in == 1:
        ratio = Fraction(p, q)
        if ratio.denominator != 1:
            raise ValueError("a length-1 signal can only be repeated an integer number of times")
        return [{0: 1.0} for _ in range(ratio.numerator)]
    rows = []
    t_up = t_in * p
    t_out = t_up // q
    scale = np.float32(1.0 / p)
    for m in range(t_out):
        acc: Dict[int, float] = {}
        for j in range(m * q, m * q + q):
            if p == 1:
                acc[j] = acc.get(j, 0.0) + 1.0 / q
                continue
            src = np.float32(scale * np.float32(j + 0.5) - np.float32(0.5))
            if src < 0:
                src = np.float32(0.0)
            i0 = int(np.floor(src))
            i1 = min(i0 + 1, t_in - 1)
            lam = float(np.float32(src - np.float32(i0)))
            acc[i0] = acc.get(i0, 0.0) + (1.0 - lam) / q
            acc[i1] = acc.get(i1, 0.0) + lam / q
        rows.append(acc)
    return rows


def _pack(rows, n_cols_unused=None):
    taps = max(1, max(len(r) for r in rows))
    idx = np.zeros((len(rows), taps), dtype=np.int32)
    coef = np.zeros((len(rows), taps), dtype=np.float32)
    for m, r in enumerate(rows):
        for k, (i, c) in enumerate(sorted(r.items())):
            idx[m, k] = i
            coef[m, k] = c
    return idx, coef, taps


def build_tables(t_in: int, p: int, q: int):
    """(forward, transpose) tables as numpy arrays: each is (idx[int32], coef[float32], Tin, Tout, taps)."""
    rows = composite_rows(t_in, p, q)
    t_out = len(rows)
    idx, coef, taps = _pack(rows)
    trans = [dict() for _ in range(t_in)]
    for m, r in enumerate(rows):
        for i, c in r.items():
            trans[i][m] = trans[i].get(m, 0.0) + c
    tidx, tcoef, ttaps = _pack(trans)
    return (idx, coef, t_in, t_out, taps), (tidx, tcoef, t_out, t_in, ttaps)


_cache: Dict[Tuple, Tuple] = {}


def device_tables(t_in: int, p: int, q: int, device: torch.device):
    """Device-resident tables, built and uploaded once per (length, ratio, device)."""
    fr = Fraction(p, q)
    key = (t_in, fr.numerator, fr.denominator, str(device))
    hit = _cache.get(key)
    if hit is None:
        f, b = build_tables(t_in, fr.numerator, fr.denominator)
        up = lambda t: (torch.from_numpy(t[0]).to(device), torch.from_numpy(t[1]).to(device), t[2], t[3], t[4])
        hit = (up(f), up(b))
        _cache[key] = hit
    return hit
