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


QUAD_MAX_NV = 4


def _pack_quads(rows, t_in: int):
    """Quad form of a sparse-row table (see ``eeggan_resample_quads``): per 4 consecutive output
    positions the first input sample of their span (a multiple of 4, moved left at the right edge so
    that the span stays inside the row) and a dense 4 x 4nv block of weights, nv the widest span of the
    table in 16-byte vectors.  None when the operator is not local enough or a length is no multiple of 4."""
    t_out = len(rows)
    if t_out % 4 or t_in % 4 or t_out == 0:
        return None
    spans = []
    for q in range(t_out // 4):
        keys = [i for r in rows[4 * q:4 * q + 4] for i in r]
        lo, hi = (min(keys), max(keys)) if keys else (0, 0)
        spans.append((lo // 4 * 4, hi))
    nv = max((hi - b) // 4 + 1 for b, hi in spans)
    if nv > QUAD_MAX_NV or t_in < 4 * nv:
        return None
    base = np.zeros(t_out // 4, dtype=np.int32)
    coef = np.zeros((t_out // 4, 4, 4 * nv), dtype=np.float32)
    for q, (b, hi) in enumerate(spans):
        b = min(b, t_in - 4 * nv)
        assert b >= 0 and b % 4 == 0 and hi < b + 4 * nv
        base[q] = b
        for e in range(4):
            for i, c in rows[4 * q + e].items():
                coef[q, e, i - b] = c
    return base, coef, nv


QUAD_LANES = 64


def interleave_quads(coef: np.ndarray) -> np.ndarray:
    """The dense blocks [quads, 4, 4 nv] in the order ``eeggan_resample_quads`` reads them: groups of 64
    consecutive quads, and inside a group [output e][vector v][quad % 64][4 floats], so that the 64 lanes of a
    wavefront (consecutive quads) fetch one 16-byte vector each from 1 KB in one piece.  The last group is padded
    with zeros.  (Block by block -- a lane reading its own 64..256 bytes -- every load instruction touched 64
    cache lines and the table, 60-90 KB, was fetched again and again: that, not the signal, bounded the kernel.)"""
    nq, four, width = coef.shape
    nv = width // 4
    groups = (nq + QUAD_LANES - 1) // QUAD_LANES
    padded = np.zeros((groups * QUAD_LANES, 4, nv, 4), dtype=np.float32)
    padded[:nq] = coef.reshape(nq, 4, nv, 4)
    return np.ascontiguousarray(padded.reshape(groups, QUAD_LANES, 4, nv, 4).transpose(0, 2, 3, 1, 4)).reshape(-1)


def affine_base(base: np.ndarray, t_in: int, nv: int):
    """(A, C, D) with ``base[q] == min(max(A q + C, 0) // D // 4 * 4, t_in - 4 nv)`` for every quad, or None.

    The first tap of a quad advances by the resampling ratio, so its 16-byte aligned start is an integer-affine
    function of the quad index, clamped at both ends of the row.  With the closed form the kernel needs no table
    look-up before it can issue its loads (``eeggan_resample_quads_affine_t``); a table the form does not reproduce
    exactly keeps the look-up."""
    from math import gcd
    nq = len(base)
    lim = t_in - 4 * nv
    if lim <= 0 or nq == 0:
        return (0, 0, 1) if (base == 0).all() else None
    g = gcd(4 * t_in, 4 * nq)
    q = np.arange(nq, dtype=np.int64)
    b = base.astype(np.int64)
    for k in (1, 2, 4, 8):
        A, D = 4 * t_in // g * k, 4 * nq // g * k
        lo, hi = -(1 << 40), 1 << 40
        inner = (b > 0) & (b < lim)
        if inner.any():
            lo = max(lo, int((b[inner] * D - A * q[inner]).max()))
            hi = min(hi, int(((b[inner] + 4) * D - 1 - A * q[inner]).min()))
        left, right = b == 0, b == lim
        if left.any():
            hi = min(hi, int((4 * D - 1 - A * q[left]).min()))
        if right.any():
            lo = max(lo, int((lim * D - A * q[right]).max()))
        if lo > hi:
            continue
        C = lo if hi >= 1 << 40 else (hi if lo <= -(1 << 40) else (lo + hi) // 2)
        got = np.minimum(np.maximum(A * q + C, 0) // D // 4 * 4, lim)
        if np.array_equal(got, b) and abs(C) < 1 << 30 and A * nq + abs(C) < 1 << 31:
            return int(A), int(C), int(D)
    return None


def build_tables(t_in: int, p: int, q: int):
    """(forward, transpose) tables as numpy arrays: each is (idx[int32], coef[float32], Tin, Tout, taps,
    quads) with quads = (base[int32], coef[float32], nv) or None."""
    rows = composite_rows(t_in, p, q)
    t_out = len(rows)
    idx, coef, taps = _pack(rows)
    trans = [dict() for _ in range(t_in)]
    for m, r in enumerate(rows):
        for i, c in r.items():
            trans[i][m] = trans[i].get(m, 0.0) + c
    tidx, tcoef, ttaps = _pack(trans)
    return ((idx, coef, t_in, t_out, taps, _pack_quads(rows, t_in)),
            (tidx, tcoef, t_out, t_in, ttaps, _pack_quads(trans, t_out)))


_cache: Dict[Tuple, Tuple] = {}


def device_tables(t_in: int, p: int, q: int, device: torch.device):
    """Device-resident tables, built and uploaded once per (length, ratio, device); the quad coefficients in the
    kernel's lane-interleaved order (``interleave_quads``), followed by the closed form of the quad starts
    (``affine_base``) or None."""
    fr = Fraction(p, q)
    key = (t_in, fr.numerator, fr.denominator, str(device))
    hit = _cache.get(key)
    if hit is None:
        f, b = build_tables(t_in, fr.numerator, fr.denominator)
        dv = lambda a: torch.from_numpy(a).to(device)
        up = lambda t: (dv(t[0]), dv(t[1]), t[2], t[3], t[4],
                        None if t[5] is None else (dv(t[5][0]), dv(interleave_quads(t[5][1])), t[5][2],
                                                   affine_base(t[5][0], t[2], t[5][2])))
        hit = (up(f), up(b))
        _cache[key] = hit
    return hit
