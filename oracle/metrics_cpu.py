"""CPU oracle for the evaluation side of the path: the 1-D sliced Wasserstein distance, the CPC feature
network's inference pass and the Frechet distance on its features.

TEST INFRASTRUCTURE ONLY (like ``oracle/ref_cpu.py``): a CPU restatement, in stock numpy / PyTorch ops, of
what the reference computes; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.

Parity status: PINNED.  ``tools/make_golden_metrics.py`` imports the reference's ``plugins.SlicedWDistance``,
``plugins.FidCalculator`` and ``cpc.cpc_network.Network`` in the build container, runs them on seeded inputs
and checks these functions against them before writing ``tests/golden/metrics.npz``;
``tests/test_oracle_golden.py`` re-checks wherever the tests run.

Reference lines are cited per function (paths relative to the reference repository root)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

EPSILON = 1e-8  # utils.py:14


# --------------------------------------------------------------------------- sliced Wasserstein distance
def swd_sliced_wasserstein(A: torch.Tensor, B: torch.Tensor, dir_repeats: int = 4, dirs_per_repeat: int = 128,
                           log: List[torch.Tensor] = None) -> float:
    """plugins.py:383-395.  Directions come from torch's global CPU generator, one [c, dirs] draw per repeat."""
    results = []
    for _ in range(dir_repeats):
        dirs = torch.randn(A.shape[1], dirs_per_repeat)
        dirs = dirs / torch.sqrt((dirs * dirs).sum(dim=0, keepdim=True) + EPSILON)
        if log is not None:
            log.append(dirs.clone())
        pa = torch.sort(torch.matmul(A, dirs.to(A.dtype)), dim=0)[0]
        pb = torch.sort(torch.matmul(B, dirs.to(B.dtype)), dim=0)[0]
        results.append((pa - pb).abs().mean())
    return float(torch.mean(torch.stack(results)))


def swd_descriptors(batch: torch.Tensor, items: int, patches_per_item: int, patch_size: int,
                    starts_log: List[np.ndarray] = None) -> torch.Tensor:
    """plugins.py:438-447 for one batch at its current decimation: ``patches_per_item`` patches per item at
    positions drawn one by one from numpy's global generator, stacked as [N, c, patch], READ as rows of c
    numbers (a plain reshape, not a transpose), then centred and divided by the unbiased standard deviation
    per column."""
    c = batch.shape[1]
    max_index = batch.shape[2] - patch_size
    patches, starts = [], []
    for j in range(items):
        for _ in range(patches_per_item):
            r = np.random.randint(0, max_index)
            starts.append(r)
            patches.append(batch[j, :, r:r + patch_size])
    if starts_log is not None:
        starts_log.append(np.asarray(starts, dtype=np.int32))
    d = torch.stack(patches, dim=0).reshape((-1, c))
    d = d - torch.mean(d, dim=0, keepdim=True)
    return d / (torch.std(d, dim=0, keepdim=True) + EPSILON)


def swd_get_descriptors(batch1: torch.Tensor, batch2: torch.Tensor, progression_scale: int, patches_per_item: int = 16,
                        patch_size: int = 49, dir_repeats: int = 4, dirs_per_repeat: int = 128) -> List[float]:
    """plugins.py:426-451: one distance per "pyramid level"; the number of levels follows ``t //= scale`` while
    ``t >= patch_size``.  As written, the reference rebuilds its list of batches from the ORIGINAL ones at the
    top of every level (plugins.py:436) and decimates them only after taking the patches (plugins.py:449), so
    the decimated signals are never used: every level draws fresh patches from the full-rate batches."""
    b, _, t = batch1.shape
    levels = 0
    while t >= patch_size:
        levels += 1
        t //= progression_scale
    out = []
    for _ in range(levels):
        descs = [swd_descriptors(batch, b, patches_per_item, patch_size) for batch in (batch1, batch2)]
        out.append(swd_sliced_wasserstein(descs[0], descs[1], dir_repeats, dirs_per_repeat))
    return out


# --------------------------------------------------------------------------- CPC feature network, inference
def cpc_conv_encoder(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "encoder.") -> torch.Tensor:
    """cpc_network.py:137-158 in eval mode: three strided convs (kernel 2s-1, padding s-1 for s = 5, 4, 3), each
    followed by ReLU, the first two also by (inactive) dropout and BatchNorm1d on its running statistics."""
    h = x
    for conv_i, bn_i, stride in ((0, 3, 5), (4, 7, 4), (8, None, 3)):
        w, b = sd[f"{prefix}{conv_i}.weight"], sd[f"{prefix}{conv_i}.bias"]
        h = F.relu(F.conv1d(h, w, b, stride=stride, padding=(w.shape[2] - 1) // 2))
        if bn_i is not None:
            h = F.batch_norm(h, sd[f"{prefix}{bn_i}.running_mean"], sd[f"{prefix}{bn_i}.running_var"],
                             sd[f"{prefix}{bn_i}.weight"], sd[f"{prefix}{bn_i}.bias"], training=False, eps=1e-5)
    return h


def cpc_pnorm_pool(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str) -> torch.Tensor:
    """cpc_network.py:161-188 with p = [1, inf]: the two norms over time, concatenated, through
    Linear -> ReLU -> [BatchNorm1d (eval)] -> Linear.  The normalisation is there for the z pooler only: the
    network passes False in its place for the c pooler (cpc_network.py:296)."""
    n = torch.cat([x.abs().sum(dim=2), x.abs().amax(dim=2)], dim=1)
    h = F.relu(F.linear(n, sd[f"{prefix}network.0.weight"], sd[f"{prefix}network.0.bias"]))
    last = 2
    if f"{prefix}network.2.running_mean" in sd:
        h = F.batch_norm(h, sd[f"{prefix}network.2.running_mean"], sd[f"{prefix}network.2.running_var"],
                         sd[f"{prefix}network.2.weight"], sd[f"{prefix}network.2.bias"], training=False, eps=1e-5)
        last = 3
    return F.linear(h, sd[f"{prefix}network.{last}.weight"], sd[f"{prefix}network.{last}.bias"])


def cpc_gru(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "contextualizer.forward_rnn.") -> torch.Tensor:
    """cpc_network.py:191-216: one-layer GRU over time from a zero state (torch.nn.GRU's equations written
    out), input and output [B, C, T]."""
    w_ih, w_hh = sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"]
    b_ih, b_hh = sd[prefix + "bias_ih_l0"], sd[prefix + "bias_hh_l0"]
    B, C, T = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H, dtype=x.dtype)
    out = []
    for t in range(T):
        gi = F.linear(x[:, :, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=2)


def cpc_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor):
    """Network.forward(x, no_loss=True), cpc_network.py:316-322: (z, c, z_pooled, c_pooled)."""
    z = cpc_conv_encoder(sd, x)
    z_p = cpc_pnorm_pool(sd, z, "z_pooler.")
    c = cpc_gru(sd, z)
    c_p = cpc_pnorm_pool(sd, c, "c_pooler.")
    return z, c, z_p, c_p


# --------------------------------------------------------------------------- Frechet distance
def torch_cov(m: torch.Tensor) -> torch.Tensor:
    """plugins.py:503-514 with rowvar=False: unbiased covariance of the columns of m [N, F]."""
    m = m.t()
    m = m - torch.mean(m, dim=1, keepdim=True)
    return (1.0 / (m.size(1) - 1)) * m.matmul(m.t())


def sqrt_newton_schulz(A: torch.Tensor, iters: int) -> torch.Tensor:
    """plugins.py:540-555 for one matrix."""
    n = A.shape[0]
    norm = A.mul(A).sum().sqrt()
    Y, eye, Z = A / norm, torch.eye(n, dtype=A.dtype), torch.eye(n, dtype=A.dtype)
    for _ in range(iters):
        T = 0.5 * (3.0 * eye - Z.mm(Y))
        Y = Y.mm(T)
        Z = T.mm(Z)
    return Y * torch.sqrt(norm)


def frechet_distance(mu1, sigma1, mu2, sigma2) -> torch.Tensor:
    """plugins.py:565-572."""
    diff = mu1 - mu2
    covmean = sqrt_newton_schulz(sigma1.mm(sigma2), 50)
    return diff.dot(diff) + torch.trace(sigma1) + torch.trace(sigma2) - 2 * torch.trace(covmean)


def numpy_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """The reference's second method (plugins.py:517-537): scipy's sqrtm of the covariance product on the host,
    with its retry on a non-finite root and its treatment of a complex result."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(np.asarray(mu1, dtype=np.float64)), np.atleast_1d(np.asarray(mu2, dtype=np.float64))
    sigma1, sigma2 = np.atleast_2d(np.asarray(sigma1, dtype=np.float64)), np.atleast_2d(np.asarray(sigma2, dtype=np.float64))
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-1):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)
