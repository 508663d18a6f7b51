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
