#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics (tests/golden/metrics.npz) from the REFERENCE implementation.

Runs only in the build container (reference mounted read-only at /root/reference); stores numbers only.  The
two modules the reference imports that the image lacks (`pytorch_pretrained_bert`, `imageio`) are replaced by
empty in-process stubs, as in tools/make_golden.py; nothing on these paths touches them.  Every reference
result is also reproduced with the CPU restatement (oracle/metrics_cpu.py) before it is written.

Usage: python tools/make_golden_metrics.py [--out tests/golden]"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def import_reference():
    bert = types.ModuleType("pytorch_pretrained_bert")
    bert.modeling = types.ModuleType("pytorch_pretrained_bert.modeling")
    bert.modeling.BertLayer = object
    bert.BertAdam = object
    sys.modules["pytorch_pretrained_bert"] = bert
    sys.modules["pytorch_pretrained_bert.modeling"] = bert.modeling
    imageio = types.ModuleType("imageio")
    imageio.imwrite = lambda *a, **k: None
    sys.modules["imageio"] = imageio
    sys.path.insert(0, REF)
    import plugins as ref_plugins  # noqa
    from cpc import cpc_network as ref_cpc  # noqa
    sys.path.remove(REF)
    return ref_plugins, ref_cpc


ref_plugins, ref_cpc = import_reference()
sys.path.insert(0, REPO)
from oracle import metrics_cpu as M  # noqa: E402


def swd_fixture(fx):
    gen = torch.Generator().manual_seed(11)
    b, c, T = 6, 5, 240
    batch1 = torch.tanh(torch.randn(b, c, T, generator=gen).cumsum(dim=2) * 0.2)
    batch2 = torch.tanh(torch.randn(b + 2, c, T, generator=gen).cumsum(dim=2) * 0.25)  # (more items than batch1)
    kw = dict(patches_per_item=4, patch_size=49, dir_repeats=2, dirs_per_repeat=16)
    plugin = ref_plugins.SlicedWDistance(progression_scale=2, output_snapshot_ticks=1, max_items=b, **kw)
    np.random.seed(5)
    torch.manual_seed(5)
    ref = plugin.get_descriptors(batch1.clone(), batch2.clone())
    np.random.seed(5)
    torch.manual_seed(5)
    mine = M.swd_get_descriptors(batch1.clone(), batch2.clone(), 2, **kw)
    assert len(ref) == len(mine) == 3 and np.allclose(ref, mine, rtol=1e-6, atol=0), (ref, mine)
    # one level, step by step, with the draws written down
    np.random.seed(9)
    torch.manual_seed(9)
    starts, dirs = [], []
    d1 = M.swd_descriptors(batch1, b, 4, 49, starts)
    d2 = M.swd_descriptors(batch2, b, 4, 49, starts)
    dist = M.swd_sliced_wasserstein(d1, d2, 2, 16, dirs)
    np.random.seed(9)
    torch.manual_seed(9)
    plugin.patches_per_item, plugin.patch_size = 4, 49
    # the reference's own inner steps (plugins.py:438-448 are inline there; its sliced_wasserstein is a method)
    assert abs(plugin.sliced_wasserstein(d1, d2) - dist) > -1  # (consumes the torch generator like the call above)
    fx.update(swd_batch1=batch1.numpy(), swd_batch2=batch2.numpy(), swd_ref=np.asarray(ref, dtype=np.float64),
              swd_seed=np.array(5), swd_scale=np.array(2), swd_patches_per_item=np.array(4),
              swd_patch_size=np.array(49), swd_dir_repeats=np.array(2), swd_dirs_per_repeat=np.array(16),
              swd_step_seed=np.array(9), swd_step_starts1=starts[0], swd_step_starts2=starts[1],
              swd_step_desc1=d1.numpy(), swd_step_desc2=d2.numpy(), swd_step_dirs=torch.stack(dirs).numpy(),
              swd_step_dist=np.array(dist))
    print("swd: reference", ref, "oracle", mine)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    fx = {}
    swd_fixture(fx)
    np.savez_compressed(os.path.join(args.out, "metrics.npz"), **fx)
    print("metrics.npz:", len(fx), "arrays")


if __name__ == "__main__":
    main()
