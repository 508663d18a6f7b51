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


def randomize_norms(net, seed):
    """Non-trivial BatchNorm state (running statistics and affine parameters), drawn from a seed: the same call
    on this package's mirror of the network gives it the same state."""
    gen = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            n = m.num_features
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(n, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(n, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(n, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(n, generator=gen) * 0.2)


def cpc_fixture(fx):
    sys.path.insert(0, REPO)
    from eeggan_amd import cpc as my_cpc
    seed = 21
    torch.manual_seed(seed)
    ref = ref_cpc.Network(5).eval()
    torch.manual_seed(seed)
    mine = my_cpc.Network(5).eval()
    randomize_norms(ref, 3)
    randomize_norms(mine, 3)
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    for k, v in sd_mine.items():
        assert torch.equal(v, sd_ref[k]), f"{k}: the mirror's initial state differs from the reference's"
    gen = torch.Generator().manual_seed(4)
    x = torch.tanh(torch.randn(6, 5, 1920, generator=gen).cumsum(dim=2) * 0.05)
    with torch.no_grad():
        lat64 = ref.double()(x.double(), no_loss=True).latents
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in ref.state_dict().items()}
        got = M.cpc_forward(sd64, x.double())
        for a, b, name in zip(got, lat64, ("z", "c", "z_p", "c_p")):
            assert rel(a, b) < 1e-12, (name, rel(a, b))
        lat32 = ref.float()(x, no_loss=True).latents
    fx.update(cpc_seed=np.array(seed), cpc_norm_seed=np.array(3), cpc_x=x.numpy(),
              cpc_check=np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd_mine.values()
                                  if v.is_floating_point()]))
    for name, a, b in zip(("z", "c", "zp", "cp"), lat64, lat32):
        fx[f"cpc_{name}64"], fx[f"cpc_{name}32"] = a.numpy(), b.numpy()
    print("cpc: latents", [tuple(v.shape) for v in lat64])


def fid_fixture(fx):
    gen = torch.Generator().manual_seed(8)
    N, Fdim = 400, 64
    mix1, mix2 = torch.randn(Fdim, Fdim, generator=gen) * 0.3, torch.randn(Fdim, Fdim, generator=gen) * 0.3
    a = torch.relu(torch.randn(N, Fdim, generator=gen) @ mix1 + 0.2)
    b = torch.relu(torch.randn(N, Fdim, generator=gen) @ mix2 + 0.1)
    F_ = ref_plugins.FidCalculator
    out = {}
    for dt, tag in ((torch.float64, "64"), (torch.float32, "32")):
        a_, b_ = a.to(dt), b.to(dt)
        mu1, mu2 = a_.mean(0), b_.mean(0)
        s1, s2 = F_.torch_cov(a_.clone(), rowvar=False), F_.torch_cov(b_.clone(), rowvar=False)
        root = F_.sqrt_newton_schulz(s1.mm(s2).unsqueeze(0), 50, dtype=a_.type()).squeeze()
        fid = F_.torch_calculate_frechet_distance(mu1, s1, mu2, s2)
        out[tag] = (s1, root, fid)
    s1, root, fid = out["64"]
    assert rel(M.torch_cov(a.double()), s1) < 1e-12
    assert rel(M.sqrt_newton_schulz(s1.mm(M.torch_cov(b.double())), 50), root) < 1e-9
    assert abs(float(M.frechet_distance(a.double().mean(0), s1, b.double().mean(0), M.torch_cov(b.double()))) - float(fid)) < 1e-9
    fx.update(fid_a=a.numpy(), fid_b=b.numpy(), fid_cov64=s1.numpy(), fid_root64=root.numpy(), fid_value64=np.array(float(fid)),
              fid_value32=np.array(float(out["32"][2])), fid_root32=out["32"][1].numpy())
    print("fid: reference", float(fid), "(fp32:", float(out["32"][2]), ")")


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    fx = {}
    swd_fixture(fx)
    cpc_fixture(fx)
    fid_fixture(fx)
    np.savez_compressed(os.path.join(args.out, "metrics.npz"), **fx)
    print("metrics.npz:", len(fx), "arrays")


if __name__ == "__main__":
    main()
