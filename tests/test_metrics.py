"""The evaluation metrics: the CPU oracle against the reference's golden vectors (CPU suite) and the HIP
kernels against both (GPU suite)."""
import numpy as np
import pytest
import torch

from helpers import load_fixture, rel


def test_oracle_swd_matches_reference():
    """oracle/metrics_cpu.py reproduces the reference plugin's distances (plugins.py:426-451) from the same
    seeds, and its step-by-step pieces reproduce the recorded intermediate values."""
    from oracle import metrics_cpu as M
    fx = load_fixture("metrics")
    b1, b2 = torch.from_numpy(fx["swd_batch1"]), torch.from_numpy(fx["swd_batch2"])
    kw = dict(patches_per_item=int(fx["swd_patches_per_item"]), patch_size=int(fx["swd_patch_size"]),
              dir_repeats=int(fx["swd_dir_repeats"]), dirs_per_repeat=int(fx["swd_dirs_per_repeat"]))
    np.random.seed(int(fx["swd_seed"]))
    torch.manual_seed(int(fx["swd_seed"]))
    got = M.swd_get_descriptors(b1, b2, int(fx["swd_scale"]), **kw)
    np.testing.assert_allclose(got, fx["swd_ref"], rtol=1e-6)
    np.random.seed(int(fx["swd_step_seed"]))
    torch.manual_seed(int(fx["swd_step_seed"]))
    starts = []
    d1 = M.swd_descriptors(b1, b1.shape[0], kw["patches_per_item"], kw["patch_size"], starts)
    assert np.array_equal(starts[0], fx["swd_step_starts1"])  # index work: bit-exact
    assert rel(d1, fx["swd_step_desc1"]) < 1e-6


@pytest.mark.gpu
def test_swd_kernels_step_by_step(dev):
    """Patch gather (bit-exact), normalisation, projection + sort, distance against the recorded steps."""
    from eeggan_amd import _lib
    from eeggan_amd.metrics import SlicedWDistance
    fx = load_fixture("metrics")
    ppi, ps = int(fx["swd_patches_per_item"]), int(fx["swd_patch_size"])
    plugin = SlicedWDistance(2, 1, patches_per_item=ppi, patch_size=ps, dir_repeats=int(fx["swd_dir_repeats"]),
                             dirs_per_repeat=int(fx["swd_dirs_per_repeat"]))
    b1, b2 = torch.from_numpy(fx["swd_batch1"]).to(dev), torch.from_numpy(fx["swd_batch2"]).to(dev)
    items = b1.shape[0]
    np.random.seed(int(fx["swd_step_seed"]))
    torch.manual_seed(int(fx["swd_step_seed"]))
    # raw patches: pure index work, must be bit-exact
    lib = _lib.load()
    starts = torch.from_numpy(fx["swd_step_starts1"]).to(dev)
    raw = torch.empty(items * ppi * ps * 5, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.eeggan_swd_patches(b1.data_ptr(), starts.data_ptr(), raw.data_ptr(), items, 5, b1.shape[2], 1, ppi, ps,
                                  st) == 0
    want = torch.stack([b1[j, :, s:s + ps] for j in range(items) for s in fx["swd_step_starts1"][j * ppi:(j + 1) * ppi]])
    assert torch.equal(raw.view(-1, 5, ps), want)
    # descriptors (same host draws as the oracle run that made the fixture)
    d1 = plugin._descriptors(b1, items)
    d2 = plugin._descriptors(b2, items)
    assert rel(d1, fx["swd_step_desc1"]) < 2e-6 and rel(d2, fx["swd_step_desc2"]) < 2e-6
    # projections, sorted per direction: against a host sort of the same products
    dirs = torch.from_numpy(fx["swd_step_dirs"][0]).to(dev)
    keys = plugin._sorted_projections(d1, dirs)
    D, rows = dirs.shape[1], d1.shape[0]
    assert torch.equal(keys >> 32, torch.arange(D, device=dev).repeat_interleave(rows))  # contiguous per direction
    low = (keys & 0xFFFFFFFF).view(D, rows)
    assert bool((low[:, 1:] >= low[:, :-1]).all())  # ascending inside every direction
    ref = torch.sort(torch.from_numpy(fx["swd_step_desc1"]).double() @ torch.from_numpy(fx["swd_step_dirs"][0]).double(),
                     dim=0)[0].T
    u = low.to(torch.int64)
    bits = torch.where(u >= 2 ** 31, u - 2 ** 31, (2 ** 32 - 1) - u).to(torch.int32)  # undo the order-preserving map
    got = bits.view(torch.float32) if bits.dtype == torch.int32 else None
    assert rel(got, ref) < 5e-6
    dist = plugin.sliced_wasserstein(d1, d2)  # consumes the directions of the recorded run
    assert abs(dist - float(fx["swd_step_dist"])) < 1e-5 * max(1.0, abs(float(fx["swd_step_dist"])))


@pytest.mark.gpu
def test_swd_plugin_against_reference(dev):
    """The whole plugin path (reference plugins.py:426-451) from the reference's seeds: distances to 1e-5."""
    from eeggan_amd import plugins
    fx = load_fixture("metrics")
    plugin = plugins.SlicedWDistance(int(fx["swd_scale"]), 1, patches_per_item=int(fx["swd_patches_per_item"]),
                                     patch_size=int(fx["swd_patch_size"]), dir_repeats=int(fx["swd_dir_repeats"]),
                                     dirs_per_repeat=int(fx["swd_dirs_per_repeat"]))
    np.random.seed(int(fx["swd_seed"]))
    torch.manual_seed(int(fx["swd_seed"]))
    got = plugin.get_descriptors(torch.from_numpy(fx["swd_batch1"]).to(dev), torch.from_numpy(fx["swd_batch2"]).to(dev))
    np.testing.assert_allclose(got, fx["swd_ref"], rtol=1e-5)


@pytest.mark.gpu
def test_swd_full_size_properties(dev):
    """Configuration-5 sizes (1024 items of [5, 1920], 16 patches of 49, 4 x 128 directions): size-independent
    properties instead of an oracle run -- identical sets are at distance 0, the distance is symmetric, and a
    constant offset d between the sets' normalised descriptors cannot be seen (they are centred per set)."""
    from eeggan_amd.metrics import SlicedWDistance
    plugin = SlicedWDistance(2, 1)
    gen = torch.Generator(device=dev).manual_seed(3)
    a = torch.tanh(torch.randn(1024, 5, 1920, device=dev, generator=gen).cumsum(dim=2) * 0.05)
    b = torch.tanh(torch.randn(1024, 5, 1920, device=dev, generator=gen).cumsum(dim=2) * 0.08)
    np.random.seed(1)
    da, db = plugin._descriptors(a, 1024), plugin._descriptors(b, 1024)
    assert tuple(da.shape) == (1024 * 16 * 49, 5)
    assert float(da.mean(dim=0).abs().max()) < 1e-4 and float((da.std(dim=0) - 1).abs().max()) < 1e-4
    torch.manual_seed(2)
    assert plugin.sliced_wasserstein(da, da) == 0.0
    torch.manual_seed(2)
    ab = plugin.sliced_wasserstein(da, db)
    torch.manual_seed(2)
    ba = plugin.sliced_wasserstein(db, da)
    assert ab > 0 and ab == ba
