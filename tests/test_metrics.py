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


# ---------------------------------------------------------------------------------------------- CPC features, FID
def _cpc_network(fx):
    """This package's mirror of the reference's feature network in the state the fixture was made with: same
    seed (same initial weights, checked by their checksums) and the same seeded BatchNorm state."""
    from eeggan_amd import cpc
    torch.manual_seed(int(fx["cpc_seed"]))
    net = cpc.Network(5).eval()
    gen = torch.Generator().manual_seed(int(fx["cpc_norm_seed"]))
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            n = m.num_features
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(n, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(n, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(n, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(n, generator=gen) * 0.2)
    got = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in net.state_dict().values()
                    if v.is_floating_point()])
    np.testing.assert_allclose(got, fx["cpc_check"], rtol=1e-5, atol=1e-3,
                               err_msg="the feature network's initial state differs from the reference's")
    return net


def test_oracle_cpc_and_fid_match_reference():
    from oracle import metrics_cpu as M
    fx = load_fixture("metrics")
    net = _cpc_network(fx)
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()}
    got = M.cpc_forward(sd, torch.from_numpy(fx["cpc_x"]).double())
    for a, name in zip(got, ("z", "c", "zp", "cp")):
        assert rel(a, fx[f"cpc_{name}64"]) < 1e-6, name  # (weights re-drawn here: ~1 ulp of the fp32 init per element)
    a, b = torch.from_numpy(fx["fid_a"]).double(), torch.from_numpy(fx["fid_b"]).double()
    s1, s2 = M.torch_cov(a), M.torch_cov(b)
    assert rel(s1, fx["fid_cov64"]) < 1e-12
    assert rel(M.sqrt_newton_schulz(s1.mm(s2), 50), fx["fid_root64"]) < 1e-9
    fid = float(M.frechet_distance(a.mean(0), s1, b.mean(0), s2))
    assert abs(fid - float(fx["fid_value64"])) < 1e-9 * abs(fid)


def test_cpc_mirror_rejects_what_it_does_not_run():
    from eeggan_amd import cpc
    with pytest.raises(NotImplementedError):
        cpc.Network(5, use_transformer=True)
    with pytest.raises(NotImplementedError):
        cpc.Network(5, residual_encoder=True)
    net = cpc.Network(5)
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 5, 1920), no_loss=False)
    assert set(k.split(".")[0] for k in net.state_dict()) == {"encoder", "z_pooler", "contextualizer", "c_pooler"}


@pytest.mark.gpu
def test_cpc_network_against_reference(dev):
    """Network.forward(x, no_loss=True) (reference cpc_network.py:316-322) on the device against the reference's
    latents: 1e-4 relative (fp32; measured far below)."""
    fx = load_fixture("metrics")
    net = _cpc_network(fx).to(dev)
    x = torch.from_numpy(fx["cpc_x"]).to(dev)
    with torch.no_grad():
        lat = net(x, no_loss=True).latents
    for got, name in zip(lat, ("z", "c", "zp", "cp")):
        e = min(rel(got, fx[f"cpc_{name}64"]), rel(got, fx[f"cpc_{name}32"]))
        print(f"cpc latent {name}: {e:.2e}")
        assert tuple(got.shape) == fx[f"cpc_{name}64"].shape
        assert e < 1e-4, (name, e)
    # a batch that does not fill the last workgroup of the recurrence, and a shorter window
    from oracle import metrics_cpu as M
    sd = {k: (v.detach().double().cpu() if v.is_floating_point() else v.cpu()) for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(2)
    for shape in ((37, 5, 1920), (3, 5, 960)):
        xs = torch.tanh(torch.randn(*shape, generator=gen))
        with torch.no_grad():
            lat = net(xs.to(dev), no_loss=True).latents
        for got, want, name in zip(lat, M.cpc_forward(sd, xs.double()), ("z", "c", "zp", "cp")):
            assert rel(got, want) < 1e-4, (shape, name, rel(got, want))


@pytest.mark.gpu
def test_dense_kernel(dev):
    from eeggan_amd.cpc import dense
    gen = torch.Generator().manual_seed(6)
    for N, K, M in ((100, 128, 384), (7, 20, 5), (65, 256, 128), (64, 64, 64)):
        x, w, b = torch.randn(N, K, generator=gen), torch.randn(M, K, generator=gen), torch.randn(M, generator=gen)
        sc, sh = torch.rand(M, generator=gen) + 0.5, torch.randn(M, generator=gen)
        ref = torch.relu(x.double() @ w.double().T + b.double()) * sc.double() + sh.double()
        got = dense(x.to(dev), w.to(dev), b.to(dev), sc.to(dev), sh.to(dev), relu=True)
        assert rel(got, ref) < 2e-6, (N, K, M)
        if N == M == K:
            ref = -0.5 * (x.double() @ w.double()) + 1.5 * torch.eye(N, dtype=torch.float64)
            assert rel(dense(x.to(dev), w.to(dev), w_is_kn=True, alpha=-0.5, eye=1.5), ref) < 2e-6


@pytest.mark.gpu
def test_fid_statistics_against_reference(dev):
    """Covariance, Newton-Schulz square root and the Frechet distance (reference plugins.py:503-572)."""
    from eeggan_amd.plugins import FidCalculator
    fx = load_fixture("metrics")
    a, b = torch.from_numpy(fx["fid_a"]).to(dev), torch.from_numpy(fx["fid_b"]).to(dev)
    mu1, s1 = FidCalculator.mean_and_cov(a)
    mu2, s2 = FidCalculator.mean_and_cov(b)
    assert rel(s1, fx["fid_cov64"]) < 1e-5 and rel(mu1, fx["fid_a"].mean(0)) < 1e-6
    assert rel(FidCalculator.torch_cov(a.t().contiguous(), rowvar=True), fx["fid_cov64"]) < 1e-5
    from eeggan_amd.cpc import dense
    root = FidCalculator.sqrt_newton_schulz(dense(s1, s2, w_is_kn=True).unsqueeze(0), 50).squeeze(0)
    e = min(rel(root, fx["fid_root64"]), rel(root, fx["fid_root32"]))
    assert e < 1e-3, e  # fifty fp32 iterations: the reference's own fp32 run is this far from its fp64 run
    fid = FidCalculator.calc_fid(mu1, s1, mu2, s2)
    ref64, ref32 = float(fx["fid_value64"]), float(fx["fid_value32"])
    assert min(abs(fid - ref64), abs(fid - ref32)) < 1e-4 * abs(ref64), (fid, ref64, ref32)
    # many rows: several partial blocks, a last block that is not full
    gen = torch.Generator().manual_seed(1)
    big = torch.randn(10000, 128, generator=gen) * torch.rand(128, generator=gen) + torch.randn(128, generator=gen)
    mu, cov = FidCalculator.mean_and_cov(big.to(dev))
    assert rel(mu, big.double().mean(0)) < 1e-5
    assert rel(cov, torch.cov(big.double().T)) < 1e-5


@pytest.mark.gpu
def test_fid_when_newton_schulz_breaks_down(dev):
    """Rank-deficient covariances: the fp32 Newton-Schulz iteration ends in NaN and calc_fid takes its second
    method, whose value is checked against the reference's second method (scipy's sqrtm on the host)."""
    from oracle import metrics_cpu as M
    from eeggan_amd.plugins import FidCalculator
    gen = torch.Generator().manual_seed(8)
    mix = torch.randn(24, 128, generator=gen)  # 128 features that live in a 24-dimensional subspace
    a = torch.randn(3000, 24, generator=gen) @ mix
    b = (torch.randn(3000, 24, generator=gen) * 1.3 + 0.2) @ mix
    sa, sb = FidCalculator.mean_and_cov(a.to(dev)), FidCalculator.mean_and_cov(b.to(dev))
    want = M.numpy_frechet_distance(a.double().mean(0).numpy(), np.cov(a.double().numpy().T),
                                    b.double().mean(0).numpy(), np.cov(b.double().numpy().T))
    eig = float(FidCalculator.eig_calculate_frechet_distance(*sa, *sb))
    assert abs(eig - want) < 1e-4 * abs(want), (eig, want)
    got = FidCalculator.calc_fid(*sa, *sb)
    assert got == got and abs(got - want) < 1e-3 * abs(want), (got, want)
    # and on the well-conditioned fixture both methods agree with the reference
    fx = load_fixture("metrics")
    s1 = FidCalculator.mean_and_cov(torch.from_numpy(fx["fid_a"]).to(dev))
    s2 = FidCalculator.mean_and_cov(torch.from_numpy(fx["fid_b"]).to(dev))
    ref = float(fx["fid_value64"])
    assert abs(float(FidCalculator.eig_calculate_frechet_distance(*s1, *s2)) - ref) < 1e-4 * ref


@pytest.mark.gpu
def test_swd_direction_chunks(dev):
    """Directions taken in several chunks (the bound on keys per sort) give the distance of the single sort."""
    from eeggan_amd.plugins import SlicedWDistance
    gen = torch.Generator().manual_seed(12)
    a = torch.randn(3000, 5, generator=gen).to(dev)
    b = (torch.randn(3000, 5, generator=gen) * 1.1 + 0.1).to(dev)
    whole = SlicedWDistance(2, 1)
    parts = SlicedWDistance(2, 1)
    parts.MAX_SORT_KEYS = 3000 * 50  # 128 directions as 50 + 50 + 28
    torch.manual_seed(5)
    want = whole.sliced_wasserstein(a, b)
    torch.manual_seed(5)
    got = parts.sliced_wasserstein(a, b)
    assert abs(got - want) < 1e-6 * abs(want), (got, want)


@pytest.mark.gpu
def test_feature_pipeline_at_the_stated_size(dev):
    """The evaluation workload at the size the configuration states (100 000 generated [5, 1920] signals: generator
    inference at the final stage, ``Network.forward(no_loss=True)``, mean and covariance of the four latents, Frechet
    distances; reference network.py:354-396, cpc_network.py:316-322, plugins.py:503-572), checked through properties
    that do not depend on the size: the latents of signals picked from all over the 100 000 equal the CPU
    restatement's (1e-4); the statistics of the 3.2 M x 128 feature matrix equal a float64 evaluation on the device
    (mean 1e-5, covariance 1e-4 -- single precision sums of millions of rows, in a fixed order); the distance is
    symmetric, and zero from a set to itself, to 1e-3 of tr(cov_1) + tr(cov_2): the distance is what is left of those
    traces after the cross term is taken off them (plugins.py:566-572), and the cross term comes out of fifty
    single-precision Newton-Schulz iterations, whose root is good to 1e-3 (the bar of the fixture test above; measured
    here: 3e-4 of the traces from a set to itself, 5e-5 between the two orders)."""
    from eeggan_amd import config, cpc, layers
    from eeggan_amd.dataset import synthetic_eeg
    from eeggan_amd.plugins import FidCalculator
    from oracle import metrics_cpu as M
    N, B = 100000, 2048
    layers.set_drop_mode("device")
    try:
        g, _ = config.build_models(config.RANDOM_SEED)
        g = g.to(dev).eval()
        g.depth, g.alpha = 7, 1.0
        T = config.seq_len(7)
        torch.manual_seed(21)
        net = cpc.Network(config.NUM_CHANNELS).to(dev).eval()
        gen = torch.Generator(device=dev).manual_seed(1373)
        fake = torch.empty(N, config.NUM_CHANNELS, T, device=dev)
        with torch.no_grad():
            for i in range(0, N, B):
                n = min(B, N - i)
                fake[i:i + n] = g({"z": torch.randn(n, config.LATENT_SIZE, device=dev, generator=gen)})[0]["x"]
    finally:
        layers.set_drop_mode("numpy")
    assert bool(torch.isfinite(fake).all()) and float(fake.abs().max()) <= 1.0
    pool = torch.from_numpy(synthetic_eeg(2048, config.NUM_CHANNELS, T, np.random.default_rng(1373))).to(dev)
    real = pool[torch.randint(0, 2048, (N,), device=dev, generator=gen)]

    def features(x):
        z = torch.empty(N, 128, T // 60, device=dev)
        c, zp = torch.empty_like(z), torch.empty(N, 128, device=dev)
        cp = torch.empty_like(zp)
        with torch.no_grad():
            for i in range(0, N, B):
                lat = net(x[i:i + B], no_loss=True).latents
                z[i:i + B], c[i:i + B], zp[i:i + B], cp[i:i + B] = lat.z, lat.c, lat.z_p, lat.c_p
        return {"z": z, "c": c, "zp": zp, "cp": cp}

    f_fake, f_real = features(fake), features(real)
    # latents of signals from all over the set (the last, partly filled batch included) against the CPU restatement
    pick = torch.tensor([0, 1, 2047, 2048, 50000, 77777, N - 2, N - 1], device=dev)
    sd = {k: (v.detach().double().cpu() if v.is_floating_point() else v.cpu()) for k, v in net.state_dict().items()}
    want = M.cpc_forward(sd, fake[pick].double().cpu())
    for name, w in zip(("z", "c", "zp", "cp"), want):
        assert rel(f_fake[name][pick], w) < 1e-4, (name, rel(f_fake[name][pick], w))
    # (the [N, 128, 32] latents as 128-column matrices, reshaped the way the plugin's feature extraction does)
    flat = lambda f: {k: v.reshape(-1, 128) for k, v in f.items()}  # noqa: E731
    s_fake = {k: FidCalculator.mean_and_cov(v.contiguous()) for k, v in flat(f_fake).items()}
    s_real = {k: FidCalculator.mean_and_cov(v.contiguous()) for k, v in flat(f_real).items()}
    for k, v in flat(f_fake).items():
        v64 = v.double()
        mu = v64.mean(0)
        cov = (v64 - mu).T @ (v64 - mu) / (v64.shape[0] - 1)
        assert v.shape[0] == (N * (T // 60) if k in ("z", "c") else N)
        assert rel(s_fake[k][0], mu) < 1e-5 and rel(s_fake[k][1], cov) < 1e-4, (k, rel(s_fake[k][1], cov))
    for k in s_fake:
        ab = float(FidCalculator.calc_fid(*s_fake[k], *s_real[k]))
        ba = float(FidCalculator.calc_fid(*s_real[k], *s_fake[k]))
        aa = float(FidCalculator.calc_fid(*s_fake[k], *s_fake[k]))
        scale = float(torch.trace(s_fake[k][1]) + torch.trace(s_real[k][1]))
        print(f"100 000 signals, latent {k}: distance {ab:.5f}, reversed {ba:.5f}, to itself {aa:.2e}; traces {scale:.3f}")
        assert ab > 0 and abs(ab - ba) < 1e-3 * scale and abs(aa) < 1e-3 * scale, (k, ab, ba, aa, scale)
