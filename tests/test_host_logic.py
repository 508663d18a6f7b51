"""Host-side logic of the package that needs no GPU: module construction and state-dict layout,
the progression scheduler, resampling tables, batch sharding for data parallelism."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as O


def test_state_dict_layout_and_parameter_counts():
    from eeggan_amd import config
    g, d = config.build_models(1)
    gk, dk = list(g.state_dict()), list(d.state_dict())
    assert len(gk) == 48 and len(dk) == 50
    assert sum(p.numel() for p in g.parameters()) == 2976360
    assert sum(p.numel() for p in d.parameters()) == 2978497
    assert gk[0] == "block0.convs.0.conv.conv.weight" and "blocks.6.to_rgb.0.conv.conv.bias" in gk
    assert "blocks.7.net.1.conv.conv.weight" in dk and "blocks.7.net.2.conv.conv.bias" in dk
    assert "blocks.0.from_rgb.conv.conv.weight" in dk and dk[-1] == "linear.conv.conv.bias"
    assert d.state_dict()["blocks.7.net.1.conv.conv.weight"].shape == (256, 257, 3)
    assert g.state_dict()["block0.convs.0.conv.conv.weight"].shape == (256, 256, 32)
    assert g.max_depth == 7 and d.max_depth == 7 and g.input_latent_size == 256
    # unit-RMS weights, zero bias, the RMS kept as the runtime input scale
    c = g.block0.convs[1].conv
    assert abs(float((c.conv.weight ** 2).mean()) - 1.0) < 1e-5 and float(c.conv.bias.abs().sum()) == 0.0
    assert 0.0 < c.scale < 1.0
    # the oracle's table of conv shapes describes the same networks
    cfg = O.NetConfig()
    for shapes, sd in ((O.conv_shapes_generator(cfg), g.state_dict()), (O.conv_shapes_discriminator(cfg), d.state_dict())):
        assert len(shapes) * 2 == len(sd)
        for path, cin, cout, k in shapes:
            assert tuple(sd[f"{path}.conv.conv.weight"].shape) == (cout, cin, k)


def test_unsupported_options_raise():
    from eeggan_amd import config
    from eeggan_amd.network import Discriminator, Generator
    kw = config.shared_model_kwargs()
    noisy = Generator(**kw, z_distribution="normal", **{**config.GENERATOR_KWARGS, "per_channel_noise": True})
    assert "blocks.6.noises.1" in noisy.state_dict() and float(noisy.block0.noises[0].abs().sum()) == 0.0
    with pytest.raises(NotImplementedError):
        Generator(**kw, z_distribution="normal", **{**config.GENERATOR_KWARGS, "conv_only": True, "split_z": True, "act_norm": "batch"})
    with pytest.raises(NotImplementedError):
        Discriminator(**kw, **{**config.DISCRIMINATOR_KWARGS, "act_norm": "layer"})
    with pytest.raises(NotImplementedError):  # (classes enter the generator through its batch norms only)
        Generator(**{**kw, "num_classes": 10}, z_distribution="normal", **config.GENERATOR_KWARGS)
    with pytest.raises(NotImplementedError):
        Discriminator(**kw, **{**config.DISCRIMINATOR_KWARGS, "sinc": True})


def test_depth_manager_schedule():
    from eeggan_amd.plugins import DepthManager

    class _Net:
        depth, alpha = 0, 1.0

    class _Trainer:
        def __init__(self):
            self.cur_nimg, self.stats = 0, {}
            self.discriminator, self.generator, self.dataset = _Net(), _Net(), _Net()
            self.optimizer_d = self.optimizer_g = None

    lrs, batches = [], []
    tr = _Trainer()
    dm = DepthManager(lambda mb: batches.append(mb) or [None], lambda mb: iter(()), 7, 5.0,
                      lambda lr: lrs.append(lr) or ("g", "d", None, None), 0.001)
    assert dm.alpha_map == O.alpha_map(7)
    dm.register(tr)
    assert (tr.generator.depth, tr.generator.alpha) == (0, 1.0) and batches == [256] and lrs == [0.001]
    amap = O.alpha_map(7)
    for nimg in [0, 1, 199999, 200000, 200001, 300000, 400000, 400001, 1000000, 1700000, 3600000, 3700000, 4000000,
                 4000001, 5999999, 7000000]:
        assert dm.calc_progress(nimg) == O.calc_progress(nimg, amap, 7), nimg
    tr.cur_nimg = 3800000
    dm.iteration()
    assert tr.discriminator.depth == 7 and tr.discriminator.alpha == 0.5 and tr.dataset.model_depth == 7
    assert batches[-1] == 32 and abs(lrs[-1] - 0.008) < 1e-12 and tr.tick_duration_nimg == 3000
    assert tr.optimizer_g == "g" and tr.stats["minibatch_size"] == 32
    # every image count of a schedule with zero-length phases (repeated boundaries) against the reference's loop
    for amap in ([0, 5, 5, 9], [3, 3, 8, 8], [2, 6, 6, 6, 6, 11]):
        dm.alpha_map = amap
        for nimg in range(0, 14):
            assert dm.calc_progress(nimg) == O.calc_progress(nimg, amap, 7), (amap, nimg)
    # resuming keeps the optimizers that were handed in; a disabled progression sits at the last stage
    tr2 = _Trainer()
    tr2.optimizer_d = tr2.optimizer_g = "kept"
    dm2 = DepthManager(lambda mb: [None], lambda mb: iter(()), 7, 5.0, lambda lr: 1 / 0, 0.001,
                       disable_progression=True)
    dm2.register(tr2)
    assert tr2.optimizer_d == "kept" and (tr2.generator.depth, tr2.generator.alpha) == (7, 1.0)
    assert tr2.stats["depth"] == 7 and tr2.stats["alpha"]["val"] == 1.0 and dm2.stage_settings(4)[0] == 48


def test_progress_counts_the_global_batch_under_data_parallelism():
    """``cur_nimg`` (what the depth manager and the tick logic read, reference trainer.py:62-77, plugins.py:78-97)
    advances by the reals shown to the whole job: a rank of a two-rank job that holds 8 of 16 counts 16."""
    from eeggan_amd.trainer import Trainer

    class _Sync:
        world = 2

    tr = Trainer(None, None, None, None, None, None, 100, None, None)
    tr._count_reals(8)
    assert tr.cur_nimg == 108
    tr.grad_sync = _Sync()
    tr._count_reals(8)
    assert tr.cur_nimg == 124


def test_resample_tables_match_interpolate_then_pool():
    from eeggan_amd.resample_tables import build_tables
    up, dn = [4, 2, 2, 3, 4, 5, 3], [1, 1, 1, 2, 3, 4, 2]
    T = 32
    for i in range(7):
        for p, q, t_in in ((up[i], dn[i], T), (dn[i], up[i], T * up[i] // dn[i])):
            x = torch.randn(2, 3, t_in)
            y = x
            if p != 1:
                y = F.interpolate(y, scale_factor=p, mode="linear", align_corners=False)
            if q != 1:
                y = F.avg_pool1d(y, q, q, 0, False, True)
            (idx, coef, a, b, taps, quads), (tidx, tcoef, c, d, ttaps, tquads) = build_tables(t_in, p, q)
            assert (a, b) == (t_in, y.shape[2]) and (c, d) == (y.shape[2], t_in) and taps <= 16 and ttaps <= 16
            got = (x.double()[..., idx.astype(np.int64)] * torch.from_numpy(coef).double()).sum(-1)
            assert float((got - y.double()).norm() / y.double().norm()) < 2e-6
            # the quad form (dense 4 x 4nv blocks on 16-byte aligned spans) is the same operator
            for (sidx, scoef, s_in, s_out), qd, src in (((idx, coef, a, b), quads, x), ((tidx, tcoef, c, d), tquads, None)):
                if qd is None:
                    continue
                base, dense, nv = qd
                assert base.shape == (s_out // 4,) and dense.shape == (s_out // 4, 4, 4 * nv) and 1 <= nv <= 4
                assert (base % 4 == 0).all() and (base >= 0).all() and (base + 4 * nv <= s_in).all()
                full = np.zeros((s_out, s_in))
                for m in range(s_out):
                    np.add.at(full[m], sidx[m], scoef[m].astype(np.float64))
                qfull = np.zeros((s_out, s_in))
                for qq in range(s_out // 4):
                    qfull[4 * qq:4 * qq + 4, base[qq]:base[qq] + 4 * nv] = dense[qq]
                assert np.array_equal(full, qfull)
                # the device layout: groups of 64 quads, [e][v][quad % 64][4] inside a group, zero padding at the end
                from eeggan_amd.resample_tables import interleave_quads
                flat = interleave_quads(dense)
                nq = s_out // 4
                assert flat.shape == ((nq + 63) // 64 * 64 * 16 * nv,)
                for qq in {0, nq // 2, nq - 1}:
                    for e in range(4):
                        for v in range(nv):
                            at = ((((qq // 64) * 4 + e) * nv + v) * 64 + qq % 64) * 4
                            assert np.array_equal(flat[at:at + 4], dense[qq, e, 4 * v:4 * v + 4])
                assert np.array_equal(np.sort(flat[flat != 0]), np.sort(dense[dense != 0]))  # nothing else is non-zero
                # the quad starts in closed form (what the kernel evaluates instead of reading ``base``)
                from eeggan_amd.resample_tables import affine_base
                form = affine_base(base, s_in, nv)
                assert form is not None
                fa, fc, fd = form
                qs = np.arange(nq, dtype=np.int64)
                assert np.array_equal(np.minimum(np.maximum(fa * qs + fc, 0) // fd // 4 * 4, s_in - 4 * nv), base)
                assert affine_base(base[::-1].copy(), s_in, nv) is None or nq < 3 or (base == base[0]).all()
            assert quads is not None or t_in < 8  # every forward resampler of the schedule has a quad form
            # the transposed table is the adjoint: <A x, g> == <x, A^T g>
            g = torch.randn_like(y)
            back = (g.double()[..., tidx.astype(np.int64)] * torch.from_numpy(tcoef).double()).sum(-1)
            lhs, rhs = float((got * g.double()).sum()), float((x.double() * back).sum())
            assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs))
        T = T * up[i] // dn[i]
    (idx, coef, a, b, taps, _q), _ = build_tables(1, 4, 1)
    assert b == 4 and taps == 1 and coef.tolist() == [[1.0]] * 4


def test_shard_batch_keeps_minibatch_stddev_groups():
    from eeggan_amd.parallel import shard_batch
    B, world = 32, 2
    x = torch.randn(B, 6, 32, dtype=torch.float64)
    full = O.minibatch_stddev(x, 4, 32)[:, -1, 0]  # the statistic every sample sees in the global batch
    seen = {}
    for r in range(world):
        xs = shard_batch(x, r, world)
        assert xs.shape[0] == B // world
        stat = O.minibatch_stddev(xs, 4, 32)[:, -1, 0]
        for row, s in zip(xs, stat):
            j = int(torch.nonzero((x == row).flatten(1).all(dim=1))[0])
            seen[j] = float(s)
    assert sorted(seen) == list(range(B))
    for j, s in seen.items():
        assert abs(s - float(full[j])) < 1e-12
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(12, 1, 1), 0, 8)


def test_adam_descriptor_table_layout():
    import ctypes
    from eeggan_amd import optim
    assert optim.DESC_BYTES == 48  # five pointers + int64, as declared in include/eeggan_hip.h
    arr = optim.build_table([(1, 2, 3, 4, 5, 6), (7, 8, 9, 10, 11, 12)])
    raw = bytes(arr)
    assert raw[40:48] == (6).to_bytes(8, "little") and raw[48:56] == (7).to_bytes(8, "little")
    assert raw[88:96] == (12).to_bytes(8, "little")


def test_trainer_plugin_schedule():
    """Plugin firing rule of the reference (trainer.py:39-60): a plugin fires when its next time is
    due, earliest first and in registration order among equals, and is re-armed one period after the
    time it fired at."""
    from eeggan_amd.plugins import Plugin
    from eeggan_amd.trainer import Trainer

    calls = []

    class P(Plugin):
        def __init__(self, name, interval):
            super().__init__(interval)
            self.name = name

        def register(self, trainer):
            self.trainer = trainer

        def iteration(self, time, *args):
            calls.append((self.name, time, args))

        def epoch(self, time, *args):
            calls.append((self.name + "/tick", time, args))

    tr = Trainer(None, None, None, None, None, iter(()), 0, None, None)
    a, b, c = P("a", [(1, "iteration")]), P("b", [(2, "iteration"), (1, "epoch")]), P("c", (3, "iteration"))
    for p in (a, b, c):
        tr.register_plugin(p)
    assert a.trainer is tr
    for it in range(1, 7):
        tr.call_plugins("iteration", it, "g", "d")
    fired = [(n, t) for n, t, _ in calls]
    assert fired == [("a", 1), ("a", 2), ("b", 2), ("a", 3), ("c", 3), ("a", 4), ("b", 4), ("a", 5), ("a", 6), ("b", 6),
                     ("c", 6)]
    assert calls[0][2] == ("g", "d")
    calls.clear()
    tr.call_plugins("epoch", 1)
    tr.call_plugins("end", 1)  # nobody registered: no-op
    assert [(n, t) for n, t, _ in calls] == [("b/tick", 1)]


def test_direct_gradient_queue_recovers_from_a_failed_backward():
    """A backward pass that raises after queueing weight-gradient work never reaches the end-of-backward
    callback; the next pass must start from an empty queue and register the callback again."""
    from eeggan_amd import ops
    d = ops._direct
    d.queued, d.pending, d.jobs, d.touched, d.keep = True, {"stale": [1]}, [(0, None)], {1: [None, 1]}, [object()]
    with ops.direct_weight_grads():
        assert not d.queued and not d.pending and not d.jobs and not d.touched and not d.keep
    # ... and an exception inside the context leaves nothing behind either
    try:
        with ops.direct_weight_grads():
            d.queued, d.pending = True, {"x": [1]}
            raise RuntimeError("backward failed")
    except RuntimeError:
        pass
    assert not d.queued and not d.pending and not d.enabled


def test_flip_cache_copy_follows_the_copied_parameter():
    """copy.deepcopy of a layer (an EMA copy of a network) must give the copy a cache of its own that tracks
    the COPIED weight and starts stale, and a cache that never ran __init__ must still register itself."""
    import copy
    import torch
    from eeggan_amd import ops
    from eeggan_amd.layers import EqualizedConv1d
    conv = EqualizedConv1d(4, 6, 3, 1, False, True, "kaiming_normal", 0.2, True, 1, 1)
    conv._flip.tag, conv._flip.wt = ("pretend", 0, 0), torch.zeros(4, 6, 3)
    twin = copy.deepcopy(conv)
    assert twin._flip is not conv._flip
    assert twin._flip.param is twin.conv.weight and twin._flip.param is not conv.conv.weight
    assert twin._flip.tag is None and twin._flip.wt is None and twin._flip in ops.FlipCache._all
    assert twin._flip._stale()


def test_eval_discriminator_plugin_logs_the_mean_score():
    """EvalDiscriminator (reference plugins.py:205-232): on its ticks the mean of the critic's scores over the batches
    the loader yields, under stats['memorization']; other ticks leave it alone."""
    import torch
    from eeggan_amd.plugins import EvalDiscriminator

    class _D:
        def __call__(self, data):
            x = data["x"] if isinstance(data, dict) else data
            return x.mean(dim=(1, 2)), None, {}

    class _Set:
        model_depth, alpha = 3, 0.5

    class _Trainer:
        stats = {"minibatch_size": 4000}
        dataset = _Set()
        discriminator = _D()

    asked = []

    def loader(minibatch, training, depth, alpha):
        asked.append((minibatch, training, depth, alpha))
        return [torch.full((2, 5, 8), 1.0), torch.full((2, 5, 8), 3.0)]

    import eeggan_amd.utils as U
    plug = EvalDiscriminator(loader, output_snapshot_ticks=2)
    tr = _Trainer()
    plug.register(tr)
    assert tr.stats["memorization"]["log_name"] == "memorization"
    saved, U.cudize = U.cudize, (lambda t: t)
    try:
        plug.epoch(1)
        assert not asked and tr.stats["memorization"]["epoch"] == 0
        plug.epoch(2)
    finally:
        U.cudize = saved
    assert asked == [(1024, False, 3, 0.5)]
    assert abs(tr.stats["memorization"]["val"] - 2.0) < 1e-6 and tr.stats["memorization"]["epoch"] == 2
