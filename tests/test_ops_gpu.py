"""Parity of every HIP kernel, called through the C ABI, against the CPU oracle (fp64 evaluation of
the same fp32 inputs) and against the per-op golden vectors produced by the reference.

Tolerance: 1e-4 relative in fp32 is the contract of the path; single kernels are held to 2e-5
(norm-wise) because their fp32 rounding noise is ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_fixture, rel

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _ops():
    from eeggan_amd import ops
    return ops


def _oracle():
    from oracle import ref_cpu
    return ref_cpu


def conv_ref(x, w, b, scale, pad, drop=None, act=None, slope=0.2, tanh_scale=0.5):
    y = F.conv1d(x.double() * scale, w.double(), None if b is None else b.double(), padding=pad)
    if act == "lrelu":
        y = F.leaky_relu(y, slope)
    elif act == "tanh":
        y = torch.tanh(y * tanh_scale)
    if drop is not None:
        y = y * drop.double()[:, :, None]
    return y


CONV_SHAPES = [
    # B, Cin, Cout, T, K
    (2, 64, 64, 1920, 3), (3, 64, 64, 96, 3), (2, 128, 64, 512, 3), (2, 64, 128, 512, 3), (2, 128, 128, 256, 3),
    (2, 256, 256, 128, 3), (4, 257, 256, 32, 3), (2, 256, 128, 256, 3), (2, 64, 5, 1920, 1), (2, 5, 64, 1920, 1),
    (3, 5, 256, 32, 1), (4, 256, 1, 1, 1), (2, 7, 9, 50, 3), (1, 33, 70, 37, 3), (2, 6, 10, 48, 1), (5, 20, 40, 4, 3),
    # >= 256 subtiles of 32 samples at <= 64 channels: the weight-stationary kernel (one unit per wave,
    # several units per wave, ragged channel counts and a ragged last subtile)
    (8, 64, 64, 1024, 3), (34, 64, 64, 1920, 3), (9, 48, 40, 1000, 3), (16, 64, 64, 512, 3),
    (16, 256, 256, 128, 3), (8, 256, 256, 128, 3), (16, 128, 128, 256, 3),
]


@pytest.mark.parametrize("B,Cin,Cout,T,K", CONV_SHAPES)
def test_conv_fwd_dgrad_wgrad(dev, B, Cin, Cout, T, K):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g)
    b = torch.randn(Cout, generator=g)
    drop = torch.rand(B, Cout, generator=g) + 0.5
    gy = torch.randn(B, Cout, T, generator=g)
    scale, pad = 0.37, (K - 1) // 2
    xd, wd, bd, dd, gyd = (t.to(dev) for t in (x, w, b, drop, gy))

    # forward, all epilogues
    y = ops.conv1d_raw(xd, wd, bd, None, in_scale=scale)
    assert rel(y, conv_ref(x, w, b, scale, pad)) < TOL
    y = ops.conv1d_raw(xd, wd, bd, dd, in_scale=scale, act=1, act_param=0.2)
    yref = conv_ref(x, w, b, scale, pad, drop, "lrelu")
    assert rel(y, yref) < TOL
    yt = ops.conv1d_raw(xd, wd, bd, None, in_scale=scale, act=2, act_param=0.5)
    assert rel(yt, conv_ref(x, w, b, scale, pad, None, "tanh")) < TOL

    # input gradient (transposed view of the forward weight), with and without the fused mask
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    lin = F.conv1d(x64 * scale, w64, None, padding=pad)
    gx_ref, gw_ref = torch.autograd.grad(lin, [x64, w64], gy.double(), retain_graph=True)
    gx = ops.conv1d_raw(gyd, wd, None, None, in_scale=scale, transposed=True)
    assert rel(gx, gx_ref) < TOL
    mask = torch.where(yref > 0, 1.0, 0.2) * drop.double()[:, :, None]
    gxm_ref, gwm_ref = torch.autograd.grad(lin, [x64, w64], gy.double() * mask)
    yd = y  # device copy of the layer output (sign == sign of the pre-activation)
    gxm = ops.conv1d_raw(gyd, wd, None, None, in_scale=scale, transposed=True, mask_y=yd, mask_drop=dd,
                         mask_slope=0.2)
    assert rel(gxm, gxm_ref) < TOL

    # weight / bias gradient
    gw, gb = ops.conv1d_wgrad_raw(xd, gyd, K, in_scale=scale)
    assert rel(gw, gw_ref) < TOL
    assert rel(gb, gy.double().sum(dim=(0, 2))) < TOL
    gwm, gbm = ops.conv1d_wgrad_raw(xd, gyd, K, in_scale=scale, mask_y=yd, mask_drop=dd, mask_slope=0.2)
    assert rel(gwm, gwm_ref) < TOL
    assert rel(gbm, (gy.double() * mask).sum(dim=(0, 2))) < TOL

    # output-masked forward (second-order op): conv(q, w) * mask
    q = torch.randn(B, Cin, T, generator=g)
    om = ops.conv1d_raw(q.to(dev), wd, None, dd, in_scale=scale, act=3, act_param=0.2, act_aux=yd)
    assert rel(om, conv_ref(q, w, None, scale, pad) * mask) < TOL


@pytest.mark.parametrize("B,Cin,Cout,T", [(4, 257, 256, 32), (64, 257, 256, 32), (3, 258, 128, 64), (40, 131, 192, 32),
                                          (4, 256, 257, 32), (64, 256, 257, 32), (5, 128, 130, 96), (520, 257, 256, 32),
                                          (520, 256, 257, 32)])
def test_conv_ragged_channels(dev, B, Cin, Cout, T):
    """k = 3 layers with a few input channels beyond a multiple of 32 (the layer behind MinibatchStddev: 256 + 1; the
    left-over channels' terms are added in the epilogue) or with a partly filled last tile of output channels (its
    input gradient), on the predicate-free kernels: every epilogue, and the fused input mask, small and large batches."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + Cin * 7 + Cout * 13 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, 3, generator=g)
    b = torch.randn(Cout, generator=g)
    drop = torch.rand(B, Cout, generator=g) + 0.5
    my = torch.randn(B, Cin, T, generator=g)
    md = torch.rand(B, Cin, generator=g) + 0.5
    aux = torch.randn(B, Cout, T, generator=g)
    xd, wd, bd, dd, myd, mdd, auxd = (t.to(dev) for t in (x, w, b, drop, my, md, aux))
    sel = sorted({0, 1, B // 2, B - 2, B - 1})
    scale = 0.37
    y = ops.conv1d_raw(xd, wd, bd, dd, in_scale=scale, act=1, act_param=0.2)
    assert rel(y[sel], conv_ref(x[sel], w, b, scale, 1, drop[sel], "lrelu")) < TOL
    y = ops.conv1d_raw(xd, wd, bd, None, in_scale=scale)
    assert rel(y[sel], conv_ref(x[sel], w, b, scale, 1)) < TOL
    xm = x[sel].double() * torch.where(my[sel] > 0, 1.0, 0.2).double() * md[sel].double()[:, :, None]
    ref = F.conv1d(xm * scale, w.double(), None, padding=1)
    y = ops.conv1d_raw(xd, wd, None, None, in_scale=scale, mask_y=myd, mask_drop=mdd, mask_slope=0.2)
    assert rel(y[sel], ref) < TOL
    y = ops.conv1d_raw(xd, wd, None, None, in_scale=scale, mask_y=myd, mask_slope=0.2)
    xm1 = x[sel].double() * torch.where(my[sel] > 0, 1.0, 0.2).double()
    assert rel(y[sel], F.conv1d(xm1 * scale, w.double(), None, padding=1)) < TOL
    refm = ref * torch.where(aux[sel] > 0, 1.0, 0.2).double() * drop[sel].double()[:, :, None]
    y = ops.conv1d_raw(xd, wd, None, dd, in_scale=scale, mask_y=myd, mask_drop=mdd, mask_slope=0.2, act=3, act_param=0.2,
                       act_aux=auxd)
    assert rel(y[sel], refm) < TOL
    # weight gradient (input channels that are not a multiple of 64 take the split kernel with clamped rows), plain and
    # with the fused mask on the output gradient
    gy = torch.randn(B, Cout, T, generator=g)
    x64, w64 = x.double(), w.double().requires_grad_(True)
    lin = F.conv1d(x64 * scale, w64, None, padding=1)
    (gw_ref,) = torch.autograd.grad(lin, w64, gy.double(), retain_graph=True)
    gw, gb = ops.conv1d_wgrad_raw(xd, gy.to(dev), 3, in_scale=scale)
    assert rel(gw, gw_ref) < TOL and rel(gb, gy.double().sum(dim=(0, 2))) < TOL
    mask = torch.where(aux > 0, 1.0, 0.2).double() * drop.double()[:, :, None]
    (gwm_ref,) = torch.autograd.grad(lin, w64, gy.double() * mask)
    gwm, gbm = ops.conv1d_wgrad_raw(xd, gy.to(dev), 3, in_scale=scale, mask_y=auxd, mask_drop=dd, mask_slope=0.2)
    assert rel(gwm, gwm_ref) < TOL and rel(gbm, (gy.double() * mask).sum(dim=(0, 2))) < TOL
    # (and through the grouped entry point, the one the training step submits its layers to)
    (gw2, gb2), (gwm2, gbm2) = ops.conv1d_wgrad_group_raw([
        dict(x=xd, gy=gy.to(dev), K=3, in_scale=scale),
        dict(x=xd, gy=gy.to(dev), K=3, in_scale=scale, mask_y=auxd, mask_drop=dd, mask_slope=0.2)])
    assert rel(gw2, gw_ref) < TOL and rel(gwm2, gwm_ref) < TOL
    assert rel(gb2, gy.double().sum(dim=(0, 2))) < TOL and rel(gbm2, (gy.double() * mask).sum(dim=(0, 2))) < TOL
    # the left-over channels alone must be right too: weights that are zero everywhere else
    if Cin % 32:
        w2 = w.clone()
        w2[:, :Cin - Cin % 32] = 0
        y = ops.conv1d_raw(xd, w2.to(dev), None, None, in_scale=scale)
        assert rel(y[sel], conv_ref(x[sel], w2, None, scale, 1)) < TOL


@pytest.mark.parametrize("Cin,Cout", [(5, 64), (64, 5), (8, 8)])
def test_conv_thin_large_batch(dev, Cin, Cout):
    """1x1 convs with at most 8 channels on one side run a bandwidth-bound kernel of their own once the
    batch is large (B*T/4 >= 2 workgroups per CU); all epilogues and the fused input mask, checked on the
    first and last samples against the fp64 reference."""
    ops = _ops()
    B, T = 288, 1920
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, 1, generator=g)
    b = torch.randn(Cout, generator=g)
    drop = torch.rand(B, Cout, generator=g) + 0.5
    my = torch.randn(B, Cin, T, generator=g)
    md = torch.rand(B, Cin, generator=g) + 0.5
    aux = torch.randn(B, Cout, T, generator=g)
    xd, wd, bd, dd, myd, mdd, auxd = (t.to(dev) for t in (x, w, b, drop, my, md, aux))
    sel = [0, 1, 2, B - 2, B - 1]
    scale = 0.37
    cases = {
        "lrelu+drop": (dict(act=1, act_param=0.2), lambda: conv_ref(x[sel], w, b, scale, 0, drop[sel], "lrelu")),
        "tanh": (dict(act=2, act_param=0.5), lambda: conv_ref(x[sel], w, b, scale, 0, None, "tanh")),
        "linear": (dict(), lambda: conv_ref(x[sel], w, b, scale, 0)),
    }
    for name, (kw, ref) in cases.items():
        y = ops.conv1d_raw(xd, wd, bd, dd if name == "lrelu+drop" else None, in_scale=scale, **kw)
        assert rel(y[sel], ref()) < TOL, name
    # fused input mask (a gradient entering through the LeakyReLU/GDrop of the layer below) and output mask
    xm = x[sel].double() * torch.where(my[sel] > 0, 1.0, 0.2).double() * md[sel].double()[:, :, None]
    ref = F.conv1d(xm * scale, w.double(), None)
    y = ops.conv1d_raw(xd, wd, None, None, in_scale=scale, mask_y=myd, mask_drop=mdd, mask_slope=0.2)
    assert rel(y[sel], ref) < TOL
    refm = F.conv1d(x[sel].double() * scale, w.double(), None) * torch.where(aux[sel] > 0, 1.0, 0.2).double() \
        * drop[sel].double()[:, :, None]
    y = ops.conv1d_raw(xd, wd, None, dd, in_scale=scale, act=3, act_param=0.2, act_aux=auxd)
    assert rel(y[sel], refm) < TOL


def test_conv_split_products_are_fp32_exact(dev):
    """The 64 -> 64, k = 3 layers multiply on the bf16 matrix pipe, every fp32 operand as the sum of three
    bf16 pieces and six of the nine piece products kept.  Against fp64 the result must be as good as an
    fp32 computation: held here to 5e-7 norm-wise and 2e-6 of the largest output element-wise, on inputs
    with a wide dynamic range (so that the low pieces matter) -- a plain bf16 product would sit at 3e-3,
    two pieces per operand at 1e-5."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, T = 16, 1024
    x = torch.randn(B, 64, T, generator=g) * torch.exp(2.0 * torch.randn(B, 64, 1, generator=g))
    w = torch.randn(64, 64, 3, generator=g) * torch.exp(torch.randn(64, 64, 1, generator=g))
    b = torch.randn(64, generator=g)
    y = ops.conv1d_raw(x.to(dev), w.to(dev), b.to(dev), None, in_scale=0.37)
    ref = conv_ref(x, w, b, 0.37, 1)
    assert rel(y, ref) < 5e-7
    assert float((y.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # the input gradient takes the same kernel on the flipped weights
    gy = torch.randn(B, 64, T, generator=g)
    x64 = x.double().requires_grad_(True)
    (gx_ref,) = torch.autograd.grad(F.conv1d(x64 * 0.37, w.double(), None, padding=1), x64, gy.double())
    gx = ops.conv1d_raw(gy.to(dev), w.to(dev), None, None, in_scale=0.37, transposed=True)
    assert rel(gx, gx_ref) < 5e-7
    # the weight gradient of every predicate-free k = 3 layer splits its operands the same way
    w64 = w.double().requires_grad_(True)
    (gw_ref,) = torch.autograd.grad(F.conv1d(x.double(), w64 * 0.37, None, padding=1), w64, gy.double())
    gw, gb = ops.conv1d_wgrad_raw(x.to(dev), gy.to(dev), 3, in_scale=0.37)
    assert rel(gw, gw_ref) < 5e-7
    assert rel(gb, gy.double().sum((0, 2))) < 5e-7
    # the wider predicate-free layers (tiled kernel): the eight-wave tile and the narrow tile of the short layers
    for B2, C, T2 in ((64, 128, 512), (8, 256, 32), (3, 128, 96)):
        x = torch.randn(B2, C, T2, generator=g) * torch.exp(2.0 * torch.randn(B2, C, 1, generator=g))
        w = torch.randn(C, C, 3, generator=g) * torch.exp(torch.randn(C, C, 1, generator=g))
        b = torch.randn(C, generator=g)
        y = ops.conv1d_raw(x.to(dev), w.to(dev), b.to(dev), None, in_scale=0.21)
        ref = conv_ref(x, w, b, 0.21, 1)
        assert rel(y, ref) < 5e-7, (B2, C, T2)
        assert float((y.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6, (B2, C, T2)


def test_conv_wgrad_deterministic(dev):
    ops = _ops()
    x = torch.randn(8, 64, 768, device=dev)
    gy = torch.randn(8, 64, 768, device=dev)
    a, _ = ops.conv1d_wgrad_raw(x, gy, 3)
    b, _ = ops.conv1d_wgrad_raw(x, gy, 3)
    assert torch.equal(a, b)


def test_conv_wgrad_grouped(dev):
    """Several layers' weight gradients in one grouped submission (shared launches, fewer splits per layer)
    against the fp64 reference, for every kernel variant in one group: predicate-free K = 3 layers, a ragged
    one, K = 1 layers, with and without the fused mask and the bias."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    shapes = [(2, 64, 64, 1920, 3), (32, 64, 64, 512, 3), (4, 257, 256, 32, 3), (2, 128, 64, 256, 3),
              (2, 5, 64, 1920, 1), (3, 64, 5, 512, 1), (8, 256, 256, 128, 3), (2, 7, 9, 50, 3), (2, 64, 128, 512, 3)]
    jobs, refs = [], []
    for n, (B, Cin, Cout, T, K) in enumerate(shapes):
        x = torch.randn(B, Cin, T, generator=g)
        gy = torch.randn(B, Cout, T, generator=g)
        scale = 0.2 + 0.1 * n
        masked, need_bias = n % 2 == 0, n % 3 != 1
        job = {"x": x.to(dev), "gy": gy.to(dev), "K": K, "in_scale": scale, "need_bias": need_bias}
        geff = gy.double()
        if masked:
            y = torch.randn(B, Cout, T, generator=g)
            drop = torch.rand(B, Cout, generator=g) + 0.5
            job.update(mask_y=y.to(dev), mask_drop=drop.to(dev), mask_slope=0.2)
            geff = geff * torch.where(y > 0, 1.0, 0.2).double() * drop.double()[:, :, None]
        w64 = torch.zeros(Cout, Cin, K, dtype=torch.float64, requires_grad=True)
        lin = F.conv1d(x.double() * scale, w64, None, padding=(K - 1) // 2)
        (gw_ref,) = torch.autograd.grad(lin, [w64], geff)
        jobs.append(job)
        refs.append((gw_ref, geff.sum(dim=(0, 2)) if need_bias else None))
    outs = ops.conv1d_wgrad_group_raw(jobs)
    for (gw, gb), (gw_ref, gb_ref), shape in zip(outs, refs, shapes):
        assert rel(gw, gw_ref) < TOL, shape
        assert (gb is None) == (gb_ref is None)
        if gb is not None:
            assert rel(gb, gb_ref) < TOL, shape
    # a group of one takes the same route as the single call
    one = ops.conv1d_wgrad_group_raw(jobs[:1])[0][0]
    single, _ = ops.conv1d_wgrad_raw(jobs[0]["x"], jobs[0]["gy"], 3, in_scale=jobs[0]["in_scale"],
                                     mask_y=jobs[0]["mask_y"], mask_drop=jobs[0]["mask_drop"], mask_slope=0.2)
    assert rel(one, single.double()) < 1e-6


def test_direct_wgrad_solo_submission(dev):
    """A layer that keeps the chip busy for >= 16 rounds by itself (here 140 samples of the 64 -> 64, T = 1920
    layer) is not queued with others: its weight gradient is submitted alone, right away, and still lands
    in the persistent gradient buffer at the end of the backward pass.  Against fp64."""
    ops = _ops()
    B, C, T, scale = 140, 64, 1920, 0.31
    g = torch.Generator().manual_seed(140)
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(C, C, 3, generator=g)
    b = torch.randn(C, generator=g)
    drop = torch.rand(B, C, generator=g) + 0.5
    gy = torch.randn(B, C, T, generator=g)
    units = ((B * (T // 32) + 1) // 2)
    assert units >= ops.WGRAD_SOLO_UNITS
    wd = torch.nn.Parameter(w.to(dev))
    bd = torch.nn.Parameter(b.to(dev))
    y = ops.conv1d(x.to(dev), wd, bd, drop.to(dev), scale, 1, 0.2)
    with ops.direct_weight_grads():
        y.backward(gy.to(dev))
    assert wd.grad is wd._eeggan_grad
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    yref = F.leaky_relu(F.conv1d(x.double() * scale, w64, b64, padding=1), 0.2) * drop.double()[:, :, None]
    gw_ref, gb_ref = torch.autograd.grad(yref, [w64, b64], gy.double())
    assert rel(wd.grad, gw_ref) < TOL and rel(bd.grad, gb_ref) < TOL


@pytest.mark.parametrize("shape", [(140, 5, 64, 1920, True), (140, 64, 5, 1920, False), (70, 5, 256, 3840, True),
                                   (280, 128, 5, 1024, False), (140, 7, 100, 1920, True)])
def test_thin_wgrad_large_batch(dev, shape):
    """1x1 layers with a thin side (from_rgb, to_rgb) at batch sizes where their weight gradient takes the vector
    kernel of its own (B * T >= 2^18): against fp64, with the fused backward mask where the layer has one, and against
    the matrix-core route on the same inputs."""
    import os
    ops = _ops()
    B, Cin, Cout, T, masked = shape
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(B, Cin, T, generator=g)
    gy = torch.randn(B, Cout, T, generator=g)
    job = {"x": x.to(dev), "gy": gy.to(dev), "K": 1, "in_scale": 0.37, "need_bias": True}
    geff = gy.double()
    if masked:
        y = torch.randn(B, Cout, T, generator=g)
        drop = torch.rand(B, Cout, generator=g) + 0.5
        job.update(mask_y=y.to(dev), mask_drop=drop.to(dev), mask_slope=0.2)
        geff = geff * torch.where(y > 0, 1.0, 0.2).double() * drop.double()[:, :, None]
    ref_w = 0.37 * torch.einsum("bot,bit->oi", geff.to(dev), x.double().to(dev))[:, :, None]
    ref_b = geff.sum(dim=(0, 2))
    (gw, gb), = ops.conv1d_wgrad_group_raw([job])
    assert rel(gw, ref_w) < TOL and rel(gb, ref_b) < TOL, (rel(gw, ref_w), rel(gb, ref_b))


def test_device_gdrop_multipliers(dev):
    """The one-launch generator of the GDrop multipliers: log(out) / log(base) is standard normal (moments, a
    Kolmogorov-Smirnov distance), successive launches on one state differ (the kernel advances its epochs), two states give
    unrelated streams, and the same seed reproduces the same values."""
    from scipy import stats
    ops = _ops()
    n, base = 1 << 18, 1.2

    def draw(state, salt, seed=1234):
        return ops.lognormal_fill_raw(torch.empty(n, device=dev), base, seed, salt, state)

    new_state = lambda: torch.zeros(ops.LOGNORMAL_STATE_WORDS, device=dev, dtype=torch.int32)
    s0, s1 = new_state(), new_state()
    a, b, c = draw(s0, 0), draw(s0, 0), draw(s1, 1)
    assert bool((s0 == 2).all()) and bool((s1 == 1).all())  # one epoch per workgroup, 256 workgroups here
    for t in (a, b, c):
        z = (torch.log(t.double()) / np.log(base)).cpu().numpy()
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02 and abs(stats.kurtosis(z)) < 0.05
        assert stats.kstest(z, "norm").statistic < 0.005
    za, zb, zc = (torch.log(t.double()) for t in (a, b, c))
    for u, v in ((za, zb), (za, zc), (za[1:], za[:-1])):
        assert abs(float(torch.corrcoef(torch.stack([u, v]))[0, 1])) < 0.01
    again = draw(new_state(), 0)
    assert torch.equal(again, a)
    s2 = new_state()
    odd = ops.lognormal_fill_raw(torch.empty(1001, device=dev), base, 7, 3, s2)
    assert torch.isfinite(odd).all() and float(odd.min()) > 0 and s2[:3].cpu().tolist() == [1, 0, 0]


def test_drop_pool_in_captured_graphs(dev):
    """The multiplier pool under graph capture: every captured draw owns a generator state, so two graphs
    captured on one stream and replayed side by side never share one, each replay draws new values, and the
    values stay log-normal.  (A first eager draw sets the state table up, as a trainer's warm-up run does.)"""
    from eeggan_amd import layers
    from eeggan_amd.graphs import capture
    pool = layers._DropPool()
    n, base = 50_000, 1.2
    first = pool.take(n, base, dev).clone()
    outs, graphs = [], []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with capture(g):
            pool.reset()
            outs.append(pool.take(n, base, dev))
        graphs.append(g)
    side = torch.cuda.Stream()
    seen = [first]
    for _ in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graphs[1].replay()
        graphs[0].replay()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        seen += [o.clone() for o in outs]
    for i, u in enumerate(seen):
        z = torch.log(u.double()) / np.log(base)
        assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1.0) < 0.04
        for v in seen[:i]:
            assert abs(float(torch.corrcoef(torch.stack([torch.log(u.double()), torch.log(v.double())]))[0, 1])) < 0.02
    table = pool._tables[str(dev)][0]
    assert table[:3, 0].cpu().tolist() == [1, 3, 3] and int(table[3:].abs().sum()) == 0


def test_conv_rejects_bad_arguments(dev):
    ops = _ops()
    from eeggan_amd._lib import EegganError
    x = torch.randn(2, 4, 8, device=dev)
    with pytest.raises(EegganError):
        ops.conv1d_raw(x, torch.randn(3, 4, 5, device=dev))  # kernel size 5
    with pytest.raises(RuntimeError):
        ops.conv1d_raw(x, torch.randn(3, 5, 3, device=dev))  # channel mismatch
    with pytest.raises(RuntimeError):
        ops.conv1d_raw(x.cpu(), torch.randn(3, 4, 3))  # no CPU path


@pytest.mark.parametrize("B", [4, 32, 40, 64, 71, 130, 131, 512])
def test_head_tail(dev, B):
    ops = _ops()
    from eeggan_amd._lib import call
    g = torch.Generator().manual_seed(B)
    Cin, Cout, KT = 256, 256, 32
    st = torch.cuda.current_stream().cuda_stream
    w = torch.randn(Cout, Cin, KT, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    scale = 0.8
    # ---- head: conv(k=KT, pad=KT-1) on a length-1 signal
    z = torch.randn(B, Cin, generator=g)
    z64 = z.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    lin = F.conv1d(z64[:, :, None] * scale, w64, b.double(), padding=KT - 1)
    yref = F.leaky_relu(lin, 0.2)
    zd, wd, bd = z.to(dev), w.to(dev), b.to(dev)
    y = torch.empty(B, Cout, KT, device=dev)
    call("eeggan_head_fwd", zd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, y.data_ptr(), B, Cin, Cout, KT, scale,
         1, 0.2, st)
    assert rel(y, yref) < TOL
    gy = torch.randn(B, Cout, KT, generator=g)
    gz_ref, gw_ref = torch.autograd.grad(lin, [z64, w64], gy.double())
    gw = torch.empty_like(wd)
    gb = torch.empty(Cout, device=dev)
    gyd = gy.to(dev)
    call("eeggan_head_wgrad", zd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, Cin, Cout, KT, scale, st)
    assert rel(gw, gw_ref) < TOL
    assert rel(gb, gy.double().sum(dim=(0, 2))) < TOL
    gz = torch.empty_like(zd)
    call("eeggan_head_dgrad", gyd.data_ptr(), wd.data_ptr(), gz.data_ptr(), B, Cin, Cout, KT, scale, st)
    assert rel(gz, gz_ref) < TOL
    # ---- tail: conv(k=KT, pad=0) on a length-KT signal
    x = torch.randn(B, Cin, KT, generator=g)
    drop = torch.rand(B, Cout, generator=g) + 0.5
    x64 = x.double().requires_grad_(True)
    lin = F.conv1d(x64 * scale, w64, b.double())
    yref = F.leaky_relu(lin, 0.2) * drop.double()[:, :, None]
    yt = ops._tail_fwd_raw(x.to(dev), wd, bd, drop.to(dev), scale, 1, 0.2)
    assert rel(yt, yref) < TOL
    gy = torch.randn(B, Cout, 1, generator=g)
    gx_ref, gw_ref = torch.autograd.grad(lin, [x64, w64], gy.double())
    assert rel(ops._tail_dgrad_raw(gy.to(dev), wd, scale), gx_ref) < TOL
    gw, gb = ops._tail_wgrad_raw(x.to(dev), gy.to(dev), scale, True)
    assert rel(gw, gw_ref) < TOL
    assert rel(gb, gy.double().sum(dim=(0, 2))) < TOL


@pytest.mark.parametrize("B,C,T", [(4, 64, 1920), (3, 256, 32), (2, 9, 20), (8, 256, 1), (2, 128, 250), (1280, 256, 32), (640, 128, 128),
                                   (300, 192, 64), (32, 128, 256), (32, 256, 128), (64, 64, 1920),
                                   (400, 64, 1920)])  # (the last: tensors beyond the caches, the streaming forms)
def test_pixelnorm_and_masks(dev, B, C, T):
    ops = _ops()
    O = _oracle()
    g = torch.Generator().manual_seed(C + T)
    c = torch.randn(B, C, T, generator=g)
    drop = torch.rand(B, C, generator=g) + 0.5
    c64 = c.double().requires_grad_(True)
    h = F.leaky_relu(c64, 0.2)
    yref = O.pixel_norm(h) * drop.double()[:, :, None]
    y, d = ops.pixelnorm_drop_fwd_raw(F.leaky_relu(c, 0.2).to(dev), drop.to(dev))
    assert rel(y, yref) < TOL
    gy = torch.randn(B, C, T, generator=g)
    (gc_ref,) = torch.autograd.grad(yref, c64, gy.double())
    gc = ops.pixelnorm_drop_bwd_raw(gy.to(dev), y, drop.to(dev), d, 0.2)
    assert rel(gc, gc_ref) < TOL
    # leaky-relu + drop backward
    yl = F.leaky_relu(c, 0.2) * drop[:, :, None]
    out = ops.lrelu_drop_bwd_raw(gy.to(dev), yl.to(dev), drop.to(dev), 0.2)
    assert rel(out, gy.double() * torch.where(c > 0, 1.0, 0.2) * drop.double()[:, :, None]) < TOL
    # axpby / per-sample mix
    a, b2 = torch.randn(B, C, T, generator=g), torch.randn(B, C, T, generator=g)
    assert rel(ops.axpby_raw(a.to(dev), 0.3, b2.to(dev), 0.7), 0.3 * a.double() + 0.7 * b2.double()) < 1e-6
    mix = torch.rand(B, 1, 1, generator=g)
    assert rel(ops.mix_batch_raw(a.to(dev), b2.to(dev), mix.to(dev)), mix.double() * a + (1 - mix.double()) * b2) < 1e-6


def test_resample_all_ratios(dev):
    from eeggan_amd.utils import resample_signal
    O = _oracle()
    up, dn = [4, 2, 2, 3, 4, 5, 3], [1, 1, 1, 2, 3, 4, 2]
    T = 32
    g = torch.Generator().manual_seed(5)
    for i in range(7):
        x = torch.randn(3, 4, T, generator=g)
        x64 = x.double().requires_grad_(True)
        yref = O.resample(x64, dn[i], up[i])
        xd = x.to(dev).requires_grad_(True)
        y = resample_signal(xd, dn[i], up[i])
        assert y.shape == yref.shape
        # exact-arithmetic oracle: the float32 interpolation weights of the reference framework
        # (reproduced by the table builder) differ from exact ones by up to ~3e-5 at T ~ 1000
        assert rel(y, yref) < 5e-5
        # same float32 arithmetic as the reference's CPU path
        y32 = x
        if up[i] != 1:
            y32 = F.interpolate(y32, scale_factor=up[i], mode="linear", align_corners=False)
        if dn[i] != 1:
            y32 = F.avg_pool1d(y32, dn[i], dn[i], 0, False, True)
        assert rel(y, y32) < 5e-6
        gy = torch.randn(yref.shape, generator=g)
        (gref,) = torch.autograd.grad(yref, x64, gy.double())
        (gx,) = torch.autograd.grad(y, xd, gy.to(dev))
        assert rel(gx, gref) < 5e-5
        # and back down
        z = torch.randn(3, 4, y.shape[2], generator=g)
        assert rel(resample_signal(z.to(dev), up[i], dn[i]), O.resample(z.double(), up[i], dn[i])) < 5e-5
        T = y.shape[2]
    assert T == 1920
    x = torch.randn(2, 3, 1, generator=g)
    assert rel(resample_signal(x.to(dev), 1, 4), x.repeat(1, 1, 4)) == 0.0


def test_resample_quad_form_matches_sparse_form(dev):
    """The 16-byte quad kernel against the sparse-tap kernel (which the test above pins to the oracle), both
    directions of every stage and their transposes, a row count that does not fill the last workgroup.  The
    same taps in the same order; the compiler fuses the multiply-adds of the two loops differently, so the
    results agree to the last bit or two (5e-7), not bit for bit."""
    ops = _ops()
    from eeggan_amd.resample_tables import device_tables
    up, dn = [4, 2, 2, 3, 4, 5, 3], [1, 1, 1, 2, 3, 4, 2]
    T, used = 32, 0
    for i in range(7):
        for p, q, t_in in ((up[i], dn[i], T), (dn[i], up[i], T * up[i] // dn[i])):
            for table in device_tables(t_in, p, q, dev):
                if table[5] is None:
                    continue
                x = torch.randn(3, 7, table[2], device=dev)
                try:
                    ops.USE_RESAMPLE_QUADS = False
                    ref = ops._gather_raw(x, table)
                finally:
                    ops.USE_RESAMPLE_QUADS = True
                got = ops._gather_raw(x, table)
                assert rel(got, ref.double()) < 5e-7, (t_in, p, q)
                # quad starts from the closed form (the default) and from the table: the same loads, bit for bit
                assert table[5][3] is not None
                try:
                    ops.USE_RESAMPLE_AFFINE = False
                    assert torch.equal(ops._gather_raw(x, table), got), (t_in, p, q)
                finally:
                    ops.USE_RESAMPLE_AFFINE = True
                used += 1
        T = T * up[i] // dn[i]
    assert used >= 24


def test_resample_beyond_the_caches(dev):
    """From 192 MB of input + output on, the quad kernel stores with the streaming hint: the first rows equal a
    small launch's bit for bit, the last ones the sparse-tap kernel's to the last bit or two."""
    ops = _ops()
    from eeggan_amd.resample_tables import device_tables
    fwd, _ = device_tables(1280, 3, 2, dev)
    x = torch.randn(280, 64, 1280, device=dev)  # 92 MB in, 138 MB out
    y = ops._gather_raw(x, fwd)
    assert torch.equal(y[:3], ops._gather_raw(x[:3].contiguous(), fwd))
    try:
        ops.USE_RESAMPLE_QUADS = False
        ref = ops._gather_raw(x[-3:].contiguous(), fwd)
    finally:
        ops.USE_RESAMPLE_QUADS = True
    assert rel(y[-3:], ref.double()) < 5e-7


@pytest.mark.parametrize("B", [4, 8, 32, 2])
def test_minibatch_stddev_orders(dev, B):
    ops = _ops()
    O = _oracle()
    g = torch.Generator().manual_seed(B)
    C, T = 24, 32
    x = torch.randn(B, C, T, generator=g)
    x64 = x.double().requires_grad_(True)
    yref = O.minibatch_stddev(x64, 4, 32)
    xd = x.to(dev).requires_grad_(True)
    y = ops.minibatch_stddev(xd, min(B, 4))
    assert rel(y, yref) < TOL
    gy = torch.randn(B, C + 1, T, generator=g)
    gy64 = gy.double().requires_grad_(True)
    gyd = gy.to(dev).requires_grad_(True)
    (gx_ref,) = torch.autograd.grad(yref, x64, gy64, create_graph=True)
    (gx,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
    assert rel(gx, gx_ref) < TOL
    v = torch.randn(B, C, T, generator=g)
    ggy_ref, gxx_ref = torch.autograd.grad(gx_ref, [gy64, x64], v.double())
    ggy, gxx = torch.autograd.grad(gx, [gyd, xd], v.to(dev))
    assert rel(ggy, ggy_ref) < TOL
    # with two members per group the second derivative w.r.t. x vanishes identically: measure the
    # error against the scale of the cotangent, not against a zero reference
    err = (gxx.double().cpu() - gxx_ref).norm() / (gxx_ref.norm() + (1.0 if B == 2 else 1e-3) * v.double().norm())
    assert float(err) < 5 * TOL


@pytest.mark.parametrize("t_in,t_out", [(16, 256), (8, 128), (256, 16), (4, 512)])
def test_resample_large_ratios(dev, t_in, t_out):
    """Ratios far beyond the progression's (a conditioning signal of a few steps stretched to a layer's length,
    reference layers.py:128, 132): the transposed table of an upsampling by 16 has 40 taps per position, more than
    the register-resident kernel holds; forward and transpose against torch's interpolate / avg_pool in fp64."""
    from eeggan_amd.utils import resample_signal
    g = torch.Generator().manual_seed(t_in + t_out)
    x = torch.randn(3, 7, t_in, generator=g)
    x64 = x.double().requires_grad_(True)
    if t_out > t_in:
        ref = F.interpolate(x64, scale_factor=t_out // t_in, mode="linear", align_corners=False)
    else:
        ref = F.avg_pool1d(x64, t_in // t_out, t_in // t_out)
    xd = x.to(dev).requires_grad_(True)
    y = resample_signal(xd, t_in, t_out)
    assert tuple(y.shape) == tuple(ref.shape) and rel(y, ref) < 5e-6
    gy = torch.randn(ref.shape, generator=g)
    (gx_ref,) = torch.autograd.grad(ref, x64, gy.double())
    (gx,) = torch.autograd.grad(y, xd, gy.to(dev))
    assert rel(gx, gx_ref) < 5e-6


@pytest.mark.parametrize("T,stride,segments", [(32, 16, 1), (32, 6, 1), (48, 32, 2), (40, 8, 2)])
def test_minibatch_stddev_temporal_windows(dev, T, stride, segments):
    """Statistics per temporal window of ``stride`` steps (reference layers.py:72, 81-89; a last window shorter
    than the others when the stride does not divide the signal), to second order, alone and on stacked minibatches."""
    ops = _ops()
    O = _oracle()
    g = torch.Generator().manual_seed(T + stride)
    B, C = 8, 24
    x = torch.randn(segments * B, C, T, generator=g)
    x64 = x.double().requires_grad_(True)
    yref = torch.cat([O.minibatch_stddev(part, 4, stride) for part in x64.chunk(segments)])
    xd = x.to(dev).requires_grad_(True)
    y = ops.minibatch_stddev(xd, 4, segments, stride)
    assert rel(y, yref) < TOL
    gy = torch.randn(segments * B, C + 1, T, generator=g)
    gy64, gyd = gy.double().requires_grad_(True), gy.to(dev).requires_grad_(True)
    (gx_ref,) = torch.autograd.grad(yref, x64, gy64, create_graph=True)
    (gx,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
    assert rel(gx, gx_ref) < TOL
    v = torch.randn(segments * B, C, T, generator=g)
    ggy_ref, gxx_ref = torch.autograd.grad(gx_ref, [gy64, x64], v.double())
    ggy, gxx = torch.autograd.grad(gx, [gyd, xd], v.to(dev))
    assert rel(ggy, ggy_ref) < TOL
    assert float((gxx.double().cpu() - gxx_ref).norm() / (gxx_ref.norm() + 1e-3 * v.double().norm())) < 5 * TOL
    from eeggan_amd._lib import EegganError, call
    with pytest.raises(EegganError):  # a window that leaves the signal is refused on the host
        call("eeggan_mbstd_bwd_window", xd.data_ptr(), gyd.data_ptr(), gx.data_ptr(), B, C, T, T - 2, 4, 4, 0)


def test_losses(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, per = 32, 5 * 1920
    dr = torch.randn(B, generator=g)
    df = torch.randn(B, generator=g)
    gg = torch.randn(B, per, generator=g) * 0.01
    a, b, c = (t.double().requires_grad_(True) for t in (dr, df, gg))
    ref = b.mean() - a.mean() + (a ** 2).mean() * 1e-3 + ((c.norm(p=2, dim=1) - 1.0) ** 2).mean() * 10.0
    ra, rb, rc = torch.autograd.grad(ref, [a, b, c])
    ad, bd, cd = (t.to(dev).requires_grad_(True) for t in (dr, df, gg))
    loss = ops.wgan_gp_d_loss(ad, bd, cd, 1e-3, 10.0, 1.0)
    assert rel(loss, ref) < TOL
    ga, gb, gc = torch.autograd.grad(loss, [ad, bd, cd])
    assert rel(ga, ra) < TOL and rel(gb, rb) < TOL and rel(gc, rc) < TOL
    gl = ops.mean_scale(bd, -1.0)
    assert rel(gl, -b.mean()) < TOL
    (gbb,) = torch.autograd.grad(gl, bd)
    assert rel(gbb, torch.full((B,), -1.0 / B)) < 1e-6


def test_adam_matches_torch(dev):
    from eeggan_amd.optim import Adam
    g = torch.Generator().manual_seed(9)
    shapes = [(256, 256, 32), (64, 64, 3), (64,), (5, 64, 1), (1,), (7,), (4099,), (3, 1367)]  # (whole chunks, ragged ends, < 4)
    ps = [torch.randn(s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    ref = [torch.nn.Parameter(p.clone().double()) for p in ps]
    om = Adam(mine, 0.008, betas=(0.0, 0.99), eps=1e-8)
    orf = torch.optim.Adam(ref, 0.008, betas=(0.0, 0.99), eps=1e-8)
    for it in range(3):
        for pm, pr in zip(mine, ref):
            gr = torch.randn(pr.shape, generator=g)
            if it == 1 and pr.dim() == 1:
                pm.grad, pr.grad = None, None  # parameters without a gradient are skipped
                continue
            pm.grad, pr.grad = gr.to(dev), gr.double()
        om.step()
        orf.step()
    for pm, pr in zip(mine, ref):
        assert rel(pm, pr) < 1e-6
    sd = om.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == 3.0 and float(sd["state"][2]["step"]) == 2.0


def test_per_op_golden(dev):
    """The reference's own per-op vectors, through the HIP kernels."""
    ops = _ops()
    from eeggan_amd.utils import resample_signal
    fx = load_fixture("per_op")
    t = lambda k: torch.from_numpy(fx[k]).float().to(dev)
    for tag in ("k3", "k1"):
        y = ops.conv1d_raw(t(f"conv_{tag}_x"), t(f"conv_{tag}_w"), t(f"conv_{tag}_b"), None,
                           in_scale=float(fx[f"conv_{tag}_scale"]))
        assert rel(y, fx[f"conv_{tag}_y"]) < TOL
        gx = ops.conv1d_raw(t(f"conv_{tag}_gy"), t(f"conv_{tag}_w"), None, None,
                            in_scale=float(fx[f"conv_{tag}_scale"]), transposed=True)
        assert rel(gx, fx[f"conv_{tag}_gx"]) < TOL
        gw, gb = ops.conv1d_wgrad_raw(t(f"conv_{tag}_x"), t(f"conv_{tag}_gy"), fx[f"conv_{tag}_w"].shape[2],
                                      in_scale=float(fx[f"conv_{tag}_scale"]))
        assert rel(gw, fx[f"conv_{tag}_gw"]) < TOL and rel(gb, fx[f"conv_{tag}_gb"]) < TOL
    y, _ = ops.pixelnorm_drop_fwd_raw(t("pn_h"), None)
    assert rel(y, fx["pn_y"]) < TOL
    for i in range(7):
        up, dn = [4, 2, 2, 3, 4, 5, 3][i], [1, 1, 1, 2, 3, 4, 2][i]
        assert rel(resample_signal(t(f"rs_up{i}_x"), dn, up), fx[f"rs_up{i}_y"]) < 5e-5
        assert rel(resample_signal(t(f"rs_dn{i}_x"), up, dn), fx[f"rs_dn{i}_y"]) < 5e-5
    assert rel(ops.minibatch_stddev(t("mb_x"), 4), fx["mb_y"]) < TOL
    # MinibatchStddev to second order: gradient, and the gradient of <gradient, v> w.r.t. both of its inputs
    xd, gyd = t("mb_x").requires_grad_(True), t("mb_gy").requires_grad_(True)
    (gx,) = torch.autograd.grad(ops.minibatch_stddev(xd, 4), xd, gyd, create_graph=True)
    assert rel(gx, fx["mb_gx"]) < TOL
    ggy, gxx = torch.autograd.grad(gx, [gyd, xd], t("mb_v"))
    assert rel(ggy, fx["mb_ggy"]) < TOL and rel(gxx, fx["mb_gxx"]) < 5 * TOL
    # pixel norm backward (drop absent, slope 1: the norm alone)
    yd, dd = ops.pixelnorm_drop_fwd_raw(t("pn_h"), None)
    assert rel(ops.pixelnorm_drop_bwd_raw(t("pn_gy"), yd, None, dd, 1.0), fx["pn_gh"]) < TOL
    # a length-1 signal is repeated
    assert rel(resample_signal(t("rs_rep_x"), 1, 4), fx["rs_rep_y"]) < 1e-7
    # the full-length kernels: generator head (k = 32, pad = 31 on a length-1 signal) and critic tail (pad = 0)
    from eeggan_amd._lib import call
    st = torch.cuda.current_stream().cuda_stream
    x, w, b, gy = t("conv_head_x"), t("conv_head_w"), t("conv_head_b"), t("conv_head_gy")
    B, Cin = x.shape[0], x.shape[1]
    Cout, _, KT = w.shape
    sc = float(fx["conv_head_scale"])
    y = torch.empty(B, Cout, KT, device=dev)
    call("eeggan_head_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, Cin, Cout, KT, sc, 0, 0.0, st)
    assert rel(y, fx["conv_head_y"]) < TOL
    gz, gw, gb = torch.empty(B, Cin, device=dev), torch.empty_like(w), torch.empty(Cout, device=dev)
    call("eeggan_head_dgrad", gy.data_ptr(), w.data_ptr(), gz.data_ptr(), B, Cin, Cout, KT, sc, st)
    call("eeggan_head_wgrad", x.data_ptr(), gy.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, Cin, Cout, KT, sc, st)
    assert rel(gz, fx["conv_head_gx"]) < TOL and rel(gw, fx["conv_head_gw"]) < TOL and rel(gb, fx["conv_head_gb"]) < TOL
    x, w, b, gy = t("conv_tail_x"), t("conv_tail_w"), t("conv_tail_b"), t("conv_tail_gy")
    sc = float(fx["conv_tail_scale"])
    assert rel(ops._tail_fwd_raw(x, w, b, None, sc, 0, 0.0), fx["conv_tail_y"]) < TOL
    assert rel(ops._tail_dgrad_raw(gy, w, sc), fx["conv_tail_gx"]) < TOL
    gw, gb = ops._tail_wgrad_raw(x, gy, sc, True)
    assert rel(gw, fx["conv_tail_gw"]) < TOL and rel(gb, fx["conv_tail_gb"]) < TOL
    y = ops.conv1d_raw(t("tanh_c"), torch.eye(5, device=dev)[:, :, None].contiguous(), None, None, act=2,
                       act_param=0.5)
    assert rel(y, fx["tanh_y"]) < TOL


def test_real_data_side_on_device(dev):
    """DepthResampler (the device form of the reference dataset's per-sample resampling and fade,
    dataset.py:161-187) against the reference's own outputs for the same windows; the resampling
    tables reproduce ATen's float32 interpolation weights, hence the tight bar."""
    from eeggan_amd import config
    from eeggan_amd.dataset import DepthResampler
    fx = load_fixture("dataset_ops")
    w = torch.from_numpy(fx["windows"]).to(dev)
    rs = DepthResampler(config.PROGRESSION_SCALE_UP, config.PROGRESSION_SCALE_DOWN)
    checked = 0
    for depth in range(8):
        for alpha in (1.0, 0.3):
            key = f"x_d{depth}_a{alpha:g}"
            if key not in fx:
                continue
            rs.model_depth, rs.alpha = depth, alpha
            got = rs(w)["x"]
            assert tuple(got.shape) == fx[key].shape
            e = rel(got, fx[key])
            assert e < 1e-5, (key, e)
            checked += 1
    assert checked == 15
    rs.model_depth, rs.alpha = 0, 0.5
    with pytest.raises(ValueError):
        rs(w)


@pytest.mark.parametrize("B,Cin,Cout,T", [(8, 64, 64, 1024), (9, 48, 40, 1000), (34, 64, 64, 1920)])
def test_conv_pixelnorm_fused(dev, B, Cin, Cout, T):
    """The generator layer in one kernel: conv -> LeakyReLU -> PixelNorm over channels -> multiplier
    (layers.py:196-213, utils.py:133-136), against float64; also the normaliser it hands to backward."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, 3, generator=g)
    b = torch.randn(Cout, generator=g)
    drop = torch.rand(B, Cout, generator=g) + 0.5
    out = ops.conv1d_pixelnorm_raw(x.to(dev), w.to(dev), b.to(dev), drop.to(dev), in_scale=0.21, slope=0.2)
    assert out is not None
    y, d = out
    h = F.leaky_relu(F.conv1d(x.double() * 0.21, w.double(), b.double(), padding=1), 0.2)
    dref = torch.rsqrt((h * h).mean(dim=1) + 1e-8)
    assert rel(d, dref) < TOL
    assert rel(y, h * dref[:, None, :] * drop.double()[:, :, None]) < TOL
    assert ops.conv1d_pixelnorm_raw(x[:2].to(dev), w.to(dev), None, None, in_scale=1.0, slope=0.2) is None  # too few subtiles


def test_scaled_tanh_module(dev):
    """ScaledTanh on its own (reference layers.py:17-23): tanh(scale * x), with its gradient."""
    from eeggan_amd.layers import ScaledTanh
    fx = load_fixture("per_op")
    x = torch.from_numpy(fx["tanh_c"]).float().to(dev).requires_grad_(True)
    y = ScaledTanh()(x)
    assert rel(y, fx["tanh_y"]) < TOL
    gy = torch.randn(y.shape, device=dev)
    (gx,) = torch.autograd.grad(y, x, gy)
    ref = gy.double().cpu() * 0.5 * (1 - torch.tanh(0.5 * torch.from_numpy(fx["tanh_c"])) ** 2)
    assert rel(gx, ref) < TOL


def test_flip_cache_of_a_copied_network(dev):
    """An input gradient through a deep copy of a layer uses the COPY's current weights: after the copy's
    weight changes, its flipped view is rebuilt even though the original's caches are up to date."""
    import copy
    from eeggan_amd.layers import EqualizedConv1d
    ops = _ops()
    torch.manual_seed(3)
    conv = EqualizedConv1d(8, 12, 3, 1, False, True, "kaiming_normal", 0.2, True, 1, 1).to(dev)
    x = torch.randn(2, 8, 40, device=dev, requires_grad=True)
    gy = torch.randn(2, 12, 40, device=dev)
    torch.autograd.grad(conv(x), x, gy)  # builds the original's view
    twin = copy.deepcopy(conv)
    with torch.no_grad():
        twin.conv.weight.mul_(-2.0)
    (gx,) = torch.autograd.grad(twin(x), x, gy)
    ref = F.conv_transpose1d(gy.double().cpu(), twin.conv.weight.detach().double().cpu(), padding=1) * twin.scale
    assert rel(gx, ref) < TOL
    (gx0,) = torch.autograd.grad(conv(x), x, gy)  # the original still answers with its own weights
    ref0 = F.conv_transpose1d(gy.double().cpu(), conv.conv.weight.detach().double().cpu(), padding=1) * conv.scale
    assert rel(gx0, ref0) < TOL
