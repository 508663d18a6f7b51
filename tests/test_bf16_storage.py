"""The bf16 storage mode (``ops.set_storage('bf16')``): the wide activations between the conv layers ([B, C > 8,
T > 32]) and their gradients live in HBM as bfloat16; the signals ([B, 5, T]), the 32-sample stage, weights, weight
gradients, optimizer state and all arithmetic stay float32.  It implies the bf16 matrix precision.

(1) Every typed kernel does what it says: on bfloat16 inputs it equals the fp64 result of the same (rounded) inputs
up to fp32 summation, and its bfloat16 output is that result rounded once (2^-9 relative per element).
(2) The mode end to end on the reference fixtures, against the reference's fp64 run, to the bounds stated below
(the arithmetic behind tenths instead of 2^-9 is in test_bf16_mode.py: LeakyReLU branch flips)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel

pytestmark = pytest.mark.gpu

# stated bounds of the mode against the reference's fp64 run (forward; critic's summed gradient on the scale of its
# terms; single critic terms and generator gradients norm-wise)
FWD_BOUND, SUMMED_BOUND, GRAD_BOUND = 3e-2, 1.5e-1, 6e-1  # measured: 1.3e-2, 8.4e-2, 4.4e-1
ROUND = 2.0 ** -8  # a bfloat16 store: half an ulp of 8 significant bits, norm-wise well below this


@pytest.fixture()
def bf16_storage():
    from eeggan_amd import ops
    prev_p, prev_s = ops.get_precision(), ops.set_storage("bf16")
    try:
        yield
    finally:
        ops.set_storage(prev_s)
        ops.set_precision(prev_p)


def _r(t):  # a float tensor rounded to bfloat16, as fp64
    return t.bfloat16().double()


@pytest.mark.parametrize("shape", [(8, 64, 64, 1920), (16, 64, 64, 480), (4, 256, 256, 128), (6, 128, 256, 256),
                                   (4, 128, 64, 512)])
def test_conv_layers_on_bfloat16_tensors(dev, bf16_storage, shape):
    from eeggan_amd import ops
    from eeggan_amd._lib import ACT_LRELU, ACT_MASK
    B, Cin, Cout, T = shape
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T, generator=gen).bfloat16()
    w = torch.randn(Cout, Cin, 3, generator=gen)
    b = torch.randn(Cout, generator=gen)
    r = torch.rand(B, Cout, generator=gen) + 0.5
    gy = torch.randn(B, Cout, T, generator=gen).bfloat16()
    s = 0.07
    xd, wd, bd, rd, gyd = (t.to(dev) for t in (x, w, b, r, gy))
    # forward with the fused epilogue
    y = ops.conv1d_raw(xd, wd, bd, rd, in_scale=s, act=ACT_LRELU, act_param=0.2)
    assert y.dtype == torch.bfloat16
    ref = F.leaky_relu(F.conv1d(x.double(), _r(w * s), b.double(), padding=1), 0.2) * r.double()[:, :, None]
    assert rel(y, ref) < ROUND, rel(y, ref)
    assert rel(y, _r(ref.float())) < 2e-3  # ... i.e. the exact result rounded once, up to ties that fp32 sums move
    # input gradient with the fused backward mask of the layer's output
    gx = ops.conv1d_raw(gyd, wd, None, None, in_scale=s, transposed=True, mask_y=y, mask_drop=rd, mask_slope=0.2)
    assert gx.dtype == torch.bfloat16
    m = torch.where(y.double().cpu() > 0, 1.0, 0.2) * r.double()[:, :, None]
    ref = F.conv_transpose1d(gy.double() * m, _r(w * s), padding=1)
    assert rel(gx, ref) < ROUND, rel(gx, ref)
    # the second-order form: a conv whose OUTPUT takes the mask
    q = torch.randn(B, Cin, T, generator=gen).bfloat16()
    gg = ops.conv1d_raw(q.to(dev), wd, None, rd, in_scale=s, act=ACT_MASK, act_param=0.2, act_aux=y)
    ref = F.conv1d(q.double(), _r(w * s), padding=1) * m
    assert rel(gg, ref) < ROUND, rel(gg, ref)
    # weight gradient (float32 out): exact products of the stored values, fp32 sums
    (gw, gb), = ops.conv1d_wgrad_group_raw([dict(x=xd, gy=gyd, K=3, in_scale=s, mask_y=y, mask_drop=rd, mask_slope=0.2)])
    gm = gy.double() * m
    xp = F.pad(x.double(), (1, 1))
    ref = s * torch.stack([torch.einsum("bot,bit->oi", _r(gm.float()), xp[:, :, j:j + T]) for j in range(3)], dim=2)
    assert gw.dtype == torch.float32 and rel(gw, ref) < 2e-3, rel(gw, ref)  # (the masked gradient is rounded to bf16 as an operand)
    assert rel(gb, _r(gm.float()).sum(dim=(0, 2))) < 1e-4  # (the bias gradient sums the same rounded operand)


def test_fused_generator_layer_on_bfloat16(dev, bf16_storage):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(6)
    B, C, T = 8, 64, 1280
    x = torch.randn(B, C, T, generator=gen).bfloat16()
    w, b = torch.randn(C, C, 3, generator=gen), torch.randn(C, generator=gen)
    r = torch.rand(B, C, generator=gen) + 0.5
    out = ops.conv1d_pixelnorm_raw(x.to(dev), w.to(dev), b.to(dev), r.to(dev), in_scale=0.05, slope=0.2)
    assert out is not None
    y, d = out
    h = F.leaky_relu(F.conv1d(x.double(), _r(w * 0.05), b.double(), padding=1), 0.2)
    dn = torch.rsqrt((h * h).mean(dim=1) + 1e-8)
    assert y.dtype == torch.bfloat16 and d.dtype == torch.float32
    assert rel(d, dn) < 1e-5 and rel(y, h * dn[:, None, :] * r.double()[:, :, None]) < ROUND
    # the un-fused pair on a wide layer: conv, then norm
    x2 = torch.randn(4, 128, 256, generator=gen).bfloat16()
    y2, d2 = ops.pixelnorm_drop_fwd_raw(x2.to(dev), None)
    dn2 = torch.rsqrt((x2.double() ** 2).mean(dim=1) + 1e-8)
    assert y2.dtype == torch.bfloat16 and rel(d2, dn2) < 1e-5 and rel(y2, x2.double() * dn2[:, None, :]) < ROUND
    gy2 = torch.randn(4, 128, 256, generator=gen).bfloat16()
    gc = ops.pixelnorm_drop_bwd_raw(gy2.to(dev), y2, None, d2, 0.2)
    yv, dv = y2.double().cpu(), d2.double().cpu()
    S = (gy2.double() * yv).mean(dim=1, keepdim=True)
    ref = dv[:, None, :] * (gy2.double() - yv * S) * torch.where(yv > 0, 1.0, 0.2)
    assert gc.dtype == torch.bfloat16 and rel(gc, ref) < ROUND, rel(gc, ref)


@pytest.mark.parametrize("xb,yb", [(False, True), (True, False), (True, True), (False, False)])
def test_thin_layers_between_storage_types(dev, bf16_storage, xb, yb):
    """to_rgb / from_rgb and their gradients: float32 signals on the thin side, bfloat16 on the wide side."""
    from eeggan_amd import ops
    from eeggan_amd._lib import ACT_LRELU
    gen = torch.Generator().manual_seed(7)
    B, T = 4, 640
    cast = lambda t, bf: t.bfloat16() if bf else t
    for Cin, Cout in ((5, 64), (64, 5)):
        x = cast(torch.randn(B, Cin, T, generator=gen), xb)
        w, b = torch.randn(Cout, Cin, 1, generator=gen), torch.randn(Cout, generator=gen)
        out = torch.empty(B, Cout, T, device=dev, dtype=torch.bfloat16 if yb else torch.float32)
        y = ops.conv1d_raw(x.to(dev), w.to(dev), b.to(dev), None, in_scale=0.3, act=ACT_LRELU, act_param=0.2, out=out)
        ref = F.leaky_relu(F.conv1d(x.double(), (w * 0.3).double(), b.double()), 0.2)
        assert rel(y, ref) < (ROUND if yb else 1e-5), (Cin, Cout, rel(y, ref))
        # weight gradient with mixed operands
        gy = cast(torch.randn(B, Cout, T, generator=gen), yb)
        (gw, gb), = ops.conv1d_wgrad_group_raw([dict(x=x.to(dev), gy=gy.to(dev), K=1, in_scale=0.3)])
        ref = 0.3 * torch.einsum("bot,bit->oi", gy.double(), x.double())[:, :, None]
        assert rel(gw, ref) < 1e-5 and rel(gb, gy.double().sum(dim=(0, 2))) < 1e-5


def test_default_rules_for_output_types(dev, bf16_storage):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(8)
    sig = torch.randn(4, 5, 256, generator=gen).to(dev)
    w = torch.randn(64, 5, 1, generator=gen).to(dev)
    h = ops.conv1d_raw(sig, w, None, None)              # from_rgb: float32 signal -> bfloat16 activation
    assert h.dtype == torch.bfloat16
    back = ops.conv1d_raw(h, w, None, None, transposed=True)  # its input gradient: back to float32
    assert back.dtype == torch.float32
    small = ops.conv1d_raw(torch.randn(4, 256, 32, generator=gen).to(dev), torch.randn(256, 256, 3, generator=gen).to(dev),
                           None, None)
    assert small.dtype == torch.float32                 # the 32-sample stage stays float32
    ops.set_storage("fp32")
    assert ops.conv1d_raw(sig, w, None, None).dtype == torch.float32


@pytest.mark.parametrize("t_in,t_out", [(32, 128), (128, 32), (1280, 1920), (1920, 1280), (512, 768)])
def test_resampler_converts_between_storage_types(dev, bf16_storage, t_in, t_out):
    from eeggan_amd import ops
    from eeggan_amd.utils import resample_signal
    from oracle import ref_cpu as O
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 64, t_in, generator=gen)
    xin = x.bfloat16() if t_in > 32 else x
    y = resample_signal(xin.to(dev), t_in, t_out)
    assert y.dtype == (torch.bfloat16 if t_out > 32 else torch.float32)
    ref = O.resample(xin.double(), t_in, t_out)
    assert rel(y, ref) < (ROUND if t_out > 32 else 1e-4), rel(y, ref)
    sig = resample_signal(torch.randn(3, 5, t_in, generator=gen).to(dev), t_in, t_out)
    assert sig.dtype == torch.float32  # signals stay float32 whatever their length


def test_elementwise_forms(dev, bf16_storage):
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(10)
    a, b = (torch.randn(4, 64, 256, generator=gen).bfloat16() for _ in range(2))
    r = torch.rand(4, 64, generator=gen) + 0.5
    ad, bd = a.to(dev), b.to(dev)
    out = ops.lrelu_drop_bwd_raw(ad, bd, r.to(dev), 0.2)
    ref = a.double() * r.double()[:, :, None] * torch.where(b.double() > 0, 1.0, 0.2)
    assert out.dtype == torch.bfloat16 and rel(out, ref) < ROUND
    out = ops.axpby_raw(ad, 0.3, bd, 0.7)
    assert out.dtype == torch.bfloat16 and rel(out, 0.3 * a.double() + 0.7 * b.double()) < ROUND
    alpha = torch.tensor([0.25], device=dev)
    out = ops.blend(ad, bd, alpha)
    assert out.dtype == torch.bfloat16 and rel(out, 0.25 * a.double() + 0.75 * b.double()) < ROUND
    f = torch.randn(1000 + 3, generator=gen)
    assert torch.equal(ops.cast_raw(f.to(dev), torch.bfloat16).cpu(), f.bfloat16())  # round to nearest even, like torch
    assert torch.equal(ops.cast_raw(f.bfloat16().to(dev), torch.float32).cpu(), f.bfloat16().float())


def test_elementwise_forms_beyond_the_caches(dev, bf16_storage):
    """The streaming (non-temporal) forms, taken from 192 MB of tensors on: the same arithmetic, so the results equal
    those of the cached forms on the same values bit for bit (checked on the first rows, which a small launch of the
    cached form recomputes) and the float64 reference to bfloat16 rounding."""
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(11)
    B, C, T = 512, 64, 1920  # 126 MB per bfloat16 tensor
    a, b = (torch.randn(B, C, T, generator=gen).bfloat16() for _ in range(2))
    r = torch.rand(B, C, generator=gen) + 0.5
    ad, bd, rd = a.to(dev), b.to(dev), r.to(dev)
    head = slice(0, 4)
    out = ops.lrelu_drop_bwd_raw(ad, bd, rd, 0.2)
    assert torch.equal(out[head], ops.lrelu_drop_bwd_raw(ad[head].contiguous(), bd[head].contiguous(), rd[head].contiguous(), 0.2))
    ref = a[-2:].double() * r[-2:].double()[:, :, None] * torch.where(b[-2:].double() > 0, 1.0, 0.2)
    assert rel(out[-2:], ref) < ROUND
    out = ops.axpby_raw(ad, 0.3, bd, 0.7)
    assert torch.equal(out[head], ops.axpby_raw(ad[head].contiguous(), 0.3, bd[head].contiguous(), 0.7))
    assert rel(out[-2:], 0.3 * a[-2:].double() + 0.7 * b[-2:].double()) < ROUND
    # (the channel norm sums in another order at another batch size: its small launch agrees to rounding)
    y, d = ops.pixelnorm_drop_fwd_raw(ad, rd)
    y4, d4 = ops.pixelnorm_drop_fwd_raw(ad[head].contiguous(), rd[head].contiguous())
    assert rel(y[head], y4.double()) < ROUND and rel(d[head], d4.double()) < 1e-6
    g = ops.pixelnorm_drop_bwd_raw(bd, y, rd, d, 0.2)
    g4 = ops.pixelnorm_drop_bwd_raw(bd[head].contiguous(), y[head].contiguous(), rd[head].contiguous(), d[head].contiguous(), 0.2)
    assert rel(g[head], g4.double()) < ROUND


def test_unsupported_combinations_fail_loudly(dev, bf16_storage):
    from eeggan_amd import ops
    from eeggan_amd._lib import EegganError
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(4, 96, 64, generator=gen).bfloat16().to(dev)   # 96 output channels: not a predicate-free layer
    w = torch.randn(96, 96, 3, generator=gen).to(dev)
    with pytest.raises(EegganError):
        ops.conv1d_raw(x, w, None, None)
    ops.set_precision("fp32")  # bfloat16 tensors with fp32 products: refused as well
    x = torch.randn(4, 64, 256, generator=gen).bfloat16().to(dev)
    with pytest.raises(EegganError):
        ops.conv1d_raw(x, torch.randn(64, 64, 3, generator=gen).to(dev), None, None)


@pytest.mark.parametrize("name", ["e2e_d4_a1", "e2e_d7_a1", "e2e_d7_a05"])
def test_stated_bounds_against_reference_fixture(dev, bf16_storage, name):
    from test_bf16_mode import run_bounds
    run_bounds(dev, name, "bf16 storage", FWD_BOUND, SUMMED_BOUND, GRAD_BOUND)


BRANCH_BOUND, BRANCH_BIAS_BOUND = 2e-2, 1.5e-1


@pytest.mark.parametrize("depth,alpha,B", [(7, 1.0, 8), (7, 0.5, 8)])
def test_gradients_under_reference_branches(dev, bf16_storage, depth, alpha, B):
    """The mode with every LeakyReLU branch of the critic taken from the fp64 oracle: every term's gradient of every
    tensor within 2e-2 of its own norm (bf16 operands and bf16-stored activations, no branch flips); the penalty
    term's bias gradients, sums that cancel to a few per cent of their terms, within 1.5e-1 (test_bf16_mode.py)."""
    from test_bf16_mode import run_branch_bound
    run_branch_bound(dev, depth, alpha, B, "bf16 storage", BRANCH_BOUND, BRANCH_BIAS_BOUND)


def test_graph_trainer_runs_in_the_mode(dev, bf16_storage):
    """The benchmarked launch path (graphs, stacked pass, grouped weight gradients) with bfloat16 activations:
    finite losses, parameters that move, float32 parameters and gradients."""
    import bench
    from eeggan_amd import layers
    layers.set_drop_mode("device")
    try:
        tr = bench.make_trainer(dev, 7, 1.0, 8, 1373, 1, 0)
        before = [p.detach().clone() for p in tr.discriminator.parameters()]
        for _ in range(3):
            tr.train()
        torch.cuda.synchronize()
        moved = sum(float((p.detach() - q).abs().sum()) for p, q in zip(tr.discriminator.parameters(), before))
        assert np.isfinite(moved) and moved > 0
        assert all(p.dtype == torch.float32 and (p.grad is None or p.grad.dtype == torch.float32) for p in tr.discriminator.parameters())
        assert all(torch.isfinite(p).all() for p in list(tr.discriminator.parameters()) + list(tr.generator.parameters()))
    finally:
        layers.set_drop_mode("numpy")


def test_layer_at_512_samples(dev, bf16_storage):
    """The 64 -> 64 layer of the final stage at 512 samples per GPU (the configuration the mode is for): forward and
    masked input gradient on a few samples against fp64 on the host, the weight gradient (which at this size is
    submitted alone, in many splits) against an fp64 product of the same stored values on the device."""
    from eeggan_amd import ops
    from eeggan_amd._lib import ACT_LRELU
    B, C, T, s = 512, 64, 1920, 0.07
    gen = torch.Generator(device=dev).manual_seed(12)
    x = torch.randn(B, C, T, generator=gen, device=dev).bfloat16()
    gy = torch.randn(B, C, T, generator=gen, device=dev).bfloat16()
    w = torch.randn(C, C, 3, generator=gen, device=dev)
    b = torch.randn(C, generator=gen, device=dev)
    r = torch.rand(B, C, generator=gen, device=dev) + 0.5
    y = ops.conv1d_raw(x, w, b, r, in_scale=s, act=ACT_LRELU, act_param=0.2)
    gx = ops.conv1d_raw(gy, w, None, None, in_scale=s, transposed=True, mask_y=y, mask_drop=r, mask_slope=0.2)
    (gw, gb), = ops.conv1d_wgrad_group_raw([dict(x=x, gy=gy, K=3, in_scale=s, mask_y=y, mask_drop=r, mask_slope=0.2)])
    assert y.dtype == gx.dtype == torch.bfloat16 and gw.dtype == torch.float32
    pick = torch.tensor([0, 1, 255, 256, 510, 511])
    wq = _r((w * s).cpu())
    xs, gys, ys, rs = (t[pick.to(dev)].double().cpu() for t in (x, gy, y, r))
    ref = F.leaky_relu(F.conv1d(xs, wq, b.double().cpu(), padding=1), 0.2) * rs[:, :, None]
    assert rel(y[pick.to(dev)], ref) < ROUND
    m = torch.where(ys > 0, 1.0, 0.2) * rs[:, :, None]
    assert rel(gx[pick.to(dev)], F.conv_transpose1d(gys * m, wq, padding=1)) < ROUND
    # weight gradient: every sample contributes; the masked gradient is an operand rounded to bfloat16
    gm = (gy.float() * torch.where(y.float() > 0, 1.0, 0.2) * r[:, :, None])
    gm = gm.bfloat16().double()
    gb_ref = gm.sum(dim=(0, 2))  # (the bias gradient sums the same rounded operand)
    xp = F.pad(x.double(), (1, 1))
    ref = s * torch.stack([torch.einsum("bot,bit->oi", gm, xp[:, :, j:j + T]) for j in range(3)], dim=2)
    assert rel(gw, ref) < 1e-5, rel(gw, ref)
    assert rel(gb, gb_ref) < 1e-5, rel(gb, gb_ref)


def test_critic_iteration_at_512_samples_against_float32(dev, bf16_storage):
    """One critic iteration at the final stage with 512 samples, eager, the reference's host noise: the mode against
    the float32 arithmetic on the same inputs, to the bounds stated for it (the float32 step itself is held to the
    oracle at sizes the oracle finishes; every kernel here takes the launch shapes of the large batch)."""
    from eeggan_amd import config, layers, losses, ops
    layers.set_drop_mode("numpy")
    gen = torch.Generator().manual_seed(13)
    B, depth = 512, 7
    T = config.seq_len(depth)
    real = (0.5 * torch.randn(B, config.NUM_CHANNELS, T, generator=gen)).to(dev)
    z = torch.randn(B, config.LATENT_SIZE, generator=gen).to(dev)
    mix = torch.rand(B, 1, 1, generator=gen).to(dev)
    g, d = config.build_models(4)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = 1.0
    results = {}
    for mode in ("bf16", "fp32"):
        ops.set_storage(mode)
        ops.set_precision(mode)
        d.zero_grad(set_to_none=True)
        np.random.seed(99)
        with ops.direct_weight_grads():
            loss = losses.discriminator_loss(d, g, {"x": real}, {"z": z}, "wgan_gp", 1e-3, 10.0, 1.0, mixing=mix)
            loss.backward()
        results[mode] = (float(loss), {k: p.grad.detach().clone() for k, p in d.named_parameters() if p.grad is not None})
    ops.set_storage("bf16")
    (l16, g16), (l32, g32) = results["bf16"], results["fp32"]
    assert np.isfinite(l16) and abs(l16 - l32) < FWD_BOUND * max(1.0, abs(l32)), (l16, l32)
    assert set(g16) == set(g32) and len(g32) > 30
    worst = max((rel(g16[k], g32[k]), k) for k in g32 if float(g32[k].abs().max()) > 0)
    print(f"512 samples, bf16 storage against float32: loss {l16:.5f} / {l32:.5f}, worst gradient {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < GRAD_BOUND, worst


@pytest.mark.parametrize("xb,yb", [(False, True), (True, False), (True, True)])
def test_thin_wgrad_large_batch_between_storage_types(dev, bf16_storage, xb, yb):
    """The thin 1x1 layers' weight gradient at a large batch (its own vector kernel) with bfloat16 on either side."""
    from eeggan_amd import ops
    gen = torch.Generator().manual_seed(17)
    B, T = 140, 1920
    cast = lambda t, bf: t.bfloat16() if bf else t
    for Cin, Cout, masked in ((5, 64, True), (64, 5, False)):
        x = cast(torch.randn(B, Cin, T, generator=gen), xb)
        gy = cast(torch.randn(B, Cout, T, generator=gen), yb)
        job = dict(x=x.to(dev), gy=gy.to(dev), K=1, in_scale=0.3)
        geff = gy.double()
        if masked:
            y = cast(torch.randn(B, Cout, T, generator=gen), yb)
            drop = torch.rand(B, Cout, generator=gen) + 0.5
            job.update(mask_y=y.to(dev), mask_drop=drop.to(dev), mask_slope=0.2)
            geff = geff * torch.where(y.double() > 0, 1.0, 0.2) * drop.double()[:, :, None]
        (gw, gb), = ops.conv1d_wgrad_group_raw([job])
        ref = 0.3 * torch.einsum("bot,bit->oi", geff.to(dev), x.double().to(dev))[:, :, None]
        assert rel(gw, ref) < 1e-5 and rel(gb, geff.sum(dim=(0, 2))) < 1e-5, (Cin, Cout, rel(gw, ref))
