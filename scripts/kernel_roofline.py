"""Per-kernel roofline table: each kernel family of the training step timed stand-alone with HIP events
on the launch stream, at the per-GPU batch given on the command line, against the bound that limits it
(fp32 MFMA 157.3 TFLOP/s for the contractions, HBM 8 TB/s for the rest).  Work per launch is the
algorithmic figure: 2*B*Cout*Cin*K*T flop, or the bytes of the logical tensors read once and written once.

usage: python scripts/kernel_roofline.py [B ...]      (default: 32 512)
"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops, utils  # noqa: E402

MFMA_PEAK = 157.3e12
HBM_PEAK = 8.0e12
dev = torch.device("cuda", 0)


def timeit(fn, n):
    """Average duration of one call: n calls captured into one HIP graph (so that the host's per-call cost
    does not leave the GPU idle between them), the replay bracketed by events on the replay stream."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(n):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def rotating(fn, sets):
    """fn(*sets[i]) with i advancing on every call: the launches of a timed loop take their inputs in turn, so that
    none finds its input in a cache because the launch before it read the same bytes (with streaming stores a
    250 MB input otherwise stays in the 256 MB memory-side cache from launch to launch: 0.83 instead of 0.66 of the
    HBM peak for the resampler at 512 samples)."""
    turn = itertools.cycle(range(len(sets)))
    return lambda: fn(*sets[next(turn)])


def row(name, t, flop=None, nbytes=None):
    if flop is not None:
        print(f"{name:46s} {t * 1e6:9.1f} us  {flop / t / 1e12:7.1f} TFLOP/s  {flop / t / MFMA_PEAK:5.2f} of MFMA peak", flush=True)
    else:
        print(f"{name:46s} {t * 1e6:9.1f} us  {nbytes / t / 1e9:7.0f} GB/s     {nbytes / t / HBM_PEAK:5.2f} of HBM peak", flush=True)


def run(B):
    n = 30 if B <= 64 else 24  # (a graph launch costs ~20 us once per replay: 24 launches keep that under 1 % at 100 us each)
    print(f"--- per-GPU batch {B} ---", flush=True)
    convs = [(64, 64, 1920, 3), (64, 64, 1024, 3), (128, 128, 256, 3), (256, 256, 128, 3)]
    for cin, cout, T, K in convs:
        x = torch.randn(B, cin, T, device=dev)
        w = torch.randn(cout, cin, K, device=dev)
        b = torch.randn(cout, device=dev)
        r = torch.rand(B, cout, device=dev) + 0.5
        gy = torch.randn(B, cout, T, device=dev)
        fl = 2.0 * B * cin * cout * K * T
        tag = f"{cin}->{cout} k={K} T={T}"
        row(f"conv fwd+lrelu+gdrop {tag}", timeit(lambda: ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2), n), flop=fl)
        row(f"conv dgrad           {tag}", timeit(lambda: ops.conv1d_raw(gy, w, None, None, in_scale=0.1, transposed=True), n), flop=fl)
        row(f"conv wgrad+reduce    {tag}", timeit(lambda: ops.conv1d_wgrad_raw(x, gy, K, in_scale=0.1), n), flop=fl)
        if cin == 64 and ops.conv1d_pixelnorm_raw(x, w, b, r, in_scale=0.1, slope=0.2) is not None:
            row(f"conv fwd+lrelu+pixelnorm+gdrop {tag}",
                timeit(lambda: ops.conv1d_pixelnorm_raw(x, w, b, r, in_scale=0.1, slope=0.2), n), flop=fl)
        del x, w, b, r, gy
    # The weight gradients as the training step submits them: the k = 3 layers of one critic backward pass in ONE
    # grouped launch (the workgroups dealt out over the layers in proportion to their work) plus the multi-job slab
    # sums -- the single-layer "wgrad+reduce" rows above are the stand-alone entry point, which the step does not use.
    if B <= 64:
        layers = [(64, 64, 1920)] * 2 + [(64, 64, 1280)] * 2 + [(64, 64, 1024)] * 2 + [(64, 64, 768)] * 2 + \
                 [(64, 64, 512), (64, 128, 512), (128, 128, 256), (128, 256, 256), (256, 256, 128), (256, 256, 128),
                  (256, 256, 32)]
        gen = torch.Generator(device=dev).manual_seed(1)
        jobs = []
        for cin, cout, T in layers:
            x = torch.randn(B, cin, T, device=dev, generator=gen)
            gy = torch.randn(B, cout, T, device=dev, generator=gen)
            y = torch.randn(B, cout, T, device=dev, generator=gen)
            r = torch.rand(B, cout, device=dev, generator=gen) + 0.5
            jobs.append(dict(x=x, gy=gy, K=3, in_scale=0.1, mask_y=y, mask_drop=r, mask_slope=0.2))
        fl = sum(2.0 * B * cin * cout * 3 * T for cin, cout, T in layers)
        row(f"grouped wgrad + slab sums, {len(layers)} critic layers", timeit(lambda: ops.conv1d_wgrad_group_raw(jobs), n), flop=fl)
        del jobs
    # 1x1 convs on five channels: bandwidth-bound
    for cin, cout in ((64, 5), (5, 64)):
        T = 1920
        x = torch.randn(B, cin, T, device=dev)
        w = torch.randn(cout, cin, 1, device=dev)
        b = torch.randn(cout, device=dev)
        row(f"conv 1x1 {cin}->{cout} T={T}", timeit(lambda: ops.conv1d_raw(x, w, b, None, in_scale=0.1, act=1, act_param=0.2), n),
            nbytes=4.0 * B * T * (cin + cout))
        del x
    # bandwidth-bound kernels: four sets of inputs taken in turn (see rotating)
    C, T, NSET = 64, 1920, 4
    hs = [torch.randn(B, C, T, device=dev) for _ in range(NSET)]
    gs = [torch.randn(B, C, T, device=dev) for _ in range(NSET)]
    r = torch.rand(B, C, device=dev) + 0.5
    h = hs[0]
    y, d = ops.pixelnorm_drop_fwd_raw(h, r)
    pairs = list(zip(gs, hs))
    row("pixelnorm+gdrop fwd [B,64,1920]", timeit(rotating(lambda a: ops.pixelnorm_drop_fwd_raw(a, r), [(a,) for a in hs]), n),
        nbytes=2.0 * 4 * B * C * T)
    row("pixelnorm+gdrop+lrelu bwd [B,64,1920]", timeit(rotating(lambda g, a: ops.pixelnorm_drop_bwd_raw(g, a, r, d, 0.2), pairs), n),
        nbytes=3.0 * 4 * B * C * T)
    row("lrelu+gdrop bwd [B,64,1920]", timeit(rotating(lambda g, a: ops.lrelu_drop_bwd_raw(g, a, r, 0.2), pairs), n),
        nbytes=3.0 * 4 * B * C * T)
    row("axpby (blend) [B,64,1920]", timeit(rotating(lambda g, a: ops.axpby_raw(a, 0.3, g, 0.7), pairs), n),
        nbytes=3.0 * 4 * B * C * T)
    del gs, pairs
    with torch.no_grad():
        hcs = [(a[:, :, :1280].contiguous(),) for a in hs]
        row("resample 1280->1920 (x3/2) [B,64,.]", timeit(rotating(lambda a: utils.resample_signal(a, 1280, 1920), hcs), n),
            nbytes=4.0 * B * C * (1280 + 1920))
        del hcs
        row("resample 1920->1280 (x2/3) [B,64,.]", timeit(rotating(lambda a: utils.resample_signal(a, 1920, 1280), [(a,) for a in hs]), n),
            nbytes=4.0 * B * C * (1280 + 1920))
        del hs
        xm = torch.randn(B, 256, 32, device=dev)
        row("minibatch stddev fwd [B,256,32]", timeit(lambda: ops.minibatch_stddev(xm, 4), n), nbytes=4.0 * B * 32 * (256 + 257))
    real = torch.randn(B, 5, T, device=dev)
    fake = torch.randn(B, 5, T, device=dev)
    mix = torch.rand(B, device=dev)
    row("x_hat mix [B,5,1920]", timeit(lambda: ops.mix_batch_raw(real, fake, mix), n), nbytes=3.0 * 4 * B * 5 * T)


if __name__ == "__main__":  # (importable: bench_hbm.py reuses timeit/row)
    for B in ([int(a) for a in sys.argv[1:]] or [32, 512]):
        run(B)
