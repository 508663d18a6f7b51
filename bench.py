#!/usr/bin/env python3
"""Benchmark of the EEG-GAN training step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; gradients all-reduced over RCCL)

One step = one ``Trainer.train()``: 5 discriminator iterations (WGAN-GP, fresh reals and latents)
+ 1 generator iteration, each with Adam, at the FINAL progression stage (depth 7, T = 1920,
5 channels), alpha = 1, fp32, 32 samples per GPU (the reference's default minibatch at that
stage), synthetic EEG-like reals resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  ``value`` counts local-batch training steps per second summed
over all ranks (weak scaling: the global batch is 32 x N).
"""
import argparse
import functools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, ALPHA, BATCH = 7, 1.0, 32
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak of MI355X (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2516.6  # 256 CUs x 4 SIMDs x 32*32*16*2 flop / 32 cycles x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--drop-mode", default="device", choices=["device", "numpy"],
                    help="'device': GDrop noise drawn on the GPU (same distribution); "
                         "'numpy': the reference's host RNG stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying HIP graphs")
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--alpha", type=float, default=ALPHA, help="fade-in factor (< 1: mid-transition step)")
    ap.add_argument("--per-iteration-fakes", action="store_true",
                    help="one generator pass per critic iteration instead of one per step (comparison runs)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="reals start in pinned host memory and cross PCIe inside the timed region "
                         "(a side measurement; the headline value keeps them resident in HBM)")
    return ap.parse_args()


def make_trainer(dev, depth, alpha, batch, seed, world, rank, eager=False, host_inputs=False):
    from eeggan_amd import config, layers
    from eeggan_amd.optim import Adam
    from eeggan_amd.parallel import DataParallelStep
    from eeggan_amd.graphs import GraphedTrainer
    from eeggan_amd.trainer import Trainer
    from eeggan_amd.dataset import synthetic_eeg

    g, d = config.build_models(config.RANDOM_SEED)
    g, d = g.to(dev), d.to(dev)
    g.depth = d.depth = depth
    g.alpha = d.alpha = alpha
    lr = config.stage_lr(depth)
    opt_g = Adam(g.parameters(), lr, **config.ADAM)
    opt_d = Adam(d.parameters(), lr, **config.ADAM)
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed + rank)
    pool = [torch.from_numpy(synthetic_eeg(batch, config.NUM_CHANNELS, T, rng)) for _ in range(8)]
    pool = [t.pin_memory() if host_inputs else t.to(dev) for t in pool]
    gen = torch.Generator(device=dev).manual_seed(seed + rank)

    def reals():
        i = 0
        while True:
            yield {"x": pool[i % len(pool)]}
            i += 1

    def latents():
        while True:
            yield {"z": torch.randn(batch, config.LATENT_SIZE, device=dev, generator=gen)}

    class _DS:
        model_depth, alpha = depth, 1.0

    d_loss, g_loss = config.loss_closures()
    cls = Trainer if (eager or layers.get_drop_mode() != "device") else GraphedTrainer
    tr = cls(d, g, d_loss, g_loss, _DS(), latents(), 0, opt_g, opt_d)
    tr.dataiter = reals()
    tr.grad_sync = DataParallelStep() if world > 1 else None
    return tr


def conv_roofline(dev, batch, T):
    """Dominant kernel: the 64->64, k=3 same-padded conv at T=1920 (54 % of the step's FLOPs are
    64-channel k=3 convs).  Average launch duration measured with HIP events on the launch stream."""
    from eeggan_amd import ops
    C, K = 64, 3
    x = torch.randn(batch, C, T, device=dev)
    w = torch.randn(C, C, K, device=dev)
    b = torch.randn(C, device=dev)
    r = torch.rand(batch, C, device=dev) + 0.5
    for _ in range(5):
        ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2)
    n = 50
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2)
    e1.record(st)
    e1.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / n
    flops = 2.0 * batch * C * C * K * T
    achieved = flops / sec / 1e12
    # HBM bytes per launch from the PMC passes kept in profiles/*pmc_conv_ws_64x64_T1920_B32.csv
    # (2 x FETCH_SIZE + WRITE_SIZE = 2 x 11588.4 KiB + 15360 KiB; the algorithmic figure is 31.5 MB:
    # input + output once, weights, multipliers).  Only valid for the profiled shape.
    traffic = (2 * 9822.5625 + 15360.0) * 1024 if (batch, T) == (32, 1920) else None
    # The kernel forms every fp32 product from three bf16 pieces per operand (six bf16 MFMAs per 16
    # reduction indices, fp32-exact to 2^-24): `achieved`/`frac` price the ALGORITHMIC fp32 flops against
    # the fp32 matrix peak (the arithmetic type of the path); `pipe_frac` prices the six-fold executed
    # work against the bf16 pipe it runs on.  EEGGAN_FP32_MFMA=1 selects the fp32 instructions instead.
    fp32_mfma = os.environ.get("EEGGAN_FP32_MFMA") is not None
    out = {"bound": "mfma", "kernel": "conv_ws_kernel<K=3> (64->64 ch, k=3, T=1920, B=%d, fused bias+lrelu+drop)" % batch,
           "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
           "avg_launch_us": round(sec * 1e6, 2), "flop_per_launch": flops,
           "matrix_instruction": "v_mfma_f32_32x32x2_f32" if fp32_mfma else
           "6 x v_mfma_f32_32x32x16_bf16 per 16 reduction indices (three bf16 pieces per fp32 operand)"}
    if not fp32_mfma:
        out["pipe_frac"] = round(6.0 * achieved / BF16_MFMA_PEAK_TFLOPS, 4)
    return out


def cpu_baseline(depth, alpha, batch, seed):
    """The CPU restatement (oracle, kind 'port') timed on the host cores: whole steps for about ten
    seconds after one warm-up discriminator iteration."""
    from eeggan_amd import config
    from oracle import ref_cpu as O
    g, d = config.build_models(config.RANDOM_SEED)
    sc = lambda m: {n[:-5]: mod.scale for n, mod in m.named_modules() if type(mod).__name__ == "EqualizedConv1d"}
    gp = {k: v.detach().clone() for k, v in g.state_dict().items()}
    dp = {k: v.detach().clone() for k, v in d.state_dict().items()}
    tr = O.OracleTrainer(dp, sc(d), gp, sc(g), O.NetConfig(), depth, alpha, config.stage_lr(depth))
    T = config.seq_len(depth)
    rng = np.random.default_rng(seed)
    reals = [torch.from_numpy(O.synthetic_eeg(batch, 5, T, rng)) for _ in range(5)]
    zs = [torch.randn(batch, 256) for _ in range(6)]
    mixes = [torch.rand(batch, 1, 1) for _ in range(5)]
    src = O.DropSource("numpy")
    tr.d_iter(reals[0], zs[0], mixes[0], src, src)  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:  # whole steps until ~10 s of CPU work are on the clock (at most 8 steps)
        tr.step(reals, zs, mixes, src, src)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or steps >= 8:
            break
    return {"value": round(steps / dt, 5), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full steps (5 D + 1 G iterations with Adam each) after 1 warm-up D iteration, "
                      f"depth {depth}, B={batch}, fp32, torch CPU ops ({dt:.2f} s)"}


def main():
    args = parse()
    if os.environ.get("EEGGAN_HANG_DUMP"):  # debugging aid: stack dump of every thread after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["EEGGAN_HANG_DUMP"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        # EEGGAN_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
        backend = os.environ.get("EEGGAN_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the training step has no CPU path")
    dev = torch.device("cuda", local_rank)
    from eeggan_amd import layers
    layers.set_drop_mode(args.drop_mode)
    tr = make_trainer(dev, args.depth, args.alpha, args.batch, 1373, world, rank, args.eager, args.host_inputs)
    if args.per_iteration_fakes:
        tr.batch_fakes = False

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        tr.train()  # not timed, not counted: the first call captures the graphs
    except Exception as exc:  # noqa: BLE001 -- any capture failure: same kernels, launched one by one
        if args.eager:
            raise
        print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        tr = make_trainer(dev, args.depth, args.alpha, args.batch, 1373, world, rank, eager=True,
                          host_inputs=args.host_inputs)
    for _ in range(args.warmup):
        tr.train()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        from eeggan_amd import config
        T = config.seq_len(args.depth)
        out = {
            "metric": "G+D WGAN-GP train steps/sec (whole node) at final stage",
            "value": round(world * args.steps / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (reals copied from pinned host memory each iteration)" if args.host_inputs else ""),
            "config": {"workload": f"final stage depth {args.depth} (T={T}, 5 ch), alpha={args.alpha:g}, 5 D + 1 G iterations "
                                   f"with Adam per step, batch {args.batch}/GPU, global batch {args.batch * world}",
                       "parallelism": f"dp{world}", "drop_noise": args.drop_mode,
                       "launch": type(tr).__name__},
        }
        out["roofline"] = conv_roofline(dev, args.batch, T)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.depth, args.alpha, args.batch, 1373)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
