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
# (algorithmic conv flops of one step: config.step_flops, 8 F_G + 62 F_D per sample at every depth)
PMC_CONV_WS = os.path.join("profiles", "r03_pmc_conv_ws_64x64_T1920_B32.csv")
PER_KERNEL = os.path.join("profiles", "r03_bench_per_kernel.txt")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 steps are ~1.7 s at the headline configuration)")
    ap.add_argument("--warmup", type=int, default=5)
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
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the conv layers' matrix products: f32 (fp32 products, the headline) or bf16 "
                         "(operands rounded to bf16, fp32 accumulation and storage: the large-batch mode, looser bounds)")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16"],
                    help="storage of the wide activations between the conv layers and of their gradients: f32 (the "
                         "headline) or bf16 (large-batch mode; implies --dtype bf16; signals, the 32-sample stage, weights, "
                         "weight gradients and optimizer state stay float32)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="reals start in pinned host memory and cross PCIe inside the timed region "
                         "(a side measurement; the headline value keeps them resident in HBM)")
    return ap.parse_args()


def make_trainer(dev, depth, alpha, batch, seed, world, rank, eager=False, host_inputs=False, force_dist=False):
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
        # (the latents of a whole step -- five critic iterations and the generator iteration -- in one draw)
        while True:
            yield from ({"z": z} for z in torch.randn(6 * batch, config.LATENT_SIZE, device=dev, generator=gen).chunk(6))

    class _DS:
        model_depth, alpha = depth, 1.0

    d_loss, g_loss = config.loss_closures()
    cls = Trainer if (eager or layers.get_drop_mode() != "device") else GraphedTrainer
    tr = cls(d, g, d_loss, g_loss, _DS(), latents(), 0, opt_g, opt_d)
    tr.dataiter = reals()
    tr.grad_sync = DataParallelStep(always_bucket=force_dist) if (world > 1 or force_dist) else None
    return tr


HBM_PEAK_GBS = 8000.0


def conv_roofline(dev, batch, T):
    """Dominant kernel: the 64->64, k=3 same-padded conv at T=1920 (54 % of the step's FLOPs are
    64-channel k=3 convs), launched 50 times on the launch stream.  ``avg_launch_us`` is one pair of HIP events
    around the whole run of launches divided by their number; ``event_pair_us`` brackets EVERY launch with its
    own pair of events (and so includes the cost of the two event records)."""
    from eeggan_amd import ops
    C, K = 64, 3
    x = torch.randn(batch, C, T, device=dev)
    esize = 4.0
    if ops.get_storage() == "bf16":  # the layer's input and output are bfloat16 in this mode
        x, esize = x.bfloat16(), 2.0
    w = torch.randn(C, C, K, device=dev)
    b = torch.randn(C, device=dev)
    r = torch.rand(batch, C, device=dev) + 0.5
    launch = lambda: ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2)
    for _ in range(5):
        launch()
    n = 50
    st = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    for i in range(n):
        ev[2 * i].record(st)
        launch()
        ev[2 * i + 1].record(st)
    ev[-1].synchronize()
    per_launch = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n)) * 1e-3 / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        launch()
    e1.record(st)
    e1.synchronize()
    back_to_back = e0.elapsed_time(e1) * 1e-3 / n
    flops = 2.0 * batch * C * C * K * T
    # `achieved` uses the back-to-back figure: it is the one that agrees with the kernel's own duration in a
    # rocprofv3 kernel trace of this command (14.96 vs 14.84 us); a pair of events around every single launch also
    # times the two event records (17.4 us) and is reported beside it as `event_pair_us`.
    achieved = flops / back_to_back / 1e12
    algorithmic_bytes = esize * 2 * batch * C * T + 4.0 * (C * C * K + C + batch * C)
    # HBM bytes per launch of THIS kernel and shape from the two counter passes kept under profiles/ (the
    # algorithmic figure is 31.5 MB: input + output once, weights, multipliers).  Only for the profiled shape.
    pmc = measured_traffic() if (batch, T, esize) == (32, 1920, 4.0) else None
    traffic = pmc["bytes"] if pmc else None
    # The kernel forms every fp32 product from three bf16 pieces per operand (six bf16 MFMAs per 16
    # reduction indices, fp32-exact to 2^-24).  `achieved`/`frac` price the ALGORITHMIC fp32 flops against
    # the fp32 matrix peak (the arithmetic type of the path); `pipe_frac` prices the six-fold executed
    # work against the bf16 pipe it runs on; `bound_frac` prices the algorithmic flops against the tighter
    # of the two ceilings this formulation has: a sixth of the bf16 pipe, and HBM bandwidth x the layer's
    # arithmetic intensity.  EEGGAN_FP32_MFMA=1 selects the fp32 instructions instead.
    mode = ops.get_precision()
    fp32_mfma = mode == "fp32_mfma"
    if mode == "bf16":
        # bf16 compute mode: one bf16 MFMA per 16 reduction indices on the operands rounded to bf16; tensors stay
        # fp32 in memory, so the ceiling is the tighter of the bf16 pipe and HBM bandwidth x arithmetic intensity
        ceiling = min(BF16_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS * (flops / algorithmic_bytes) / 1e3)
        hbm_bound = ceiling < BF16_MFMA_PEAK_TFLOPS
        return {"bound": "hbm" if hbm_bound else "mfma",
                "kernel": "conv_ws_kernel<K=3> (64->64 ch, k=3, T=1920, B=%d, fused bias+lrelu+drop), bf16 products, %s tensors"
                          % (batch, "bfloat16" if esize == 2.0 else "float32"),
                "achieved": round(algorithmic_bytes / back_to_back / 1e9, 1) if hbm_bound else round(achieved, 3),
                "peak": HBM_PEAK_GBS if hbm_bound else BF16_MFMA_PEAK_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(achieved / ceiling, 4), "traffic": None, "avg_launch_us": round(back_to_back * 1e6, 2),
                "event_pair_us": round(per_launch * 1e6, 2), "flop_per_launch": flops,
                "algorithmic_bytes_per_launch": algorithmic_bytes, "tflops": round(achieved, 2),
                "pipe_frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                "matrix_instruction": "1 x v_mfma_f32_32x32x16_bf16 per 16 reduction indices (operands rounded to bf16)"}
    out = {"bound": "mfma", "kernel": "conv_ws_kernel<K=3> (64->64 ch, k=3, T=1920, B=%d, fused bias+lrelu+drop)" % batch,
           "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
           "avg_launch_us": round(back_to_back * 1e6, 2), "event_pair_us": round(per_launch * 1e6, 2),
           "flop_per_launch": flops, "algorithmic_bytes_per_launch": algorithmic_bytes,
           "matrix_instruction": "v_mfma_f32_32x32x2_f32" if fp32_mfma else
           "6 x v_mfma_f32_32x32x16_bf16 per 16 reduction indices (three bf16 pieces per fp32 operand)"}
    if pmc:
        out["traffic_over_algorithmic"] = round(traffic / algorithmic_bytes, 3)
        out["traffic_source"] = pmc["source"]
    if not fp32_mfma:
        out["pipe_frac"] = round(6.0 * achieved / BF16_MFMA_PEAK_TFLOPS, 4)
        ceiling = min(BF16_MFMA_PEAK_TFLOPS / 6.0, HBM_PEAK_GBS * (flops / algorithmic_bytes) / 1e3)
        out["bound_tflops"] = round(ceiling, 1)
        out["bound_frac"] = round(achieved / ceiling, 4)
    return out


def measured_traffic():
    """(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch of the stand-alone weight-stationary launch, from the
    rocprofv3 counter passes committed under profiles/ (one pass per counter; gfx950 tallies a 128-byte read
    request at 64 bytes, hence the factor two on the read side).  The file's header names the library build the
    passes ran (``# library <id>``); the figure is reported with that id and marked when it is not the build in
    use.  None when the file is missing."""
    from eeggan_amd import _lib
    path = os.path.join(ROOT, PMC_CONV_WS)
    if not os.path.exists(path):
        return None
    sums, built = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}, None
    for line in open(path):
        if line.startswith("# library"):
            built = line.split()[2]
        if line.startswith("#") or "conv_ws_kernel" not in line:
            continue
        cells = line.rstrip().rsplit(",", 3)  # ..., counter, value_kib, duration_ns
        if cells[1] in sums:
            sums[cells[1]][0] += float(cells[2])
            sums[cells[1]][1] += 1
    if not (sums["FETCH_SIZE"][1] and sums["WRITE_SIZE"][1]):
        return None
    fetch, write = (sums[k][0] / sums[k][1] for k in ("FETCH_SIZE", "WRITE_SIZE"))
    same = built == _lib.library_id()
    return {"bytes": (2.0 * fetch + write) * 1024.0,
            "source": f"{PMC_CONV_WS} (library {built}{'' if same else ', NOT the build in use: ' + _lib.library_id()})"}


def conv_ws_family_flops(trainer_factory):
    """Algorithmic flops per training step of all launches that take the weight-stationary kernel (k = 3, at
    most 64 channels on both sides, >= 256 subtiles: the library's own dispatch rule), counted over one eager
    step with the same models, and the list of those launches (shapes and which optional operands they carry)."""
    from eeggan_amd import ops
    total = [0.0, 0, []]
    raw = ops.conv1d_raw

    def counted(x, w, *a, **k):
        B, Cin, T = x.shape
        K = w.shape[2]
        Cout = w.shape[1] if k.get("transposed") else w.shape[0]
        if K == 3 and Cin <= 64 and 32 < Cout <= 64 and T % 4 == 0 and B * ((T + 31) // 32) >= 256:
            total[0] += 2.0 * B * Cin * Cout * K * T
            total[1] += 1
            bias, drop = (list(a) + [None, None])[:2] if a else (k.get("bias"), k.get("drop"))
            total[2].append(("conv", (B, Cin, Cout, T), bool(k.get("transposed")), int(k.get("act", 0)), bias is not None,
                             drop is not None, k.get("mask_y") is not None, k.get("mask_drop") is not None))
        return raw(x, w, *a, **k)

    fused = ops.conv1d_pixelnorm_raw

    def counted_fused(x, w, *a, **k):  # the generator's conv + norm layers take the same kernel when covered
        out = fused(x, w, *a, **k)
        if out is not None:
            total[0] += 2.0 * x.shape[0] * x.shape[1] * w.shape[0] * w.shape[2] * x.shape[2]
            total[1] += 1
            total[2].append(("conv_pn", (x.shape[0], x.shape[1], w.shape[0], x.shape[2])))
        return out

    ops.conv1d_raw, ops.conv1d_pixelnorm_raw = counted, counted_fused
    try:
        tr = trainer_factory()
        tr.train()
        torch.cuda.synchronize()
    finally:
        ops.conv1d_raw, ops.conv1d_pixelnorm_raw = raw, fused
    return total


def family_chain(dev, launches):
    """The weight-stationary family's launches of one training step (as recorded by ``conv_ws_family_flops``: same
    shapes, same epilogues and masks, same order) captured as ONE dependent chain in a HIP graph and replayed:
    milliseconds per replay by HIP events on the replay stream.  Unlike the stand-alone figure this one pays what
    every launch of the step pays -- the launch boundary inside a graph, a prologue and a partly filled round per
    layer, all of the step's layer sizes -- but nothing of the other kernels of the step or of the second chain that
    shares the chip with it; unlike a profiler's per-kernel durations it is measured on the unprofiled launch path."""
    from eeggan_amd import ops
    cache, flips = {}, {}

    def flipped(w):  # the flipped view an input gradient reads: kept per weight, as the layers keep theirs
        if id(w) not in flips:
            flips[id(w)] = ops.FlipCache(w)
        return flips[id(w)]

    def tensor(*shape):
        if shape not in cache:
            cache[shape] = torch.randn(*shape, device=dev) if len(shape) != 2 else torch.rand(*shape, device=dev) + 0.5
        return cache[shape]

    def run():
        for rec in launches:
            if rec[0] == "conv_pn":
                B, Cin, Cout, T = rec[1]
                ops.conv1d_pixelnorm_raw(tensor(B, Cin, T), tensor(Cout, Cin, 3), tensor(Cout), tensor(B, Cout),
                                         in_scale=0.1, slope=0.2)
                continue
            _, (B, Cin, Cout, T), transposed, act, has_bias, has_drop, has_mask, has_mask_drop = rec
            w = tensor(Cin, Cout, 3) if transposed else tensor(Cout, Cin, 3)
            ops.conv1d_raw(tensor(B, Cin, T), w, tensor(Cout) if has_bias else None,
                           tensor(B, Cout) if has_drop else None, in_scale=0.1, transposed=transposed, act=act,
                           act_param=0.2, act_aux=tensor(B, Cout, T) if act == 3 else None,
                           mask_y=tensor(B, Cin, T) if has_mask else None,
                           mask_drop=tensor(B, Cin) if has_mask_drop else None, mask_slope=0.2,
                           flip=flipped(w) if transposed else None)

    run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def device_info(dev):
    """What the run was measured on (boxes of one pool differ by several per cent at equal CU counts)."""
    p = torch.cuda.get_device_properties(dev)
    info = {"name": p.name, "cus": p.multi_processor_count}
    for key, attr in (("clock_mhz", "clock_rate"), ("memory_clock_mhz", "memory_clock_rate")):
        v = getattr(p, attr, None)
        if v:
            info[key] = round(v / 1000)
    return info


def in_step_figures(flops, launches, chain_ms=None):
    """The weight-stationary family inside the training step: flops counted live; kernel time per step from the
    committed rocprofv3 kernel-trace summary of this command -- a figure of the session and box that wrote that
    file, not of this run, which is why it carries the file's library id and is left out (``stale``) when the
    summary belongs to another build of the library than the one in use."""
    from eeggan_amd import _lib
    out = {"family": "conv_ws_kernel<*>", "flop_per_step": flops, "launches_per_step": launches}
    if chain_ms:
        # measured live: the family's launches of a step replayed as one dependent chain of a HIP graph
        out["chain"] = {"ms_per_step": round(chain_ms, 3), "achieved": round(flops / (chain_ms * 1e-3) / 1e12, 2),
                        "frac": round(flops / (chain_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "how": "the step's launches of the family, in order, as one captured chain; HIP events"}
    path = os.path.join(ROOT, PER_KERNEL)
    if not os.path.exists(path):
        return out
    ms, built = 0.0, None
    for line in open(path):
        if line.startswith("# library"):
            built = line.split()[2]
        if "conv_ws_kernel" in line and "ms/step=" in line:
            ms += float(line.split("ms/step=")[1].split()[0])
    out["source"] = f"{PER_KERNEL} (library {built})"
    if built != _lib.library_id():
        out["stale"] = f"the summary belongs to another build than the one in use ({_lib.library_id()})"
    elif ms > 0:
        out.update(kernel_ms_per_step=round(ms, 3), achieved=round(flops / (ms * 1e-3) / 1e12, 2),
                   frac=round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4))
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_steps(depth, alpha, batch, seed, budget_s, max_steps):
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
    while True:  # whole steps until the budget of CPU work is on the clock
        tr.step(reals, zs, mixes, src, src)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= max_steps:
            break
    return steps, dt


def cpu_baseline(depth, alpha, batch, seed):
    """The CPU restatement (oracle, kind 'port') timed on the host cores: whole steps for about ten
    seconds after one warm-up discriminator iteration at the benchmarked configuration, and a few seconds
    at the first progression stage (depth 0, the reference's own CPU-runnable case)."""
    steps, dt = _cpu_steps(depth, alpha, batch, seed, 10.0, 8)
    out = {"value": round(steps / dt, 5), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": cpu_model(), "host_cpus_visible": len(os.sched_getaffinity(0)),
           "sample": f"{steps} full steps (5 D + 1 G iterations with Adam each) after 1 warm-up D iteration, "
                     f"depth {depth}, B={batch}, fp32, torch CPU ops on {torch.get_num_threads()} threads ({dt:.2f} s)"}
    steps0, dt0 = _cpu_steps(0, 1.0, 32, seed, 4.0, 12)
    out["first_stage"] = {"value": round(steps0 / dt0, 5), "unit": "steps/s",
                          "sample": f"{steps0} full steps at depth 0 (T=32), B=32, alpha=1, fp32 ({dt0:.2f} s)"}
    return out


def main():
    args = parse()
    if os.environ.get("EEGGAN_HANG_DUMP"):  # debugging aid: stack dump of every thread after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["EEGGAN_HANG_DUMP"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # EEGGAN_FORCE_DIST=1 takes the multi-rank path (process group, gradient buckets, collectives between graph
    # replays) with however many ranks there are, one included: a rehearsal of that path on a one-GPU box
    force_dist = os.environ.get("EEGGAN_FORCE_DIST") is not None
    if world > 1 or force_dist:
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
    if args.storage == "bf16":
        args.dtype = "bf16"
    if args.dtype == "bf16":
        from eeggan_amd import ops
        ops.set_precision("bf16")
        if args.storage == "bf16":
            ops.set_storage("bf16")
    tr = make_trainer(dev, args.depth, args.alpha, args.batch, 1373, world, rank, args.eager, args.host_inputs,
                      force_dist=force_dist)
    if args.per_iteration_fakes:
        tr.batch_fakes = False

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # (rank 0 counts the flops of the weight-stationary family over one eager step BEFORE the timed region, so
    # that a kernel trace of this command ends with the timed steps and the stand-alone roofline launches)
    fam = None
    if rank == 0:
        fam = conv_ws_family_flops(lambda: make_trainer(dev, args.depth, args.alpha, args.batch, 1373, 1, 0, eager=True))
        torch.cuda.empty_cache()
    # not timed, not counted: the first call captures the graphs.  A capture that fails ends the run with its
    # error (a non-zero exit): the measured launch path is the captured one, or --eager when asked for.
    tr.train()
    for _ in range(args.warmup):
        tr.train()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train()
    barrier()
    dt = time.perf_counter() - t0
    ranks_reporting = 1
    if dist is not None:
        on = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], device=on, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank reports its GPU (local rank): the count of distinct reports is the number of ranks that ran
        seen = torch.zeros(dist.get_world_size(), device=on, dtype=torch.float64)
        seen[rank] = 1.0
        dist.all_reduce(seen)
        ranks_reporting = int(seen.sum().item())
    if rank == 0:
        from eeggan_amd import config
        T = config.seq_len(args.depth)
        out = {
            "metric": "G+D WGAN-GP train steps/sec (whole node) at final stage",
            "value": round(world * args.steps / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" + (" (reals copied from pinned host memory each iteration)" if args.host_inputs else ""),
            "library": __import__("eeggan_amd")._lib.library_id(),
            "config": {"workload": f"final stage depth {args.depth} (T={T}, 5 ch), alpha={args.alpha:g}, 5 D + 1 G iterations "
                                   f"with Adam per step, batch {args.batch}/GPU, global batch {args.batch * world}",
                       "parallelism": f"dp{world}", "drop_noise": args.drop_mode, "activation_storage": args.storage,
                       "launch": type(tr).__name__, "device": device_info(dev)},
        }
        if dist is not None:
            # what the process group itself reports, so that a multi-GPU line can be checked against --gpus
            out["config"]["dist"] = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                                     "ranks_reporting": ranks_reporting}
        out["roofline"] = conv_roofline(dev, args.batch, T)
        # (the committed per-kernel profile belongs to the headline configuration only)
        headline = args.batch == 32 and args.depth == 7 and args.alpha == 1.0 and args.dtype == "f32" and args.storage == "f32"
        if headline:
            chain_ms = family_chain(dev, fam[2]) if fam[2] else None
            out["roofline"]["in_step"] = in_step_figures(fam[0], fam[1], chain_ms)
        peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS
        step_flop = config.step_flops(args.depth, args.batch)
        out["roofline"]["whole_step"] = {
            "flop_per_step": step_flop, "achieved": round(step_flop / (dt / args.steps) / 1e12, 2),
            "frac": round(step_flop / (dt / args.steps) / 1e12 / peak, 4), "peak": peak}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.depth, args.alpha, args.batch, 1373)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
