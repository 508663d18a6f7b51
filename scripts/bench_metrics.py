#!/usr/bin/env python3
"""Evaluation-metrics workload: N generated [5, 1920] signals -> feature-network latents -> Frechet distance
against the same number of synthetic "real" signals, plus the sliced Wasserstein distance of the two sets.

    python scripts/bench_metrics.py [--samples 100000] [--batch 2048] [--cpu-sample 64]

Stages (each bracketed by events on the launch stream, inputs resident in HBM):
  generate   Generator.forward at the final depth under no_grad (reference network.py:354-396)
  features   cpc Network.forward(x, no_loss=True) (reference cpc_network.py:316-322) -> z, c, z_p, c_p
  stats      mean + covariance of the four feature matrices (reference plugins.py:503-538)
  frechet    Newton-Schulz square root (50 iterations) + distance, four feature kinds (plugins.py:540-572)
  swd        SlicedWDistance.get_descriptors + sliced_wasserstein, 4 x 128 directions (plugins.py:383-451)
Prints one JSON line.  The CPU leg times the numpy/torch restatement of the feature network on a bounded
sample (it is the checker, not a shipped path)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timed(fn, stream):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    out = fn()
    b.record(stream)
    b.synchronize()
    return out, a.elapsed_time(b) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--swd-items", type=int, default=8192, help="items per set for the sliced Wasserstein distance")
    args = ap.parse_args()

    from eeggan_amd import config, cpc, layers
    layers.set_drop_mode("device")  # generator noise multipliers drawn on the GPU, as in the timed training step
    from eeggan_amd.dataset import synthetic_eeg
    from eeggan_amd.plugins import FidCalculator, SlicedWDistance

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream()
    g, _ = config.build_models(config.RANDOM_SEED)
    g = g.to(dev).eval()
    g.depth, g.alpha = len(config.PROGRESSION_SCALE_UP), 1.0
    T = config.seq_len(g.depth)
    torch.manual_seed(21)
    net = cpc.Network(config.NUM_CHANNELS).to(dev).eval()
    N, B = args.samples, args.batch
    gen = torch.Generator(device=dev).manual_seed(1373)
    rng = np.random.default_rng(1373)
    real_pool = torch.from_numpy(synthetic_eeg(2048, config.NUM_CHANNELS, T, rng)).to(dev)

    def generate():
        out = torch.empty(N, config.NUM_CHANNELS, T, device=dev)
        with torch.no_grad():
            for i in range(0, N, B):
                n = min(B, N - i)
                z = torch.randn(n, config.LATENT_SIZE, device=dev, generator=gen)
                out[i:i + n] = g({"z": z})[0]["x"]
        return out

    def features(x):
        z = torch.empty(N, 128, T // 60, device=dev)
        c = torch.empty_like(z)
        zp = torch.empty(N, 128, device=dev)
        cp = torch.empty_like(zp)
        with torch.no_grad():
            for i in range(0, N, B):
                lat = net(x[i:i + B], no_loss=True).latents
                z[i:i + B], c[i:i + B], zp[i:i + B], cp[i:i + B] = lat.z, lat.c, lat.z_p, lat.c_p
        return {"z": z.view(-1, 128), "c": c.view(-1, 128), "zp": zp, "cp": cp}

    def stats(f):
        return {k: FidCalculator.mean_and_cov(v) for k, v in f.items()}

    def frechet(sa, sb):
        return {k: FidCalculator.calc_fid(*sa[k], *sb[k]) for k in sa}

    for _ in range(2):  # warm-up at a small size: first launches, allocator
        n_keep, N = N, min(N, 2 * B)
        f = features(generate())
        frechet(stats(f), stats(f))
        N = n_keep
    torch.cuda.synchronize()

    fake, t_gen = timed(generate, stream)
    real = real_pool[torch.randint(0, real_pool.size(0), (N,), device=dev, generator=gen)]
    f_fake, t_feat = timed(lambda: features(fake), stream)
    f_real = features(real)
    s_fake, t_stats = timed(lambda: stats(f_fake), stream)
    s_real = stats(f_real)
    fid, t_fid = timed(lambda: {k: float(v) for k, v in frechet(s_fake, s_real).items()}, stream)

    swd = SlicedWDistance(progression_scale=2, output_snapshot_ticks=1)
    n_swd = min(args.swd_items, N)

    def run_swd():
        return swd.get_descriptors(fake[:n_swd], real[:n_swd])  # one distance per level
    run_swd()
    torch.cuda.synchronize()
    swd_vals, t_swd = timed(run_swd, stream)
    swd_vals = [float(v) for v in swd_vals]

    # feature-network algorithmic figures per sample: conv MACs x 2 + GRU + dense layers
    t1, t2, t3 = T // 5, T // 20, T // 60
    flops = 2 * (5 * 9 * 32 * t1 + 32 * 7 * 64 * t2 + 64 * 5 * 128 * t3) + 2 * t3 * 2 * 128 * 384 \
        + 2 * 2 * (256 * 128 + 128 * 128)
    bytes_in = config.NUM_CHANNELS * T * 4
    bytes_out = (2 * 128 * t3 + 2 * 128) * 4

    cpu = None
    if args.cpu_sample:
        from oracle import metrics_cpu as M
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        xs = fake[:args.cpu_sample].cpu()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        M.cpc_forward(sd, xs[:4])
        t0 = time.perf_counter()
        M.cpc_forward(sd, xs)
        cpu = {"value": args.cpu_sample / (time.perf_counter() - t0), "unit": "samples/s (features only)",
               "cores": torch.get_num_threads(), "kind": "port", "sample": f"{args.cpu_sample} signals"}

    total = t_gen + t_feat + t_stats + t_fid
    print(json.dumps({
        "workload": f"{N} generated [5,{T}] signals: generate + feature network + statistics + Frechet distance",
        "samples_per_s": N / total, "seconds": {"generate": t_gen, "features": t_feat, "stats": t_stats,
                                                "frechet_4_kinds": t_fid, "swd": t_swd},
        "generate_samples_per_s": N / t_gen, "features_samples_per_s": N / t_feat,
        "features": {"gflop_per_s": flops * N / t_feat * 1e-9, "algorithmic_gb_per_s": (bytes_in + bytes_out) * N / t_feat * 1e-9,
                     "hbm_frac": (bytes_in + bytes_out) * N / t_feat / 8e12},
        "swd": {"items": n_swd, "values": swd_vals, "items_per_s": n_swd / t_swd},
        "fid": fid, "batch": B, "dtype": "f32", "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
