"""Algorithmic bytes of the weight-gradient jobs and of the full-length (k = 32) layer's kernels over one eager
training step at the benchmarked configuration: every input tensor read once, every output written once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeggan_amd import layers, ops

dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
tr = bench.make_trainer(dev, 7, 1.0, 32, 1373, 1, 0, eager=True)
tot = {"wgrad_full": [0, 0], "wgrad_other": [0, 0]}
orig = ops._direct.wgrad


def counted(w, bias, x, gy, in_scale, mask_y, mask_drop, slope):
    B, Cin, T = x.shape
    Cout, K = gy.shape[1], w.shape[2]
    key = "wgrad_full" if (K == 3 and Cin % 64 == 0 and Cout % 64 == 0 and T % 32 == 0) else "wgrad_other"
    nbytes = 4 * (x.numel() + gy.numel() + (mask_y.numel() if mask_y is not None else 0) + w.numel())
    tot[key][0] += nbytes
    tot[key][1] += 1
    return orig(w, bias, x, gy, in_scale, mask_y, mask_drop, slope)


ops._direct.wgrad = counted
tr.train()
torch.cuda.synchronize()
for k, (b, n) in tot.items():
    print(f"{k}: {n} jobs per step, {b / 2**20:.1f} MiB read+written per step (inputs once, weight-shaped output once)")
