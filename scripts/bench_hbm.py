"""Timing of the bandwidth-bound kernels (pixel norm, resampler) at two batch sizes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops, utils
from kernel_roofline import timeit, row
dev = torch.device("cuda", 0)
for B in (32, 512):
    n = 30 if B <= 64 else 8
    C, T = 64, 1920
    h = torch.randn(B, C, T, device=dev); g = torch.randn(B, C, T, device=dev); r = torch.rand(B, C, device=dev) + 0.5
    y, d = ops.pixelnorm_drop_fwd_raw(h, r)
    row(f"B={B} pixelnorm fwd", timeit(lambda: ops.pixelnorm_drop_fwd_raw(h, r), n), nbytes=2.0 * 4 * B * C * T)
    row(f"B={B} pixelnorm bwd", timeit(lambda: ops.pixelnorm_drop_bwd_raw(g, y, r, d, 0.2), n), nbytes=3.0 * 4 * B * C * T)
    with torch.no_grad():
        hc = h[:, :, :1280].contiguous()
        row(f"B={B} resample 1280->1920", timeit(lambda: utils.resample_signal(hc, 1280, 1920), n), nbytes=4.0 * B * C * 3200)
        row(f"B={B} resample 1920->1280", timeit(lambda: utils.resample_signal(h, 1920, 1280), n), nbytes=4.0 * B * C * 3200)
        h2 = h[:, :, :1024].contiguous()
        row(f"B={B} resample 1024->1280", timeit(lambda: utils.resample_signal(h2, 1024, 1280), n), nbytes=4.0 * B * C * 2304)
        row(f"B={B} resample 1280->1024", timeit(lambda: utils.resample_signal(hc, 1280, 1024), n), nbytes=4.0 * B * C * 2304)
