"""Timing of the bandwidth-bound kernels (pixel norm, LeakyReLU backward, blend, resampler) at [B, 64, 1920].

Every launch of a timed loop takes the next of NBUF input sets, so that no launch finds its input in a cache because
the launch before it read the same bytes (NBUF = 1 is the loop over one tensor of `kernel_roofline.py`).

usage: python scripts/bench_hbm.py [B ...] [--nbuf N] [--bf16]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops, utils  # noqa: E402
from kernel_roofline import rotating, timeit, row  # noqa: E402

dev = torch.device("cuda", 0)


def run(B, nbuf, bf16=False):
    n = 32 if B <= 64 else 24
    C, T = 64, 1920
    r = torch.rand(B, C, device=dev) + 0.5
    dt = torch.bfloat16 if bf16 else torch.float32
    hs = [torch.randn(B, C, T, device=dev).to(dt) for _ in range(nbuf)]
    gs = [torch.randn(B, C, T, device=dev).to(dt) for _ in range(nbuf)]
    tag = f"B={B} x{nbuf}" + (" bf16" if bf16 else "")
    esz = 2.0 if bf16 else 4.0
    y, d = ops.pixelnorm_drop_fwd_raw(hs[0], r)
    row(f"{tag} pixelnorm fwd", timeit(rotating(lambda h: ops.pixelnorm_drop_fwd_raw(h, r), [(h,) for h in hs]), n),
        nbytes=2.0 * esz * B * C * T)
    row(f"{tag} pixelnorm bwd", timeit(rotating(lambda g, h: ops.pixelnorm_drop_bwd_raw(g, h, r, d, 0.2), list(zip(gs, hs))), n),
        nbytes=3.0 * esz * B * C * T)
    row(f"{tag} lrelu+gdrop bwd", timeit(rotating(lambda g, h: ops.lrelu_drop_bwd_raw(g, h, r, 0.2), list(zip(gs, hs))), n),
        nbytes=3.0 * esz * B * C * T)
    row(f"{tag} axpby (blend)", timeit(rotating(lambda g, h: ops.axpby_raw(h, 0.3, g, 0.7), list(zip(gs, hs))), n),
        nbytes=3.0 * esz * B * C * T)
    del gs
    with torch.no_grad():
        for tin, tout in ((1280, 1920), (1920, 1280), (1024, 1280), (1280, 1024)):
            xs = [(h[:, :, :tin].contiguous(),) for h in hs]
            row(f"{tag} resample {tin}->{tout}", timeit(rotating(lambda x: utils.resample_signal(x, tin, tout), xs), n),
                nbytes=esz * B * C * (tin + tout))
            del xs


if __name__ == "__main__":
    args = sys.argv[1:]
    nbuf = 4
    if "--nbuf" in args:
        i = args.index("--nbuf")
        nbuf = int(args[i + 1])
        del args[i:i + 2]
    bf16 = "--bf16" in args  # bfloat16 tensors (the bf16 storage mode's forms of the same kernels)
    if bf16:
        args.remove("--bf16")
        ops.set_precision("bf16")
        ops.set_storage("bf16")
    for B in ([int(a) for a in args] or [32, 512]):
        run(B, nbuf, bf16)
