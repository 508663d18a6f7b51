"""The 1x1 layers with a 5-channel side at the batch sizes of the training step (32: critic passes, 64: the stacked
pass, 160: the generator pass for five critic iterations): time per launch of whichever kernel the library's rule picks
(EEGGAN_THIN_MIN_WG_PER_CU moves the rule)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeggan_amd import ops
dev = torch.device("cuda", 0)
for B in (32, 64, 96, 128, 160, 256):
    for cin, cout, act in ((64, 5, 2), (5, 64, 1)):
        x = torch.randn(B, cin, 1920, device=dev); w = torch.randn(cout, cin, 1, device=dev); b = torch.randn(cout, device=dev)
        f = lambda: ops.conv1d_raw(x, w, b, None, in_scale=0.1, act=act, act_param=0.5 if act == 2 else 0.2)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); e1.synchronize()
        print(f"B={B:4d} {cin:2d}->{cout:2d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
