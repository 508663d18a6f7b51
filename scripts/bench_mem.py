import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops
dev = torch.device("cuda", 0)
for n in (32*64*1920, 32*64*1920*4, 32*64*1920*16):
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev)
    for _ in range(20): ops.axpby_raw(a, 0.5, b, 0.5)
    for _ in range(20): ops.axpby_raw(a, 0.5)
    y = a.view(32, -1, 1920)
    r = torch.rand(32, y.shape[1], device=dev)
    for _ in range(20): ops.lrelu_drop_bwd_raw(y, y, r, 0.2)
    for _ in range(20): a.clone()
torch.cuda.synchronize()
