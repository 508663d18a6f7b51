import sys, os, ctypes as C
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eeggan_amd import ops
old = C.CDLL(os.path.join(R, "scripts", "libold.so"))
f, i, fl, st = C.c_void_p, C.c_int, C.c_float, C.c_void_p
old.eeggan_conv1d.argtypes = [f, f, f, f, f, i, i, i, i, i, fl, i, i, fl, f, f, f, fl, st]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (B, cin, cout, T, K) in [(4, 64, 64, 1920, 3), (4, 64, 64, 512, 3), (4, 128, 128, 256, 3), (4, 256, 256, 128, 3), (4, 5, 64, 1920, 1), (32, 64, 64, 1920, 3)]:
    x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin, K, device=dev); b = torch.randn(cout, device=dev)
    r = torch.rand(B, cout, device=dev) + 0.5
    ymask = torch.randn(B, cin, T, device=dev); md = torch.rand(B, cin, device=dev) + 0.5
    for mask in (False, True):
        y_new = ops.conv1d_raw(x, w, b, r, in_scale=0.37, act=1, act_param=0.2, mask_y=ymask if mask else None, mask_drop=md if mask else None, mask_slope=0.2)
        y_old = torch.empty_like(y_new)
        rc = old.eeggan_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), y_old.data_ptr(), B, cin, cout, T, K, 0.37, 0, 1, 0.2, None,
                               ymask.data_ptr() if mask else None, md.data_ptr() if mask else None, 0.2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        d = (y_new - y_old).abs()
        print((B, cin, cout, T, K), "mask" if mask else "nomask", "rc", rc, "max abs diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel())
