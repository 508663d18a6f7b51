import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import ops
dev = torch.device("cuda", 0)
B, cin, cout, T, K = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin, K, device=dev)
b = torch.randn(cout, device=dev); r = torch.rand(B, cout, device=dev) + 0.5
for _ in range(40): ops.conv1d_raw(x, w, b, r, in_scale=0.1, act=1, act_param=0.2)
torch.cuda.synchronize()
