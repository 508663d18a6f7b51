"""Are dependencies between parallel branches of a captured HIP graph honoured on back-to-back replays?
A (main): x += 1 on a large tensor; fork; B (side): y = x (small prefix); join; C (main): err += |y - x[:n]|.
If B ever runs before A of the same replay, err becomes non-zero.  Also: an eager write right before each
replay that the side branch reads (ordering of a branch against earlier work of the launch stream)."""
import torch
dev = torch.device("cuda", 0)
N, n = 1 << 24, 1 << 12
x = torch.zeros(N, device=dev)
y = torch.zeros(n, device=dev)
err = torch.zeros((), device=dev)
inp = torch.zeros(n, device=dev)      # written eagerly before every replay
seen = torch.zeros(n, device=dev)     # side branch: copy of inp
err2 = torch.zeros((), device=dev)
want = torch.zeros(n, device=dev)     # what inp should be during this replay (written eagerly as well, read on main)
side = torch.cuda.Stream()


w = torch.zeros(N, device=dev)
err3 = torch.zeros((), device=dev)
MAIN_BETWEEN = 3
seen0 = torch.zeros(n, device=dev)
err0 = torch.zeros((), device=dev)
ROOT_READS = True


def body():
    main = torch.cuda.current_stream()
    if ROOT_READS:
        torch.add(inp, 0, out=seen0)   # the root node of the graph reads the eager input
    x.add_(1.0)                        # A
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.add(x[:n], 0, out=y)     # B reads A's result
        torch.add(inp, 0, out=seen)    # B' reads the eager input
    for _ in range(MAIN_BETWEEN):      # main-stream work between the fork and the join: w = x + 0.5, in place chain
        torch.add(x, 0.5, out=w)
    main.wait_stream(side)
    err3.add_((w[:n] - x[:n] - 0.5).abs().sum())   # the join must also wait for the main-stream work before it
    err.add_((y - x[:n]).abs().sum())  # C
    err2.add_((seen - want).abs().sum())
    err0.add_((seen0 - want).abs().sum())


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body()
torch.cuda.synchronize()
x.zero_(); err.zero_(); err2.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
x.zero_(); err.zero_(); err2.zero_(); err0.zero_(); err3.zero_(); w.zero_()
vals = [torch.full((n,), float(i), device=dev) for i in range(64)]
big = torch.zeros(1 << 26, device=dev)
for i in range(64):
    big.add_(1.0)                      # keeps the launch stream busy so that the copies below start late
    inp.copy_(vals[i]); want.copy_(vals[i])
    g.replay()
torch.cuda.synchronize()
print(f"x = {float(x[0])} (64 expected); branch-vs-main error {float(err)}; eager-input error seen by the side "
      f"branch {float(err2)}, by the root node on the origin stream {float(err0)}; main-stream work between fork and "
      f"join seen late by the join: {float(err3)}")
