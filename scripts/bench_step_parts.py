"""Where the training step's time goes, graph by graph: the generator pass that makes the fakes, the two chains
of a critic iteration (each alone, then side by side), the closing graph, the generator iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeggan_amd import layers

dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = bench.make_trainer(dev, 7, 1.0, batch, 1373, 1, 0)
for _ in range(3):
    tr.train()
torch.cuda.synchronize()
it = tr._d_parts


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    main = torch.cuda.current_stream()
    it.sides[0].wait_stream(main)
    with torch.cuda.stream(it.sides[0]):
        it.graphs[1].replay()
    it.graphs[0].replay()
    main.wait_stream(it.sides[0])


rows = [("fakes (generator pass, 5 x batch)", timed(tr._fakes_call)),
        (f"critic chain '{it.parts[0]}' alone", timed(it.graphs[0].replay)),
        (f"critic chain '{it.parts[1]}' alone", timed(it.graphs[1].replay)),
        ("both chains side by side", timed(both)),
        ("closing graph (slab sums, loss, Adam, flipped views)", timed(it.graph_late.replay)),
        ("critic iteration", timed(it)),
        ("generator iteration", timed(lambda: tr._replay(tr._g_parts))),
        ("whole step", timed(tr.train, 10))]
for name, ms in rows:
    print(f"{name:55s} {ms:8.3f} ms", flush=True)
for name, g in (("chain0", it.graphs[0]), ("chain1", it.graphs[1])):
    pass
