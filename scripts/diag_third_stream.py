"""What a generator pass costs when it runs on a third stream beside the critic iterations of a step (the pricing of
hoisting the generator iteration's forward pass off the critical path): five critic iterations alone, a 32-sample
generator pass alone, and the five iterations with that pass started beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeggan_amd import layers
from eeggan_amd.graphs import GraphedCall

dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
tr = bench.make_trainer(dev, 7, 1.0, 32, 1373, 1, 0)
for _ in range(3):
    tr.train()
torch.cuda.synchronize()
it = tr._d_parts
z = tr._z


def g_pass():
    with torch.no_grad():
        return tr.generator({"z": z})[0]["x"]


gcall = GraphedCall(g_pass)
third = torch.cuda.Stream()


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


def five():
    for _ in range(5):
        it()


def five_with_pass():
    main = torch.cuda.current_stream()
    third.wait_stream(main)
    with torch.cuda.stream(third):
        gcall.graph.replay()
    for _ in range(5):
        it()
    main.wait_stream(third)


a, b, c = timed(five), timed(gcall.graph.replay), timed(five_with_pass)
print(f"five critic iterations                      {a:8.3f} ms")
print(f"generator pass, 32 samples, no grad, alone  {b:8.3f} ms")
print(f"five iterations with the pass beside them   {c:8.3f} ms   (hidden: {a + b - c:6.3f} of {b:6.3f} ms)")
