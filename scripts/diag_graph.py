"""Eager vs graph-replay weight trajectories (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from eeggan_amd import layers
from eeggan_amd.graphs import GraphedTrainer
from eeggan_amd.trainer import Trainer
import test_e2e_gpu as T
dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
runs = [("eager1", Trainer), ("eager2", Trainer), ("graph1", GraphedTrainer), ("graph2", GraphedTrainer), ("eager3", Trainer)]
res = {n: T._two_steps(dev, c) for n, c in runs}
def diff(a, b):
    return max(float((a[i][k] - b[i][k]).abs().max()) for i in (0, 1) for k in a[i])
for a in res:
    print(a, " ".join(f"{diff(res[a], res[b]):.2e}" for b in res))
