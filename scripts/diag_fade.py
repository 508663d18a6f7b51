"""Fade-in: eager vs eager vs graph moments with lr = 0 (debugging aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from eeggan_amd import layers
from eeggan_amd.graphs import GraphedTrainer
from eeggan_amd.trainer import Trainer
import test_e2e_gpu as T
dev = torch.device("cuda", 0)
layers.set_drop_mode("device")
fade = (0.3, 0.6)
class DevAlpha(Trainer):
    def train(self):
        GraphedTrainer._sync_alpha(self)
        return super().train()
runs = {"devalpha": T._two_steps(dev, DevAlpha, lr=0.0, alphas=fade), "eager1": T._two_steps(dev, Trainer, lr=0.0, alphas=fade), "eager2": T._two_steps(dev, Trainer, lr=0.0, alphas=fade),
        "graph1": T._two_steps(dev, GraphedTrainer, lr=0.0, alphas=fade), "graph2": T._two_steps(dev, GraphedTrainer, lr=0.0, alphas=fade)}
def worst(a, b):
    w = (0, "")
    for k in a:
        if k.endswith((".m", ".v")):
            e = float((a[k] - b[k]).abs().max()) / max(float(a[k].abs().max()), 1e-30)
            w = max(w, (e, k))
    return w
for x in runs:
    print(x, [f"{worst(runs[x], runs[y])[0]:.1e}" for y in runs], worst(runs[x], runs["eager1"])[1])
