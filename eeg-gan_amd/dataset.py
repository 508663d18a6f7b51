"""The real-data side of the step on the device: what ``EEGDataset.__getitem__`` does to a window
after it is cut from a recording (reference dataset.py:161-187) -- bring it down to the current
stage's sampling rate, and during a fade-in blend it with its own low-passed copy -- applied to a
whole batch of full-rate windows with the resampling and blend kernels of this package, instead of
sample by sample in host data-loader workers.

Reading, channel picking and normalising the recordings (dataset.py:53-118) stays on the host: it
happens once, before training."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .utils import resample_signal


class DepthResampler:
    """Callable with the attribute protocol of the reference dataset (``model_depth``, ``alpha``,
    set by ``DepthManager``): ``resampler(windows)`` maps full-rate windows [B, C, T_max] to the
    reals of the current stage ``{'x': [B, C, T_depth]}``."""

    def __init__(self, progression_scale_up, progression_scale_down):
        if len(progression_scale_up) != len(progression_scale_down):
            raise ValueError("progression_scale_up and progression_scale_down must have equal lengths")
        self.progression_scale_up = list(progression_scale_up)
        self.progression_scale_down = list(progression_scale_down)
        self.max_dataset_depth = len(self.progression_scale_up)
        self.model_depth = self.max_dataset_depth
        self.alpha = 1.0

    def resample_data(self, data, index, forward=True, alpha_fade=False):
        """One stage transition: up by scale_up/scale_down (forward) or back down (dataset.py:153-159)."""
        k = index - (1 if alpha_fade else 0)
        up, down = self.progression_scale_up[k], self.progression_scale_down[k]
        return resample_signal(data, down, up) if forward else resample_signal(data, up, down)

    def create_datapoint_from_depth(self, datapoint, target_depth):
        """Full-rate window -> stage ``target_depth``, one transition at a time like the reference
        (dataset.py:169-173), so that each step carries the same float32 interpolation rounding."""
        if not 0 <= target_depth <= self.max_dataset_depth:
            raise ValueError(f"target depth {target_depth} outside [0, {self.max_dataset_depth}]")
        for index in range(self.max_dataset_depth - 1, target_depth - 1, -1):
            datapoint = self.resample_data(datapoint, index, False)
        return datapoint

    def alpha_fade(self, datapoint):
        """x + (up(down(x)) - x) * (1 - alpha): the reals lose the new stage's detail in step with the
        networks' fade-in (dataset.py:175-180)."""
        if self.alpha == 1:
            return datapoint
        if self.model_depth < 1:
            raise ValueError("alpha_fade: the first stage has nothing to fade in from")
        low = self.resample_data(datapoint, self.model_depth, False, alpha_fade=True)
        low = self.resample_data(low, self.model_depth, True, alpha_fade=True)
        # = alpha * x + (1 - alpha) * low
        return ops.blend(datapoint, low, float(self.alpha))

    @torch.no_grad()
    def __call__(self, windows):
        if windows.dim() != 3:
            raise ValueError("DepthResampler: expected full-rate windows [B, C, T]")
        x = windows
        if self.model_depth != self.max_dataset_depth:
            x = self.create_datapoint_from_depth(x, self.model_depth)
        return {"x": self.alpha_fade(x)}


def synthetic_eeg(batch: int, channels: int, length: int, rng: np.random.Generator, full_rate_hz: float = 60.0,
                  full_length: int = 1920) -> np.ndarray:
    """Stand-in recordings for benchmarks and smoke runs when no dataset is at hand: [batch, channels, length]
    float32 windows, each channel six sinusoids of 1-30 Hz with 1/f amplitudes and random phases plus white
    noise (sigma 0.1), sampled at the rate of the stage whose window is ``length`` long
    (``full_rate_hz * length / full_length``), then min-max scaled to [-1, 1] per window like the
    reference's recordings (dataset.py:139-144)."""
    rate = full_rate_hz * length / full_length
    time = np.arange(length)[None, None, None, :] / rate
    freq = rng.uniform(1.0, 30.0, size=(batch, channels, 6, 1))
    phase = rng.uniform(0, 2 * np.pi, size=(batch, channels, 6, 1))
    x = (np.sin(2 * np.pi * freq * time + phase) / freq).sum(axis=2)
    x = x + rng.normal(0, 0.1, size=(batch, channels, length))
    lo, hi = x.min(axis=(1, 2), keepdims=True), x.max(axis=(1, 2), keepdims=True)
    return ((x - lo) / (hi - lo) * 2.0 - 1.0).astype(np.float32)
