"""The real-data side of the step on the device: what ``EEGDataset.__getitem__`` does to a window
after it is cut from a recording (reference dataset.py:161-187) -- bring it down to the current
stage's sampling rate, and during a fade-in blend it with its own low-passed copy -- applied to a
whole batch of full-rate windows with the resampling and blend kernels of this package, instead of
sample by sample in host data-loader workers.

Reading, channel picking and normalising the recordings (dataset.py:53-118) stays on the host: it
happens once, before training."""
from __future__ import annotations

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
