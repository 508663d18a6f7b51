"""The plugin that steers the hot path: ``DepthManager`` (reference plugins.py:28-119) maps the
number of images shown so far to (depth, alpha) of the progressive schedule and, on a depth
change, installs the stage's minibatch size, learning rate (256 * lr / minibatch), optimizers,
data loader and latent generator.  ``Plugin`` is the small base class of the reference's plugin
protocol (torch_utils.py:4-12).  Monitoring / checkpoint / metric plugins are outside this package.
"""
from __future__ import annotations

import bisect


class Plugin(object):
    def __init__(self, interval=None):
        if interval is None:
            interval = []
        self.trigger_interval = interval

    def register(self, trainer):
        raise NotImplementedError


class DepthManager(Plugin):
    """Progressive-growing schedule (reference plugins.py:28-119).  The schedule is a sorted list of image counts
    (``alpha_map``: end of a stable phase, end of the following fade-in, ...), looked up by bisection; a stage's
    settings (minibatch, learning rate, tick length) come from ``stage_settings``; ``iteration`` applies what changed
    to the trainer it was registered with."""
    minibatch_override = {0: 256, 1: 256, 2: 128, 3: 128, 4: 48, 5: 32, 6: 32, 7: 32, 8: 16, 9: 16, 10: 8, 11: 8}
    tick_kimg_override = {4: 4, 5: 4, 6: 4, 7: 3, 8: 3, 9: 2, 10: 2, 11: 1}
    training_kimg_override = {1: 200, 2: 200, 3: 200, 4: 200}
    transition_kimg_override = {1: 200, 2: 200, 3: 200, 4: 200}

    def __init__(self, create_dataloader_fun, create_rlg, max_depth, tick_kimg_default, get_optimizer, default_lr,
                 reset_optimizer: bool = True, disable_progression=False, minibatch_default=256, depth_offset=0,
                 lod_training_kimg=400, lod_transition_kimg=400):
        super().__init__([(1, "iteration")])
        # what a stage change builds
        self.create_dataloader_fun, self.create_rlg, self.get_optimizer = create_dataloader_fun, create_rlg, get_optimizer
        # the schedule
        self.depth_offset, self.max_depth, self.disable_progression = depth_offset, max_depth, disable_progression
        self.alpha_map = self.pre_compute_alpha_map(depth_offset, max_depth, lod_training_kimg,
                                                    self.training_kimg_override, lod_transition_kimg,
                                                    self.transition_kimg_override)
        # per-stage settings
        self.minibatch_default, self.tick_kimg_default, self.default_lr = minibatch_default, tick_kimg_default, default_lr
        self.reset_optimizer = reset_optimizer
        # state: nothing applied yet, so the first ``iteration`` installs both depth and alpha
        self.trainer, self.depth, self.alpha = None, -1, -1

    def register(self, trainer):
        resuming = trainer.optimizer_d is not None  # optimizers handed in: keep them for the first stage
        trainer.stats.update(minibatch_size=self.minibatch_default,
                             alpha=dict(log_name="alpha", log_epoch_fields=["{val:.2f}"], val=self.alpha))
        self.trainer = trainer
        self.iteration(is_resuming=resuming)

    @staticmethod
    def pre_compute_alpha_map(start_depth, max_depth, lod_training_kimg, lod_training_kimg_overrides,
                              lod_transition_kimg, lod_transition_kimg_overrides):
        """Image counts at which a stage stops being stable (even entries) and at which the
        following fade-in completes (odd entries)."""
        boundaries, shown = [], 0
        for stage in range(start_depth + 1, max_depth + 1):
            shown += int(lod_training_kimg_overrides.get(stage, lod_training_kimg) * 1000)
            boundaries.append(shown)
            shown += int(lod_transition_kimg_overrides.get(stage, lod_transition_kimg) * 1000)
            boundaries.append(shown)
        return boundaries

    def calc_progress(self, cur_nimg=None):
        """(depth, alpha) after ``cur_nimg`` images.  ``alpha_map[2s]`` ends stage s's stable phase and
        ``alpha_map[2s + 1]`` the fade-in of stage s + 1; an image count exactly on a boundary still belongs to
        the phase that ends there, with alpha 1."""
        if self.disable_progression:
            return self.max_depth, 1.0
        if cur_nimg is None:
            cur_nimg = self.trainer.cur_nimg
        phase = bisect.bisect_left(self.alpha_map, cur_nimg)  # first boundary not yet passed
        depth, alpha = self.depth_offset + (phase + 1) // 2, 1.0
        if phase % 2 == 1 and phase < len(self.alpha_map) and self.alpha_map[phase] != cur_nimg:
            fade_start, fade_end = self.alpha_map[phase - 1], self.alpha_map[phase]
            alpha = (cur_nimg - fade_start) / (fade_end - fade_start)
        return min(self.max_depth, depth), alpha

    def stage_settings(self, depth):
        """(minibatch, learning rate, images per tick) of a stage: the learning rate scales with
        ``minibatch_default / minibatch`` (reference plugins.py:105-113)."""
        stage = depth - self.depth_offset
        minibatch = self.minibatch_override.get(stage, self.minibatch_default)
        return (minibatch, self.minibatch_default * self.default_lr / minibatch,
                int(self.tick_kimg_override.get(stage, self.tick_kimg_default) * 1000))

    def _enter_stage(self, depth, keep_optimizers):
        tr = self.trainer
        minibatch, lr, tick_nimg = self.stage_settings(depth)
        for holder, attr in ((tr.discriminator, "depth"), (tr.generator, "depth"), (tr.dataset, "model_depth"),
                             (self, "depth")):
            setattr(holder, attr, depth)
        if self.reset_optimizer and not keep_optimizers:
            tr.optimizer_g, tr.optimizer_d, tr.lr_scheduler_g, tr.lr_scheduler_d = self.get_optimizer(lr)
        self.data_loader = self.create_dataloader_fun(minibatch)
        tr.dataiter = iter(self.data_loader)
        tr.random_latents_generator = self.create_rlg(minibatch)
        tr.tick_duration_nimg = tick_nimg
        tr.stats["minibatch_size"] = minibatch

    def iteration(self, is_resuming=False, *args):
        depth, alpha = self.calc_progress()
        if self.depth != depth:
            self._enter_stage(depth, keep_optimizers=is_resuming)
        if self.alpha != alpha:
            tr = self.trainer
            tr.discriminator.alpha = tr.generator.alpha = tr.dataset.alpha = self.alpha = alpha
        self.trainer.stats["depth"] = depth
        self.trainer.stats["alpha"]["val"] = alpha


class OutputGenerator(Plugin):
    """The running average of the generator's weights and samples drawn from it (reference plugins.py:235-315):
    every epoch tick ``avg = old_weight * avg + (1 - old_weight) * weight`` for all parameters (one multi-tensor
    launch), and every ``output_snapshot_ticks`` ticks a batch of samples generated with the averaged weights
    swapped in.  The reference then plots those samples and writes PNG files and a checkpoint (matplotlib,
    imageio, plugins.py:271-315); this class stops at the samples (``self.last_samples``, a device tensor)
    and, when ``checkpoints_dir`` is given, the ``smooth_generator`` checkpoint."""

    def __init__(self, sample_fn, checkpoints_dir, seq_len: int, max_freq: float, samples_count: int = 8,
                 output_snapshot_ticks: int = 25, old_weight: float = 0.59):
        super().__init__([(1, "epoch")])
        self.old_weight = old_weight
        self.sample_fn = sample_fn
        self.samples_count = samples_count
        self.checkpoints_dir = checkpoints_dir
        self.seq_len = seq_len
        self.max_freq = max_freq
        self.my_g_clone = None
        self.output_snapshot_ticks = output_snapshot_ticks
        self.last_samples = None

    @staticmethod
    def flatten_params(model):
        return [p.detach().clone() for p in model.parameters()]

    @staticmethod
    def load_params(flattened, model):
        import torch
        from . import ops
        with torch.no_grad():
            for p, avg_p in zip(model.parameters(), flattened):
                p.copy_(avg_p)
        ops.bump_weights_epoch()  # derived weight views are stale

    def register(self, trainer):
        self.trainer = trainer
        self.my_g_clone = self.flatten_params(self.trainer.generator)

    def epoch(self, epoch_index):
        import os
        import torch
        from . import ops
        from .utils import cudize
        gen = self.trainer.generator
        ops.ema_update(self.my_g_clone, list(gen.parameters()), self.old_weight)
        if epoch_index % self.output_snapshot_ticks != 0:
            return
        z = cudize(next(self.sample_fn(self.samples_count)))
        original = self.flatten_params(gen)
        self.load_params(self.my_g_clone, gen)
        try:
            if self.checkpoints_dir:
                dest = os.path.join(self.checkpoints_dir, "network-snapshot-smooth_generator-{:06}.dat".format(
                    self.trainer.cur_nimg // 1000))
                torch.save({"cur_nimg": self.trainer.cur_nimg, "model": gen.state_dict()}, dest)
            with torch.no_grad():
                self.last_samples = gen(z)[0]["x"]
        finally:
            self.load_params(original, gen)


class EvalDiscriminator(Plugin):
    """The critic's mean score over the validation set, logged as ``memorization`` (reference plugins.py:205-232;
    train.py:116-117 registers it when the critic's layers have a norm).  ``create_dataloader_fun(minibatch, is_training,
    depth, alpha)`` yields batches as the reference's does."""

    def __init__(self, create_dataloader_fun, output_snapshot_ticks):
        super().__init__([(1, "epoch")])
        self.create_dataloader_fun = create_dataloader_fun
        self.output_snapshot_ticks = output_snapshot_ticks

    def register(self, trainer):
        self.trainer = trainer
        self.trainer.stats["memorization"] = {
            "log_name": "memorization",
            "log_epoch_fields": ["{val:.2f}", "{epoch:.2f}"],
            "val": float("nan"), "epoch": 0,
        }

    def epoch(self, epoch_index):
        import numpy as np
        import torch
        from . import ops
        from .utils import cudize
        if epoch_index % self.output_snapshot_ticks != 0:
            return
        tr = self.trainer
        values = []
        with torch.no_grad():
            for data in self.create_dataloader_fun(min(tr.stats["minibatch_size"], 1024), False, tr.dataset.model_depth,
                                                   tr.dataset.alpha):
                d_real, _, _ = tr.discriminator(cudize(data))
                values.append(float(ops.mean_scale(d_real, 1.0)) if d_real.is_cuda else float(d_real.mean()))
        tr.stats["memorization"]["val"] = np.array(values).mean()
        tr.stats["memorization"]["epoch"] = epoch_index


def __getattr__(name):
    # the metric plugins keep the reference's import path (plugins.SlicedWDistance, plugins.FidCalculator) but
    # live in metrics.py, which imports this module for the Plugin base class
    if name in ("SlicedWDistance", "FidCalculator"):
        from . import metrics
        return getattr(metrics, name)
    raise AttributeError(name)
