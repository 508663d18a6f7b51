"""The training loop that owns the hot path.  ``Trainer`` keeps the constructor, attributes and
plugin protocol of the reference's ``trainer.py`` (7-96): per step ``d_training_repeats``
discriminator iterations (fresh reals and latents each) then one generator iteration; plugins are
fired per iteration, per tick ("epoch") and at the end."""
from __future__ import annotations

import bisect
import itertools

from . import ops
from .utils import cudize


class _Schedule:
    """Plugins of one trigger unit, kept sorted by (next firing time, registration order)."""

    def __init__(self, unit):
        self.unit = unit
        self.entries = []
        self._order = itertools.count()

    def add(self, plugin, first_time):
        bisect.insort(self.entries, (first_time, next(self._order), plugin))

    def _period(self, plugin):
        triggers = plugin.trigger_interval
        if not isinstance(triggers, list):
            triggers = [triggers]
        period = None
        for length, unit in triggers:
            if unit == self.unit:
                period = length  # the last matching trigger wins, as in the reference (trainer.py:56-58)
        return period

    def fire(self, time, *args):
        """Call every plugin that is due at ``time`` (earliest first) and re-arm it one period later."""
        while self.entries and self.entries[0][0] <= time:
            _, order, plugin = self.entries.pop(0)
            getattr(plugin, self.unit)(time, *args)
            bisect.insort(self.entries, (time + self._period(plugin), order, plugin))

    def __len__(self):
        return len(self.entries)


class Trainer(object):
    # conv weight gradients are summed straight into persistent buffers (ops.direct_weight_grads)
    direct_grads = True

    def __init__(self, discriminator, generator, d_loss, g_loss, dataset, random_latents_generator, resume_nimg,
                 optimizer_g, optimizer_d, d_training_repeats: int = 5, tick_kimg_default: float = 5.0):
        if d_training_repeats < 1:
            raise AssertionError("d_training_repeats must be at least 1")
        self.discriminator, self.generator = discriminator, generator
        self.d_loss, self.g_loss = d_loss, g_loss
        self.optimizer_d, self.optimizer_g = optimizer_d, optimizer_g
        self.lr_scheduler_d = self.lr_scheduler_g = None
        self.dataset = dataset
        self.dataiter = None
        self.random_latents_generator = random_latents_generator
        self.d_training_repeats = d_training_repeats
        # progress counters (reals shown, ticks of tick_kimg_default thousand reals, steps)
        self.cur_nimg = self.tick_start_nimg = resume_nimg
        self.tick_duration_nimg = int(tick_kimg_default * 1000)
        self.cur_tick = self.iterations = self.time = 0
        self.stats = {
            "kimg_stat": dict(val=self.cur_nimg / 1000., log_epoch_fields=["{val:8.3f}"], log_name="kimg"),
            "tick_stat": dict(val=self.cur_tick, log_epoch_fields=["{val:5}"], log_name="tick"),
        }
        self.plugin_queues = {unit: _Schedule(unit) for unit in ("iteration", "epoch", "end")}
        # hook run after each backward and before the optimizer step (gradient all-reduce)
        self.grad_sync = None

    # ---- plugins -----------------------------------------------------------------------------
    def register_plugin(self, plugin):
        plugin.register(self)
        triggers = plugin.trigger_interval
        for length, unit in (triggers if isinstance(triggers, list) else [triggers]):
            self.plugin_queues[unit].add(plugin, length)

    def call_plugins(self, queue_name, time, *args):
        self.plugin_queues[queue_name].fire(time, *args)

    # ---- outer loop ----------------------------------------------------------------------------
    def _tick_finished(self, total_nimg):
        return self.cur_nimg >= min(self.tick_start_nimg + self.tick_duration_nimg, total_nimg)

    def run(self, total_kimg=1):
        total_nimg = int(total_kimg * 1000)
        try:
            while self.cur_nimg < total_nimg:
                self.train()
                if self._tick_finished(total_nimg):
                    self.cur_tick += 1
                    self.tick_start_nimg = self.cur_nimg
                    self.stats["kimg_stat"]["val"] = self.cur_nimg / 1000.
                    self.stats["tick_stat"]["val"] = self.cur_tick
                    self.call_plugins("epoch", self.cur_tick)
        except KeyboardInterrupt:
            return
        self.call_plugins("end", 1)

    # ---- one step --------------------------------------------------------------------------------
    def _step(self, loss, optimizer, module):
        with ops.direct_weight_grads(self.direct_grads):
            loss.backward()
        if self.grad_sync is None:
            optimizer.step()
        else:
            self.grad_sync(module, optimizer)

    def _next_latents(self):
        return cudize(next(self.random_latents_generator))

    def _count_reals(self, local_batch: int) -> None:
        """The progress counter follows the reals shown to the WHOLE job: with a data-parallel ``grad_sync`` every
        rank holds ``local_batch`` of a global batch of ``world`` times as many, and the schedule the plugins read
        (depth, fade-in, ticks: reference plugins.py:78-119, trainer.py:62-77) is defined in reals."""
        self.cur_nimg += local_batch * int(getattr(self.grad_sync, "world", 1) or 1)

    def train(self):
        """5 x (critic loss, backward, Adam) then generator loss, backward, Adam: reference trainer.py:79-96."""
        if self.lr_scheduler_g is not None:
            self.lr_scheduler_g.step(self.cur_nimg / self.d_training_repeats)
        latents = self._next_latents()
        for _ in range(self.d_training_repeats):
            if self.lr_scheduler_d is not None:
                self.lr_scheduler_d.step(self.cur_nimg)
            reals = cudize(next(self.dataiter))
            self._count_reals(reals["x"].size(0))
            d_loss = self.d_loss(self.discriminator, self.generator, reals, latents)
            self._step(d_loss, self.optimizer_d, self.discriminator)
            latents = self._next_latents()
        g_loss = self.g_loss(self.discriminator, self.generator, reals, latents)
        self._step(g_loss, self.optimizer_g, self.generator)
        self.iterations += 1
        self.call_plugins("iteration", self.iterations, g_loss, d_loss)
