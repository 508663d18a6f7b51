"""HIP-graph execution of the training step.

At the final stage one ``Trainer.train()`` enqueues ~4000 kernels; launched one by one from Python
the host (autograd bookkeeping + one C-ABI call per kernel) takes longer than the GPU.  Every entry
point of the kernel library only enqueues work on the current stream (no allocation, no copies,
no synchronisation), the optimizer keeps its step counters on the device and the GDrop noise can be
drawn on the device, so a whole discriminator iteration (three critic passes, the gradient-penalty
double backward, backward, Adam) and a whole generator iteration can each be captured once into a
hipGraph and replayed with fresh inputs copied into static buffers.
"""
from __future__ import annotations

import contextlib
import functools
import gc
import inspect
import os
from typing import Callable, Optional

import torch

from . import layers, ops
from .trainer import Trainer
from .utils import cudize


@contextlib.contextmanager
def capture(graph):
    """``torch.cuda.graph(graph)`` with the garbage collector out of the way.  A collection that starts while a stream
    is capturing (on any thread: autograd's backward thread allocates Python objects too) can free device objects of
    earlier, unrelated work -- the graphs, streams and tensors of a discarded trainer that sat in a reference cycle --
    and the runtime calls that release them are not allowed while a capture in the default (global) mode is under
    way: the process aborts.  So: collect first, outside the capture, and keep the collector off until it ends."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was_enabled:
            gc.enable()


class GraphedCall:
    """Capture ``fn()`` into a HIP graph after ``warmup`` eager runs on a side stream.  ``fn`` must
    read its inputs from tensors that stay alive and at the same address (static buffers)."""

    def __init__(self, fn: Callable[[], torch.Tensor], warmup: int = 3):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        ops.bump_weights_epoch()  # every derived weight view is rebuilt inside the captured region
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        # a replay may step an optimizer (weights change with no Python in between) and rewrites the shared
        # flipped-weight views with the weights it started from: eager code that runs next must rebuild them
        ops.bump_weights_epoch()
        return self.out


class _SplitCriticIteration:
    """One critic iteration as several single-chain captured graphs instead of one.

    The critic loss is a sum of a term of the stacked [reals; fakes] pass alone and a term of the
    interpolates' pass alone (``losses.discriminator_loss(part=...)``), so the two passes -- forward and
    backward -- never meet before the weight gradients are summed.  Each is a chain of short kernels (5-25 us
    at 32 samples per GPU) whose launch gaps and partly filled last rounds the other chain can fill:

        graph A (launch stream): stacked pass, its loss term, its backward; weight-gradient slabs only
        graph B (side stream)  : interpolates' pass, create_graph input gradient, penalty, double backward
        closing graphs (launch stream, behind both): sums over the chains of the gradients autograd
                                 accumulated; slab sums of A and B, loss sum, [Adam + new flipped weight views for
                                 the next iteration | data-parallel: one gradient bucket per closing graph, each
                                 with its all-reduce started right behind it and its Adam in a graph of its own]

    (``parts`` names the chains; the score term can be split once more into a reals' and a fakes' chain.)
    Every graph is a single chain captured on one stream; the chains are replayed on real streams of their
    own and ordered by ordinary stream waits.  (One captured graph with two parallel branches was measured first: on
    ROCm 7.2 its origin-stream branch started before earlier work of the launch stream had finished.)"""

    serial = False  # replay the chains one after the other on the launch stream (comparison runs)

    def __init__(self, trainer, d_loss_fn, sync, bucketed, warmup=3, parts=("scores", "penalty")):
        self.trainer, self.sync, self.bucketed = trainer, sync, bucketed
        dis, gen, opt = trainer.discriminator, trainer.generator, trainer.optimizer_d
        self.parts = tuple(parts)
        self.sides = [torch.cuda.Stream() for _ in self.parts[1:]]  # default priority

        def chain(part):
            loss = d_loss_fn(dis, gen, {"x": trainer._real}, {"z": trainer._z}, part=part,
                             zero_grad=(part == self.parts[0]))
            with ops.direct_weight_grads(trainer.direct_grads, deferred=True):
                loss.backward(trainer._seed)  # (a persistent one: backward() alone fills a fresh one per replay)
            # gradients that took autograd's own route (the full-length layer's): kept per chain
            stash = [(p, p.grad) for p in dis.parameters() if p.grad is not None]
            for p, _ in stash:
                p.grad = None
            return loss.detach(), stash

        def close_early(results):
            """Gradients that are complete once the chains are (autograd's route: the full-length layer, the
            linear layer): summed over the chains and, data-parallel, packed into a bucket of their own."""
            grads = {}
            for level, (_, stash) in enumerate(results):
                # the first chain's tensors take the sums, in place: one launch per further chain, not one per tensor
                # (a chain rewrites its tensors at its next replay)
                both = [(grads[p], g) for p, g in stash if p in grads]
                if both:
                    ops.multi_add_([a for a, _ in both], [b for _, b in both])
                for p, g in stash:
                    grads.setdefault(p, g)
            for p, g in grads.items():
                p.grad = g
            early = list(grads)
            if bucketed and early:
                self.bucket_early = sync.pack(dis, params=early, slot="early")
            return early

        def close_late(results, early):
            """Slab sums of all chains -> p.grad of the conv weights, the loss, and [their bucket | Adam + new
            flipped weight views]."""
            ops.flush_direct_grads()
            loss = results[0][0]
            for other, _ in results[1:]:
                loss = loss + other
            if bucketed:
                taken = {id(p) for p in early}
                late = [p for p in dis.parameters() if p.grad is not None and id(p) not in taken]
                if late:
                    self.bucket_late = sync.pack(dis, params=late, slot="late")
            else:
                self._step(opt, dis)
            return loss

        def apply_early():
            if self.bucket_early is not None:
                sync.apply(opt, self.bucket_early)

        def apply_late():
            if self.bucket_late is not None:
                sync.apply(opt, self.bucket_late)
            ops.refresh_flipped_weights(dis)  # for the next iteration's chains

        self.bucket_early = self.bucket_late = None
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream()
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            for _ in range(warmup):
                results = [chain(part) for part in self.parts]
                close_late(results, close_early(results))
                if bucketed:
                    apply_early()
                    apply_late()
        cur.wait_stream(warm)
        torch.cuda.synchronize()
        # the chains read the flipped weight views the closing graph of the previous iteration has built
        # (the warm-up's last step just did): nothing may rebuild them inside a chain, they run side by side
        ops.refresh_flipped_weights(dis)
        torch.cuda.synchronize()
        try:
            self.graphs = [torch.cuda.CUDAGraph() for _ in self.parts]
            results = []
            for graph, part in zip(self.graphs, self.parts):
                with capture(graph):
                    results.append(chain(part))
            # Data-parallel: the closing work is two graphs with a collective started behind each, so that the
            # first bucket (the full-length layer's weights: most of the bytes) is summed over the ranks while
            # the slab sums of the conv weights run, and its Adam runs while the second bucket is summed.
            # (On one GPU both halves of the closing work are one graph.)
            self.graph_early = torch.cuda.CUDAGraph() if bucketed else None
            self.graph_late = torch.cuda.CUDAGraph()
            if bucketed:
                with capture(self.graph_early):
                    early = close_early(results)
                with capture(self.graph_late):
                    self.out = close_late(results, early)
            else:
                with capture(self.graph_late):
                    self.out = close_late(results, close_early(results))
        except BaseException:
            ops.discard_direct_grads()
            raise
        self.apply_early = GraphedCall(apply_early) if bucketed and self.bucket_early is not None else None
        self.apply_late = GraphedCall(apply_late) if bucketed else None

    def _step(self, opt, dis):
        opt.step()
        ops.refresh_flipped_weights(dis)  # for the next iteration's chains

    def __call__(self):
        main = torch.cuda.current_stream()
        if self.serial:
            for graph in self.graphs:
                graph.replay()
        else:
            for side, graph in zip(self.sides, self.graphs[1:]):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    graph.replay()
            self.graphs[0].replay()
            for side in self.sides:
                main.wait_stream(side)
        if not self.bucketed:
            self.graph_late.replay()
        else:
            self.graph_early.replay()
            # collectives are started between replays, never inside a capture
            sync = self.sync
            first = sync.reduce(self.bucket_early, async_op=True) if self.bucket_early is not None else None
            self.graph_late.replay()
            second = sync.reduce(self.bucket_late, async_op=True) if self.bucket_late is not None else None
            if self.apply_early is not None:
                sync.wait(first)
                self.apply_early()
            sync.wait(second)
            self.apply_late()
        ops.bump_weights_epoch()
        return self.out


def _snapshot(modules, optimizers):
    params = [(p, p.detach().clone()) for m in modules for p in m.parameters()]
    params += [(b, b.detach().clone()) for m in modules for b in m.buffers()]  # (running statistics of batch norms)
    states = []
    for opt in optimizers:
        for p, st in opt.state.items():
            states.append((opt, p, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}))
    had = [(opt, set(opt.state.keys())) for opt in optimizers]
    return params, states, had


def _restore(snap):
    params, states, had = snap
    with torch.no_grad():
        for p, v in params:
            p.copy_(v)
        for opt, p, st in states:
            for k, v in st.items():
                if torch.is_tensor(v):
                    opt.state[p][k].copy_(v)
        for opt, keys in had:  # state created by the warm-up runs starts from zero again
            for p, st in opt.state.items():
                if p not in keys:
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
    ops.bump_weights_epoch()


def _accepts(fn, name: str) -> bool:
    """Whether the callable takes a keyword argument ``name``."""
    try:
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return name in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


def _names(fn, name: str) -> bool:
    """Whether the callable has a parameter called ``name`` that its caller has not bound already (a
    ``functools.partial`` with that keyword, or a wrapper that only forwards ``**kwargs``, decides it itself)."""
    if isinstance(fn, functools.partial) and name in (fn.keywords or {}):
        return False
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class GraphedTrainer(Trainer):
    """``Trainer`` whose discriminator and generator iterations replay captured HIP graphs.

    Graphs are (re)captured whenever (depth, alpha, batch shape, optimizers) change; the warm-up
    runs needed for capture are undone (parameters and optimizer state are restored), so the
    sequence of updates equals the eager trainer's.  During fade-in (alpha < 1) the factor is read
    by the blend kernels from a one-element device tensor that is refreshed before every step, so
    the same graphs serve the whole transition.  Requires the
    device noise mode (``layers.set_drop_mode('device')``): host RNG draws cannot be captured.

    With a data-parallel ``grad_sync`` (``parallel.DataParallelStep``) every iteration is two
    graphs -- forward + backward + gradient-bucket pack, and the Adam step reading the bucket --
    with the gradient all-reduce issued eagerly between the two replays."""

    # generate the fakes of all critic iterations of a step in one generator pass
    batch_fakes = True
    # run the two independent halves of a critic iteration side by side on two streams (needs batch_fakes and
    # a critic loss that can be asked for its parts); EEGGAN_SPLIT_CRITIC=0 keeps one graph on one stream
    split_critic = os.environ.get("EEGGAN_SPLIT_CRITIC", "1") != "0"
    critic_chains = tuple(os.environ.get("EEGGAN_CRITIC_CHAINS", "scores,penalty").split(","))

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._key = None
        self._d_parts = None
        self._g_parts = None
        self._real: Optional[torch.Tensor] = None
        self._z: Optional[torch.Tensor] = None
        # fakes of all critic iterations of a step from one generator pass (None: one pass per iteration)
        self._fakes_call: Optional[GraphedCall] = None
        self._zall: Optional[torch.Tensor] = None
        self._fake: Optional[torch.Tensor] = None
        self._mix: Optional[torch.Tensor] = None

    def _capture(self, real, z, key):
        if layers.get_drop_mode() != "device":
            raise RuntimeError("GraphedTrainer needs layers.set_drop_mode('device')")
        sync = self.grad_sync
        if sync is not None and not (hasattr(sync, "pack") and hasattr(sync, "reduce") and hasattr(sync, "apply")):
            raise NotImplementedError("GraphedTrainer: grad_sync must offer pack/reduce/apply phases")
        bucketed = sync is not None and getattr(sync, "bucketed", True)
        # static inputs: reals and (when they are made ahead of the critic iterations) fakes back to back in one
        # buffer, so that the stacked critic pass reads them without a concatenation
        n = real.shape[0]
        self._pair = torch.empty((2 * n,) + tuple(real.shape[1:]), device=real.device, dtype=real.dtype)
        self._real = self._pair[:n]
        self._real.copy_(real)
        self._z = z.clone()
        self._seed = torch.ones((), device=real.device, dtype=torch.float32)  # d loss / d loss
        snap = _snapshot([self.discriminator, self.generator], [self.optimizer_d, self.optimizer_g])

        # The generator does not change during the critic iterations of a step (reference
        # trainer.py:83-91), so their fakes come from ONE generator pass over all their latents:
        # a fifth of the launches, five times the work per launch.  Needs a critic loss that takes
        # ready-made fakes (losses.discriminator_loss does).
        d_loss_fn = self.d_loss
        self._fakes_call = self._zall = self._fake = None
        # (a generator with batch norm layers couples the samples of a pass: its fakes are made per iteration)
        # (a spectral norm's power iteration runs once per forward: as many generator passes as the eager trainer)
        per_sample = not any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) or getattr(m, "spectral", False)
                             for m in self.generator.modules())
        if self.batch_fakes and per_sample and self.d_training_repeats > 1 and _accepts(self.d_loss, "fake"):
            self._zall = z.repeat(self.d_training_repeats, *([1] * (z.dim() - 1)))

            def all_fakes():
                layers.drop_pool_reset()
                with torch.no_grad():
                    return self.generator({"z": self._zall})[0]["x"]

            self._fakes_call = GraphedCall(all_fakes)
            self._fake = self._pair[n:]
            self._fake.copy_(self._fakes_call.out[:n])
            d_loss_fn = lambda dis, gen, r, zz, **kw: self.d_loss(dis, gen, r, zz, fake=self._fake, **kw)  # noqa: E731
        # The interpolation factors of the penalty come from a static buffer that train() fills from ONE draw per
        # step.  Drawn inside the captured iteration (the loss's own uniform_()) they cost a launch in the penalty
        # chain plus two fills of the generator's seed and offset cells on the launch stream before every replay.
        self._mix = None
        if _names(self.d_loss, "mixing"):
            self._mix = torch.empty((n, 1, 1), device=real.device, dtype=torch.float32).uniform_()
            with_fakes = d_loss_fn
            d_loss_fn = lambda dis, gen, r, zz, **kw: with_fakes(dis, gen, r, zz, mixing=self._mix, **kw)  # noqa: E731

        def iteration(loss_fn, module, optimizer):
            """(graph of forward + backward [+ bucket pack], bucket, graph of the optimizer step)."""
            box = {}

            def fwd_bwd():
                loss = loss_fn(self.discriminator, self.generator, {"x": self._real}, {"z": self._z})
                with ops.direct_weight_grads(self.direct_grads):
                    loss.backward(self._seed)
                if bucketed:
                    box["bucket"] = sync.pack(module)
                else:
                    optimizer.step()
                return loss.detach()

            first = GraphedCall(fwd_bwd)
            if not bucketed:
                return first, None, None
            bucket = box["bucket"]
            return first, bucket, GraphedCall(lambda: sync.apply(optimizer, bucket))

        self._d_parts = None
        # (with spectral norm every critic pass advances the layers' power-iteration vectors in place: passes side by
        # side on two streams would race on them)
        spectral_d = any(getattr(m, "spectral", False) for m in self.discriminator.modules())
        if self.split_critic and not spectral_d and self._fake is not None and _accepts(self.d_loss, "part"):
            from .losses import PartsUnavailable
            try:
                self._d_parts = _SplitCriticIteration(self, d_loss_fn, sync, bucketed, parts=self.critic_chains)
            except PartsUnavailable:  # a loss whose terms do not separate: one graph per iteration
                _restore(snap)
        if self._d_parts is None:
            self._d_parts = iteration(d_loss_fn, self.discriminator, self.optimizer_d)
        self._g_parts = iteration(self.g_loss, self.generator, self.optimizer_g)
        _restore(snap)
        # (a split critic iteration reads the views its predecessor's closing graph built: the first one
        # needs them built here, from the restored weights)
        ops.refresh_flipped_weights(self.discriminator)
        self._key = key

    def _sync_alpha(self):
        """The fade-in factor is a kernel operand read from device memory: refresh it before the
        replays of this step (the depth manager moves it between steps, reference plugins.py:115-117)."""
        for net in (self.generator, self.discriminator):
            if net.alpha < 1.0:
                if net.alpha_dev is None:
                    net.alpha_dev = torch.empty(1, device=next(net.parameters()).device, dtype=torch.float32)
                net.alpha_dev.fill_(float(net.alpha))
            # (at alpha == 1 the networks take the branch without a blend and never read alpha_dev)

    @staticmethod
    def _stage(dsts, srcs):
        """Fresh inputs into the static buffers the graphs read: one multi-tensor launch when everything is a
        contiguous float32 device tensor (the usual case), torch's copies otherwise (another dtype, host memory)."""
        plain = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == d.device
                    for d, t in zip(dsts, srcs))
        if plain:
            ops.multi_copy_(dsts, srcs)
        else:
            for d, t in zip(dsts, srcs):
                d.copy_(t)

    def _replay(self, parts):
        if callable(parts):
            return parts()
        first, bucket, second = parts
        loss = first()
        if bucket is not None:
            # the collective runs between two replays, never inside a captured region
            self.grad_sync.reduce(bucket)
            second()
        return loss

    def train(self):
        gen, dis = self.generator, self.discriminator
        if self.lr_scheduler_d is not None or self.lr_scheduler_g is not None:
            self._key = None
            for net in (gen, dis):
                net.alpha_dev = None
            return super().train()
        self._sync_alpha()
        repeats = self.d_training_repeats
        real = cudize(next(self.dataiter))["x"]
        zs = [cudize(next(self.random_latents_generator))["z"] for _ in range(repeats + 1)]
        # (the options that fold the fade-in factor into kernel arguments instead of reading it from device memory
        # -- the generator's summed signals, the critic's input at every stage -- need a new capture per value)
        baked = (float(gen.alpha) if getattr(gen, "rgb_generation_mode", "pggan") != "pggan" else None,
                 float(dis.alpha) if getattr(dis, "input_to_all_layers", False) else None)
        key = (gen.depth, gen.alpha < 1.0, dis.alpha < 1.0, baked, tuple(real.shape), tuple(zs[0].shape), repeats,
               id(self.optimizer_d), id(self.optimizer_g),
               self.optimizer_d.param_groups[0]["lr"], self.optimizer_g.param_groups[0]["lr"])
        if key != self._key:
            self._capture(real, zs[0], key)
        fakes = None
        if self._fakes_call is not None:
            ops.multi_copy_(list(self._zall.chunk(repeats)), zs[:repeats])
            fakes = self._fakes_call()
        d_loss = None
        n = real.size(0)
        mixes = None
        if self._mix is not None:
            mixes = torch.rand((repeats,) + tuple(self._mix.shape), device=self._mix.device, dtype=torch.float32)
        for i in range(repeats):
            if i > 0:
                real = cudize(next(self.dataiter))["x"]
                if real.shape != self._real.shape:
                    raise RuntimeError("GraphedTrainer: the batch shape changed inside a step")
            self._count_reals(n)
            extra = ([self._mix], [mixes[i]]) if mixes is not None else ([], [])
            if fakes is not None:  # (the iteration's latents went into the fakes already); all inputs in one launch
                self._stage([self._real, self._fake] + extra[0], [real, fakes[i * n:(i + 1) * n]] + extra[1])
            else:
                self._stage([self._real, self._z] + extra[0], [real, zs[i]] + extra[1])
            d_loss = self._replay(self._d_parts)
        self._stage([self._z], [zs[repeats]])
        g_loss = self._replay(self._g_parts)
        self.last_losses = (g_loss, d_loss)  # device scalars of the step's last iterations
        self.iterations += 1
        self.call_plugins("iteration", self.iterations, *(g_loss, d_loss))
