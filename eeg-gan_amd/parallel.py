"""Data-parallel training of the step across the GPUs of one node: one process per GPU,
gradients summed with one RCCL all-reduce per optimizer step over a persistent flat bucket.

The reference is single-device (reference utils.py:116-126); this module is the MI355X-side
addition.  Every loss term of the step is a mean over samples (reference losses.py:94-100,118),
so averaging per-rank gradients of equal local batches reproduces the global-batch gradient --
except that MinibatchStddev groups samples in fours with stride B/4 (reference layers.py:79-87):
``shard_batch`` therefore hands each rank whole groups of the global batch.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist

from . import optim
from ._lib import call


def shard_batch(global_batch: torch.Tensor, rank: int, world: int, group: int = 4) -> torch.Tensor:
    """Rows of ``global_batch`` that rank ``rank`` owns so that the union over ranks of the
    per-rank minibatch-stddev groups equals the groups of the single-process global batch:
    global sample j + k*(B/G) (k < G) belongs to the rank that owns column j of [0, B/G)."""
    B = global_batch.shape[0]
    G = min(B, group)
    if B % G or (B // G) % world:
        raise ValueError(f"global batch {B} cannot be split into whole groups of {G} over {world} ranks")
    M = B // G
    per = M // world
    cols = torch.arange(rank * per, (rank + 1) * per)
    idx = (torch.arange(G)[:, None] * M + cols[None, :]).reshape(-1)
    return global_batch[idx.to(global_batch.device)]


class GradBucket:
    """A flat fp32 buffer holding the gradients of the parameters that currently receive one.
    Rebuilt whenever that set changes (depth or fade-in phase change: reference plugins.py:102-117)."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.views: Dict[torch.nn.Parameter, torch.Tensor] = {}
        off = 0
        for p in params:
            self.views[p] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def key(self):
        return tuple(id(p) for p in self.params)

    def pack(self):
        """Copy every ``p.grad`` into its slice with ONE multi-tensor kernel launch."""
        rows = []
        for p in self.params:
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            rows.append((self.views[p].data_ptr(), g.data_ptr(), 0, 0, 0, p.numel()))
        call("eeggan_multi_copy", optim.build_table(rows), len(rows), torch.cuda.current_stream().cuda_stream)


class DataParallelStep:
    """Callable installed as ``Trainer.grad_sync``: pack -> all-reduce(sum) -> Adam reading the
    bucket with grad_scale = 1/world.  With ``world == 1`` (or no process group) it only steps,
    unless ``always_bucket`` asks for the full pack/reduce/apply sequence (used by tests).

    The three phases are also exposed one by one (``pack``, ``reduce``, ``apply``) so that a graph
    trainer can capture ``pack`` with the backward and ``apply`` on its own, and issue the
    collective between the two replays, outside any capture."""

    def __init__(self, process_group=None, always_bucket: bool = False):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.bucketed = self.world > 1 or always_bucket
        self.buckets: Dict[tuple, GradBucket] = {}

    def pack(self, module: torch.nn.Module, params=None, slot=None) -> GradBucket:
        """Gradients of ``module`` (or of the given subset of its parameters) into their persistent bucket.
        ``slot`` names the bucket when a module's gradients travel in several."""
        if params is None:
            params = [p for p in module.parameters() if p.grad is not None]
        elif any(p.grad is None for p in params):
            raise RuntimeError("DataParallelStep.pack: a listed parameter has no gradient")
        key = (id(module), slot)
        bucket = self.buckets.get(key)
        if bucket is None or bucket.key() != tuple(id(p) for p in params):
            bucket = GradBucket(params)
            self.buckets[key] = bucket
        bucket.pack()
        return bucket

    def _staged(self, bucket: GradBucket) -> bool:
        """A host-side backend (rehearsals of the multi-rank path without one GPU per rank): the bucket
        crosses to host memory and back around the collective."""
        return bucket.flat.is_cuda and dist.get_backend(self.pg) != "nccl"

    def reduce(self, bucket: GradBucket, async_op: bool = False):
        """Sum the bucket over the ranks.  With ``async_op`` the collective is only enqueued (behind the work
        already on the current stream, on the communicator's own stream) and a handle is returned for
        ``wait``: kernels launched in between run beside it."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if self._staged(bucket):
            host = bucket.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
            bucket.flat.copy_(host)
            return None
        work = dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        return work if async_op else None

    @staticmethod
    def wait(work) -> None:
        """Order the current stream behind a collective started with ``reduce(..., async_op=True)`` (the host
        does not block)."""
        if work is not None:
            work.wait()

    def apply(self, optimizer, bucket: GradBucket) -> None:
        optimizer.grad_scale = 1.0 / self.world
        optimizer.step(grads=bucket.views)

    def __call__(self, module: torch.nn.Module, optimizer):
        if not self.bucketed:
            optimizer.step()
            return
        bucket = self.pack(module)
        self.reduce(bucket)
        self.apply(optimizer, bucket)
