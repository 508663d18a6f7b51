"""Autograd operators over the HIP kernels of ``libeeggan_hip.so``.

PyTorch is used here for device memory (``torch.empty`` through the caching allocator),
for the current HIP stream and for the autograd graph; every number is produced by a
kernel of the shared library, called through its C ABI.  There is no fallback: tensors
that are not float32 HIP tensors are rejected.

Second-order support (needed by the WGAN-GP penalty, reference losses.py:35-37,113-118):
the operators of the discriminator path build their backward out of other operators of
this module, so ``autograd.grad(..., create_graph=True)`` followed by ``backward()`` works.
Operators that only ever see a first-order backward on the path (generator layers, loss
heads) raise if differentiated twice instead of returning wrong numbers.
"""
from __future__ import annotations

import contextlib
import ctypes
import functools
import os
import weakref
from typing import Optional

import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ACT_LRELU, ACT_MASK, ACT_NONE, ACT_TANH, call

# When set, backward passes skip weight/bias gradients.  Set around the create_graph pass of
# the gradient penalty (only d(pred)/d(x_hat) is wanted there; reference losses.py:35-37).
_input_grad_only = False


@contextlib.contextmanager
def input_grad_only():
    global _input_grad_only
    prev = _input_grad_only
    _input_grad_only = True
    try:
        yield
    finally:
        _input_grad_only = prev


# --------------------------------------------------------------------------------------- helpers
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, bf16_ok: bool = False) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP device tensor, got {t.device}; this package has no CPU path")
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16):
        raise RuntimeError(f"{name}: expected float32{' or bfloat16' if bf16_ok else ''}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _bf(t) -> int:
    """Storage flag of a tensor for the typed entry points (1: bfloat16)."""
    return int(t is not None and t.dtype == torch.bfloat16)


# Storage of the wide activations ([B, C > 8, T > 32] tensors between the conv layers, and their gradients):
# float32 (the reference's, layers.py:150-174) or bfloat16 (the large-batch mode).  The signals themselves
# ([B, 5, T]) and everything at the 32-sample stage (the generator's first block, the critic's last block with
# MinibatchStddev and the full-length layer, scores, losses) stay float32, as do weights, gradients of weights
# and optimizer state: the resampler and the 1x1 to_rgb / from_rgb layers convert on the way.
_storage = torch.float32


def set_storage(mode: str) -> str:
    """'fp32' (default) or 'bf16' (which also selects the bf16 matrix precision: exact products of values that
    carry 8 bits would be wasted work).  Process-wide, for the tensors created from now on; captured graphs keep
    the mode they were captured in.  Returns the previous mode."""
    global _storage
    if mode not in ("fp32", "bf16"):
        raise ValueError(f"storage {mode!r}: expected 'fp32' or 'bf16'")
    previous = get_storage()
    _storage = torch.bfloat16 if mode == "bf16" else torch.float32
    if mode == "bf16":
        set_precision("bf16")
    return previous


def get_storage() -> str:
    return "bf16" if _storage == torch.bfloat16 else "fp32"


def _act_dtype(channels: int, length: int):
    return _storage if (channels > 8 and length > 32) else torch.float32


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------- raw kernels
# Input gradients run the forward kernel on wt[ci, co, j] = w[co, ci, K-1-j].  A layer keeps that
# view in a FlipCache next to its parameter and rebuilds it only when the weight changes: torch's
# version counter catches in-place torch updates, the epoch catches updates made by this package's
# own optimizer kernel, the data pointer catches a parameter moved to other storage.
_weights_epoch = 0


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def _flip_into(w, wt):
    cout, cin, k = w.shape
    call("eeggan_conv1d_flip_weights", w.data_ptr(), wt.data_ptr(), cout, cin, k, _stream())
    return wt


class _FlipDesc(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("wt", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("K", ctypes.c_int), ("reserved", ctypes.c_int)]


class FlipCache:
    """Flipped view of one weight.  All caches are registered: when one is found stale, every stale
    cache on the same device is rebuilt by the same launch (the views of a network go stale together,
    at its optimizer step, and each is needed a few kernels later)."""

    _all = weakref.WeakSet()

    def __init__(self, param):
        self.param = param
        self.tag = None
        self.wt = None
        self.used = False  # only views that some input-gradient launch has asked for are kept up to date
        FlipCache._all.add(self)

    def __deepcopy__(self, memo):
        # the copy follows the COPIED parameter and starts without a view (a cloned view would carry the
        # source's tag and never be found stale by the registry)
        import copy
        return FlipCache(copy.deepcopy(self.param, memo))

    def __reduce__(self):
        return (FlipCache, (self.param,))

    def _current_tag(self):
        p = self.param
        return (p.data_ptr(), p._version, _weights_epoch)

    def _stale(self):
        return self.tag != self._current_tag()

    def get(self):
        self.used = True
        if self._stale():
            dev = self.param.device
            # a cache made by copy.deepcopy or unpickling never ran __init__: (re-)register it, and always
            # rebuild the view that was asked for, whatever the registry holds
            FlipCache._all.add(self)
            todo = [c for c in FlipCache._all if c is not self and c.used and c.param.device == dev
                    and c.param.is_cuda and c._stale()]
            todo.insert(0, self)
            table = (_FlipDesc * len(todo))()
            for i, c in enumerate(todo):
                p = c.param
                cout, cin, k = p.shape
                if c.wt is None or c.wt.device != dev or c.wt.numel() != p.numel():
                    c.wt = torch.empty((cin, cout, k), device=dev, dtype=torch.float32)
                table[i].w, table[i].wt = p.data_ptr(), c.wt.data_ptr()
                table[i].Cout, table[i].Cin, table[i].K = cout, cin, k
            call("eeggan_conv1d_flip_weights_multi", table, len(todo), _stream())
            for c in todo:
                c.tag = c._current_tag()
        return self.wt


_PRECISIONS = {"fp32": 3, "bf16": 1, "fp32_mfma": 0}


def set_precision(mode: str) -> str:
    """Arithmetic of the conv layers' matrix products for the launches that follow (process-wide; captured graphs
    keep the mode they were captured in): 'fp32' (default: fp32 products from three bf16 pieces per operand),
    'bf16' (operands rounded to bf16, fp32 accumulation and storage: the large-batch mode, with its looser
    bounds) or 'fp32_mfma' (fp32 matrix instructions; comparison runs).  Returns the previous mode."""
    if mode not in _PRECISIONS:
        raise ValueError(f"precision {mode!r}: expected one of {sorted(_PRECISIONS)}")
    previous = get_precision()
    call("eeggan_set_matrix_precision", _PRECISIONS[mode])
    return previous


def get_precision() -> str:
    pieces = _lib.load().eeggan_get_matrix_precision()
    return next(k for k, v in _PRECISIONS.items() if v == pieces)


def refresh_flipped_weights(module):
    """Rebuild, on the current stream, every stale flipped view of ``module`` that an input gradient has asked
    for before (one launch).  For callers that run passes over the module on several streams: the views are
    shared, so they are rebuilt once, ahead of the passes, on a stream all of them wait for."""
    for m in module.modules():
        c = getattr(m, "_flip", None)
        if isinstance(c, FlipCache) and c.used and c._stale():
            c.get()
            return  # one launch rebuilt every stale view of this device


def flipped_weights(w, cache=None):
    if cache is not None:
        return cache.get()
    cout, cin, k = w.shape
    return _flip_into(w, torch.empty((cin, cout, k), device=w.device, dtype=torch.float32))


def conv1d_raw(x, w, bias=None, drop=None, *, in_scale=1.0, transposed=False, act=ACT_NONE, act_param=0.0,
               act_aux=None, mask_y=None, mask_drop=None, mask_slope=0.0, flip=None, out=None):
    x = _chk(x, "conv1d x", True)
    w = _chk(w, "conv1d w")
    B, Cin, T = x.shape
    K = w.shape[2]
    if transposed:
        if w.shape[0] != Cin:
            raise RuntimeError(f"conv1d(transposed): weight {tuple(w.shape)} does not match input channels {Cin}")
        w = flipped_weights(w.detach(), flip)
    elif w.shape[1] != Cin:
        raise RuntimeError(f"conv1d: weight {tuple(w.shape)} does not match input channels {Cin}")
    Cout = w.shape[0]
    ydt = out.dtype if out is not None else _act_dtype(Cout, T)
    for name, t, shape, dt in (("bias", bias, (Cout,), torch.float32), ("drop", drop, (B, Cout), torch.float32),
                               ("act_aux", act_aux, (B, Cout, T), ydt), ("mask_y", mask_y, (B, Cin, T), x.dtype),
                               ("mask_drop", mask_drop, (B, Cin), torch.float32)):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != dt or not t.is_cuda):
            raise RuntimeError(f"conv1d: {name} must be a contiguous {dt} device tensor of shape {shape}, "
                               f"got {t.dtype} {tuple(t.shape)}")
    if out is None:
        y = torch.empty((B, Cout, T), device=x.device, dtype=ydt)
    else:
        y = out
        if (tuple(y.shape) != (B, Cout, T) or not y.is_contiguous() or y.device != x.device
                or y.dtype not in (torch.float32, torch.bfloat16)):
            raise RuntimeError(f"conv1d: out must be a contiguous float32 or bfloat16 tensor of shape {(B, Cout, T)}")
    call("eeggan_conv1d_t", x.data_ptr(), w.data_ptr(), _p(bias), _p(drop), y.data_ptr(), B, Cin, Cout, T, K,
         float(in_scale), int(act), float(act_param), _p(act_aux), _p(mask_y), _p(mask_drop), float(mask_slope),
         _bf(x), _bf(y), _stream())
    return y


def conv1d_pixelnorm_raw(x, w, bias, drop, *, in_scale, slope, eps=1e-8):
    """(y, d) of the fused conv + LeakyReLU + PixelNorm + multiplier kernel, or None when the shape is
    not covered (the caller then runs the conv and the norm as two kernels)."""
    x = _chk(x, "conv1d x", True)
    w = _chk(w, "conv1d w")
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    if w.shape[1] != Cin or not _lib.load().eeggan_conv1d_pixelnorm_supported(B, Cin, Cout, T, K):
        return None
    if x.dtype != _act_dtype(Cout, T) or (_bf(x) and not (Cin == 64 and Cout == 64 and T % 32 == 0)):
        return None  # (mixed storage types, or a bfloat16 shape the fused kernel has no form for: two kernels)
    y = torch.empty((B, Cout, T), device=x.device, dtype=x.dtype)
    d = torch.empty((B, T), device=x.device, dtype=torch.float32)
    call("eeggan_conv1d_pixelnorm_t", x.data_ptr(), w.data_ptr(), _p(bias), _p(drop), y.data_ptr(), d.data_ptr(), B, Cin,
         Cout, T, K, float(in_scale), float(slope), float(eps), _bf(x), _stream())
    return y, d


def conv1d_wgrad_raw(x, gy, K, *, in_scale=1.0, need_bias=True, mask_y=None, mask_drop=None, mask_slope=0.0,
                     out_w=None, out_b=None, beta_w=0.0, beta_b=0.0):
    """Weight/bias gradient.  ``out_w``/``out_b``: write into these buffers instead of new tensors,
    as ``out = beta * out + gradient`` (beta 0: plain overwrite)."""
    x = _chk(x, "wgrad x", True)
    gy = _chk(gy, "wgrad gy", True)
    B, Cin, T = x.shape
    if gy.shape[0] != B or gy.shape[2] != T:
        raise RuntimeError(f"conv1d_wgrad: x {tuple(x.shape)} and gy {tuple(gy.shape)} disagree")
    Cout = gy.shape[1]
    if _bf(x) or _bf(gy):
        # bfloat16 activations: the grouped entry points have the typed kernels
        (gw, gb), = conv1d_wgrad_group_raw([dict(x=x, gy=gy, K=K, in_scale=in_scale, need_bias=need_bias,
                                                 mask_y=mask_y, mask_drop=mask_drop, mask_slope=mask_slope)])
        if out_w is not None:
            gw = axpby_raw(out_w, beta_w, gw, 1.0, out=out_w) if beta_w else out_w.copy_(gw)
        if out_b is not None and gb is not None:
            gb = axpby_raw(out_b, beta_b, gb, 1.0, out=out_b) if beta_b else out_b.copy_(gb)
        return gw, gb
    for name, t, shape in (("mask_y", mask_y, (B, Cout, T)), ("mask_drop", mask_drop, (B, Cout))):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous()):
            raise RuntimeError(f"conv1d_wgrad: {name} must be contiguous with shape {shape}, got {tuple(t.shape)}")
    nbytes = _lib.load().eeggan_conv1d_wgrad_workspace(B, Cin, Cout, T, K)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    gw = out_w if out_w is not None else torch.empty((Cout, Cin, K), device=x.device, dtype=torch.float32)
    gb = out_b if out_b is not None else (
        torch.empty((Cout,), device=x.device, dtype=torch.float32) if need_bias else None)
    if tuple(gw.shape) != (Cout, Cin, K) or not gw.is_contiguous() or (gb is not None and gb.numel() != Cout):
        raise RuntimeError("conv1d_wgrad: output buffers do not match the weight/bias shapes")
    call("eeggan_conv1d_wgrad", x.data_ptr(), gy.data_ptr(), gw.data_ptr(), _p(gb), B, Cin, Cout, T, K,
         float(in_scale), _p(mask_y), _p(mask_drop), float(mask_slope), float(beta_w), float(beta_b),
         ws.data_ptr(), nbytes, _stream())
    return gw, gb


# --------------------------------------------------------------------------------------- direct weight gradients
# Inside ``direct_weight_grads()`` a plain ``loss.backward()`` does not hand conv weight gradients to
# autograd.  Each one only runs its partial-slab kernel; the slab sums of the whole backward pass are
# done by a few multi-job launches in an end-of-backward callback, straight into a persistent
# per-parameter buffer that then becomes ``p.grad`` (accumulating when the parameter already holds a
# gradient, like torch: ``p.grad`` is summed into until someone sets it to None).  Against the
# autograd route this removes, per contribution and tensor, one reduce launch, autograd's
# accumulation kernel and the allocation of a gradient tensor -- ~500 of the ~2300 launches of a
# training step, each worth ~4 us of dispatch latency.  Not used while a graph for double backward
# is being built (grad mode on) nor outside the context (``torch.autograd.grad`` keeps its meaning).
class _ReduceJob(ctypes.Structure):
    _fields_ = [("slabs", ctypes.c_void_p), ("out", ctypes.c_void_p), ("elems", ctypes.c_int64),
                ("nsplit", ctypes.c_int), ("scale", ctypes.c_float), ("beta", ctypes.c_float),
                ("reserved", ctypes.c_int)]


class _WgradJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("gy", ctypes.c_void_p), ("mask_y", ctypes.c_void_p),
                ("mask_drop", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("B", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int), ("T", ctypes.c_int),
                ("K", ctypes.c_int), ("want_bias", ctypes.c_int), ("in_scale", ctypes.c_float),
                ("mask_slope", ctypes.c_float), ("x_bf16", ctypes.c_int), ("gy_bf16", ctypes.c_int)]


def _wgrad_mask_check(mask_y, mask_drop, gy, B, Cout, T):
    for name, t, shape, dt in (("mask_y", mask_y, (B, Cout, T), gy.dtype), ("mask_drop", mask_drop, (B, Cout), torch.float32)):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != dt):
            raise RuntimeError(f"conv1d_wgrad: {name} must be contiguous {dt} with shape {shape}, "
                               f"got {t.dtype} {tuple(t.shape)}")


# Weight gradients of a backward pass are submitted in groups of this many layers per launch (the
# library deals the chip's workgroups out to the jobs of a group): each layer is then cut into fewer
# splits, so there are fewer partial slabs to write and sum, and a group pays one launch latency.
WGRAD_GROUP = int(os.environ.get("EEGGAN_WGRAD_GROUP", "16"))
# ... unless the layer keeps the chip busy for many rounds by itself (work units = 64x64-channel regions x
# pairs of 32-sample subtiles; 256 workgroups take one unit each per round): grouping then only delays
# the launch and holds on to the layer's activations (measured at 512 samples per GPU: 4 % slower grouped).
WGRAD_SOLO_UNITS = 16 * 256


class _DirectGrads:
    def __init__(self):
        self.enabled = False
        self.deferred = False  # leave the slab sums of the pass to a later flush() (several passes, one sum)
        self.streams = {}   # stream handle -> torch stream: every stream a slab kernel of this pass ran on
        self.keep = []      # workspaces the pending reductions still read
        self.jobs = []      # (round, job): jobs of one round write distinct buffers
        self.touched = {}   # id(param) -> [param, contributions so far in this backward pass]
        self.pending = {}   # kernel variant -> [(job, tensors it reads, parameters)] not yet submitted
        self.queued = False

    def reset(self):
        """Forget everything queued: a backward pass that raised never ran ``finish``."""
        self.touched, self.keep, self.jobs, self.pending, self.queued = {}, [], [], {}, False
        self.streams = {}

    def usable(self, w):
        return self.enabled and not torch.is_grad_enabled() and w.is_leaf and w.requires_grad and w.is_cuda

    def _slot(self, p):
        buf = getattr(p, "_eeggan_grad", None)
        if buf is None or buf.shape != p.shape or buf.device != p.device:
            buf = torch.empty_like(p, memory_format=torch.contiguous_format)
            p._eeggan_grad = buf
        ent = self.touched.get(id(p))
        if ent is not None:
            beta = 1.0
        elif p.grad is None:
            beta = 0.0
        elif p.grad is buf:
            beta = 1.0
        else:  # a gradient that came from elsewhere: continue from a copy of it
            buf.copy_(p.grad)
            beta = 1.0
        if ent is None:
            ent = self.touched[id(p)] = [p, 0]
        rnd = ent[1]
        ent[1] += 1
        return buf, beta, rnd

    def wgrad(self, w, bias, x, gy, in_scale, mask_y, mask_drop, slope):
        x, gy = _chk(x, "wgrad x", True), _chk(gy, "wgrad gy", True)
        B, Cin, T = x.shape
        Cout, K = gy.shape[1], w.shape[2]
        if gy.shape[0] != B or gy.shape[2] != T or tuple(w.shape) != (Cout, Cin, K):
            raise RuntimeError(f"conv1d_wgrad: x {tuple(x.shape)}, gy {tuple(gy.shape)}, w {tuple(w.shape)} disagree")
        _wgrad_mask_check(mask_y, mask_drop, gy, B, Cout, T)
        nbytes = _lib.load().eeggan_conv1d_wgrad_workspace(B, Cin, Cout, T, K)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        job = _WgradJob(x.data_ptr(), gy.data_ptr(), _p(mask_y), _p(mask_drop), ws.data_ptr(), nbytes, B, Cin, Cout,
                        T, K, int(bias is not None), float(in_scale), float(slope), _bf(x), _bf(gy))
        # the tensors stay referenced until the group is submitted: nothing may recycle their memory
        # (jobs that run the same kernel variant ON THE SAME STREAM share a launch: backward nodes run on the
        # stream of their forward pass, and two critic passes may run side by side on two streams)
        stream = torch.cuda.current_stream()
        self.streams[stream.cuda_stream] = stream
        variant = (K, Cin % 64 == 0 and Cout % 64 == 0 and T % 32 == 0, stream.cuda_stream, _bf(x), _bf(gy))
        units = ((Cout + 63) // 64) * ((Cin + 63) // 64) * ((B * ((T + 31) // 32) + 1) // 2)
        if units >= WGRAD_SOLO_UNITS:
            variant = variant + (len(self.jobs),)  # a queue of its own, submitted right away
        queue = self.pending.setdefault(variant, [])
        queue.append((job, (x, gy, mask_y, mask_drop, ws), (w, bias)))
        if not self.queued:
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            self.queued = True
        if len(queue) >= WGRAD_GROUP or units >= WGRAD_SOLO_UNITS:
            self.submit(variant)

    def submit(self, variant=None):
        """Launch the queued weight-gradient jobs (of one kernel variant, or all) and register their slab
        reductions."""
        if variant is None:
            for v in list(self.pending):
                self.submit(v)
            return
        pending = self.pending.pop(variant, [])
        if not pending:
            return
        n = len(pending)
        table = (_WgradJob * n)(*[p[0] for p in pending])
        jw, jb = (_ReduceJob * n)(), (_ReduceJob * n)()
        call("eeggan_conv1d_wgrad_partial_multi", table, n, jw, jb, variant[2])
        for i, (_, tensors, (w, bias)) in enumerate(pending):
            for p, src in ((w, jw[i]), (bias, jb[i])):
                if p is not None:
                    job = _ReduceJob(src.slabs, None, src.elems, src.nsplit, src.scale, 0.0, 0)
                    buf, beta, rnd = self._slot(p)
                    job.out, job.beta = buf.data_ptr(), beta
                    self.jobs.append((rnd, job))
            self.keep.append(tensors[4])

    def finish(self):
        if self.deferred:
            # only the slab kernels of this pass; their sums wait for flush() (which the caller orders behind
            # the streams of all the passes it collects)
            try:
                self.submit()
            except BaseException:
                self.reset()
                raise
            self.queued = False
            return
        self.flush(sync_streams=True)

    def flush(self, sync_streams=False):
        """Sum the slabs of everything collected since the last flush into the parameters' gradient buffers
        and make those ``p.grad``.  ``sync_streams``: first make the current stream wait for the streams the
        slab kernels ran on (otherwise the caller has ordered the current stream behind them)."""
        try:
            self.submit()
            # the slab sums run on the current stream, behind the slab kernels of every stream of this pass
            here = torch.cuda.current_stream()
            for handle, stream in (self.streams.items() if sync_streams else ()):
                if handle != here.cuda_stream:
                    event = torch.cuda.Event()
                    event.record(stream)
                    here.wait_event(event)
            for rnd in sorted({r for r, _ in self.jobs}):
                jobs = [j for r, j in self.jobs if r == rnd]
                table = (_ReduceJob * len(jobs))(*jobs)
                call("eeggan_wgrad_reduce_multi", table, len(jobs), here.cuda_stream)
            for p, _ in self.touched.values():
                p.grad = p._eeggan_grad
        finally:
            self.reset()


_direct = _DirectGrads()


def conv1d_wgrad_group_raw(jobs):
    """Weight/bias gradients of several independent layers through the grouped entry points.
    ``jobs``: dicts with x, gy, K and optionally in_scale, need_bias, mask_y, mask_drop, mask_slope (as
    ``conv1d_wgrad_raw``); returns a list of (gw, gb)."""
    n = len(jobs)
    table, keep, outs = (_WgradJob * n)(), [], []
    for i, j in enumerate(jobs):
        x, gy, K = _chk(j["x"], "wgrad x", True), _chk(j["gy"], "wgrad gy", True), int(j["K"])
        B, Cin, T = x.shape
        Cout = gy.shape[1]
        if gy.shape[0] != B or gy.shape[2] != T:
            raise RuntimeError(f"conv1d_wgrad: x {tuple(x.shape)} and gy {tuple(gy.shape)} disagree")
        mask_y, mask_drop = j.get("mask_y"), j.get("mask_drop")
        _wgrad_mask_check(mask_y, mask_drop, gy, B, Cout, T)
        nbytes = _lib.load().eeggan_conv1d_wgrad_workspace(B, Cin, Cout, T, K)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        need_bias = bool(j.get("need_bias", True))
        table[i] = _WgradJob(x.data_ptr(), gy.data_ptr(), _p(mask_y), _p(mask_drop), ws.data_ptr(), nbytes, B, Cin,
                             Cout, T, K, int(need_bias), float(j.get("in_scale", 1.0)), float(j.get("mask_slope", 0.0)),
                             _bf(x), _bf(gy))
        keep.append((x, gy, mask_y, mask_drop, ws))
        gw = torch.empty((Cout, Cin, K), device=x.device, dtype=torch.float32)
        gb = torch.empty((Cout,), device=x.device, dtype=torch.float32) if need_bias else None
        outs.append((gw, gb))
    jw, jb = (_ReduceJob * n)(), (_ReduceJob * n)()
    call("eeggan_conv1d_wgrad_partial_multi", table, n, jw, jb, _stream())
    red = []
    for i, (gw, gb) in enumerate(outs):
        jw[i].out = gw.data_ptr()
        red.append(jw[i])
        if gb is not None:
            jb[i].out = gb.data_ptr()
            red.append(jb[i])
    call("eeggan_wgrad_reduce_multi", (_ReduceJob * len(red))(*red), len(red), _stream())
    return outs


def flush_direct_grads():
    """Close a series of backward passes run under ``direct_weight_grads(deferred=True)``: one set of slab
    sums for all of them, on the current stream, which the caller has ordered behind those passes."""
    _direct.flush(sync_streams=False)


def discard_direct_grads():
    """Forget what deferred passes have collected (their slabs are never summed)."""
    _direct.reset()


@contextlib.contextmanager
def direct_weight_grads(enabled=True, deferred=False):
    if _direct.queued:
        # left over from a backward pass that raised after queueing work (the engine then never ran the
        # end-of-backward callback): without this every later pass would skip registering it
        _direct.reset()
    prev, _direct.enabled = (_direct.enabled, _direct.deferred), bool(enabled)
    _direct.deferred = bool(deferred)
    try:
        yield
    except BaseException:
        _direct.reset()
        raise
    finally:
        _direct.enabled, _direct.deferred = prev


def _same_dtype(what, *tensors):
    if len({t.dtype for t in tensors}) != 1:
        raise RuntimeError(f"{what}: tensors of different storage types {[str(t.dtype) for t in tensors]}")


def lrelu_drop_bwd_raw(g, y, drop, slope):
    g = _chk(g, "lrelu_drop_bwd g", True)
    _same_dtype("lrelu_drop_bwd", g, y)
    B, Cc, T = y.shape
    out = torch.empty_like(y)
    call("eeggan_lrelu_drop_bwd_t", g.data_ptr(), y.data_ptr(), _p(drop), out.data_ptr(), B, Cc, T, float(slope),
         _bf(y), _stream())
    return out


def pixelnorm_drop_fwd_raw(h, drop, eps=1e-8):
    h = _chk(h, "pixelnorm h", True)
    B, Cc, T = h.shape
    y = torch.empty_like(h)
    d = torch.empty((B, T), device=h.device, dtype=torch.float32)
    call("eeggan_pixelnorm_drop_fwd_t", h.data_ptr(), _p(drop), y.data_ptr(), d.data_ptr(), B, Cc, T, float(eps),
         _bf(h), _stream())
    return y, d


def pixelnorm_drop_bwd_raw(gy, y, drop, d, slope):
    gy = _chk(gy, "pixelnorm gy", True)
    _same_dtype("pixelnorm_drop_bwd", gy, y)
    B, Cc, T = y.shape
    gc = torch.empty_like(y)
    call("eeggan_pixelnorm_drop_bwd_t", gy.data_ptr(), y.data_ptr(), _p(drop), d.data_ptr(), gc.data_ptr(), B, Cc, T,
         float(slope), _bf(y), _stream())
    return gc


def axpby_raw(a, ca, b=None, cb=0.0, out=None):
    a = _chk(a, "axpby a", True)
    if b is not None:
        b = _chk(b, "axpby b", True)
        if b.shape != a.shape:
            raise RuntimeError(f"axpby: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        _same_dtype("axpby", a, b)
    if out is None:
        out = torch.empty_like(a)
    elif out.shape != a.shape or out.dtype != a.dtype or not out.is_contiguous():
        raise RuntimeError("axpby: out does not match the operands")
    call("eeggan_axpby_t", a.data_ptr(), float(ca), _p(b), float(cb), out.data_ptr(), a.numel(), _bf(a), _stream())
    return out


def cast_raw(x, dtype):
    """x converted to float32 or bfloat16 storage (round to nearest even)."""
    x = _chk(x, "cast x", True)
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    call("eeggan_cast", x.data_ptr(), out.data_ptr(), x.numel(), _bf(x), _bf(out), _stream())
    return out


def mix_batch_raw(real, fake, mix, out=None):
    real = _chk(real, "mix real")
    fake = _chk(fake, "mix fake")
    mix = _chk(mix.reshape(-1), "mix factors")
    if real.shape != fake.shape or mix.numel() != real.shape[0]:
        raise RuntimeError("mix_batch: shape mismatch")
    if out is None:
        out = torch.empty_like(real)
    elif out.shape != real.shape or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError("mix_batch: out does not match the inputs")
    call("eeggan_mix_batch", real.data_ptr(), fake.data_ptr(), mix.data_ptr(), out.data_ptr(), real.shape[0],
         real[0].numel(), _stream())
    return out


LOGNORMAL_STATE_WORDS = 256


def lognormal_fill_raw(out, base, seed, salt, state):
    """out[i] = base ** N(0, 1) drawn on the device in one launch; ``state``: LOGNORMAL_STATE_WORDS int32 words of
    device memory, zero before the first call, advanced by the kernel (one state per call site that may run beside
    others)."""
    if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
        raise RuntimeError("lognormal_fill: out must be a contiguous float32 device tensor")
    if not (state.is_cuda and state.dtype == torch.int32 and state.numel() >= LOGNORMAL_STATE_WORDS and state.is_contiguous()):
        raise RuntimeError(f"lognormal_fill: state must be {LOGNORMAL_STATE_WORDS} int32 words on the device")
    call("eeggan_lognormal_fill", out.data_ptr(), out.numel(), float(base), int(seed) & 0xFFFFFFFFFFFFFFFF, int(salt) & 0xFFFFFFFF,
         state.data_ptr(), _stream())
    return out


def pow_base_raw(n, base):
    n = _chk(n, "pow_base n")
    out = torch.empty_like(n)
    call("eeggan_pow_base", n.data_ptr(), out.data_ptr(), n.numel(), float(base), _stream())
    return out


# --------------------------------------------------------------------------------------- D-path conv
class _WGrad(Function):
    """Weight/bias gradient; a leaf of the second-order graph (never differentiated on the path)."""

    @staticmethod
    def forward(ctx, x, gy, K, in_scale, need_bias, mask_y, mask_drop, mask_slope):
        gw, gb = conv1d_wgrad_raw(x, gy, K, in_scale=in_scale, need_bias=need_bias, mask_y=mask_y,
                                  mask_drop=mask_drop, mask_slope=mask_slope)
        if gb is None:
            ctx.mark_non_differentiable()
            return gw, None
        return gw, gb

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("third-order differentiation through the conv weight gradient is not supported")


class _ConvT(Function):
    """gx = in_scale * conv_transpose(mask(gy), w): input gradient of the fused conv layer,
    itself differentiable w.r.t. gy and w (the double-backward of the gradient penalty)."""

    @staticmethod
    def forward(ctx, gy, w, mask_y, mask_drop, in_scale, slope, flip=None):
        gx = conv1d_raw(gy, w, None, None, in_scale=in_scale, transposed=True, mask_y=mask_y, mask_drop=mask_drop,
                        mask_slope=slope, flip=flip)
        ctx.save_for_backward(gy, w, mask_y, mask_drop)
        ctx.in_scale, ctx.slope = in_scale, slope
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, q):
        gy, w, mask_y, mask_drop = ctx.saved_tensors
        q = _chk(q, "ConvT cotangent", True)
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            if mask_y is not None:
                g_gy = conv1d_raw(q, w, None, mask_drop, in_scale=ctx.in_scale, act=ACT_MASK, act_param=ctx.slope,
                                  act_aux=mask_y)
            else:
                g_gy = conv1d_raw(q, w, None, None, in_scale=ctx.in_scale)
        if ctx.needs_input_grad[1] and not _input_grad_only:
            if _direct.usable(w):
                _direct.wgrad(w, None, q, gy, ctx.in_scale, mask_y, mask_drop, ctx.slope)
            else:
                g_w, _ = conv1d_wgrad_raw(q, gy, w.shape[2], in_scale=ctx.in_scale, need_bias=False, mask_y=mask_y,
                                          mask_drop=mask_drop, mask_slope=ctx.slope)
        return g_gy, g_w, None, None, None, None, None


class _Conv(Function):
    """y = drop * act(bias + conv(in_scale * x, w)), act in {none, leaky-relu}; twice differentiable."""

    @staticmethod
    def forward(ctx, x, w, bias, drop, in_scale, act, slope, flip=None, forced=None):
        ctx.flip = flip
        if act not in (ACT_NONE, ACT_LRELU):
            raise ValueError("_Conv supports linear and leaky-relu epilogues")
        if act == ACT_NONE and drop is not None:
            raise ValueError("a drop multiplier needs the leaky-relu epilogue")
        if forced is not None:
            # the LeakyReLU branch of every output is GIVEN (sign of ``forced``) instead of taken from the
            # pre-activation's own sign: the layer is then linear in x and w, forward and backward
            if act != ACT_LRELU:
                raise ValueError("a forced mask needs the leaky-relu epilogue")
            y = conv1d_raw(x, w, bias, drop, in_scale=in_scale, act=ACT_MASK, act_param=slope, act_aux=forced)
            ctx.save_for_backward(x, w, drop, forced, bias)
        else:
            y = conv1d_raw(x, w, bias, drop, in_scale=in_scale, act=act, act_param=slope)
            ctx.save_for_backward(x, w, drop, y if act == ACT_LRELU else None, bias)
        ctx.in_scale, ctx.slope, ctx.has_bias = in_scale, slope, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, drop, y, bias = ctx.saved_tensors
        # The layer output only selects the LeakyReLU branch below (zero derivative): detached, so that a
        # twice-differentiated pass does not tie its backward graph to this node's own backward -- the
        # engine would otherwise run it once more on gradients that are identically zero.
        y = None if y is None else y.detach()
        gy = _chk(gy, "conv grad", True)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _ConvT.apply(gy, w, y, drop, ctx.in_scale, ctx.slope, ctx.flip)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and not _input_grad_only:
            if _direct.usable(w) and (bias is None or _direct.usable(bias)):
                _direct.wgrad(w, bias, x, gy, ctx.in_scale, y, drop, ctx.slope)
            else:
                gw, gb = _WGrad.apply(x, gy, w.shape[2], ctx.in_scale, ctx.has_bias, y, drop, ctx.slope)
        return gx, gw, gb, None, None, None, None, None, None


def conv1d(x, w, bias, drop, in_scale, act, slope=0.0, flip=None, forced_mask=None):
    """Fused equalized conv layer of the discriminator path (twice differentiable).  ``forced_mask`` ([B, Cout, T],
    its sign picks each output's LeakyReLU branch): the branches are given instead of following the pre-activations
    (parity tests hold the kernels to a reference under the reference's own branches)."""
    return _Conv.apply(x, w, bias, drop, float(in_scale), int(act), float(slope), flip, forced_mask)


# --------------------------------------------------------------------------------------- depthwise conv
def _dw_raw(x, w, bias, pad, in_scale, flip):
    x, w = _chk(x, "depthwise x"), _chk(w, "depthwise w")
    B, Cc, T = x.shape
    K = w.shape[2]
    if tuple(w.shape) != (Cc, 1, K):
        raise RuntimeError(f"depthwise conv: weight {tuple(w.shape)} does not match {Cc} channels")
    y = torch.empty((B, Cc, T + 2 * pad - K + 1), device=x.device, dtype=torch.float32)
    call("eeggan_dwconv1d", x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), B, Cc, T, K, int(pad), float(in_scale),
         int(flip), _stream())
    return y


class _DwWGrad(Function):
    """Weight/bias gradient of the depthwise conv; a leaf of the second-order graph."""

    @staticmethod
    def forward(ctx, x, gy, K, pad, in_scale, flip, need_bias):
        x, gy = _chk(x, "depthwise x"), _chk(gy, "depthwise gy")
        B, Cc, T = x.shape
        gw = torch.empty((Cc, 1, K), device=x.device, dtype=torch.float32)
        gb = torch.empty((Cc,), device=x.device, dtype=torch.float32) if need_bias else None
        call("eeggan_dwconv1d_wgrad", x.data_ptr(), gy.data_ptr(), gw.data_ptr(), _p(gb), B, Cc, T, K, int(pad),
             float(in_scale), int(flip), _stream())
        return gw, gb

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("third-order differentiation through the depthwise weight gradient is not supported")


class _DwConv(Function):
    """y = bias + in_scale * depthwise_conv(x, w) (``flip``: with the taps reversed); linear in x and in w, so its
    backward is built from itself and the weight-gradient op: differentiable to any order in x, twice in w."""

    @staticmethod
    def forward(ctx, x, w, bias, pad, in_scale, flip):
        ctx.save_for_backward(x, w)
        ctx.cfg = (pad, in_scale, flip, bias is not None)
        return _dw_raw(x, w, bias, pad, in_scale, flip)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        pad, in_scale, flip, has_bias = ctx.cfg
        gy = _chk(gy, "depthwise grad")
        K = w.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _DwConv.apply(gy, w, None, K - 1 - pad, in_scale, not flip)
        if (ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])) and not _input_grad_only:
            gw, gb = _DwWGrad.apply(x, gy, K, pad, in_scale, flip, has_bias)
        return gx, gw, gb, None, None, None


def dwconv1d(x, w, bias, in_scale, pad):
    """Depthwise (groups = channels) equalized conv: bias + conv(in_scale * x, w) with zero padding ``pad``."""
    return _DwConv.apply(x, w, bias, int(pad), float(in_scale), False)


# --------------------------------------------------------------------------------------- D tail
class _MaskMul(Function):
    """out = g * drop * (y > 0 ? 1 : slope); linear in g, so its backward is itself."""

    @staticmethod
    def forward(ctx, g, y, drop, slope):
        ctx.save_for_backward(y, drop)
        ctx.slope = slope
        return lrelu_drop_bwd_raw(g, y, drop, slope)

    @staticmethod
    def backward(ctx, q):
        y, drop = ctx.saved_tensors
        return _MaskMul.apply(_chk(q, "maskmul cotangent", True), y, drop, ctx.slope), None, None, None


def _tail_fwd_raw(x, w, bias, drop, in_scale, act, slope):
    B, Cin, KT = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, Cout, 1), device=x.device, dtype=torch.float32)
    need = _tail_fwd_ws_bytes(B, Cin, Cout, KT)
    if need:  # matrix-core form: the reduction is split over workgroups, partial sums go through a workspace
        ws = torch.empty((need // 4,), device=x.device, dtype=torch.float32)
        call("eeggan_tail_fwd_ws", x.data_ptr(), w.data_ptr(), _p(bias), _p(drop), y.data_ptr(), B, Cin, Cout, KT,
             float(in_scale), int(act), float(slope), ws.data_ptr(), need, _stream())
    else:
        call("eeggan_tail_fwd", x.data_ptr(), w.data_ptr(), _p(bias), _p(drop), y.data_ptr(), B, Cin, Cout, KT,
             float(in_scale), int(act), float(slope), _stream())
    return y


@functools.lru_cache(maxsize=None)
def _tail_fwd_ws_bytes(B, Cin, Cout, KT):
    from ._lib import load
    return int(load().eeggan_tail_fwd_workspace(B, Cin, Cout, KT))


def _tail_dgrad_raw(gc, w, in_scale):
    Cout, Cin, KT = w.shape
    B = gc.shape[0]
    gx = torch.empty((B, Cin, KT), device=gc.device, dtype=torch.float32)
    call("eeggan_tail_dgrad", gc.data_ptr(), w.data_ptr(), gx.data_ptr(), B, Cin, Cout, KT, float(in_scale),
         _stream())
    return gx


def _tail_wgrad_raw(x, gc, in_scale, need_bias):
    B, Cin, KT = x.shape
    Cout = gc.shape[1]
    gw = torch.empty((Cout, Cin, KT), device=x.device, dtype=torch.float32)
    gb = torch.empty((Cout,), device=x.device, dtype=torch.float32) if need_bias else None
    call("eeggan_tail_wgrad", x.data_ptr(), gc.data_ptr(), gw.data_ptr(), _p(gb), B, Cin, Cout, KT, float(in_scale),
         _stream())
    return gw, gb


class _TailWGrad(Function):
    @staticmethod
    def forward(ctx, x, gc, in_scale, need_bias):
        gw, gb = _tail_wgrad_raw(_chk(x, "tail x"), _chk(gc, "tail gc"), in_scale, need_bias)
        return gw, gb

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("third-order differentiation through the tail weight gradient is not supported")


class _TailT(Function):
    """gx = in_scale * W^T gc for the full-length (k = T, pad = 0) layer; differentiable in gc and w."""

    @staticmethod
    def forward(ctx, gc, w, in_scale):
        gc = _chk(gc, "tail gc")
        ctx.save_for_backward(gc, w)
        ctx.in_scale = in_scale
        return _tail_dgrad_raw(gc, w, in_scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, q):
        gc, w = ctx.saved_tensors
        q = _chk(q, "tail cotangent")
        g_gc = g_w = None
        if ctx.needs_input_grad[0]:
            g_gc = _tail_fwd_raw(q, w, None, None, ctx.in_scale, ACT_NONE, 0.0)
        if ctx.needs_input_grad[1] and not _input_grad_only:
            g_w, _ = _tail_wgrad_raw(q, gc, ctx.in_scale, False)
        return g_gc, g_w, None


class _Tail(Function):
    """y[B, Cout, 1] = drop * lrelu(bias + sum_{ci,t} w[co,ci,t] * in_scale * x[b,ci,t]); twice differentiable."""

    @staticmethod
    def forward(ctx, x, w, bias, drop, in_scale, slope, forced=None):
        x = _chk(x, "tail x")
        if forced is not None:  # given LeakyReLU branches (see conv1d): the linear layer, then the mask and multipliers
            lin = _tail_fwd_raw(x, w, bias, None, in_scale, ACT_NONE, 0.0)
            y = lrelu_drop_bwd_raw(lin, forced, drop, slope)
            ctx.save_for_backward(x, w, drop, forced)
        else:
            y = _tail_fwd_raw(x, w, bias, drop, in_scale, ACT_LRELU, slope)
            ctx.save_for_backward(x, w, drop, y)
        ctx.in_scale, ctx.slope, ctx.has_bias = in_scale, slope, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, drop, y = ctx.saved_tensors
        gc = _MaskMul.apply(_chk(gy, "tail grad"), y.detach(), drop, ctx.slope)  # (as in _Conv)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _TailT.apply(gc, w, ctx.in_scale)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and not _input_grad_only:
            gw, gb = _TailWGrad.apply(x, gc, ctx.in_scale, ctx.has_bias)
        return gx, gw, gb, None, None, None, None


def tail_conv(x, w, bias, drop, in_scale, slope, forced_mask=None):
    if x.shape[2] != w.shape[2]:
        raise RuntimeError(f"tail conv: signal length {x.shape[2]} must equal the kernel size {w.shape[2]}")
    return _Tail.apply(x, w, bias, drop, float(in_scale), float(slope), forced_mask)


# --------------------------------------------------------------------------------------- G-path layers
class _GLayer(Function):
    """One generator layer, first-order differentiable:
         mode 'pn'  : y = drop * pixelnorm(lrelu(bias + conv(x)))      (kind 'same' or 'head')
         mode 'tanh': y = tanh(tanh_scale * (bias + conv(x)))           (to_rgb)
         mode 'linear': y = bias + conv(x)                               (kind 'same' or 'head'; before a batch norm)
    """

    @staticmethod
    def forward(ctx, x, w, bias, drop, in_scale, slope, mode, kind, tanh_scale, flip=None):
        ctx.flip = flip
        x = _chk(x, "G layer x", True)
        d = None
        if kind == "head":
            B, Cin = x.shape[0], x.shape[1]
            Cout, _, KT = w.shape
            h = torch.empty((B, Cout, KT), device=x.device, dtype=torch.float32)
            linear = mode == "linear"
            call("eeggan_head_fwd", x.data_ptr(), w.data_ptr(), _p(bias), None, h.data_ptr(), B, Cin, Cout, KT,
                 float(in_scale), ACT_NONE if linear else ACT_LRELU, float(slope), _stream())
            if linear:
                y = h
            elif mode == "pn":
                y, d = pixelnorm_drop_fwd_raw(h, drop)
            else:
                raise ValueError(f"head layer in mode {mode}")
        elif mode == "pn":
            fused = conv1d_pixelnorm_raw(x, w, bias, drop, in_scale=in_scale, slope=slope)
            if fused is not None:
                y, d = fused
            else:
                h = conv1d_raw(x, w, bias, None, in_scale=in_scale, act=ACT_LRELU, act_param=slope)
                y, d = pixelnorm_drop_fwd_raw(h, drop)
        elif mode == "tanh":
            y = conv1d_raw(x, w, bias, None, in_scale=in_scale, act=ACT_TANH, act_param=tanh_scale)
        elif mode == "linear":
            y = conv1d_raw(x, w, bias, None, in_scale=in_scale)
        else:
            raise ValueError(mode)
        if mode == "linear" and drop is not None:
            raise ValueError("a linear layer takes no drop multipliers")
        ctx.save_for_backward(x, w, drop, y, d, bias)
        ctx.cfg = (in_scale, slope, mode, kind, tanh_scale, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, drop, y, d, bias = ctx.saved_tensors
        in_scale, slope, mode, kind, tanh_scale, has_bias = ctx.cfg
        gy = _chk(gy, "G layer grad", True)
        if mode == "tanh":
            gc = torch.empty_like(y)
            call("eeggan_tanh_bwd", gy.data_ptr(), y.data_ptr(), gc.data_ptr(), y.numel(), float(tanh_scale),
                 _stream())
        elif mode == "linear":
            gc = gy
        else:
            gc = pixelnorm_drop_bwd_raw(gy, y, drop, d, slope)
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        if kind == "head":
            B, Cin = x.shape[0], x.shape[1]
            Cout, _, KT = w.shape
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                call("eeggan_head_dgrad", gc.data_ptr(), w.data_ptr(), gx.data_ptr(), B, Cin, Cout, KT,
                     float(in_scale), _stream())
            if need_w:
                gw = torch.empty_like(w)
                gb = torch.empty((Cout,), device=x.device, dtype=torch.float32) if has_bias else None
                call("eeggan_head_wgrad", x.data_ptr(), gc.data_ptr(), gw.data_ptr(), _p(gb), B, Cin, Cout, KT,
                     float(in_scale), _stream())
        else:
            if ctx.needs_input_grad[0]:
                gx = conv1d_raw(gc, w, None, None, in_scale=in_scale, transposed=True, flip=ctx.flip)
            if need_w:
                if _direct.usable(w) and (bias is None or _direct.usable(bias)):
                    _direct.wgrad(w, bias, x, gc, in_scale, None, None, 0.0)
                else:
                    gw, gb = conv1d_wgrad_raw(x, gc, w.shape[2], in_scale=in_scale, need_bias=has_bias)
        return gx, gw, gb, None, None, None, None, None, None, None


def g_layer(x, w, bias, drop, in_scale, slope, mode="pn", kind="same", tanh_scale=0.5, flip=None):
    return _GLayer.apply(x, w, bias, drop, float(in_scale), float(slope), mode, kind, float(tanh_scale), flip)


class _BNAct(Function):
    """Batch normalisation with a generator layer's epilogue (first-order differentiable):
         y = drop * lrelu(modulate(gamma * normalise(c) + beta))
    ``gamma`` / ``beta`` [C] or None; ``gmod`` / ``bmod`` [B, C] or None (per-sample scale and shift: o + gmod * o +
    bmod); the module's running statistics are updated in training mode and used in evaluation mode."""

    @staticmethod
    def forward(ctx, c, drop, gamma, beta, gmod, bmod, run_mean, run_var, slope, eps, momentum, train):
        c = _chk(c, "batch norm input")
        B, C, T = c.shape
        for t, shape, what in ((gamma, (C,), "gamma"), (beta, (C,), "beta"), (gmod, (B, C), "scale modulation"),
                               (bmod, (B, C), "shift modulation"), (drop, (B, C), "drop"), (run_mean, (C,), "running mean"),
                               (run_var, (C,), "running variance")):
            if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous()
                                  or t.device != c.device):
                raise RuntimeError(f"batch norm layer: {what} must be a contiguous float32 {shape} tensor on the device")
        mean = torch.empty((C,), device=c.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(c)
        call("eeggan_bn_lrelu_drop_fwd", c.data_ptr(), _p(gamma), _p(beta), _p(gmod), _p(bmod), _p(drop), _p(run_mean),
             _p(run_var), mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), B, C, T, float(eps), float(momentum), float(slope),
             int(train), _stream())
        ctx.save_for_backward(drop, y, c, mean, rstd, gamma, beta, gmod)
        ctx.cfg = (slope, train)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        drop, y, c, mean, rstd, gamma, beta, gmod = ctx.saved_tensors
        slope, train = ctx.cfg
        gy = _chk(gy, "batch norm grad")
        B, C, T = c.shape
        gc = torch.empty_like(c)
        g_gamma = torch.empty((C,), device=c.device, dtype=torch.float32)
        g_beta = torch.empty_like(g_gamma)
        g_gmod = g_bmod = None
        if gmod is not None:
            g_gmod = torch.empty((B, C), device=c.device, dtype=torch.float32)
            g_bmod = torch.empty_like(g_gmod)
        call("eeggan_bn_lrelu_drop_bwd", gy.data_ptr(), y.data_ptr(), c.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma),
             _p(beta), _p(gmod), _p(drop), gc.data_ptr(), g_gamma.data_ptr(), g_beta.data_ptr(), _p(g_gmod), _p(g_bmod), B, C, T,
             float(slope), int(train), _stream())
        return (gc, None, g_gamma if gamma is not None else None, g_beta if beta is not None else None, g_gmod, g_bmod,
                None, None, None, None, None, None)


def bn_act(c, drop, gamma, beta, gmod, bmod, run_mean, run_var, slope, eps, momentum, train):
    return _BNAct.apply(c, drop, gamma, beta, gmod, bmod, run_mean, run_var, float(slope), float(eps), float(momentum),
                        bool(train))


# ---- a batch norm that can be differentiated twice (the critic's): three multilinear pieces whose backwards are
# made of each other, so autograd derives every order from them
class _ChanSum(Function):
    """[B, C, T] -> [C]: scale * the sum over (batch, time)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _chk(x, "channel sum input")
        B, C, T = x.shape
        out = torch.empty((C,), device=x.device, dtype=torch.float32)
        call("eeggan_channel_sum", x.data_ptr(), out.data_ptr(), B, C, T, float(scale), _stream())
        ctx.cfg = (B, T, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, scale = ctx.cfg
        return _ChanAffine.apply(None, None, g * scale, (B, g.shape[0], T)), None


class _ChanAffine(Function):
    """x * s[ch] + t[ch] over [B, C, T] (x None: t[ch] spread over ``shape``; s / t None: 1 / 0)."""

    @staticmethod
    def forward(ctx, x, s, t, shape):
        B, C, T = shape
        ref = x if x is not None else t
        for v, what in ((s, "scale"), (t, "shift")):
            if v is not None and not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == C):
                raise RuntimeError(f"channel affine: {what} must be a contiguous float32 device vector of {C} elements")
        if x is not None:
            x = _chk(x, "channel affine input")
        y = torch.empty((B, C, T), device=ref.device, dtype=torch.float32)
        call("eeggan_channel_affine", _p(x), _p(s), _p(t), y.data_ptr(), B, C, T, _stream())
        ctx.save_for_backward(x, s)
        ctx.shape = shape
        ctx.has = (x is not None, s is not None, t is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        has_x, has_s, has_t = ctx.has
        g = g.contiguous()
        gx = gs = gt = None
        if has_x and ctx.needs_input_grad[0]:
            gx = _ChanAffine.apply(g, s, None, ctx.shape) if has_s else g
        if has_s and ctx.needs_input_grad[1]:
            gs = _ChanSum.apply(_Product.apply(g, x), 1.0)
        if has_t and ctx.needs_input_grad[2]:
            gt = _ChanSum.apply(g, 1.0)
        return gx, gs, gt, None


class _Product(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "product a"), _chk(b, "product b")
        if a.shape != b.shape:
            raise RuntimeError("product: shapes differ")
        out = torch.empty_like(a)
        call("eeggan_product", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        return (_Product.apply(g, b) if ctx.needs_input_grad[0] else None,
                _Product.apply(g, a) if ctx.needs_input_grad[1] else None)


class _PosSum(Function):
    """[B, C, T] -> [B, 1, T]: scale * the sum over the channels."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _chk(x, "position sum input")
        B, C, T = x.shape
        out = torch.empty((B, 1, T), device=x.device, dtype=torch.float32)
        call("eeggan_position_sum", x.data_ptr(), out.data_ptr(), B, C, T, float(scale), _stream())
        ctx.cfg = (C, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        C, scale = ctx.cfg
        return _PosScale.apply(None, g.contiguous() * scale, C), None


class _PosScale(Function):
    """x * s with s [B, 1, T] spread over the channels (x None: s spread over ``C`` channels)."""

    @staticmethod
    def forward(ctx, x, s, C):
        s = _chk(s, "position scale factor")
        B, _, T = s.shape
        if x is not None:
            x = _chk(x, "position scale input")
            if tuple(x.shape) != (B, C, T):
                raise RuntimeError(f"position scale: input {tuple(x.shape)} is not {(B, C, T)}")
        y = torch.empty((B, C, T), device=s.device, dtype=torch.float32)
        call("eeggan_position_scale", _p(x), s.data_ptr(), y.data_ptr(), B, C, T, _stream())
        ctx.save_for_backward(x, s)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = g.contiguous()
        gx = gs = None
        if x is not None and ctx.needs_input_grad[0]:
            gx = _PosScale.apply(g, s, ctx.C)
        if ctx.needs_input_grad[1]:
            gs = _PosSum.apply(_Product.apply(g, x), 1.0) if x is not None else _PosSum.apply(g, 1.0)
        return gx, gs, None


def pixel_norm_twice(a, eps=1e-8):
    """a / sqrt(mean over the channels of a^2 + eps) (PixelNorm, reference layers.py:10-14), differentiable to any
    order; the per-position vector in between is handled with torch's own operations."""
    C = a.shape[1]
    mean_sq = _PosSum.apply(_Product.apply(a, a), 1.0 / C)
    return _PosScale.apply(a, (mean_sq + eps).rsqrt(), C)


def batch_norm_twice(c, gamma, beta, run_mean, run_var, eps, momentum, train):
    """gamma * normalise(c) + beta with statistics over (batch, time) per channel (biased variance; evaluation mode:
    the running ones), differentiable to any order.  The per-channel vectors in between are handled with torch's
    own (tiny) operations.  Running statistics are updated like nn.BatchNorm1d's."""
    B, C, T = c.shape
    n = B * T
    if train:
        mean = _ChanSum.apply(c, 1.0 / n)
        d = _ChanAffine.apply(c, None, -mean, (B, C, T))
        var = _ChanSum.apply(_Product.apply(d, d), 1.0 / n)
        if run_mean is not None:
            with torch.no_grad():
                run_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
                run_var.mul_(1.0 - momentum).add_(var, alpha=momentum * (n / (n - 1.0) if n > 1 else 1.0))
        rstd = (var + eps).rsqrt()
        return _ChanAffine.apply(d, (gamma * rstd).contiguous(), beta, (B, C, T))
    rstd = (run_var + eps).rsqrt()
    scale = (gamma * rstd).contiguous()
    return _ChanAffine.apply(c, scale, (beta - run_mean * scale).contiguous(), (B, C, T))


def row_scale(x, n):
    """x[b, ch, t] * n[b, ch] (``n`` given as [B, C] or [B, C, 1]), differentiable in both (any order): the
    per-channel affine op on the tensor read as one sample of B * C channels."""
    B, C, T = x.shape
    if n.numel() != B * C:
        raise RuntimeError(f"row_scale: {tuple(n.shape)} multipliers for a tensor {tuple(x.shape)}")
    flat = _ChanAffine.apply(x.reshape(1, B * C, T), n.reshape(B * C).contiguous(), None, (1, B * C, T))
    return flat.reshape(B, C, T)


def lrelu_drop(o, drop, slope):
    """drop[b, ch] * lrelu_slope(o), differentiable to any order (piecewise linear: the mask is a constant)."""
    return _MaskMul.apply(o, o.detach(), drop, float(slope))


class _SpectralNorm(Function):
    """w_orig / sigma with sigma = u^T W v of the weight as a matrix [output channels, rest], after one power iteration
    on the module's vectors ``u`` and ``v`` (in place) when ``iterate`` (torch.nn.utils.spectral_norm's arithmetic; the
    vectors are constants of the backward).  First-order differentiable: what is differentiated twice under the
    gradient penalty is the conv, with respect to the normalised weight."""

    @staticmethod
    def forward(ctx, w_orig, u, v, eps, iterate):
        w_orig = _chk(w_orig, "spectral norm weight")
        rows = w_orig.shape[0]
        cols = w_orig.numel() // rows
        for t, n, what in ((u, rows, "u"), (v, cols, "v")):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                raise RuntimeError(f"spectral norm: {what} must be a contiguous float32 device vector of {n} elements")
        w_sn = torch.empty_like(w_orig)
        u_used, v_used = torch.empty_like(u), torch.empty_like(v)
        sigma = torch.empty((1,), device=w_orig.device, dtype=torch.float32)
        scratch = torch.empty((rows,), device=w_orig.device, dtype=torch.float32)
        call("eeggan_spectral_norm_fwd", w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), w_sn.data_ptr(), u_used.data_ptr(),
             v_used.data_ptr(), sigma.data_ptr(), scratch.data_ptr(), rows, cols, float(eps), int(iterate), _stream())
        ctx.save_for_backward(w_sn, u_used, v_used, sigma)
        ctx.mark_non_differentiable(sigma)
        return w_sn, sigma

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _):
        w_sn, u_used, v_used, sigma = ctx.saved_tensors
        g = _chk(g, "spectral norm grad")
        rows = w_sn.shape[0]
        cols = w_sn.numel() // rows
        out = torch.empty_like(w_sn)
        scratch = torch.empty((1,), device=w_sn.device, dtype=torch.float32)
        call("eeggan_spectral_norm_bwd", g.data_ptr(), w_sn.data_ptr(), u_used.data_ptr(), v_used.data_ptr(), sigma.data_ptr(),
             scratch.data_ptr(), out.data_ptr(), rows, cols, _stream())
        return out, None, None, None, None


def spectral_norm(w_orig, u, v, eps=1e-12, iterate=True):
    """(normalised weight, sigma [1])."""
    return _SpectralNorm.apply(w_orig, u, v, float(eps), bool(iterate))


class _ScaledTanh(Function):
    """y = tanh(scale * x) on its own (on the training path the to_rgb conv applies it as its epilogue)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _chk(x, "scaled tanh x")
        y = torch.empty_like(x)
        call("eeggan_scaled_tanh", x.data_ptr(), y.data_ptr(), x.numel(), float(scale), _stream())
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _chk(gy, "scaled tanh grad")
        gc = torch.empty_like(y)
        call("eeggan_tanh_bwd", gy.data_ptr(), y.data_ptr(), gc.data_ptr(), y.numel(), float(ctx.scale), _stream())
        return gc, None


def scaled_tanh(x, scale):
    return _ScaledTanh.apply(x, float(scale))


class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, h):
        y, d = pixelnorm_drop_fwd_raw(h, None)
        ctx.save_for_backward(y, d)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        y, d = ctx.saved_tensors
        return pixelnorm_drop_bwd_raw(gy, y, None, d, 1.0)


def pixel_norm(h):
    """h * rsqrt(mean_c h^2 + 1e-8) for [B, C] or [B, C, T] tensors."""
    if h.dim() == 2:
        return _PixelNorm.apply(h.unsqueeze(2)).squeeze(2)
    return _PixelNorm.apply(h)


# --------------------------------------------------------------------------------------- linear helpers
USE_RESAMPLE_QUADS = os.environ.get("EEGGAN_RESAMPLE_QUADS", "1") != "0"  # (comparison switch)
USE_RESAMPLE_AFFINE = os.environ.get("EEGGAN_RESAMPLE_AFFINE", "1") != "0"  # (comparison switch)


def _gather_raw(x, table, out=None):
    x = _chk(x, "resample x", True)
    idx, coef, t_in, t_out, taps, quads = table
    if x.shape[-1] != t_in:
        raise RuntimeError(f"resample: input length {x.shape[-1]} does not match the table ({t_in})")
    rows = x.numel() // t_in
    if out is None:
        # the resampler is where the float32 tensors of the 32-sample stage meet the wide activations' storage type
        dt = _act_dtype(x.shape[-2] if x.dim() >= 2 else 1, t_out)
        out = torch.empty(x.shape[:-1] + (t_out,), device=x.device, dtype=dt)
    elif tuple(out.shape) != tuple(x.shape[:-1]) + (t_out,) or not out.is_contiguous():
        raise RuntimeError("resample: out does not match the result shape")
    if quads is not None and USE_RESAMPLE_QUADS and (x.data_ptr() | out.data_ptr()) % 16 == 0:
        qbase, qcoef, nv, affine = quads  # 16-byte accesses, four outputs per thread
        if affine is not None and USE_RESAMPLE_AFFINE:  # the quad starts in closed form: no look-up before the loads
            call("eeggan_resample_quads_affine_t", x.data_ptr(), affine[0], affine[1], affine[2], qcoef.data_ptr(),
                 out.data_ptr(), rows, t_in, t_out, nv, _bf(x), _bf(out), _stream())
        else:
            call("eeggan_resample_quads_t", x.data_ptr(), qbase.data_ptr(), qcoef.data_ptr(), out.data_ptr(), rows, t_in,
                 t_out, nv, _bf(x), _bf(out), _stream())
    else:
        call("eeggan_resample_t", x.data_ptr(), idx.data_ptr(), coef.data_ptr(), out.data_ptr(), rows, t_in, t_out,
             taps, _bf(x), _bf(out), _stream())
    return out


class _Gather(Function):
    """out[r, m] = sum_k coef[m, k] * x[r, idx[m, k]]; the backward is the same op with the transposed table."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.tables = (fwd, bwd)
        return _gather_raw(x, fwd)

    @staticmethod
    def backward(ctx, g):
        fwd, bwd = ctx.tables
        return _Gather.apply(g, bwd, fwd), None, None


class _Scale(Function):
    @staticmethod
    def forward(ctx, x, c):
        ctx.c = c
        return axpby_raw(x, c)

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.c), None


class _Axpby(Function):
    @staticmethod
    def forward(ctx, a, b, ca, cb):
        ctx.c = (ca, cb)
        return axpby_raw(a, ca, b, cb)

    @staticmethod
    def backward(ctx, g):
        ca, cb = ctx.c
        ga = _Scale.apply(g, ca) if ctx.needs_input_grad[0] else None
        gb = _Scale.apply(g, cb) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def axpby(a, ca, b, cb):
    """ca*a + cb*b (fade-in blends)."""
    return _Axpby.apply(a, b, float(ca), float(cb))


class _ChannelWindow(Function):
    """out [B, Cd, T] = (base or 0) + src[:, s0:s0+n] placed at channels [d0, d0+n) (src with one time step is
    repeated over time).  Linear in both operands and its own transpose with the roles of the windows swapped, so it
    is differentiable to any order."""

    @staticmethod
    def forward(ctx, src, base, s0, d0, n, Cd, T):
        src = _chk(src, "channel window src")
        B, Cs, Ts = src.shape
        if base is not None:
            base = _chk(base, "channel window base")
            if tuple(base.shape) != (B, Cd, T):
                raise RuntimeError(f"channel window: base {tuple(base.shape)} is not {(B, Cd, T)}")
        out = torch.empty((B, Cd, T), device=src.device, dtype=torch.float32)
        call("eeggan_channel_window", src.data_ptr(), _p(base), out.data_ptr(), B, Cs, Cd, T, Ts, int(s0), int(d0), int(n),
             _stream())
        ctx.cfg = (s0, d0, n, Cs, Ts, base is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        s0, d0, n, Cs, Ts, has_base = ctx.cfg
        g_src = None
        if ctx.needs_input_grad[0]:
            if Ts != g.shape[2]:
                raise NotImplementedError("gradient of a sample repeated over time (the latent of a deep first block)")
            g_src = _ChannelWindow.apply(g, None, d0, s0, n, Cs, Ts)
        return g_src, (g if has_base and ctx.needs_input_grad[1] else None), None, None, None, None, None


def channel_window(src, base, s0, d0, n, channels=None, length=None):
    """``base + place(src[:, s0:s0+n] at channels d0..)`` (``base`` None: into zeros of ``channels`` x ``length``)."""
    Cd = base.shape[1] if base is not None else int(channels)
    T = base.shape[2] if base is not None else int(length if length is not None else src.shape[2])
    return _ChannelWindow.apply(src, base, int(s0), int(d0), int(n), Cd, T)


def _blend_raw(a, b, alpha, mode):
    a = _chk(a, "blend a", True)
    if b is not None:
        b = _chk(b, "blend b", True)
        if b.shape != a.shape:
            raise RuntimeError(f"blend: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        _same_dtype("blend", a, b)
    if alpha.numel() != 1 or alpha.dtype != torch.float32 or alpha.device != a.device:
        raise RuntimeError("blend: alpha must be a one-element float32 tensor on the operands' device")
    out = torch.empty_like(a)
    call("eeggan_blend_t", a.data_ptr(), _p(b), alpha.data_ptr(), int(mode), out.data_ptr(), a.numel(), _bf(a),
         _stream())
    return out


class _ScaleDev(Function):
    """alpha*x (mode 1) or (1-alpha)*x (mode 2) with alpha in device memory; linear, so its own backward."""

    @staticmethod
    def forward(ctx, x, alpha, mode):
        ctx.alpha, ctx.mode = alpha, mode
        return _blend_raw(x, None, alpha, mode)

    @staticmethod
    def backward(ctx, g):
        return _ScaleDev.apply(g, ctx.alpha, ctx.mode), None, None


class _BlendDev(Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        return _blend_raw(a, b, alpha, 0)

    @staticmethod
    def backward(ctx, g):
        ga = _ScaleDev.apply(g, ctx.alpha, 1) if ctx.needs_input_grad[0] else None
        gb = _ScaleDev.apply(g, ctx.alpha, 2) if ctx.needs_input_grad[1] else None
        return ga, gb, None


def blend(a, b, alpha):
    """alpha*a + (1-alpha)*b.  ``alpha`` is a Python float or a one-element device tensor; the latter
    is read by the kernel at run time, so a captured graph follows a fade-in factor that changes."""
    if torch.is_tensor(alpha):
        return _BlendDev.apply(a, b, alpha)
    # same float32 arithmetic as the device-alpha kernel (1 - alpha rounded in float32), so that both
    # routes give bit-identical blends
    al = np.float32(alpha)
    return _Axpby.apply(a, b, float(al), float(np.float32(1.0) - al))


# --------------------------------------------------------------------------------------- self attention pieces
def _bmm_raw(a, b, ta, tb, alpha=1.0):
    a, b = _chk(a, "bmm a"), _chk(b, "bmm b")
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0]:
        raise RuntimeError(f"bmm: operands {tuple(a.shape)} and {tuple(b.shape)}")
    M, K = (a.shape[2], a.shape[1]) if ta else (a.shape[1], a.shape[2])
    Kb, N = (b.shape[2], b.shape[1]) if tb else (b.shape[1], b.shape[2])
    if K != Kb:
        raise RuntimeError(f"bmm: reduction lengths differ ({K} vs {Kb})")
    out = torch.empty((a.shape[0], M, N), device=a.device, dtype=torch.float32)
    call("eeggan_bmm", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], M, N, K, int(ta), int(tb), float(alpha),
         _stream())
    return out


class _Bmm(Function):
    """alpha * op(a) op(b) per batch entry, op = transpose of the stored matrix where the flag is set.  Its gradients
    are products of the same kind, so it is differentiable to any order."""

    @staticmethod
    def forward(ctx, a, b, ta, tb, alpha):
        ctx.save_for_backward(a, b)
        ctx.cfg = (ta, tb, alpha)
        return _bmm_raw(a, b, ta, tb, alpha)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.cfg
        g = _chk(g, "bmm cotangent")
        ga = gb = None
        if ctx.needs_input_grad[0]:  # d op(a) = g op(b)^T
            ga = _Bmm.apply(b, g, tb, True, alpha) if ta else _Bmm.apply(g, b, False, not tb, alpha)
        if ctx.needs_input_grad[1]:  # d op(b) = op(a)^T g
            gb = _Bmm.apply(g, a, True, ta, alpha) if tb else _Bmm.apply(a, g, not ta, False, alpha)
        return ga, gb, None, None, None


def bmm(a, b, trans_a=False, trans_b=False, alpha=1.0):
    return _Bmm.apply(a, b, bool(trans_a), bool(trans_b), float(alpha))


class _SoftmaxBwd(Function):
    """gx = y (gy - sum_m gy y), differentiable once more in both operands."""

    @staticmethod
    def forward(ctx, gy, y):
        gy, y = _chk(gy, "softmax cotangent"), _chk(y, "softmax output")
        ctx.save_for_backward(gy, y)
        gx = torch.empty_like(y)
        call("eeggan_softmax_dim1_bwd", gy.data_ptr(), y.data_ptr(), gx.data_ptr(), *y.shape, _stream())
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, q):
        gy, y = ctx.saved_tensors
        q = _chk(q, "softmax second-order cotangent")
        d_gy = d_y = None
        if ctx.needs_input_grad[0]:
            d_gy = torch.empty_like(y)
            call("eeggan_softmax_dim1_bwd", q.data_ptr(), y.data_ptr(), d_gy.data_ptr(), *y.shape, _stream())
        if ctx.needs_input_grad[1]:
            d_y = torch.empty_like(y)
            call("eeggan_softmax_dim1_bwd_dy", q.data_ptr(), gy.data_ptr(), y.data_ptr(), d_y.data_ptr(), *y.shape,
                 _stream())
        return d_gy, d_y


class _Softmax(Function):
    @staticmethod
    def forward(ctx, s):
        s = _chk(s, "softmax input")
        if s.dim() != 3:
            raise RuntimeError("softmax over dim 1 takes a [B, M, N] tensor")
        p = torch.empty_like(s)
        call("eeggan_softmax_dim1_fwd", s.data_ptr(), p.data_ptr(), *s.shape, _stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        p, = ctx.saved_tensors
        return _SoftmaxBwd.apply(g, p)


def softmax_dim1(s):
    """softmax over dim 1 of a [B, M, N] tensor (twice differentiable)."""
    return _Softmax.apply(s)


class _PoolSelect(Function):
    """Max pooling with the positions of the maxima fixed (``idx``): the gather (scatter = False) and the scatter are
    each other's transposes, so both are differentiable to any order."""

    @staticmethod
    def forward(ctx, src, idx, t_in, pool, scatter):
        src = _chk(src, "maxpool select src")
        ctx.cfg = (idx, t_in, pool, scatter)
        rows = idx.numel() // (t_in // pool)
        t_out = t_in if scatter else t_in // pool
        if src.numel() != rows * (t_in // pool if scatter else t_in):
            raise RuntimeError("maxpool select: source does not match the pooled shape")
        dst = torch.empty(src.shape[:-1] + (t_out,), device=src.device, dtype=torch.float32)
        call("eeggan_maxpool_select", src.data_ptr(), idx.data_ptr(), dst.data_ptr(), rows, t_in, pool, int(scatter), _stream())
        return dst

    @staticmethod
    def backward(ctx, g):
        idx, t_in, pool, scatter = ctx.cfg
        return _PoolSelect.apply(g, idx, t_in, pool, not scatter), None, None, None, None


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, pool):
        x = _chk(x, "maxpool input")
        t_in = x.shape[-1]
        if t_in % pool:
            raise NotImplementedError("max pooling of a length that is not a multiple of the window")
        y = torch.empty(x.shape[:-1] + (t_in // pool,), device=x.device, dtype=torch.float32)
        idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
        call("eeggan_maxpool_fwd", x.data_ptr(), y.data_ptr(), idx.data_ptr(), x.numel() // t_in, t_in, pool, _stream())
        ctx.cfg = (idx, t_in, pool)
        return y

    @staticmethod
    def backward(ctx, g):
        idx, t_in, pool = ctx.cfg
        return _PoolSelect.apply(g, idx, t_in, pool, True), None


def maxpool(x, pool):
    """nn.MaxPool1d(pool) along the last axis (twice differentiable: linear once the maxima are located)."""
    return _MaxPool.apply(x, int(pool))


class _Dot(Function):
    """sum(a * b) as a 0-dim tensor."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "dot a"), _chk(b, "dot b")
        if a.shape != b.shape:
            raise RuntimeError("dot: shapes differ")
        ctx.save_for_backward(a, b)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        nbytes = _lib.load().eeggan_dot_workspace()
        ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
        call("eeggan_dot", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), ws.data_ptr(), nbytes, _stream())
        return out

    @staticmethod
    def backward(ctx, q):
        a, b = ctx.saved_tensors
        q = q.contiguous()
        ga = _MulParam.apply(b, q) if ctx.needs_input_grad[0] else None
        gb = _MulParam.apply(a, q) if ctx.needs_input_grad[1] else None
        return ga, gb


class _MulParam(Function):
    """s * f with s a one-element tensor in device memory (a scalar parameter): read by the kernel at run time, so a
    captured graph follows the parameter as the optimizer moves it.  Differentiable to any order in both operands."""

    @staticmethod
    def forward(ctx, f, s):
        ctx.save_for_backward(f, s)
        return _blend_raw(f, None, s, 1)

    @staticmethod
    def backward(ctx, g):
        f, s = ctx.saved_tensors
        g = _chk(g, "scalar-multiple cotangent")
        gf = _MulParam.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _Dot.apply(g, f).reshape(s.shape) if ctx.needs_input_grad[1] else None
        return gf, gs


def mul_param(f, s):
    return _MulParam.apply(f, s)


# --------------------------------------------------------------------------------------- minibatch stddev
# ``segments``: the batch is a concatenation of that many independent minibatches (e.g. reals and
# fakes stacked into one critic pass); the group statistics never mix samples of two segments, so
# the kernels run once per segment on the corresponding slice (the batch axis is outermost).
def _mbstd_ws(bs, Cc, T, group, device):
    nbytes = _lib.load().eeggan_mbstd_workspace(bs, Cc, T, group)
    return torch.empty(max(nbytes, 4), device=device, dtype=torch.uint8)


def _seg(t, i, bs):
    """Device address of the i-th batch segment of the contiguous tensor t."""
    return t.data_ptr() + i * bs * (t.numel() // t.shape[0]) * 4


def _windows(T, stride):
    """(start, length) of the temporal windows the layer takes its statistics over (reference layers.py:81-83)."""
    return [(t0, min(stride, T - t0)) for t0 in range(0, T, stride)]


class _MbStdBwd(Function):
    @staticmethod
    def forward(ctx, x, gout, group, segments, stride):
        gout = _chk(gout, "mbstd gout")
        B, Cc, T = x.shape
        bs = B // segments
        gx = torch.empty_like(x)
        for i in range(segments):
            for t0, tw in _windows(T, stride):
                call("eeggan_mbstd_bwd_window", _seg(x, i, bs), _seg(gout, i, bs), _seg(gx, i, bs), bs, Cc, T, t0, tw,
                     group, _stream())
        ctx.save_for_backward(x, gout)
        ctx.group, ctx.segments, ctx.stride = group, segments, stride
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, q):
        x, gout = ctx.saved_tensors
        q = _chk(q, "mbstd cotangent")
        B, Cc, T = x.shape
        bs = B // ctx.segments
        ggout = torch.empty_like(gout)
        gx2 = torch.empty_like(x)
        for i in range(ctx.segments):
            for t0, tw in _windows(T, ctx.stride):
                ws = _mbstd_ws(bs, Cc, T, ctx.group, x.device)
                call("eeggan_mbstd_bwd_bwd_window", _seg(x, i, bs), _seg(gout, i, bs), _seg(q, i, bs), _seg(ggout, i, bs),
                     _seg(gx2, i, bs), ws.data_ptr(), bs, Cc, T, t0, tw, ctx.group, _stream())
        return gx2, ggout, None, None, None


class _MbStd(Function):
    @staticmethod
    def forward(ctx, x, group, segments, stride):
        x = _chk(x, "mbstd x")
        B, Cc, T = x.shape
        if B % segments:
            raise ValueError(f"minibatch_stddev: batch {B} is not {segments} equal segments")
        bs = B // segments
        out = torch.empty((B, Cc + 1, T), device=x.device, dtype=torch.float32)
        for i in range(segments):
            for t0, tw in _windows(T, stride):
                ws = _mbstd_ws(bs, Cc, T, group, x.device)
                call("eeggan_mbstd_fwd_window", _seg(x, i, bs), _seg(out, i, bs), ws.data_ptr(), bs, Cc, T, t0, tw, group,
                     _stream())
        ctx.save_for_backward(x)
        ctx.group, ctx.segments, ctx.stride = group, segments, stride
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        return _MbStdBwd.apply(x, gout, ctx.group, ctx.segments, ctx.stride), None, None, None


def minibatch_stddev(x, group, segments=1, stride=None):
    """MinibatchStddev (reference layers.py:67-90): ``stride`` is the length of a temporal window (default: the
    whole signal, one window)."""
    stride = int(x.shape[2] if stride is None else stride)
    if stride < 1:
        raise ValueError("minibatch_stddev: the temporal window must hold at least one step")
    return _MbStd.apply(x, int(group), int(segments), stride)


# --------------------------------------------------------------------------------------- losses
class _WganGpLoss(Function):
    @staticmethod
    def forward(ctx, d_real, d_fake, g, drift, lam, target):
        d_real, d_fake, g = _chk(d_real, "d_real"), _chk(d_fake, "d_fake"), _chk(g, "penalty gradient")
        B = d_real.numel()
        per = g.numel() // B
        norms = torch.empty((B,), device=g.device, dtype=torch.float32)
        loss = torch.empty((), device=g.device, dtype=torch.float32)
        call("eeggan_wgan_gp_loss_fwd", d_real.data_ptr(), d_fake.data_ptr(), g.data_ptr(), norms.data_ptr(),
             loss.data_ptr(), B, per, drift, lam, target, _stream())
        ctx.save_for_backward(d_real, g, norms)
        ctx.cfg = (drift, lam, target)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        d_real, g, norms = ctx.saved_tensors
        drift, lam, target = ctx.cfg
        gl = _chk(gl, "loss grad")
        B = d_real.numel()
        per = g.numel() // B
        g_r, g_f, g_g = torch.empty_like(d_real), torch.empty_like(d_real), torch.empty_like(g)
        call("eeggan_wgan_gp_loss_bwd", gl.data_ptr(), d_real.data_ptr(), g.data_ptr(), norms.data_ptr(),
             g_r.data_ptr(), g_f.data_ptr(), g_g.data_ptr(), B, per, drift, lam, target, _stream())
        return g_r, g_f, g_g, None, None, None


class _WganGpLossStacked(Function):
    """The same loss on the scores of a stacked critic pass ([reals; fakes] in one tensor): one gradient
    tensor goes back instead of two slices (whose backward costs autograd two zero fills, two copies and an add)."""

    @staticmethod
    def forward(ctx, scores, g, drift, lam, target):
        scores, g = _chk(scores, "scores"), _chk(g, "penalty gradient")
        B = scores.numel() // 2
        if scores.numel() != 2 * B or g.shape[0] != B:
            raise RuntimeError("wgan_gp_d_loss_stacked: scores must hold the reals' and the fakes' scores")
        per = g.numel() // B
        norms = torch.empty((B,), device=g.device, dtype=torch.float32)
        loss = torch.empty((), device=g.device, dtype=torch.float32)
        call("eeggan_wgan_gp_loss_fwd", scores.data_ptr(), scores.data_ptr() + 4 * B, g.data_ptr(), norms.data_ptr(),
             loss.data_ptr(), B, per, drift, lam, target, _stream())
        ctx.save_for_backward(scores, g, norms)
        ctx.cfg = (drift, lam, target)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        scores, g, norms = ctx.saved_tensors
        drift, lam, target = ctx.cfg
        gl = _chk(gl, "loss grad")
        B = scores.numel() // 2
        per = g.numel() // B
        g_s, g_g = torch.empty_like(scores), torch.empty_like(g)
        call("eeggan_wgan_gp_loss_bwd", gl.data_ptr(), scores.data_ptr(), g.data_ptr(), norms.data_ptr(),
             g_s.data_ptr(), g_s.data_ptr() + 4 * B, g_g.data_ptr(), B, per, drift, lam, target, _stream())
        return g_s, g_g, None, None, None


SCORE_LOSS_KINDS = {"hinge": 1, "rsgan": 2, "rasgan": 3, "rahinge": 4, "wgan_theirs": 5, "gain": 6}


class _ScoreLoss(Function):
    """A loss term of two score vectors (``eeggan_score_loss``): value and both gradients come from one launch,
    the backward only scales the stored gradients by the upstream gradient."""

    @staticmethod
    def forward(ctx, a, b, kind, eps):
        a, b = _chk(a, "scores a"), _chk(b, "scores b")
        if a.numel() != b.numel():
            raise RuntimeError(f"score loss: {a.numel()} and {b.numel()} scores")
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        call("eeggan_score_loss", kind, a.data_ptr(), b.data_ptr(), a.numel(), eps, loss.data_ptr(), ga.data_ptr(),
             gb.data_ptr(), _stream())
        ctx.save_for_backward(ga, gb)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        ga, gb = ctx.saved_tensors
        gl = _chk(gl, "loss grad")
        out = []
        for need, g in zip(ctx.needs_input_grad[:2], (ga, gb)):
            if not need:
                out.append(None)
                continue
            o = torch.empty_like(g)
            call("eeggan_scale_by_scalar", gl.data_ptr(), g.data_ptr(), o.data_ptr(), g.numel(), _stream())
            out.append(o)
        return out[0], out[1], None, None


def score_loss(kind: str, a, b, eps: float = 0.0):
    """The reference's non-default loss terms on two score vectors (see ``eeggan_score_loss``): critic losses with
    a = D(real), b = D(fake); the generator's relativistic losses with a = D(fake), b = D(real)."""
    if kind not in SCORE_LOSS_KINDS:
        raise ValueError(f"score loss {kind!r}: expected one of {sorted(SCORE_LOSS_KINDS)}")
    return _ScoreLoss.apply(a, b, SCORE_LOSS_KINDS[kind], float(eps))


class _GradPenalty(Function):
    @staticmethod
    def forward(ctx, g, lam, target, one_sided):
        g = _chk(g, "penalty gradient")
        B = g.shape[0]
        per = g.numel() // B
        norms = torch.empty((B,), device=g.device, dtype=torch.float32)
        loss = torch.empty((), device=g.device, dtype=torch.float32)
        call("eeggan_grad_penalty_fwd", g.data_ptr(), norms.data_ptr(), loss.data_ptr(), B, per, lam, target,
             int(one_sided), _stream())
        ctx.save_for_backward(g, norms)
        ctx.cfg = (lam, target, int(one_sided))
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        g, norms = ctx.saved_tensors
        lam, target, one_sided = ctx.cfg
        gl = _chk(gl, "loss grad")
        B = g.shape[0]
        g_g = torch.empty_like(g)
        call("eeggan_grad_penalty_bwd", gl.data_ptr(), g.data_ptr(), norms.data_ptr(), g_g.data_ptr(), B,
             g.numel() // B, lam, target, one_sided, _stream())
        return g_g, None, None, None


def grad_penalty(g, lam, target, one_sided=False):
    """lam / target^2 * mean((|g_b| - target)^2) over the rows of g [B, per]; ``one_sided``: relu(|g_b| - target)."""
    return _GradPenalty.apply(g, float(lam), float(target), bool(one_sided))


def ema_update(averages, params, old_weight: float):
    """averages[k] = old_weight * averages[k] + (1 - old_weight) * params[k] for two lists of tensors, in one launch
    per 64 tensors (the running average of the generator's weights, reference plugins.py:305-307)."""
    from . import optim
    rows = []
    for avg, p in zip(averages, params):
        if not avg.is_contiguous():
            raise RuntimeError("ema_update: the averages are updated in place and must be contiguous")
        avg, p = _chk(avg, "ema average"), _chk(p.detach(), "ema parameter")
        if avg.numel() != p.numel():
            raise RuntimeError("ema_update: an average and its parameter differ in size")
        rows.append((avg.data_ptr(), p.data_ptr(), 0, 0, 0, avg.numel()))
    if rows:
        call("eeggan_multi_axpby", optim.build_table(rows), len(rows), float(old_weight), 1.0 - float(old_weight),
             _stream())


def _pair_table(dsts, srcs, what):
    from . import optim
    rows = []
    for dst, src in zip(dsts, srcs):
        dst, src = _chk(dst, f"{what} destination"), _chk(src.detach(), f"{what} source")
        if dst.numel() != src.numel():
            raise RuntimeError(f"{what}: a destination and its source differ in size")
        rows.append((dst.data_ptr(), src.data_ptr(), 0, 0, 0, dst.numel()))
    return optim.build_table(rows), len(rows)


def multi_copy_(dsts, srcs):
    """dsts[k] <- srcs[k] for two lists of contiguous float32 tensors of equal sizes, in ONE launch (per 64 tensors)
    instead of one copy kernel per tensor."""
    table, n = _pair_table(dsts, srcs, "multi_copy")
    if n:
        call("eeggan_multi_copy", table, n, _stream())


def multi_add_(dsts, srcs):
    """dsts[k] += srcs[k] for two lists of tensors, in one launch."""
    table, n = _pair_table(dsts, srcs, "multi_add")
    if n:
        call("eeggan_multi_axpby", table, n, 1.0, 1.0, _stream())


def wgan_gp_d_loss_stacked(scores, g, drift, lam, target):
    """``wgan_gp_d_loss(scores[:B], scores[B:], ...)`` without slicing the scores."""
    return _WganGpLossStacked.apply(scores, g, float(drift), float(lam), float(target))


def wgan_gp_d_loss(d_real, d_fake, g, drift, lam, target):
    """mean(d_fake) - mean(d_real) + drift*mean(d_real^2) + lam/target^2 * mean((|g_b| - target)^2)."""
    return _WganGpLoss.apply(d_real, d_fake, g, float(drift), float(lam), float(target))


class _FeatureMatch(Function):
    """mean_f (mean_b real[b, f] - mean_b fake[b, f])^2, differentiable (once) in ``fake``."""

    @staticmethod
    def forward(ctx, real, fake):
        real, fake = _chk(real, "features of the reals"), _chk(fake, "features of the fakes")
        B = real.shape[0]
        F = real.numel() // B
        if fake.shape != real.shape:
            raise RuntimeError(f"feature matching: {tuple(real.shape)} and {tuple(fake.shape)} features")
        diff = torch.empty((F,), device=real.device, dtype=torch.float32)
        loss = torch.empty((), device=real.device, dtype=torch.float32)
        call("eeggan_feature_match_fwd", real.data_ptr(), fake.data_ptr(), diff.data_ptr(), loss.data_ptr(), B, F,
             _stream())
        ctx.save_for_backward(diff)
        ctx.shape = fake.shape
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        (diff,) = ctx.saved_tensors
        gl = _chk(gl, "loss grad")
        g_fake = torch.empty(ctx.shape, device=gl.device, dtype=torch.float32)
        call("eeggan_feature_match_bwd", gl.data_ptr(), diff.data_ptr(), g_fake.data_ptr(), ctx.shape[0],
             g_fake.numel() // ctx.shape[0], _stream())
        return None, g_fake


def feature_match(real_features, fake_features):
    """The feature-matching term of the generator loss (reference losses.py:62-69)."""
    return _FeatureMatch.apply(real_features.detach(), fake_features)


class _MeanScale(Function):
    @staticmethod
    def forward(ctx, v, scale):
        v = _chk(v, "mean input")
        out = torch.empty((), device=v.device, dtype=torch.float32)
        call("eeggan_mean_scale", v.data_ptr(), out.data_ptr(), v.numel(), scale, _stream())
        ctx.shape, ctx.scale = v.shape, scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        gl = _chk(gl, "loss grad")
        n = 1
        for s in ctx.shape:
            n *= s
        out = torch.empty(ctx.shape, device=gl.device, dtype=torch.float32)
        call("eeggan_mean_scale_bwd", gl.data_ptr(), out.data_ptr(), n, ctx.scale, _stream())
        return out, None


def mean_scale(v, scale):
    return _MeanScale.apply(v, float(scale))
