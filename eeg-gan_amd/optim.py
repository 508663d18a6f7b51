"""Adam with the interface and state layout of ``torch.optim.Adam`` (the optimizer the reference
builds at train.py:131-136), stepping every parameter in ONE multi-tensor HIP kernel launch.

The per-parameter step counts live on the device (like torch's ``capturable=True`` mode) and the
descriptor table travels in the kernel arguments, so ``step()`` makes no host<->device copy and no
synchronisation: it can be captured into a HIP graph and replayed."""
from __future__ import annotations

import ctypes

import torch

from ._lib import call


class _Desc(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("step", ctypes.c_void_p), ("n", ctypes.c_int64)]


DESC_BYTES = ctypes.sizeof(_Desc)


def build_table(rows):
    """rows: list of (p_ptr, g_ptr, m_ptr, v_ptr, step_ptr, n).  Returns a host ctypes array."""
    arr = (_Desc * len(rows))()
    for i, (p, g, m, v, s, n) in enumerate(rows):
        arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].step, arr[i].n = p, g, m, v, s, n
    return arr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("Adam: weight_decay and amsgrad are not supported")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.grad_scale = 1.0  # set to 1/world_size when gradients arrive as a sum over ranks

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), device=p.device, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        elif not st["step"].is_cuda:  # state restored from a checkpoint written by torch.optim.Adam
            st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
        return st

    @torch.no_grad()
    def step(self, closure=None, grads=None):
        """``grads`` optionally maps a parameter to the tensor holding its gradient (e.g. a slice of
        an all-reduced bucket); by default ``p.grad`` is used and parameters without one are skipped."""
        if closure is not None:
            raise NotImplementedError("Adam.step: closures are not supported")
        from . import ops
        for group in self.param_groups:
            rows, keep = [], []
            for p in group["params"]:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("Adam: parameters must be contiguous float32 device tensors")
                g = g if g.is_contiguous() else g.contiguous()
                st = self._init_state(p)
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             st["step"].data_ptr(), p.numel()))
                keep.append(g)
            if rows:
                b1, b2 = group["betas"]
                call("eeggan_adam_step", build_table(rows), len(rows), float(group["lr"]), float(b1), float(b2),
                     float(group["eps"]), float(self.grad_scale), torch.cuda.current_stream().cuda_stream)
                ops.bump_weights_epoch()  # cached derived weight views are stale from here on
        return None
