"""Adam with the interface and state layout of ``torch.optim.Adam`` (the optimizer the reference
builds at train.py:131-136), stepping every parameter in ONE multi-tensor HIP kernel launch."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ._lib import call


class _Desc(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_int64)]


DESC_BYTES = ctypes.sizeof(_Desc)


def build_table(rows, device):
    """rows: list of (p_ptr, g_ptr, m_ptr, v_ptr, n).  Returns (device uint8 table, host int64 sizes)."""
    arr = (_Desc * len(rows))()
    for i, (p, g, m, v, n) in enumerate(rows):
        arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n = p, g, m, v, n
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    sizes = np.array([r[4] for r in rows], dtype=np.int64)
    return host.to(device), sizes


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("Adam: weight_decay and amsgrad are not supported")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.grad_scale = 1.0  # set to 1/world_size when gradients arrive as a sum over ranks

    @torch.no_grad()
    def step(self, closure=None, grads=None):
        """``grads`` optionally maps a parameter to the tensor holding its gradient (e.g. a slice of
        an all-reduced bucket); by default ``p.grad`` is used and parameters without one are skipped."""
        if closure is not None:
            raise NotImplementedError("Adam.step: closures are not supported")
        for group in self.param_groups:
            rows, keep = [], []
            step_no = None
            for p in group["params"]:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("Adam: parameters must be contiguous float32 device tensors")
                g = g if g.is_contiguous() else g.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                s = int(st["step"].item())
                if step_no is None:
                    step_no = s
                elif s != step_no:
                    # parameters that joined later (progressive growing) have younger moments
                    self._launch(rows, keep, group, step_no)
                    rows, keep, step_no = [], [], s
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             p.numel()))
                keep.append(g)
            if rows:
                self._launch(rows, keep, group, step_no)
        return None

    def _launch(self, rows, keep, group, step_no):
        dev = torch.device("cuda", torch.cuda.current_device())
        table, sizes = build_table(rows, dev)
        b1, b2 = group["betas"]
        call("eeggan_adam_step", table.data_ptr(), sizes.ctypes.data, len(rows), float(group["lr"]), float(b1),
             float(b2), float(group["eps"]), int(step_no), float(self.grad_scale),
             torch.cuda.current_stream().cuda_stream)
        from . import ops
        ops.bump_weights_epoch()  # cached derived weight views are stale from here on
        # the table and the gradient tensors are referenced by the enqueued kernel: torch's
        # stream-ordered allocator keeps their memory valid until the kernel has run.
        del table, keep
