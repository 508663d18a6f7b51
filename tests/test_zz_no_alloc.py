"""The entry point that allocates nothing (``eeggan_mbstd_fwd_static``: partial sums in a buffer built into
the library) against ``eeggan_mbstd_fwd`` with a caller-provided workspace, through the C ABI: bit-exact, on
two calls in a row, the second with a smaller input (the shared buffer keeps the first call's contents); and
against the CPU oracle's MinibatchStddev (reference layers.py:67-90) in float64 at 1e-5."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, x, group):
    B, C, T = x.shape
    st = torch.cuda.current_stream().cuda_stream
    nbytes = lib.eeggan_mbstd_workspace(B, C, T, group)
    ws = torch.empty(max(nbytes, 4), device=x.device, dtype=torch.uint8)
    ref = torch.full((B, C + 1, T), float("nan"), device=x.device)
    got = torch.full((B, C + 1, T), float("nan"), device=x.device)
    assert lib.eeggan_mbstd_fwd(x.data_ptr(), ref.data_ptr(), ws.data_ptr(), B, C, T, group, st) == 0
    rc = lib.eeggan_mbstd_fwd_static(x.data_ptr(), got.data_ptr(), B, C, T, group, st)
    assert rc == 0, lib.eeggan_last_error()
    torch.cuda.synchronize()
    return ref, got


def test_static_scratch_entry_point_matches_workspace_entry_point(dev):
    from eeggan_amd import _lib
    from helpers import rel
    from oracle import ref_cpu as O
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(5)
    for B, C, T in ((32, 256, 32), (8, 64, 32)):  # the second call is smaller than the first
        x = torch.randn(B, C, T, device=dev, generator=gen)
        ref, got = _both(lib, x, 4)
        assert torch.isfinite(ref).all()
        assert torch.equal(ref, got), (B, C, T)
        want = O.minibatch_stddev(x.double().cpu(), 4, T)
        assert rel(got, want) < 1e-5, (B, C, T, rel(got, want))


def test_static_scratch_entry_point_rejects_oversized_inputs(dev):
    from eeggan_amd import _lib
    lib = _lib.load()
    # 16384 members x 16 chunks = 262144 partial sums > 65536: rejected on the host, nothing is launched
    rc = lib.eeggan_mbstd_fwd_static(ctypes.c_void_p(256), ctypes.c_void_p(256), 65536, 128, 32, 4, None)
    assert rc != 0 and b"scratch" in lib.eeggan_last_error()
