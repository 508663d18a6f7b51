"""The C-ABI library loads and exports every symbol that include/eeggan_hip.h declares
(no compute call is made: this runs without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "eeggan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eeggan_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert "eeggan_conv1d" in syms and "eeggan_conv1d_wgrad" in syms and "eeggan_adam_step" in syms
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from eeggan_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_abi_version_and_error_channel():
    from eeggan_amd import _lib
    lib = _lib.load()
    assert lib.eeggan_abi_version() == 4
    assert isinstance(lib.eeggan_last_error(), bytes)
    # argument validation happens on the host before any launch: safe without a GPU
    rc = lib.eeggan_conv1d(None, None, None, None, None, 1, 1, 1, 1, 5, 1.0, 0, 0.0, None, None, None, 0.0, None)
    assert rc != 0 and b"kernel size" in lib.eeggan_last_error()
    assert lib.eeggan_conv1d_wgrad_workspace(32, 64, 64, 1920, 3) > 0
    assert lib.eeggan_conv1d_wgrad_workspace(32, 64, 64, 1920, 5) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from eeggan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError):
        _lib.load()


def test_no_cpu_path():
    import pytest
    import torch
    from eeggan_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.conv1d_raw(torch.zeros(1, 2, 8), torch.zeros(3, 2, 3))
