"""CPU: the C-ABI shared library loads and exports every symbol that include/maskunet_hip.h declares, and the
ctypes signature table of the Python binding covers exactly that set (no compute calls -- no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "maskunet_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_symbols():
    names = _declared()
    assert len(names) >= 25 and "mu_attn_fwd" in names and "mu_conv_wgrad" in names


def test_library_exports_every_declared_symbol():
    from maskunet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in maskunet_hip.h but not exported"


def test_binding_table_matches_header():
    from maskunet_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert b"gfx950" in lib.mu_version_host()


def test_workspace_queries_are_host_only():
    from maskunet_amd import _lib
    lib = _lib.load()
    assert lib.mu_bn_workspace_bytes(64) > 0
    assert lib.mu_conv_wgrad_workspace_bytes(2, 16, 16, 64, 64, 9) >= 9 * 64 * 64 * 4
    assert lib.mu_attn_bwd_workspace_bytes(2, 256, 64) > 0
    assert lib.mu_ln_sample_workspace_bytes(4) > 0
    assert lib.mu_colsum_workspace_bytes(64) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from maskunet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_null_args_return_error_codes_without_gpu():
    from maskunet_amd import _lib
    lib = _lib.load()
    assert lib.mu_conv_fwd(None, None, None, None, 1, 1, 1, 32, 32, 9, 32, 32, 0, None) == -1
    assert lib.mu_transpose(None, 0, 1, None, 0, 1, 1, 1, 1, None) == -1
    assert lib.mu_attn_fwd(None, None, None, None, None, None, None, None, None, None, None, 1, 1, 64, 1, 1e-5, 0, None) == -1
