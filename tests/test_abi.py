"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports
every symbol ``include/happypose_amd.h`` declares (no compute calls -- no GPU here)."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "happypose_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from happypose_amd import _ffi
    from happypose_amd.build import build

    build()
    lib = ctypes.CDLL(str(_ffi.lib_path()))
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    # the Python binding covers exactly the header
    assert sorted(_ffi.EXPORTED_SYMBOLS) == declared


def test_no_cpu_fallback(monkeypatch, tmp_path):
    """The product path must fail loudly when the HIP library is missing."""
    from happypose_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "_LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_ffi.HipLibraryError):
        _ffi.lib()


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for py in (ROOT / "happypose_amd").rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), py


def test_argument_errors_reported_without_gpu():
    from happypose_amd import _ffi

    lib = _ffi.lib()
    assert lib.hp_version() >= 100
    rc = lib.hp_pose_update(-1, None, None, 9, None, None, None, None)
    assert rc == -1 and b"hp_pose_update" in lib.hp_last_error()
    with pytest.raises(AssertionError):
        _ffi.check(rc, "hp_pose_update")
    assert not lib.hp_net_create(7, 6, 240, 320)
    assert not lib.hp_mesh_store_create(None, None, None, None, 0, None, 0, None, 0, None, 0, None, 0)
