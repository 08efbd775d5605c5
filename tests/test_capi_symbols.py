"""CPU: the C-ABI library loads and exports every symbol include/e2e_asr_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "e2e_asr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    assert "asr_lstm_layer_fwd" in names and "asr_gemm_f32" in names and len(names) >= 10


def test_library_exports_every_declared_symbol():
    from e2e_asr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), "missing export: %s" % name
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == _declared()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from e2e_asr_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
