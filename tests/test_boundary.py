"""CPU tier: the C-ABI library loads and exports every symbol include/walnuts_hip.h declares (no compute)."""
import os
import re

import pytest

import walnuts_amd._ffi as ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "walnuts_hip.h")).read()
    text = text.replace("#define WALNUTS_HIP_EXPORT", "")
    return set(re.findall(r"WALNUTS_HIP_EXPORT[^;(]*?\b(\w+)\s*\(", text))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    bound = {name for name, _, _ in ffi.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(ffi.DEFAULT_LIB):
        pytest.fail("walnuts_amd/lib/libwalnuts_hip.so is not built: run __graft_entry__.build()")
    lib = ffi.load_library()  # binds every symbol or raises AttributeError
    for name in _declared_symbols():
        assert hasattr(lib, name)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ffi.WalnutsHipError, match="no CPU fallback"):
        ffi.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "walnuts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "wno" not in re.findall(r"import\s+(\w+)", src), f
                assert "oracle/" not in src and "wn_oracle" not in src, f
