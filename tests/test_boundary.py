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


def test_library_stands_in_for_libwalnutpy_at_load_time_and_refuses_host_models():
    """Every symbol the reference's python/src/walnutpie/_ffi.py binds at import exists; the two host-model samplers
    refuse with the reference's error protocol (rc != 0, type `config` -> ValueError there) instead of sampling on a
    CPU path that does not exist."""
    if not os.path.exists(ffi.DEFAULT_LIB):
        pytest.fail("walnuts_amd/lib/libwalnuts_hip.so is not built: run __graft_entry__.build()")
    import ctypes as C
    import numpy as np
    lib = ffi.load_library()
    for name in ("walnutpie_get_error_message", "walnutpie_get_error_type", "walnutpie_destroy_error",
                 "walnutpie_sample_cfunc", "walnutpie_sample_bridgestan", "walnutpie_ess", "walnutpie_r_hat",
                 "walnutpie_mcse", "walnutpie_separator_char"):
        assert hasattr(lib, name), name
    assert lib.walnutpie_separator_char() == b"\x1c"
    called = []

    @ffi.LOGP_CFUNC
    def logp(size, theta, grad, lp, data):
        called.append(size)
        return 0

    out = np.zeros(4 * 3)
    lengths = np.zeros(2, dtype=np.intc)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    trailing = [1, 1, 0, 2.0, None, 1, 1, 1, 1, 8, 8, 1, 0.5, 0.1, 0.1, 1.1, 1.0, 1e-5, 8.0, 1.0, 0.8, 0.2, 0.3, 0.99, 0.1,
                0.5, False, dp(out), out.size, lengths.ctypes.data_as(C.POINTER(C.c_int)), None, None, 0,
                ffi.PRINT_CALLBACK(0)]
    for call in (lambda e: lib.walnutpie_sample_cfunc(logp, None, 3, None, *trailing, C.byref(e)),
                 lambda e: lib.walnutpie_sample_bridgestan(b"model.so", b"{}", ffi.PRINT_CALLBACK(0), 1, b"", *trailing,
                                                           C.byref(e))):
        err = C.c_void_p()
        assert call(err) != 0 and err.value
        assert lib.walnutpie_get_error_type(err) == 1   # config (errors.hpp:10-24) -> ValueError in the reference's wrapper
        assert b"walnutpie_sample_device" in lib.walnutpie_get_error_message(err)
        lib.walnutpie_destroy_error(err)
    assert not called and not out.any()


def test_default_launch_geometry_by_model_and_dimension():
    """The engine's own choice of kernel (wn_launch.h, choose_geometry), asked through the C ABI without a GPU: one
    wavefront per chain up to 1 024 parameters, two / four at sixteen elements per lane up to 2 048 / 4 096 (round 6); the
    streaming kernels that hold the trajectory's moving end in registers
    (8 wavefronts) from 4 097 to 16 384 parameters for one-pass gradients and rw1, from 8 193 for the funnel (two passes,
    sums only: its (16, 8) register kernels still win below); sixteen wavefronts streaming both ends beyond 16 384."""
    if not os.path.exists(ffi.DEFAULT_LIB):
        pytest.fail("walnuts_amd/lib/libwalnuts_hip.so is not built: run __graft_entry__.build()")
    lib = ffi.load_library()
    lanes = lambda name, D: lib.wn_lanes_for_model_dim(lib.wn_model_id(name.encode()), D, 0, 0)
    assert [lanes("std_normal", D) for D in (100, 1024, 2048, 4096, 4097, 8192, 16384, 16385)] == [64, 64, 128, 256, 512, 512, 512, 1024]
    assert [lanes("diag_normal", D) for D in (1024, 6000, 16384, 40000)] == [64, 512, 512, 1024]
    assert [lanes("funnel", D) for D in (128, 6000, 8192, 8193, 16384, 16385)] == [64, 1024, 1024, 512, 512, 1024]
    # (rw1 states its hint as a function of the dimension: the default policy up to 1 024, (8,4) up to 2 048, (8,8) up to 4 096)
    assert [lanes("rw1", D) for D in (512, 1024, 2048, 4096, 4097, 16384, 16385)] == [64, 64, 512, 512, 512, 512, 1024]


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


def test_launch_geometry_answers_come_from_the_library():
    """the engine's choice of launch geometry depends on the model (one-pass gradients stream from 4 097
    parameters on the held kernels, the funnel's kind from 8 193); wn_geometry_for_model answers for a registered model
    with the engine's own rule, wn_geometry_candidates lists what a not-yet-registered (run-time compiled) model has to
    instantiate, and walnuts_amd.models derives its -D switches from that list alone.  No GPU needed: host logic."""
    import walnuts_amd as wa
    from walnuts_amd import models

    assert models.geometry_for(1024) == (1, 16, False) == models.geometry_for(1024, model=wa.MODEL_STD_NORMAL)
    assert models.geometry_for(2000) == (2, 16, False) and models.geometry_for(3000, model=wa.MODEL_STD_NORMAL) == (4, 16, False)
    assert models.geometry_for(2000, model=wa.MODEL_FUNNEL) == (2, 16, False)
    assert models.geometry_for(3000, model=wa.MODEL_FUNNEL) == (8, 8, False)  # (a carried gradient: eight per lane there)
    assert models.geometry_for(2048, model=wa.MODEL_RW1) == (8, 4, False) and models.geometry_for(1024, model=wa.MODEL_RW1) == (1, 16, False)
    assert models.geometry_for(4096, model=wa.MODEL_RW1) == (8, 8, False)
    # 6 000 parameters: (16, 8) register kernels for a model without held streaming kernels ...
    assert models.geometry_for(6000) == (16, 8, False)
    # ... the held streaming kernels (8 wavefronts) for the built-in one-pass models, (16, 8) for the funnel up to 8 192
    assert models.geometry_for(6000, model=wa.MODEL_DIAG_NORMAL) == (8, 0, True)
    assert models.geometry_for(6000, model=wa.MODEL_RW1) == (8, 0, True)
    assert models.geometry_for(6000, model=wa.MODEL_FUNNEL) == (16, 8, False)
    assert models.geometry_for(12000, model=wa.MODEL_FUNNEL) == (8, 0, True)
    assert models.geometry_for(20000, model=wa.MODEL_DIAG_NORMAL) == (16, 0, True)
    assert models.geometry_candidates(1024) == [(1, 16, False)]
    assert sorted(models.geometry_candidates(6000)) == [(8, 0, True), (16, 8, False)]
    assert sorted(models.geometry_candidates(12000)) == [(8, 0, True), (16, 0, True)]
    assert models.geometry_candidates(20000) == [(16, 0, True)]
    assert models.geometry_candidates(6000, waves_per_chain=16, elems_per_lane=8) == [(16, 8, False)]   # a request is a request
    assert list(models.geometry_defines([(16, 8, False), (8, 0, True)])) == ["-DWN_ONLY_NW=16", "-DWN_ONLY_EPL=8", "-DWN_ONLY_MEM_NW=8"]
    assert list(models.geometry_defines([(16, 0, True), (8, 0, True)])) == ["-DWN_ONLY_MEM_NW=16", "-DWN_ONLY_MEM_NW_ALSO=8"]
    # the library states what it was built with, and a run-time model is built with exactly that
    flags = models.hipcc_flags()
    assert "-ffp-contract=off" in flags and "--offload-arch=gfx950" in flags
    lib = wa.load_library()
    assert lib.wn_build_compiler().decode() != "" and lib.wn_stream_version() == 2
    models.check_compiler()   # this container's hipcc built the library
