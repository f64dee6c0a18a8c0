"""include/walnuts_hip.hpp -- the C++ mirror of the reference's surface for the many-chain path (SURVEY.md §8b:
batched AdaptiveWalnuts / WalnutsSampler, per-chain Sampler views, handler callbacks, config builders, the
top-level walnuts() call).  tests/cpp/cpp_surface.cpp exercises the contracts in C++ and dumps what the handlers
saw; here the dump is compared bit for bit with the oracle run through the same recipe.

CPU tier: linked against the workgroup emulation.  GPU tier: linked against libwalnuts_hip.so."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "cpusim"))
import parity  # noqa: E402

wno = parity.wno


sys.path.insert(0, os.path.join(HERE, "cpp"))
from build_surface import build_surface_library, surface_library  # noqa: E402


def run_and_compare(surface: str, lib_path: str, model: str, C: int, D: int, W: int, S: int, seed: int, tmp_path):
    import ctypes
    dump = str(tmp_path / "dump.bin")
    so = ctypes.CDLL(surface)
    so.cpp_surface_run.restype = ctypes.c_int
    so.cpp_surface_run.argtypes = [ctypes.c_char_p] + [ctypes.c_size_t] * 5 + [ctypes.c_char_p]
    rc = so.cpp_surface_run(model.encode(), C, D, W, S, seed, dump.encode())   # in-process: nothing is spawned
    assert rc == 0, "the C++ expectations failed (see stderr)"
    raw = np.fromfile(dump, dtype=np.float64)
    per_chain = W * D + W + W + W * D + 1 + D + S * D + S
    n_rhat = 1 if (C > 1 and S >= 2) else 0
    assert raw.size == C * per_chain + n_rhat
    got = []
    for c in range(C):
        r0 = raw[c * per_chain:(c + 1) * per_chain]
        o = 0
        rec = {}
        for name, n, shape in (("warmup_draws", W * D, (W, D)), ("warmup_lp", W, (W,)), ("warmup_step", W, (W,)),
                               ("warmup_inv_mass", W * D, (W, D)), ("final_step", 1, ()), ("final_inv_mass", D, (D,)),
                               ("draws", S * D, (S, D)), ("lp", S, (S,))):
            rec[name] = r0[o:o + n].reshape(shape)
            o += n
        got.append(rec)

    # the same recipe on the oracle (device arithmetic order): InitConfigBuilder verbs, then api.hpp:46-69
    _, om = parity.MODELS[model]
    from walnuts_amd import _ffi
    dm, _ = parity.MODELS[model]
    lanes = _ffi.load_library(lib_path).wn_lanes_for_model_dim(dm, D, 0, 0)   # reduction width of the default geometry
    import walnuts_amd as wa
    cfg = wno.default_config(rng_mode=wno.RNG_PHILOX, math_mode=wno.MATH_PORTABLE, reduce_lanes=lanes,
                             fma=int(wa.default_config(lib_path).fused_multiply_add))   # the library's arithmetic mode
    o = wno.Engine(om, D, C, cfg, params=parity.model_params(model, D))
    o.init_positions(seed + 5, 0, 2.0)
    o.init_masses_from_grad(1e-5)
    o.set_step_sizes(1.0)
    o.adapt_step(seed + 6, 0)
    o.seed_chains(seed, 0)
    for it in range(W):
        im = o.inv_mass()                       # inverse masses the transition integrates with
        o.warmup_step(8)
        pos, lp, st = o.positions(), o.logp(), o.step_sizes()
        for c in range(C):
            assert np.array_equal(got[c]["warmup_draws"][it], pos[c]), f"warmup draw it={it} chain={c}"
            assert got[c]["warmup_lp"][it] == lp[c]
            assert got[c]["warmup_step"][it] == st[c], "on_warmup step size is the post-update one"
            assert np.array_equal(got[c]["warmup_inv_mass"][it], im[c]), "on_warmup inverse mass is the pre-transition one"
    o.freeze()
    st, im = o.step_sizes(), o.inv_mass()
    for c in range(C):
        assert got[c]["final_step"] == st[c] and np.array_equal(got[c]["final_inv_mass"], im[c])
    for it in range(S):
        o.sample_step(8)
        pos, lp = o.positions(), o.logp()
        for c in range(C):
            assert np.array_equal(got[c]["draws"][it], pos[c]), f"draw it={it} chain={c}"
            assert got[c]["lp"][it] == lp[c]
    if n_rhat:
        assert raw[-1] == pytest.approx(o.rhat(), rel=1e-12)


@pytest.mark.timeout(1800)
def test_cpp_surface_under_emulation(oracle, tmp_path):
    import build as simbuild
    sim = simbuild.build()
    run_and_compare(build_surface_library(sim, "sim"), sim, "diag_normal", 2, 5, 3, 2, 77, tmp_path)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,C,D,W,S", [("std_normal", 64, 1024, 12, 8), ("funnel", 40, 128, 12, 8),
                                           ("diag_normal", 6, 9000, 6, 4)])
def test_cpp_surface_on_gpu(gpu, tmp_path, model, C, D, W, S):
    from walnuts_amd import _ffi
    surface = surface_library("hip")
    # prebuilt by __graft_entry__.build(): a process that spawns children (a compiler) on the GPU box loses its GPU
    assert os.path.exists(surface), "tests/cpp/libcpp_surface_hip.so missing: run __graft_entry__.build()"
    run_and_compare(surface, _ffi.DEFAULT_LIB, model, C, D, W, S, 4242, tmp_path)


@pytest.mark.timeout(600)
def test_example_program_builds_and_runs_under_emulation(tmp_path):
    """examples/walnuts_hip_api.cpp (the device counterpart of the reference's examples/walnutpie_api.cpp) compiles
    warning-free against include/walnuts_hip.hpp and runs end to end (tiny sizes, workgroup emulation)."""
    import subprocess

    import build as simbuild
    sim = simbuild.build()
    exe = str(tmp_path / "walnuts_hip_api")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "walnuts_hip_api.cpp"), sim,
                           f"-Wl,-rpath,{os.path.dirname(sim)}", "-pthread", "-o", exe])
    r = subprocess.run([exe, "2", "3", "2", "4"], capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FINISHED NORMALLY." in r.stdout and "# warmup_draws = 2; # draws = 4" in r.stdout
