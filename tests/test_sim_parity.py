"""CPU tier: the product's HOST logic and kernel control flow under the test-only workgroup emulation
(tests/cpusim) against the oracle.  Small on purpose: the emulation runs one OS thread per lane.
Device parity proper is tests/test_gpu_parity.py (-m gpu)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpusim"))
import build as simbuild  # noqa: E402
import parity  # noqa: E402
import walnuts_amd as wa  # noqa: E402


@pytest.fixture(scope="module")
def sim():
    return simbuild.build()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("lds_vectors", [0, 1, 2])
def test_emulated_pool_tiers(sim, oracle, lds_vectors):
    """The span pool's two tiers -- LDS vectors, then the HBM arena -- with the LDS tier squeezed so that deep trees
    reach both; turn-arounds, halvings and the reversibility check's parked candidate included."""
    parity.run_case("std_normal", 200, 2, warmup=3, sampling=4, lib_path=sim, geometry=(1, 4), step=0.11,
                    max_trajectory_doublings=6, lds_vectors=lds_vectors)
    parity.run_case("diag_normal", 130, 2, warmup=2, sampling=3, lib_path=sim, geometry=(1, 4), step=1.4,
                    max_trajectory_doublings=4, lds_vectors=lds_vectors)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("model,D,geometry,fma", [   # fma: fused multiply-adds (the default) / every product rounded
    ("std_normal", 10, None, 1),          # (1,2): D < one pair per lane, heavy padding
    ("std_normal", 10, None, 0),
    ("diag_normal", 130, (1, 4), 1),      # (1,4): four elements per lane
    ("funnel", 9, (2, 2), 1),             # two wavefronts: cross-wave reductions, broadcasts, barriers
    ("funnel", 9, (2, 2), 0),
    ("std_normal", 200, (2, 2), 1),       # two wavefronts, span's other end in registers: wavefront 0 decides (share())
    ("diag_normal", 300, (1, -1), 1),     # streaming backend (vectors in HBM scratch), 3 tiles per lane
    ("diag_normal", 300, (1, -1), 0),
    ("rw1", 70, (1, 2), 1),               # neighbour-coupled gradient through the public model interface
])
def test_emulated_engine_matches_oracle(sim, oracle, model, D, geometry, fma):
    parity.run_case(model, D, 2, warmup=4, sampling=3, lib_path=sim, geometry=geometry, step=None,
                    fused_multiply_add=fma)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fma", [1, 0])
@pytest.mark.parametrize("model,D,geometry,fused", [
    ("std_normal", 1024, (1, 16), 1),     # the headline kernel: one wavefront per chain, the span's other end parked
    ("std_normal", 1024, (1, 16), 4),     # ... four transitions per launch
    ("diag_normal", 1024, (1, 16), 1),    # config #2's kernel (per-coordinate parameters in registers)
    ("funnel", 1024, (1, 16), 3),         # general gradient at the headline dimension
    ("std_normal", 1024, (2, 8), 2),      # two wavefronts per chain at one per SIMD
    ("rw1", 900, (2, 8), 1),
    ("std_normal", 1000, (4, 4), 1),      # four wavefronts, ragged padding
    ("funnel", 1000, (4, 4), 2),          # cross-wavefront sums inside the model
    ("rw1", 1024, (4, 4), 1),             # rw1's own geometry at 1 024 dimensions
    ("std_normal", 4096, (8, 8), 1),      # eight wavefronts
    ("diag_normal", 16384, (4, -1), 3),   # config #4's dimension: streaming, four wavefronts, LDS-parked inverse mass
    ("funnel", 16384, (2, -1), 1),        # streaming, two passes per micro step
    ("rw1", 12000, (4, -1), 2),           # streaming with halo reads
    ("std_normal", 20000, (2, -1), 1),    # ragged last tile
    ("diag_normal", 2048, (1, -1), 2),    # streaming with the moving end held in registers: all 16 tiles in use
    ("std_normal", 3900, (2, -1), 1),     # ... 16 tiles, the last one ragged
    ("diag_normal", 1100, (1, -1), 1),    # ... 9 of the 16
])
def test_emulated_full_size_geometries(sim, oracle, model, D, geometry, fused, fma):
    """The geometries the benchmarks run at -- the headline's (1, 16) at 1 024 dimensions, its multi-wavefront
    neighbours, the streaming kernels at 12 000-20 000 dimensions -- bit for bit against the oracle in both arithmetic
    modes, on the CPU tier (cheap since the emulation runs lanes as fibers; the GPU tier repeats them on the device)."""
    parity.run_case(model, D, 3, warmup=2 * fused + 2, sampling=2 * fused + 1, lib_path=sim, geometry=geometry,
                    fused_multiply_add=fma, fused=fused)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,geometry,kw", [
    ("std_normal", 64, None, dict(warmup=2, sampling=4, step=0.3, max_trajectory_doublings=6)),
    ("std_normal", 200, (2, 2), dict(warmup=2, sampling=4, step=1.0, max_trajectory_doublings=6)),
    ("diag_normal", 300, (1, -1), dict(warmup=2, sampling=4, step=0.3, max_trajectory_doublings=6)),   # streaming
    ("std_normal", 200, (1, 4), dict(warmup=2, sampling=4, step=0.3, max_trajectory_doublings=6, fused_multiply_add=0)),
])
def test_emulated_span_weights_move_their_reference_energy(sim, oracle, model, D, geometry, kw):
    """combine() in the linear domain (wn_traj.h, "span weights") when a tree's energies leave the range a fixed
    reference can carry: the reference moves, every live weight is rescaled, bit for bit like the oracle."""
    parity.run_weight_rebase_case(model, D, 3, lib_path=sim, geometry=geometry, **kw)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [2025, 7])
def test_emulated_random_campaign(sim, oracle, seed):
    """The GPU tier's randomised parity campaign (tests/gpu_probes/fuzz_parity.py: random model, dimension, geometry,
    pool tiers, step size, tree depth, halvings, micro steps, error bound, arithmetic mode, transitions per launch,
    chain groups) against the emulation, over the geometries the emulation is built with: a fixed number of cases per
    seed, every one bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "wn_fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_probes", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    done, failed, tally = fuzz.campaign(seed, seconds=300.0, cases=40, lib_path=sim,
                                        geometries=[(1, 2), (1, 4), (2, 2), (1, 16), (2, 8), (4, 4), (8, 8)],
                                        mem_waves=(1, 2, 4), chain_counts=(1, 2, 3, 5))
    assert done == 40 and not failed, failed
    assert len(tally) >= 8, tally   # (the cases really spread over models and geometries)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,fma,kw", [
    ("funnel", 140, 1, dict(warmup=2, sampling=2, max_trajectory_doublings=4)),   # sums over the coordinates: two passes per micro step
    ("funnel", 140, 0, dict(warmup=1, sampling=2, step=1.6, max_trajectory_doublings=3)),   # halvings + reversibility
    ("rw1", 300, 1, dict(warmup=2, sampling=2)),                       # neighbours' values: halo reads after a barrier
    ("rw1", 260, 0, dict(warmup=0, sampling=3, step=0.9, min_micro_steps=3, max_trajectory_doublings=3)),   # ping-pong sets
])
def test_emulated_streaming_backend_for_gradients_that_are_not_elementwise(sim, oracle, model, D, fma, kw):
    """The streaming kernels (vectors in HBM, num_params > 8192 by default; forced here at small sizes) for models whose
    gradient needs sums over all coordinates (funnel) or neighbouring coordinates (rw1): wn_model_api.h's streaming
    form, two passes per micro step, bit for bit against the oracle (both arithmetic modes over the four cases; the
    full cross product runs on the GPU)."""
    parity.run_case(model, D, 2, lib_path=sim, geometry=(1, -1), fused_multiply_add=fma, **kw)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,geometry", [
    ("std_normal", 10, None),          # one wavefront per chain: program order carries the chain's state
    ("funnel", 9, (2, 2)),             # two wavefronts: a workgroup barrier between the transitions
    ("diag_normal", 300, (1, -1)),     # streaming backend
])
def test_emulated_fused_transitions_match_single_steps(sim, oracle, model, D, geometry):
    """wn_engine_warmup_steps / wn_engine_sample_steps: several transitions of every chain in one launch (the workgroup
    that fetched a chain runs them back to back) leave exactly the state the oracle reaches with single steps -- stream
    keys, warmup iteration number (the estimator's discount) and Adam state advance per transition.  More chains than
    resident workgroups would not change anything here: every chain is its own work item."""
    parity.run_case(model, D, 3, warmup=5, sampling=5, lib_path=sim, geometry=geometry, step=None, fused=3)


@pytest.mark.timeout(600)
def test_emulated_fused_transitions_write_every_draw_row(sim):
    C_, D, T = 3, 10, 4
    def engine():
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C_, wa.default_config(sim), lib_path=sim)
        e.init_positions(5, 0, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(0.7)
        e.seed_chains(6, 0)
        return e
    a, b = engine(), engine()
    rows = np.full((C_, T, D), np.nan)
    a.warmup_steps(T, rows.ctypes.data, T * D, D)
    a.synchronize()
    for k in range(T):
        b.warmup_step()
        b.synchronize()
        assert np.array_equal(rows[:, k, :], b.positions()), k
    a.freeze(); b.freeze()
    planes = np.full((T, C_, D), np.nan)   # one [C][D] plane per transition (the bench's layout)
    a.sample_steps(T, planes.ctypes.data, D, C_ * D)
    a.synchronize()
    for k in range(T):
        b.sample_step()
        b.synchronize()
        assert np.array_equal(planes[k], b.positions()), k
    assert a.iteration == b.iteration == 2 * T
    assert np.array_equal(a.grad_evals(), b.grad_evals())
    # host-fed variates cover one transition
    b.set_variates(np.zeros((C_, D)), np.full((C_, 64), 0.5))
    with pytest.raises(ValueError, match="one transition"):
        b.sample_steps(2)
    with pytest.raises(ValueError, match="at least 1"):
        a.sample_steps(0)


@pytest.mark.timeout(600)
def test_emulated_engine_lds_pool_and_arena_paths(sim, oracle):
    # same chains with the span pool in LDS, split over LDS / HBM arena, in the arena only: identical results
    outs = []
    for lds in (-1, 2, 0):
        dev, orc = parity.run_case("std_normal", 12, 2, warmup=2, sampling=2, lib_path=sim, lds_vectors=lds,
                                   max_trajectory_doublings=4)
        outs.append(dev.positions())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])


@pytest.mark.timeout(600)
def test_emulated_engine_averaged_init_masses(sim, oracle):
    # InitConfigBuilder::masses(logp_grad, s, average_masses=true), config.hpp:371-380
    dev, orc = parity.run_case("diag_normal", 7, 3, warmup=2, sampling=1, lib_path=sim, average_masses=True)


@pytest.mark.timeout(600)
def test_emulated_engine_halving_and_reversibility(sim, oracle):
    # an oversized fixed step forces step halvings and reversibility re-integrations (walnuts.hpp:254-279)
    dev, orc = parity.run_case("std_normal", 6, 2, warmup=0, sampling=4, lib_path=sim, step=3.5,
                               max_trajectory_doublings=3)
    assert dev.grad_evals().sum() > 4 * 2 * 8  # more than the no-halving count


@pytest.mark.timeout(600)
def test_emulated_batched_adam_flushes_mid_transition(sim, oracle):
    # warmup from a tiny step: more than 64 macro steps in a transition, so the batched Adam update (wn_traj.h
    # adam_record / adam_flush) flushes a full register of observations mid-transition and the rest at its end
    dev, orc = parity.run_case("std_normal", 6, 2, warmup=3, sampling=1, lib_path=sim, step=0.01,
                               max_trajectory_doublings=7, check_every=1)
    assert dev.grad_evals().max() > 64


@pytest.mark.timeout(600)
def test_emulated_streaming_backend_halvings_and_deep_trees(sim, oracle):
    # the streaming kernels' buffer hand-over under retries, reversibility passes (multi-step leaves) and deeper trees
    parity.run_case("std_normal", 140, 2, warmup=0, sampling=3, lib_path=sim, geometry=(1, -1), step=2.9,
                    max_trajectory_doublings=3)
    parity.run_case("diag_normal", 140, 2, warmup=2, sampling=2, lib_path=sim, geometry=(1, -1), step=0.15,
                    max_trajectory_doublings=5)


@pytest.mark.timeout(600)
def test_emulated_engine_host_variates(sim, oracle):
    import test_gpu_parity

    test_gpu_parity._host_variates_case(sim, D=10, C=2)


@pytest.mark.timeout(600)
def test_emulated_controller_statistics(sim, oracle):
    dev, orc = parity.make_pair("std_normal", 8, 4, sim)
    pos = np.random.default_rng(0).normal(size=(4, 8))
    for x in (dev, orc):
        x.set_positions(pos); x.set_step_sizes(0.5); x.seed_chains(3, 0)
    for _ in range(3):
        dev.warmup_step(); orc.warmup_step()
    parity.check_monitors(dev, orc, warm=True)
    dev.freeze(); orc.freeze()
    for _ in range(3):
        dev.sample_step(); orc.sample_step()
    parity.check_monitors(dev, orc, warm=False)


def test_emulated_sample_device_early_stop(sim):
    # python/tests/test_pyfunc.py:38-64: min <= length <= max, and a loose tolerance stops at the minimum
    kw = dict(num_params=4, num_chains=2, seed=7, min_warmup_iter=5, max_warmup_iter=6, min_sampling_iter=3,
              max_sampling_iter=4, lib_path=sim, save_warmup=True)
    loose = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw, step_size_converge_tol=1e6, mass_converge_tol=1e6,
                              rhat_converge_tol=1e6)
    assert all(x.shape[0] == 3 and x.warmup.warmup_draws.shape[0] == 5 for x in loose)
    tight = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw, step_size_converge_tol=1e-12, mass_converge_tol=1e-12,
                              rhat_converge_tol=1.0 + 1e-12)
    assert all(x.shape[0] == 4 and x.warmup.warmup_draws.shape[0] == 6 for x in tight)


@pytest.mark.timeout(900)
def test_emulated_reference_stream_run_tracks_reference_order_oracle(sim, oracle):
    worst, _ = parity.check_reference_stream_run("std_normal", 7, 2, seed=48, warmup=4, sampling=3, lib_path=sim)
    assert worst <= 1e-10


def test_emulated_sample_device_contract(sim):
    # python/tests/test_pyfunc.py:38-125 restated for the device entry point
    kw = dict(num_params=5, num_chains=2, seed=1234, min_warmup_iter=2, max_warmup_iter=2, min_sampling_iter=2,
              max_sampling_iter=2, save_inv_metric=True, lib_path=sim)
    a = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    b = wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "save_warmup": True})   # same seed: same chains
    c = wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "seed": 4321})
    assert len(a) == 2 and a[0].shape == (2, 5)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
        assert x.warmup.stepsize == y.warmup.stepsize
        assert np.array_equal(x.warmup.inv_metric, y.warmup.inv_metric)
    assert not np.array_equal(a[0], c[0])
    assert b[0].warmup.warmup_draws.shape == (2, 5) and b[0].shape == (2, 5)
    with pytest.raises(ValueError, match="min_iter must be"):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "min_sampling_iter": 100, "max_sampling_iter": 10})
    with pytest.raises(ValueError, match="min_iter cannot be greater"):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "min_warmup_iter": 100, "max_warmup_iter": 10})
    with pytest.raises(ValueError, match="refresh must be non-negative"):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "refresh": -1})


def test_engine_argument_and_state_errors(sim):
    """Error typing of the C ABI: std::invalid_argument -> config (ValueError), anything else -> generic
    (RuntimeError), as python/src/walnutpie/errors.hpp:42-72 and _ffi.py:195-215 map them."""
    mk = lambda *a, **k: wa.DeviceEngine(*a, lib_path=sim, **k)
    with pytest.raises(ValueError, match="num_params"):
        mk(wa.MODEL_STD_NORMAL, 0, 2, wa.default_config(sim))
    with pytest.raises(ValueError, match="num_chains"):
        mk(wa.MODEL_STD_NORMAL, 3, 0, wa.default_config(sim))
    with pytest.raises(ValueError, match="model"):
        mk(17, 3, 2, wa.default_config(sim))
    with pytest.raises(ValueError, match="diag_normal model needs a parameter vector"):
        mk(wa.MODEL_DIAG_NORMAL, 3, 2, wa.default_config(sim))
    with pytest.raises(ValueError, match="funnel"):
        mk(wa.MODEL_FUNNEL, 1, 2, wa.default_config(sim))
    with pytest.raises(ValueError, match="max_hamiltonian_error"):
        mk(wa.MODEL_STD_NORMAL, 3, 2, wa.default_config(sim, max_hamiltonian_error=-1.0))
    with pytest.raises(ValueError, match="max_nuts_depth|max_trajectory_doublings"):
        mk(wa.MODEL_STD_NORMAL, 3, 2, wa.default_config(sim, max_trajectory_doublings=0))
    mk(wa.MODEL_FUNNEL, 40, 2, wa.default_config(sim, elems_per_lane=-1))   # (the funnel streams since round 3)
    e = mk(wa.MODEL_STD_NORMAL, 3, 2, wa.default_config(sim))
    with pytest.raises(ValueError, match="masses must be positive"):
        e.set_masses(np.array([[1.0, 0.0, 1.0], [1.0, 1.0, 1.0]]))
    with pytest.raises(ValueError, match="step size"):
        e.set_step_sizes([0.1, float("inf")])
    with pytest.raises(ValueError, match="mass_smoothing"):
        e.init_masses_from_grad(1.5)
    with pytest.raises(RuntimeError, match="before freeze"):
        e.sample_step()
    e.set_masses(np.array([[4.0, 1.0, 0.25], [1.0, 1.0, 1.0]]))
    # AdaptiveWalnuts::inv_mass() before any observation: sqrt((1/m)/m), adaptive_walnuts.hpp:54-62,89-94
    assert np.allclose(e.inv_mass(), [[0.25, 1.0, 4.0], [1.0, 1.0, 1.0]], rtol=1e-15)
    e.freeze()
    with pytest.raises(RuntimeError, match="after freeze"):
        e.warmup_step()
    assert e.inv_mass().shape == (2, 3)


def test_emulated_sample_device_output_buffer_check(sim):
    lib = wa.load_library(sim)
    out = np.zeros(10)
    lens = np.zeros(4, dtype=np.intc)
    err = C.c_void_p()
    dp = C.POINTER(C.c_double)
    rc = lib.walnutpie_sample_device(0, None, 5, None, 2, 1, 1, 2.0, None, 2, 2, 2, 2, 5, 5, 1, 0.5, 0.1, 1.0, 1.01, 4.0,
                                     1e-5, 15.0, 1.0, 0.8, 0.05, 0.8, 0.9, 1e-4, 0.5, False, out.ctypes.data_as(dp),
                                     out.size, lens.ctypes.data_as(C.POINTER(C.c_int)), None, None, 0,
                                     C.cast(None, wa._ffi.PRINT_CALLBACK), C.byref(err))
    assert rc == -1
    msg = lib.walnutpie_get_error_message(err).decode()
    assert msg.startswith("Output buffer too small. Expected at least 2 chains of 10 doubles, got 10")
    assert lib.walnutpie_get_error_type(err) == 0  # generic (std::runtime_error), walnutpy.cpp:153-160
    lib.walnutpie_destroy_error(err)


@pytest.mark.timeout(600)
def test_emulated_device_errors_are_reported_once_per_check(sim, oracle):
    # host-fed uniforms that run out inside a transition: reported by wn_engine_check even when a later transition of
    # the same engine is clean (the per-transition report, depth -1, is overwritten) -- and cleared by that check, so a
    # caller who then supplies enough variates can carry on
    D, Cn = 6, 2
    dev = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, Cn, wa.default_config(sim), lib_path=sim)
    rng = np.random.default_rng(2)
    dev.set_positions(rng.normal(size=(Cn, D)))
    dev.set_step_sizes(0.4)
    dev.seed_chains(1, 0)
    dev.freeze()
    dev.sample_step()
    dev.check()                                   # nothing wrong so far
    dev.set_variates(rng.normal(size=(Cn, D)), rng.uniform(size=(Cn, 1)))   # one uniform: not enough for a tree
    dev.sample_step()
    with pytest.raises(RuntimeError, match="host-fed uniforms"):
        dev.check()
    dev.sample_step()                             # counter-based stream again: a clean transition
    assert np.all(dev.depths() >= 1)
    dev.check()                                   # the error was reported (and cleared) by the check above
    dev.set_variates(rng.normal(size=(Cn, D)), rng.uniform(size=(Cn, 1)))
    dev.sample_step()
    dev.sample_step()                             # the failure is one transition back ...
    with pytest.raises(RuntimeError, match="host-fed uniforms"):
        dev.check()                               # ... and still reported


@pytest.mark.timeout(900)
@pytest.mark.parametrize("D,geometry", [(9, None), (131, (1, 4)), (150, (2, 2))])
def test_emulated_fourth_model_added_through_the_model_interface(sim, oracle, D, geometry):
    """rw1 (the reference's examples/examples.cpp:34-49) lives in csrc/models/rw1.h + a five-line .hip file: no edit
    of the kernels.  Its gradient couples neighbouring coordinates (cx.shift: lane shuffles, wavefront edges through
    LDS, pair-row wrap-around) and is kept as a vector -- the general path of the register kernels."""
    assert wa.model_id("rw1", lib_path=sim) == wa.MODEL_RW1 and wa.model_id("std_normal", lib_path=sim) == 0
    with pytest.raises(ValueError):
        wa.model_id("no_such_model", lib_path=sim)
    parity.run_case("rw1", D, 2, warmup=3, sampling=3, lib_path=sim, geometry=geometry)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,C,geometry", [
    ("std_normal", 10, 5, None),           # uneven halves (2 + 3 chains)
    ("funnel", 9, 4, (2, 2)),              # two wavefronts per chain
    ("diag_normal", 300, 3, (1, -1)),      # streaming backend: every group has its own arena slice
])
def test_emulated_chain_groups(sim, oracle, model, D, C, geometry):
    """wn_config::chain_groups = 2: the chains launched as two halves, each with its own chain counter, arena slice and
    stream -- the same bits as one group (the oracle knows nothing of groups); single steps and fused launches, warmup,
    freeze, sampling, and the host-side operations in between (every one of them first waits for both groups)."""
    parity.run_case(model, D, C, warmup=3, sampling=3, lib_path=sim, geometry=geometry, chain_groups=2)
    parity.run_case(model, D, C, warmup=4, sampling=4, lib_path=sim, geometry=geometry, chain_groups=2, fused=2)


@pytest.mark.timeout(600)
def test_emulated_stream_ordering_calls(sim):
    """wn_engine_wait_stream / _release_stream / _wait_event (ordering a caller's stream or event against the engine's
    streams without adopting them) and wn_engine_set_stream (adopting: one chain group from then on) change nothing in
    the draws; a null engine-side handle is an error object, not a crash."""
    def engine(groups):
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, 10, 5, wa.default_config(sim, chain_groups=groups), lib_path=sim)
        e.init_positions(3, 0, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(0.4)
        e.seed_chains(4, 0)
        return e

    a, b, c = engine(1), engine(2), engine(2)
    assert (a.chain_groups, b.chain_groups) == (1, 2)
    for it in range(3):
        a.warmup_step()
        b.wait_stream(0)
        b.warmup_step()
        b.release_stream(0)
        if it == 1:
            c.set_stream(0)
            assert c.chain_groups == 1
        c.warmup_step()
    for x in (b, c):
        assert np.array_equal(a.positions(), x.positions()) and np.array_equal(a.step_sizes(), x.step_sizes())
    err = C.c_void_p()
    assert b.lib.wn_engine_wait_event(b.h, None, C.byref(err)) == 0   # (the emulation's events are no-ops)
    for e in (a, b, c):
        e.close()


@pytest.mark.timeout(600)
def test_emulated_failed_extension_flag(sim):
    """wn_engine_get_failed_extensions, the failure channel of device models (the counterpart of the reference's
    on_logp_exception events, util.hpp:336-346): a chain that starts where the model's log density overflows fails its
    first leaf at every step size, the extension fails and the chain stays put -- flagged after every transition; the
    other chains (small step: nothing fails) report nothing."""
    C, D = 3, 10
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C, wa.default_config(sim), lib_path=sim)
    pos = np.full((C, D), 0.3)
    pos[1] = 1e200
    e.set_positions(pos)
    e.set_step_sizes(0.05)
    e.seed_chains(1, 0)
    e.warmup_step()
    e.synchronize()
    assert e.failed_extensions().tolist() == [0, 1, 0]
    assert np.array_equal(e.positions()[1], pos[1]) and np.all(e.positions()[0] != pos[0])
    e.freeze()
    e.sample_steps(2)
    e.synchronize()
    assert e.failed_extensions().tolist() == [0, 1, 0]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,geometry", [("diag_normal", 130, (1, 4)), ("funnel", 9, (2, 2)), ("rw1", 70, (1, 2)),
                                              ("diag_normal", 300, (1, -1))])
def test_observation_pending_between_warmup_launches(sim, oracle, model, D, geometry):
    """The register kernels' warmup transitions leave the mass estimator's observation of their result to the NEXT
    transition's prologue (wn_chip.h kDeferObservation); between two launches the engine holds it pending and applies it
    before anything but another warmup launch (wn_engine: flush_pending_observation).  Launches of one and of several
    transitions back to back with NOTHING read in between, then every way out of the pending state -- a read of the
    estimator, a freeze, new positions -- against the oracle, which observes at the end of each transition
    (adaptive_walnuts.hpp:247-248).  (The streaming kernels observe in their own epilogue: same test, nothing pending.)"""
    parity.run_pending_observation_case(model, D, 3, lib_path=sim, geometry=geometry)
