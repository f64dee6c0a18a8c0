"""Shared parity driver: runs the same call sequence on the product engine (HIP library on a GPU, or the
test-only CPU emulation of it) and on the oracle configured to replay the device's execution order
(portable maths, counter-based stream, reductions over `engine.lanes` lanes), and demands bit equality.

Bit-exact is the bar for the element-wise leapfrog state AND for every reduced scalar, because both sides
execute the same IEEE operations in the same order; the <=1e-10 relative bar of BASELINE.json's north star
is the distance between this device-order oracle and the reference-order oracle (sequential sums, libm),
checked separately in tests/test_oracle_kat.py and test_*_parity.py::test_reference_order_tolerance.
"""
import numpy as np

import walnuts_amd as wa
import wno

MODELS = {"std_normal": (wa.MODEL_STD_NORMAL, wno.MODEL_STD_NORMAL),
          "diag_normal": (wa.MODEL_DIAG_NORMAL, wno.MODEL_DIAG_NORMAL),
          "funnel": (wa.MODEL_FUNNEL, wno.MODEL_FUNNEL),
          "rw1": (wa.MODEL_RW1, wno.MODEL_RW1)}

CFG_FIELDS = ("max_trajectory_doublings", "max_step_halvings", "min_micro_steps", "max_hamiltonian_error",
              "mass_init_count", "max_macro_steps_target", "step_accept_rate_target", "step_learning_rate",
              "step_gradient_decay", "step_sq_gradient_decay", "step_stabilization", "step_learn_rate_decay")


def model_params(model: str, D: int):
    if model not in ("diag_normal", "user_diag"):   # (user_diag: tests/test_runtime_model.py)
        return None
    return np.array([(1.0 + (d % 16)) ** 2 for d in range(D)])  # sigma_d = 1 + (d mod 16), SURVEY.md §8d cfg4


def make_pair(model: str, D: int, C: int, lib_path=None, geometry=None, **cfg_over):
    """-> (device engine, oracle engine) with identical configuration."""
    dm, om = MODELS[model]
    geo = {}
    if geometry is not None:
        geo = dict(waves_per_chain=geometry[0], elems_per_lane=geometry[1])
    extra = {k: cfg_over.pop(k) for k in ("workgroups_per_cu", "lds_vectors", "fused_multiply_add", "chain_groups")
             if k in cfg_over}
    dcfg = wa.default_config(lib_path, **cfg_over, **geo, **extra)
    params = model_params(model, D)
    dev = wa.DeviceEngine(dm, D, C, dcfg, params=params, lib_path=lib_path)
    ocfg = wno.default_config(rng_mode=wno.RNG_PHILOX, math_mode=wno.MATH_PORTABLE, reduce_lanes=dev.lanes,
                              fma=int(dcfg.fused_multiply_add), **cfg_over)
    orc = wno.Engine(om, D, C, ocfg, params=params)
    return dev, orc


def check_monitors(dev, orc, warm: bool):
    """The controller statistics (adapt.hpp:193-221, sampler.hpp:132-145), bit for bit: the device's monitor kernels sum
    over chains in runs of 256 and over dimensions in a fixed block tree, with the portable exp / log -- and the
    device-order oracle replays exactly that (oracle/wn_oracle.cpp: chain_sum, wno_warmup_spread)."""
    if warm:
        assert dev.warmup_spread() == orc.warmup_spread(), (dev.warmup_spread(), orc.warmup_spread())
    else:
        d, o = dev.rhat(), orc.rhat()
        assert d == o or (d != d and o != o), (d, o)


def same_bits_or_nan(a, b) -> bool:
    """Exact equality, NaN in the same place on both sides counting as equal."""
    a, b = np.asarray(a), np.asarray(b)
    return np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


def assert_same_state(dev, orc, where: str, warm: bool):
    dev.synchronize()
    checks = [("positions", dev.positions(), orc.positions()), ("logp", dev.logp(), orc.logp()),
              ("depths", dev.depths(), orc.depths()), ("grad_evals", dev.grad_evals(), orc.grad_evals()),
              ("rng_draws", dev.rng_draws(), orc.rng_draws())]
    if warm:
        checks.append(("adam", dev.adam(), orc.adam()))
        de, oe = dev.estimator(), orc.estimator()
        for k in ("draw_mean", "draw_ssd", "score_mean", "score_ssd", "weights"):
            checks.append(("estimator." + k, de[k], oe[k]))
        checks.append(("min_micro", dev.min_micro(), orc.min_micro()))
        checks.append(("step_sizes", dev.step_sizes(), orc.step_sizes()))
        checks.append(("inv_mass estimate", dev.inv_mass(), orc.inv_mass()))
    for name, a, b in checks:
        a, b = np.asarray(a), np.asarray(b)
        b = b.astype(a.dtype) if a.dtype != b.dtype else b
        # NaN in the same place on both sides is parity (the reference's NaN hazard, SURVEY.md section 5: a non-finite
        # energy error poisons Adam on both sides alike); everything else is compared exactly
        if not np.array_equal(a, b, equal_nan=a.dtype.kind == "f"):
            bad = np.argwhere((a != b) & ~((a != a) & (b != b)) if a.dtype.kind == "f" else (a != b))
            raise AssertionError(f"{where}: {name} differs at {bad[:4].tolist()} "
                                 f"device={a[tuple(bad[0])]!r} oracle={b[tuple(bad[0])]!r} ({len(bad)} entries)")
    assert np.all(dev.depths() >= 1), f"{where}: device reported span-pool exhaustion"


def run_case(model: str, D: int, C: int, *, warmup: int, sampling: int, lib_path=None, geometry=None, seed=1234,
             init="random", step=None, check_every=1, average_masses=False, fused=1, init_scale=2.0, lazy=False,
             **cfg_over):
    """InitConfigBuilder -> warmup -> freeze -> sampling on both sides, bit-compared along the way.  fused > 1: the
    device runs that many transitions per launch (wn_engine_warmup_steps / _sample_steps), the oracle single steps.
    lazy: nothing is read between the launches of a phase (the state is compared at the end of the phase only), so the
    register kernels' pending estimator observation crosses launch boundaries (wn_chip.h kDeferObservation)."""
    dev, orc = make_pair(model, D, C, lib_path, geometry, **cfg_over)
    rng = np.random.default_rng(seed)
    if init == "random":
        pos = rng.normal(0.0, init_scale, size=(C, D))  # init_radius 2.0, pyfunc.py:57
        for x in (dev, orc):
            x.set_positions(pos)
    elif init == "device":
        dev.init_positions(seed, 7, 2.0)
        orc.init_positions(seed, 7, 2.0)
        dev.synchronize()
        assert np.array_equal(dev.positions(), orc.positions()), "init positions differ"
    dev.init_masses_from_grad(1e-5, average_masses)
    orc.init_masses_from_grad(1e-5, average_masses)
    dev.synchronize()
    assert np.array_equal(dev.masses(), orc.masses()), "init masses differ"
    for x in (dev, orc):
        x.set_step_sizes(1.0 if step is None else step)
    if step is None:
        dev.adapt_step(seed, 11)
        orc.adapt_step(seed, 11)
        dev.synchronize()
        assert np.array_equal(dev.step_sizes(), orc.step_sizes()), \
            f"adapt_step differs: {dev.step_sizes()[:4]} vs {orc.step_sizes()[:4]}"
    for x in (dev, orc):
        x.seed_chains(seed + 1, 3)
    it = 0
    while it < warmup:
        n = min(fused, warmup - it)
        dev.warmup_step() if n == 1 else dev.warmup_steps(n)
        for _ in range(n):
            orc.warmup_step(8)
        it += n
        if it == warmup or (not lazy and (it % check_every == 0 or fused > 1)):
            assert_same_state(dev, orc, f"{model} D={D} warmup it={it - 1}", warm=True)
    dev.freeze()
    orc.freeze()
    dev.synchronize()
    assert same_bits_or_nan(dev.step_sizes(), orc.step_sizes()), "frozen step sizes differ"
    assert same_bits_or_nan(dev.inv_mass(), orc.inv_mass()), "frozen inverse mass differs"
    assert np.array_equal(dev.min_micro(), orc.min_micro().astype(np.int32)), "frozen min micro steps differ"
    it = 0
    while it < sampling:
        n = min(fused, sampling - it)
        dev.sample_step() if n == 1 else dev.sample_steps(n)
        for _ in range(n):
            orc.sample_step(8)
        it += n
        if it == sampling or (not lazy and (it % check_every == 0 or fused > 1)):
            assert_same_state(dev, orc, f"{model} D={D} sampling it={it - 1}", warm=False)
    return dev, orc


def reference_order_run(model: str, D: int, C: int, *, seed: int, id: int = 1, warmup: int, sampling: int,
                        init_radius: float = 2.0, smoothing: float = 1e-5, step_size_init: float = 1.0,
                        init_inv_metric=None, **cfg_over):
    """The oracle run the way the REFERENCE runs (walnutpy.cpp:134-222 -> run_sampler -> api.hpp:35-69): libm,
    left-to-right sums, mt19937_64 + libstdc++ distributions seeded exactly as the reference seeds them.
    Returns draws [C, warmup+sampling, D], step sizes, inverse metric."""
    _, om = MODELS[model]
    cfg = wno.default_config(rng_mode=wno.RNG_STD_MT64, math_mode=wno.MATH_LIBM, reduce_lanes=0, **cfg_over)
    o = wno.Engine(om, D, C, cfg, params=model_params(model, D))
    o.init_positions(seed, 1, init_radius)          # seed_seq{seed, 1}, walnutpy.cpp:187-189
    if init_inv_metric is not None:
        o.set_masses(np.broadcast_to(init_inv_metric, (C, D)))   # walnutpy.cpp:64-70 (handed to masses())
    else:
        o.init_masses_from_grad(smoothing)          # walnutpy.cpp:72
    o.set_step_sizes(step_size_init)
    o.adapt_step(seed, 2)                           # seed_seq{seed, 2}, walnutpy.cpp:75-80
    o.seed_chains(seed + id + C)                    # walnutpy.cpp:82, api.hpp:46-51
    draws = np.empty((C, warmup + sampling, D))
    trees = []
    for it in range(warmup):
        o.warmup_step(8)
        draws[:, it] = o.positions()
        trees.append((o.depths().copy(), o.rng_draws().copy()))
    o.freeze()
    for it in range(sampling):
        o.sample_step(8)
        draws[:, warmup + it] = o.positions()
        trees.append((o.depths().copy(), o.rng_draws().copy()))
    return draws, o.step_sizes(), o.inv_mass(), o.grad_evals(), trees


def check_reference_stream_run(model: str, D: int, C: int, *, seed: int, warmup: int, sampling: int, lib_path=None,
                               rtol: float = 1e-10, horizon: int = 0, init_inv_metric=None, init_radius: float = 2.0,
                               **cfg_over):
    """walnutpie_sample_device_reference_streams (the product's drop-in entry point fed the reference's own
    random streams) against the reference-order oracle at the same seed.  Tolerance: BASELINE.json's north star,
    <= 1e-10 relative on the per-chain state (the two sides differ in summation order and in the last ulp of
    exp/log).  MCMC trajectories amplify rounding differences from transition to transition (the survey calls
    long-run values "rounding-chaotic"), so the bound is asserted over the first `horizon` transitions (0 = all)
    and the growth curve is returned."""
    dm, _ = MODELS[model]
    mp = model_params(model, D)
    kw = dict(max_trajectory_doublings=cfg_over.get("max_trajectory_doublings", 5),
              max_step_halvings=cfg_over.get("max_step_halvings", 5),
              min_micro_steps=cfg_over.get("min_micro_steps", 1),
              max_hamiltonian_error=cfg_over.get("max_hamiltonian_error", 0.5))
    out = wa.walnuts_device(dm, model_params=mp, num_params=D, num_chains=C, seed=seed, id=1, min_warmup_iter=warmup,
                            max_warmup_iter=warmup, min_sampling_iter=sampling, max_sampling_iter=sampling,
                            save_warmup=True, save_inv_metric=True, reference_streams=True, lib_path=lib_path,
                            init_inv_metric=init_inv_metric, init_radius=init_radius, **kw)
    ref, steps, inv_metric, _, _ = reference_order_run(model, D, C, seed=seed, warmup=warmup, sampling=sampling,
                                                       init_inv_metric=init_inv_metric, init_radius=init_radius,
                                                       **cfg_over)
    assert np.any(ref[:, 1:] != ref[:, :-1]), "vacuous case: the chains never moved"
    worst = 0.0
    growth = np.zeros(warmup + sampling)
    for c in range(C):
        got = np.concatenate([out[c].warmup.warmup_draws, np.asarray(out[c])], axis=0)
        assert got.shape == ref[c].shape
        # relative to the state vector's magnitude: ||got - ref||_inf / ||ref||_inf per transition
        rel_t = np.max(np.abs(got - ref[c]), axis=1) / np.max(np.abs(ref[c]), axis=1)
        growth = np.maximum(growth, rel_t)
        rel = float(rel_t[:horizon].max()) if horizon else float(rel_t.max())
        worst = max(worst, rel)
        assert rel <= rtol, f"chain {c}: max relative difference {rel:.3e} > {rtol} within {horizon or len(rel_t)} transitions"
        if not horizon:
            assert abs(out[c].warmup.stepsize - steps[c]) <= rtol * steps[c]
            assert np.allclose(out[c].warmup.inv_metric, inv_metric[c], rtol=rtol, atol=0)
    return worst, growth


def run_weight_rebase_case(model: str, D: int, C: int, *, lib_path=None, geometry=None, **kw):
    """A run whose trees climb hundreds of units of log density (far-out starting points, no energy-error bound, i.e.
    plain NUTS): the device's span weights (walnuts_amd/csrc/wn_traj.h, "span weights") move their reference energy
    several times per chain -- every weight alive at that moment is rescaled -- and the result still equals the
    oracle's bit for bit.  The oracle counts the moves, so the case cannot pass vacuously."""
    dev, orc = run_case(model, D, C, lib_path=lib_path, geometry=geometry, init_scale=100.0,
                        max_hamiltonian_error=1e9, **kw)
    assert orc.weight_rebases() >= 3, orc.weight_rebases()
    return dev, orc


def run_pending_observation_case(model: str, D: int, C: int, *, lib_path=None, geometry=None):
    """Warmup launches back to back with NOTHING read in between (the register kernels leave the estimator's observation
    of a launch's last transition pending until the next launch's first prologue: wn_chip.h kDeferObservation), then every
    way out of the pending state -- a read of the estimator, new positions, a freeze -- against the oracle, which
    observes at the end of each transition (adaptive_walnuts.hpp:247-248)."""
    dev, orc = make_pair(model, D, C, lib_path, geometry, chain_groups=2)
    rng = np.random.default_rng(4)
    pos = rng.normal(0.0, 1.5, size=(C, D))
    for x in (dev, orc):
        x.set_positions(pos)
    dev.init_masses_from_grad(1e-5)
    orc.init_masses_from_grad(1e-5)
    for x in (dev, orc):
        x.set_step_sizes(0.3)
        x.seed_chains(5, 1)

    def both(n_launches, per_launch):
        for _ in range(n_launches):
            dev.warmup_steps(per_launch)      # no synchronize, no read: the observation stays pending across launches
            for _ in range(per_launch):
                orc.warmup_step(2)

    both(3, 1)
    both(2, 3)
    assert_same_state(dev, orc, "after five launches without a read", warm=True)   # (reads flush)
    both(2, 2)
    assert_same_state(dev, orc, "pending again, read again", warm=True)
    both(1, 2)
    new_pos = rng.normal(0.0, 1.0, size=(C, D))
    for x in (dev, orc):
        x.set_positions(new_pos)              # the pending observation is of the OLD positions: applied first
    both(2, 1)
    assert_same_state(dev, orc, "positions replaced while an observation was pending", warm=True)
    both(1, 3)
    dev.freeze()                              # freeze with an observation pending
    orc.freeze()
    dev.synchronize()
    assert same_bits_or_nan(dev.inv_mass(), orc.inv_mass()) and same_bits_or_nan(dev.step_sizes(), orc.step_sizes())
    for _ in range(2):
        dev.sample_steps(2)
        orc.sample_step(2)
        orc.sample_step(2)
    assert_same_state(dev, orc, "sampling after the freeze", warm=False)
