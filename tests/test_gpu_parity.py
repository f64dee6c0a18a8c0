"""GPU tier (-m gpu): the HIP path through the C ABI against the oracle, bit for bit, on seeded inputs
sized so that the oracle finishes in seconds, plus size-independent properties at BASELINE.json's full
sizes.  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

import parity
import walnuts_amd as wa

pytestmark = pytest.mark.gpu


def _host_variates_case(lib_path, D=96, C=8):
    dev, orc = parity.make_pair("std_normal", D, C, lib_path)
    rng = np.random.default_rng(5)
    pos = rng.normal(size=(C, D))
    for x in (dev, orc):
        x.set_positions(pos)
        x.set_step_sizes(0.4)
        x.seed_chains(1, 0)
    for it in range(3):
        z, u = rng.normal(size=(C, D)), rng.uniform(size=(C, 64))
        for x in (dev, orc):
            x.set_variates(z, u)
            x.warmup_step()
        parity.assert_same_state(dev, orc, f"variates warmup {it}", warm=True)
    dev.freeze()
    orc.freeze()
    for it in range(3):
        z, u = rng.normal(size=(C, D)), rng.uniform(size=(C, 64))
        for x in (dev, orc):
            x.set_variates(z, u)
            x.sample_step()
        parity.assert_same_state(dev, orc, f"variates sampling {it}", warm=False)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu):
    return gpu


# ---- leapfrog + tree + adaptation, every launch geometry, both arithmetic modes -------------------------
# fma = 1: the integrator's multiply-adds fused (wn_config::fused_multiply_add, the default), the oracle replaying
# std::fma at the same places; fma = 0: every product rounded (the reference's x86-64 element-wise bits)
@pytest.mark.parametrize("fma", [1, 0])
@pytest.mark.parametrize("model,D,C,geometry", [
    ("std_normal", 100, 64, None),          # BASELINE config #1 shape: (1,2)
    ("std_normal", 3, 16, None),            # odd tiny D, heavy padding
    ("std_normal", 256, 64, (1, 4)),
    ("std_normal", 500, 48, (1, 8)),
    ("std_normal", 1024, 128, (4, 4)),      # headline geometry
    ("std_normal", 1024, 64, (1, 16)),      # one wavefront per chain
    ("std_normal", 1024, 64, (2, 8)),
    ("std_normal", 1000, 32, (4, 8)),
    ("std_normal", 2048, 32, (8, 4)),
    ("std_normal", 4096, 24, (8, 8)),
    ("std_normal", 4000, 16, (16, 4)),
    ("std_normal", 2000, 24, None),         # the default beyond 1 024 dimensions: two wavefronts, 16 elements per lane
    ("diag_normal", 4096, 12, None),        # ... four wavefronts, 16 elements per lane
    ("funnel", 3000, 12, (4, 16)),
    ("rw1", 2048, 12, (2, 16)),
    ("std_normal", 8192, 12, (16, 8)),      # the largest register geometry
    ("diag_normal", 1024, 96, None),        # config #2/#4 family
    ("diag_normal", 130, 64, (1, 4)),
    ("funnel", 128, 128, None),             # config #3
    ("funnel", 1000, 32, (4, 4)),           # cross-wavefront reductions inside the model
    ("funnel", 1024, 48, None),             # general-gradient path at the headline dimension: (1,16)
    ("rw1", 1024, 48, None),                # 4th model, added through csrc/models/rw1.h: neighbour-coupled gradient
    ("rw1", 200, 64, (2, 2)),               # ... wavefront edges through LDS
    ("rw1", 2000, 16, (4, 8)),               # ... pair-row wrap-around across four wavefronts
    ("diag_normal", 16384, 12, None),       # config #4 dimension: streaming backend, the moving end held in registers
    ("diag_normal", 16384, 8, (16, -1)),    # ... sixteen wavefronts streaming both ends (the kernels beyond 16 384 dimensions)
    ("std_normal", 12100, 6, None),         # held moving end, 12 of the 16 tiles, the last one ragged
    ("diag_normal", 6000, 8, None),         # ... the default from 4 097 parameters for one-pass gradients
    ("funnel", 9000, 6, None),              # held moving end, two passes per micro step (no halo)
    ("rw1", 9000, 6, None),                 # ... with halo reads: wavefront-edge elements through LDS, the rest by lane shuffles
    ("rw1", 4500, 8, None),                 # ... the default for rw1 from 4 097 parameters
    ("std_normal", 20000, 8, (8, -1)),      # streaming, 8 wavefronts per chain, ragged last tile
    ("std_normal", 1000, 24, (2, -1)),      # streaming forced at a small dimension
    ("diag_normal", 5000, 12, (16, -1)),
    ("funnel", 16384, 8, None),             # streaming for a gradient that needs sums over all coordinates (two passes)
    ("rw1", 12000, 8, None),                # ... and one that needs neighbouring coordinates (halo reads)
    ("rw1", 3000, 12, (4, -1)),
    ("funnel", 2000, 12, (2, -1)),
])
def test_engine_matches_oracle_bitwise(model, D, C, geometry, fma):
    parity.run_case(model, D, C, warmup=12, sampling=8, geometry=geometry, check_every=2, fused_multiply_add=fma)


# ---- several transitions per launch (wn_engine_warmup_steps / wn_engine_sample_steps) ---------------------------------
@pytest.mark.parametrize("model,D,C,geometry,fused", [
    ("std_normal", 1024, 2500, (1, 16), 4),   # headline kernel, more chains than resident workgroups (1 024): the shared
                                              # counter hands every chain to ONE workgroup for all of its transitions
    ("std_normal", 100, 64, None, 8),
    ("std_normal", 1024, 64, (2, 8), 3),      # two wavefronts per chain: barrier between the transitions
    ("diag_normal", 1024, 96, None, 5),
    ("funnel", 128, 300, None, 8),            # config #3's kernel
    ("funnel", 1000, 32, (4, 4), 2),
    ("rw1", 1024, 48, None, 4),
    ("diag_normal", 16384, 12, None, 3),      # streaming backend
    ("funnel", 2000, 12, (2, -1), 4),         # streaming, two passes per micro step
])
def test_fused_launches_match_oracle_bitwise(model, D, C, geometry, fused):
    """The device runs `fused` transitions of every chain per launch, the oracle single steps: same bits after every
    launch, through adaptive warmup (Adam, the estimator's discount by iteration number) and sampling."""
    parity.run_case(model, D, C, warmup=2 * fused + 1, sampling=2 * fused + 1, geometry=geometry, fused=fused)


@pytest.mark.parametrize("model,D,C,geometry,fused", [
    ("std_normal", 1024, 301, (1, 16), 4),     # uneven halves on the headline kernel
    ("funnel", 128, 500, None, 8),             # config #3's kernel (the engine's own choice at this size is two groups)
    ("rw1", 1024, 48, None, 1),                # four wavefronts per chain, single steps
    ("diag_normal", 16384, 12, None, 3),       # streaming backend: one arena slice per group
])
def test_chain_groups_match_oracle_bitwise(model, D, C, geometry, fused):
    """wn_config::chain_groups = 2: the chains as two independently launched halves on two streams -- own chain
    counters, arena slices, no ordering between one half's launch n + 1 and the other's launch n -- against the oracle
    (which knows nothing of groups) after every launch, through warmup, freeze and sampling."""
    parity.run_case(model, D, C, warmup=2 * fused + 1, sampling=2 * fused + 1, geometry=geometry, fused=fused,
                    chain_groups=2)


def test_stream_ordering_without_adopting_the_stream():
    """wn_engine_wait_stream / _release_stream: a consumer on another stream (the place of the draws' collective at
    N > 1) reads two alternating draw buffers while the engine -- two chain groups on its own streams -- runs ahead;
    what the consumer saw equals the draws of a plain run, launch by launch."""
    import torch
    D, C, T, L = 1024, 4096, 2, 12

    def engine(groups):
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C, wa.default_config(chain_groups=groups))
        e.init_positions(3, 0, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(0.3)
        e.seed_chains(4, 0)
        e.freeze()
        return e

    ref = engine(1)
    want = torch.empty((L, T, C, D), dtype=torch.float64, device="cuda")
    for k in range(L):
        ref.sample_steps(T, want[k].data_ptr(), D, C * D)
    ref.synchronize()

    e = engine(2)
    assert e.chain_groups == 2
    side = torch.cuda.Stream()
    bufs = [torch.empty((T, C, D), dtype=torch.float64, device="cuda") for _ in range(2)]
    seen = torch.zeros((L, T, C, D), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for k in range(L):
        e.wait_stream(side.cuda_stream)          # the consumer's last read of bufs[k & 1] (launch k - 2) is done
        e.sample_steps(T, bufs[k & 1].data_ptr(), D, C * D)
        e.release_stream(side.cuda_stream)       # the consumer runs behind launch k
        with torch.cuda.stream(side):
            seen[k].copy_(bufs[k & 1])
            for _ in range(3):                   # (a slow consumer: the engine would overtake it without the waits)
                seen[k].add_(bufs[k & 1]).sub_(bufs[k & 1])
    e.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(seen, want)


def test_fused_launches_full_size_headline():
    """65 536 x 1 024: one launch of 8 transitions leaves the positions, statistics and EVERY draw plane that 8
    launches of one transition leave."""
    import torch
    D, C, T = 1024, 65536, 8

    def engine():
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C)
        e.init_positions(3, 0, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(0.3)
        e.seed_chains(4, 0)
        for _ in range(2):
            e.warmup_step()
        return e

    a, b = engine(), engine()
    a.warmup_steps(3)
    for _ in range(3):
        b.warmup_step()
    a.freeze(), b.freeze()
    pa = torch.empty((T, C, D), dtype=torch.float64, device="cuda")
    pb = torch.empty((T, C, D), dtype=torch.float64, device="cuda")
    a.sample_steps(T, pa.data_ptr(), D, C * D)
    for k in range(T):
        b.sample_step(pb[k].data_ptr(), D)
    a.synchronize(), b.synchronize()
    assert torch.equal(pa, pb)
    assert np.array_equal(a.positions(), b.positions())
    assert np.array_equal(a.positions(), pa[T - 1].cpu().numpy())
    for f in ("logp", "depths", "grad_evals", "rng_draws", "step_sizes"):
        assert np.array_equal(getattr(a, f)(), getattr(b, f)()), f
    assert a.rhat() == b.rhat()
    a.check(), b.check()


@pytest.mark.parametrize("model,D,C,geometry,lds", [
    ("std_normal", 1024, 96, (2, 8), 2),    # two LDS vectors, the rest overflows to the HBM arena
    ("std_normal", 1024, 64, (2, 8), 0),     # the whole span pool in the HBM arena
    ("diag_normal", 1000, 64, (4, 4), 1),
    ("funnel", 300, 48, (2, 4), 0),
    ("std_normal", 100, 64, None, 3),
])
def test_span_pool_tiers_match_oracle_bitwise(model, D, C, geometry, lds):
    """Where a span-pool vector lives (LDS, HBM arena) must not change a single bit."""
    parity.run_case(model, D, C, warmup=10, sampling=6, geometry=geometry, lds_vectors=lds, check_every=2)


@pytest.mark.parametrize("kw", [
    dict(model="std_normal", D=1, C=5),                                   # one parameter
    dict(model="std_normal", D=2, C=1),                                   # one chain
    dict(model="funnel", D=2, C=7),                                       # smallest funnel
    dict(model="std_normal", D=65, C=9, max_trajectory_doublings=1),      # a single doubling
    dict(model="std_normal", D=65, C=9, max_step_halvings=1, step=1.7),   # no halving allowed: many failed leaves
    dict(model="diag_normal", D=129, C=9, min_micro_steps=3),             # several micro steps per macro step
    dict(model="std_normal", D=64, C=9, max_hamiltonian_error=1e-3, step=0.9),   # nearly every level rejected
    dict(model="std_normal", D=64, C=9, max_trajectory_doublings=10, step=0.01, max_hamiltonian_error=50.0),
])
def test_edge_configurations_match_oracle(kw):
    kw = dict(kw)
    parity.run_case(kw.pop("model"), kw.pop("D"), kw.pop("C"), warmup=5, sampling=5, **kw)


@pytest.mark.parametrize("model,D,C,geometry,kw", [
    ("std_normal", 1024, 24, None, dict(step=0.3)),                         # the headline kernel
    ("std_normal", 1024, 24, None, dict(step=0.3, fused_multiply_add=0)),
    ("std_normal", 200, 16, (2, 2), dict(step=1.0)),                        # two wavefronts: each keeps its own stack
    ("diag_normal", 1000, 16, (4, 4), dict(step=0.3)),
    ("diag_normal", 9000, 6, None, dict(step=0.3)),                         # streaming backend
    ("std_normal", 100, 32, None, dict(step=0.3, lds_vectors=1)),           # deep stacks, most of them in the arena
])
def test_span_weights_move_their_reference_energy(model, D, C, geometry, kw):
    """combine() in the linear domain (wn_traj.h, "span weights") on trees whose energies leave the range a fixed
    reference can carry (far-out starts, no energy-error bound): the reference moves, every live weight is rescaled,
    and the chains equal the oracle's bit for bit; the oracle counts the moves (parity.run_weight_rebase_case)."""
    parity.run_weight_rebase_case(model, D, C, geometry=geometry, warmup=2, sampling=4, max_trajectory_doublings=6, **kw)


def test_non_finite_energies_follow_ieee_like_the_reference():
    """Positions so large that the energies overflow: logp = -inf / NaN comparisons must take the reference's
    branches (walnuts.hpp:339: a NaN difference is never <= max_error, so no such state is ever merged and the rules
    of util.hpp:176-181 are never reached), the transition ends where the reference's would and the chain keeps its
    state."""
    D, C = 32, 6
    dev, orc = parity.make_pair("std_normal", D, C)
    pos = np.random.default_rng(0).normal(size=(C, D))
    pos[0] *= 1e160   # x*x overflows -> logp = -inf
    pos[1] *= 1e200
    pos[2, 3] = 1e308
    for x in (dev, orc):
        x.set_positions(pos)
        x.set_step_sizes(0.25)
        x.seed_chains(8, 0)
    dev.freeze()
    orc.freeze()
    for it in range(3):
        dev.sample_step()
        orc.sample_step()
        dev.synchronize()
        a, b = dev.positions(), orc.positions()
        assert np.array_equal(a, b, equal_nan=True), it
        assert np.array_equal(dev.depths(), orc.depths()) and np.array_equal(dev.grad_evals(), orc.grad_evals())
        assert np.array_equal(dev.logp(), orc.logp(), equal_nan=True)


@pytest.mark.parametrize("model,D,C,warm,samp,horizon,unit_mass", [
    ("std_normal", 100, 4, 40, 20, 12, False),    # BASELINE config #1's model and shape, the reference's seeding
    ("diag_normal", 257, 6, 25, 10, 12, False),
    ("funnel", 16, 8, 25, 10, 12, False),
    ("std_normal", 1024, 4, 15, 10, 12, True),    # headline dimension, unit initial metric
    ("std_normal", 1024, 4, 15, 10, 1, False),    # gradient-based initial masses: see the docstring
])
def test_reference_streams_track_the_reference_order_oracle(model, D, C, warm, samp, horizon, unit_mass):
    """BASELINE.json north star: per-chain state matches the reference CPU path at fixed seed to <= 1e-10
    relative.  The product's drop-in entry point is fed the reference's own mt19937_64 streams; the oracle runs
    in reference order (libm, left-to-right sums, same seeding as walnutpy.cpp).  One transition from identical
    inputs agrees to ~1e-15; over consecutive transitions the two summation orders drift apart the way any two
    orders would (the survey: "long-run values are rounding-chaotic").  With the reference's gradient-based
    initial masses (mass ~ |x_i| can be ~1e-5) some coordinates are stiff by 1e5 and amplify the last-bit
    differences within a few transitions, so that case asserts the single-transition bound only; all cases
    print the measured growth."""
    metric = np.ones(D) if unit_mass else None
    worst, growth = parity.check_reference_stream_run(model, D, C, seed=48, warmup=warm, sampling=samp,
                                                      horizon=horizon, init_inv_metric=metric,
                                                      init_radius=1.0 if unit_mass else 2.0)
    assert growth[0] <= 1e-13
    print(f"{model} D={D}: worst relative difference in the first {horizon} transitions = {worst:.3e}; "
          f"per transition: {' '.join('%.0e' % g for g in growth)}")


def test_device_side_initialisation_matches_oracle():
    parity.run_case("std_normal", 777, 40, warmup=3, sampling=2, init="device")
    parity.run_case("funnel", 64, 40, warmup=3, sampling=2, init="device")


def test_averaged_initial_masses_match_oracle():
    # InitConfigBuilder::masses(logp_grad, s, average_masses=true), config.hpp:371-380 (also streaming backend)
    parity.run_case("diag_normal", 300, 48, warmup=3, sampling=2, init="device", average_masses=True)
    parity.run_case("diag_normal", 9000, 5, warmup=2, sampling=1, init="device", average_masses=True)


def test_ill_conditioned_config2_long_warmup():
    # sigma_d = d+1 (examples/examples.cpp:20-31), D=1024: the config #2 model on a chain subset
    D, C = 1024, 32
    s2 = np.array([(d + 1.0) ** 2 for d in range(D)])
    parity.MODELS  # noqa
    old = parity.model_params
    parity.model_params = lambda m, d: s2 if m == "diag_normal" else None
    try:
        parity.run_case("diag_normal", D, C, warmup=40, sampling=10, check_every=10)
    finally:
        parity.model_params = old


def test_span_pool_in_lds_or_hbm_gives_identical_chains():
    outs = []
    for lds in (-1, 4, 0):
        dev, _ = parity.run_case("std_normal", 512, 32, warmup=6, sampling=6, lds_vectors=lds, check_every=6)
        outs.append(dev.positions())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_deep_trees_and_halvings():
    parity.run_case("std_normal", 64, 64, warmup=0, sampling=6, step=0.02, max_trajectory_doublings=9,
                    check_every=3)  # tiny step: trees reach depth 9 (511 leaves)
    parity.run_case("std_normal", 64, 64, warmup=0, sampling=6, step=2.9, max_trajectory_doublings=4,
                    check_every=3)  # huge step: halving levels + reversibility checks
    parity.run_case("funnel", 32, 64, warmup=10, sampling=10, step=1.5, max_step_halvings=8, check_every=5)
    # adaptive warmup with a tiny first step: hundreds of macro steps per transition, so Adam's batched update
    # (64 observations per flush) takes its mid-transition flushes as well as the one at the end
    parity.run_case("std_normal", 64, 32, warmup=4, sampling=2, step=0.02, max_trajectory_doublings=8, check_every=1)


@pytest.mark.parametrize("kw", [
    dict(step=0.02, max_trajectory_doublings=9),                    # deep trees: the span pool hands buffers around a lot
    dict(step=2.9, max_trajectory_doublings=4),                     # halving levels + reversibility re-integrations
    dict(step=0.7, min_micro_steps=2, max_trajectory_doublings=6),  # multi-step leaves: no fused level-0 U-turn
    dict(step=1.7, max_step_halvings=1),                            # failed leaves end the transition
    dict(step=0.9, max_hamiltonian_error=1e-3),                     # nearly every level rejected
    dict(step=0.4, max_trajectory_doublings=1),                     # a single doubling
])
def test_streaming_backend_edge_configurations(kw):
    """The HBM-streaming kernels (zero-copy span pool, fused level-0 U-turn, recomputed gradient) on the same edge
    cases as the register kernels, warmup included."""
    parity.run_case("diag_normal", 700, 12, warmup=4, sampling=5, geometry=(2, -1), **kw)
    parity.run_case("std_normal", 9000, 3, warmup=2, sampling=3, **kw)           # default: streaming above 8192


def test_model_geometry_hint():
    """A model may state the elements per lane it prefers (wn_model_api.h: kPreferredElemsPerLane, or
    preferred_elems_per_lane(num_params) where the best width depends on the dimension -- models/rw1.h: the default
    policy up to 1 024 dimensions, 4 per lane up to 2 048, 8 up to 4 096): the engine takes the fewest wavefronts that
    hold num_params at that width, an explicit request wins."""
    cfg = wa.default_config()
    assert wa.DeviceEngine(wa.MODEL_RW1, 1024, 8, cfg).lanes == 64           # default policy there: 1 x 16
    assert wa.DeviceEngine(wa.MODEL_RW1, 1500, 8, cfg).lanes == 512          # 8 wavefronts x 4 elements per lane
    assert wa.DeviceEngine(wa.MODEL_RW1, 3000, 8, cfg).lanes == 512          # 8 x 8
    assert wa.DeviceEngine(wa.MODEL_FUNNEL, 3000, 8, cfg).lanes == 512       # the funnel's hint: 8 x 8 there
    assert wa.DeviceEngine(wa.MODEL_FUNNEL, 2000, 8, cfg).lanes == 128       # ... and the default 2 x 16 below
    assert wa.DeviceEngine(wa.MODEL_STD_NORMAL, 1024, 8, cfg).lanes == 64    # default policy: 1 x 16
    cfg = wa.default_config(waves_per_chain=4, elems_per_lane=4)
    assert wa.DeviceEngine(wa.MODEL_RW1, 1024, 8, cfg).lanes == 256


def test_randomised_parity_campaign():
    """A short fixed-seed run of the randomised campaign (tests/gpu_probes/fuzz_parity.py: random model, dimension,
    geometry, pool tiers, step size, depth / halving / micro-step limits): every case bit-exact against the oracle.
    The long runs of the round are filed as profiles/r02/fuzz_parity.txt (222 056 cases, 0 failing)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_probes", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    done, failed, _ = fz.campaign(seed=3, seconds=15.0)
    assert done > 100 and not failed, failed[:3]


def test_host_supplied_variates_path():
    """kRngBuffer: normals and canonical uniforms supplied by the host (the hook for exact libstdc++-stream
    runs), compared with the oracle fed the same variates."""
    _host_variates_case(None)


# ---- size-independent properties at full BASELINE sizes --------------------------------------------------
@pytest.mark.parametrize("model,D,C,geometry", [("std_normal", 1024, 96, None), ("diag_normal", 1000, 40, None),
                                                ("funnel", 128, 200, None), ("rw1", 1024, 24, None),
                                                ("funnel", 1000, 24, (4, 4)), ("std_normal", 4096, 12, (8, 8)),
                                                ("diag_normal", 3000, 12, None), ("funnel", 2048, 12, None),
                                                ("diag_normal", 6000, 6, None)])
def test_observation_pending_between_warmup_launches(model, D, C, geometry):
    """wn_chip.h kDeferObservation / wn_engine flush_pending_observation on the device: launches with nothing read in
    between, then a read, new positions and a freeze with an observation pending (tests/parity.py)."""
    parity.run_pending_observation_case(model, D, C, geometry=geometry)


def test_device_normals_moments_and_tails():
    """The device's uniform-to-normal map (Philox4x32-7 -> 52-bit open-interval uniforms -> Box-Muller with the portable
    log / sincospi and the range-free square root, wn_devmath.h) vetted statistically, not only bit-mirrored: 6.7e7
    standard normals drawn by the engine (initial positions at scale 1: the same stream_normal_pair the momentum refresh
    calls) against the normal law -- mean, variance, skewness, kurtosis within 5 standard errors, the counts beyond 3, 4
    and 5 sigma within 5 Poisson deviations of their expectations, the largest |z| where the extreme-value law of 6.7e7
    normals puts it, and no correlation between the two members of a pair or between neighbouring pairs."""
    from math import erfc, sqrt
    C, D = 65536, 1024
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C)
    e.init_positions(20261003, 0, 1.0)
    e.synchronize()
    z = e.positions()
    n = z.size
    m1 = z.mean()
    zc = z - m1
    var = float(np.mean(zc * zc))
    skew = float(np.mean(zc ** 3)) / var ** 1.5
    kurt = float(np.mean(zc ** 4)) / var ** 2
    assert abs(m1) < 5 / sqrt(n)
    assert abs(var - 1) < 5 * sqrt(2 / n)
    assert abs(skew) < 5 * sqrt(6 / n)
    assert abs(kurt - 3) < 5 * sqrt(24 / n)
    a = np.abs(z)
    for k in (3.0, 4.0, 5.0):
        expect = n * erfc(k / sqrt(2))
        got = int(np.count_nonzero(a > k))
        assert abs(got - expect) < 5 * sqrt(expect) + 1, (k, got, expect)
    assert 5.0 < a.max() < 6.7, a.max()      # P(max |z| of 6.7e7 > 6.7) = 1.4e-3, P(< 5.0) < 1e-16
    pairs = z.reshape(C, D // 2, 2)
    assert abs(float(np.mean(pairs[:, :, 0] * pairs[:, :, 1]))) < 5 / sqrt(n / 2)          # cos / sin members of a pair
    assert abs(float(np.mean(pairs[:, :-1, 1] * pairs[:, 1:, 0]))) < 5 / sqrt(n / 2)       # neighbouring counters
    assert abs(float(np.mean(z[:-1, :] * z[1:, :]))) < 5 / sqrt(n)                         # neighbouring chains


def test_range_free_square_root_equals_sqrt_on_the_device():
    """wnd::sqrt_normal (wn_devmath.h): the compiler's fp64 sqrt refinement without its range scaling and special-case
    patches -- the Box-Muller radius and the warmup prologue's inverse mass / Cholesky factor use it.  Bit for bit equal
    to the correctly rounded square root on 2.4e7 arguments: dense around the radicand's range [2^-52, 73], every
    binade from 2^-760 to 2^1023, values one ulp either side of perfect squares (the hard cases of a last-bit
    correction); the checked variant also on 0, -0, inf, NaN and negative arguments."""
    import ctypes as C
    lib = wa.load_library()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(8)

    def device_sqrt(x, checked):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        assert lib.wn_internal_sqrt_probe(x.ctypes.data_as(dp), y.ctypes.data_as(dp), x.size, checked) == 0
        return y

    squares = rng.uniform(1.0, 2.0, size=1 << 20) ** 2
    args = np.concatenate([
        -2.0 * np.log(rng.uniform(size=1 << 23)),                                     # the radicand's own law
        np.exp(rng.uniform(np.log(2.0 ** -52), np.log(73.0), size=1 << 22)),
        rng.uniform(1.0, 4.0, size=1 << 22) * 2.0 ** rng.integers(-760, 1022, size=1 << 22),
        squares, np.nextafter(squares, 0.0), np.nextafter(squares, 10.0),
    ])
    want = np.sqrt(args)
    for checked in (0, 1):
        got = device_sqrt(args, checked)
        assert np.array_equal(got, want), int(np.sum(got != want))
    special = np.array([0.0, -0.0, np.inf, np.nan, -1.0, -np.inf, 4.0])
    with np.errstate(invalid="ignore"):
        want = np.sqrt(special)
    got = device_sqrt(special, 1)
    assert np.array_equal(got, want, equal_nan=True) and np.signbit(got[1])


def _logp_std_normal(x):
    return -0.5 * np.einsum("ij,ij->i", x, x)


def test_full_size_headline_properties():
    """65 536 chains x 1 024 dims (BASELINE headline): determinism, chain independence (a chain's result is
    a function of (seed, chain id, state) only, however chains are batched) and reported logp == logp(theta)."""
    D, C = 1024, 65536
    rng = np.random.default_rng(9)
    pos = rng.normal(0, 1, size=(C, D))

    def run(chains, offset):
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, len(chains))
        e.set_positions(pos[chains])
        e.set_step_sizes(0.35)
        e.seed_chains(77, offset)
        e.freeze()
        for _ in range(3):
            e.sample_step()
        e.synchronize()
        return e.positions(), e.logp(), e.grad_evals(), e.depths()

    full = run(np.arange(C), 0)
    again = run(np.arange(C), 0)
    for a, b in zip(full, again):
        assert np.array_equal(a, b)
    sub = np.arange(1000, 1000 + 512)
    part = run(sub, 1000)
    for a, b in zip(full, part):
        assert np.array_equal(a[sub], b)
    lp = _logp_std_normal(full[0])
    assert np.allclose(full[1], lp, rtol=1e-12, atol=0)
    assert full[2].min() >= 3 * 2 and full[3].min() >= 1 and full[3].max() <= 6


def test_full_size_high_dim_properties():
    """8 192 chains x 8 192 dims, diagonal Gaussian (the register-resident kernels' largest D)."""
    D, C = 8192, 8192
    s2 = np.array([(1.0 + (d % 16)) ** 2 for d in range(D)])
    e = wa.DeviceEngine(wa.MODEL_DIAG_NORMAL, D, C, params=s2)
    e.init_positions(3, 0, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(3, 0)
    e.seed_chains(4, 0)
    for _ in range(3):
        e.warmup_step()
    e.freeze()
    e.sample_step()
    e.synchronize()
    x = e.positions()
    lp = np.zeros(C)
    for lo in range(0, C, 1024):
        xs = x[lo:lo + 1024]
        lp[lo:lo + 1024] = np.sum(-0.5 * xs * xs / s2, axis=1)
    assert np.allclose(e.logp(), lp, rtol=1e-11, atol=0)
    assert np.all(np.isfinite(x)) and e.depths().min() >= 1
    assert np.all(e.step_sizes() > 0) and np.all(np.isfinite(e.inv_mass()))


def test_full_size_config4_properties():
    """BASELINE config #4: 8 192 chains x 16 384-dim diagonal Gaussian on the streaming kernels."""
    D, C = 16384, 8192
    s2 = np.array([(1.0 + (d % 16)) ** 2 for d in range(D)])
    e = wa.DeviceEngine(wa.MODEL_DIAG_NORMAL, D, C, params=s2)
    assert e.streaming
    e.init_positions(3, 0, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(3, 0)
    e.seed_chains(4, 0)
    for _ in range(2):
        e.warmup_step()
    e.freeze()
    e.sample_step()
    e.synchronize()
    sub = slice(0, 256)
    x = e.positions()[sub]
    assert np.allclose(e.logp()[sub], np.sum(-0.5 * x * x / s2, axis=1), rtol=1e-11, atol=0)
    assert np.all(np.isfinite(x)) and e.depths().min() >= 1 and np.all(e.step_sizes() > 0)
    # chain independence: a sub-batch with the same global chain ids reproduces the same rows
    e2 = wa.DeviceEngine(wa.MODEL_DIAG_NORMAL, D, 64, params=s2)
    e2.init_positions(3, 100, 2.0)
    e2.init_masses_from_grad(1e-5)
    e2.set_step_sizes(1.0)
    e2.adapt_step(3, 100)
    e2.seed_chains(4, 100)
    for _ in range(2):
        e2.warmup_step()
    e2.freeze()
    e2.sample_step()
    assert np.array_equal(e2.positions(), e.positions()[100:164])


def _full_size_run(model_id, D, chains, offset, s2, warm, samp, fused=1):
    """InitConfigBuilder on the device keyed by GLOBAL chain ids -> `warm` adaptive + `samp` sampling transitions."""
    e = wa.DeviceEngine(model_id, D, chains, params=s2)
    e.init_positions(3, offset, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(3, offset)
    e.seed_chains(4, offset)
    for i in range(0, warm, fused):
        e.warmup_steps(min(fused, warm - i))
    e.freeze()
    for i in range(0, samp, fused):
        e.sample_steps(min(fused, samp - i))
    e.synchronize()
    e.check()
    return e


def test_full_size_config2_properties(oracle):
    """BASELINE config #2 at full size: 4 096 chains x 1 024-dim ill-conditioned diagonal Gaussian, sigma_d = d + 1
    (examples/examples.cpp:20-31).  Size-independent properties: the reported log density is the model's at the reported
    position; chain independence (a sub-batch with the same global chain ids reproduces its rows bit for bit, whatever
    the batching and the chain groups); the warmup adapts the metric towards sigma^2 on every chain; and 64 of the 4 096
    chains meet the device-order oracle bit for bit."""
    D, C = 1024, 4096
    s2 = np.array([(d + 1.0) ** 2 for d in range(D)])
    e = _full_size_run(wa.MODEL_DIAG_NORMAL, D, C, 0, s2, warm=24, samp=8, fused=8)
    x = e.positions()
    assert np.allclose(e.logp(), np.sum(-0.5 * x * x / s2, axis=1), rtol=1e-11, atol=0)
    assert np.all(np.isfinite(x)) and e.depths().min() >= 1 and e.depths().max() <= 6
    assert np.all(e.step_sizes() > 0) and np.all(np.isfinite(e.inv_mass()))
    sub = _full_size_run(wa.MODEL_DIAG_NORMAL, D, 64, 2000, s2, warm=24, samp=8, fused=8)
    for a, b in ((sub.positions(), x[2000:2064]), (sub.logp(), e.logp()[2000:2064]),
                 (sub.grad_evals(), e.grad_evals()[2000:2064]), (sub.inv_mass(), e.inv_mass()[2000:2064])):
        assert np.array_equal(a, b)
    # the same 64 chains on the oracle (device order): bit for bit
    ocfg = oracle.default_config(rng_mode=oracle.RNG_PHILOX, math_mode=oracle.MATH_PORTABLE, reduce_lanes=sub.lanes,
                                 fma=int(sub.cfg.fused_multiply_add))
    o = oracle.Engine(oracle.MODEL_DIAG_NORMAL, D, 64, ocfg, params=s2)
    o.init_positions(3, 2000, 2.0)
    o.init_masses_from_grad(1e-5)
    o.set_step_sizes(1.0)
    o.adapt_step(3, 2000)
    o.seed_chains(4, 2000)
    for _ in range(24):
        o.warmup_step(8)
    o.freeze()
    for _ in range(8):
        o.sample_step(8)
    assert np.array_equal(o.positions(), sub.positions()) and np.array_equal(o.logp(), sub.logp())
    assert np.array_equal(o.grad_evals(), sub.grad_evals())


def test_full_size_config3_properties(oracle):
    """BASELINE config #3 at full size: Neal's funnel D = 128, 16 384 chains (adaptive step size / divergence stress).
    Properties: reported log density = the funnel's at the reported position; chain independence; every chain's
    failed-extension flag is 0 or 1; step sizes adapt downwards from 1; 128 of the chains meet the device-order oracle
    bit for bit through the adaptive and the sampling transitions."""
    D, C = 128, 16384
    e = _full_size_run(wa.MODEL_FUNNEL, D, C, 0, None, warm=40, samp=8, fused=8)
    x = e.positions()
    lp = -0.5 * x[:, 0] ** 2 / 9.0 - 0.5 * np.exp(-x[:, 0]) * np.sum(x[:, 1:] ** 2, axis=1) - 0.5 * (D - 1) * x[:, 0]
    assert np.allclose(e.logp(), lp, rtol=1e-10, atol=1e-9)
    assert np.all(np.isfinite(x)) and e.depths().min() >= 1
    st = e.step_sizes()
    assert np.all(st > 0) and np.median(st) < 1.0
    assert set(np.unique(e.failed_extensions()).tolist()) <= {0, 1}
    sub = _full_size_run(wa.MODEL_FUNNEL, D, 128, 9000, None, warm=40, samp=8, fused=8)
    for a, b in ((sub.positions(), x[9000:9128]), (sub.logp(), e.logp()[9000:9128]),
                 (sub.grad_evals(), e.grad_evals()[9000:9128]), (sub.step_sizes(), st[9000:9128])):
        assert np.array_equal(a, b)
    ocfg = oracle.default_config(rng_mode=oracle.RNG_PHILOX, math_mode=oracle.MATH_PORTABLE, reduce_lanes=sub.lanes,
                                 fma=int(sub.cfg.fused_multiply_add))
    o = oracle.Engine(oracle.MODEL_FUNNEL, D, 128, ocfg)
    o.init_positions(3, 9000, 2.0)
    o.init_masses_from_grad(1e-5)
    o.set_step_sizes(1.0)
    o.adapt_step(3, 9000)
    o.seed_chains(4, 9000)
    for _ in range(40):
        o.warmup_step(8)
    o.freeze()
    for _ in range(8):
        o.sample_step(8)
    assert np.array_equal(o.positions(), sub.positions()) and np.array_equal(o.logp(), sub.logp())
    assert np.array_equal(o.grad_evals(), sub.grad_evals()) and np.array_equal(o.step_sizes(), sub.step_sizes())


def test_full_size_config5_shard_properties():
    """BASELINE config #5's shape on the one GPU there is: 262 144 chains x 1 024-dim normal as ONE engine, and rank 3's
    shard of an 8-GPU run (chains 98 304 .. 131 071, `chain_offset` = its first global id) as an engine of its own --
    the shard's rows equal the big engine's bit for bit through adaptive and sampling launches (what makes the 8-GPU run
    a pure partition: SURVEY.md section 8e), the draw planes of a fused launch land where the all-gather expects them
    ([T][rows][D]), and the reported log densities are the model's.  The exchange itself (RCCL over xGMI) needs 8 GPUs."""
    import torch
    D, C, R = 1024, 262144, 8
    lo, n = 3 * (C // R), C // R

    def run(chains, offset, draws=None):
        e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, chains)
        e.init_positions(5, offset, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(1.0)
        e.adapt_step(5, offset)
        e.seed_chains(6, offset)
        e.warmup_steps(8)
        e.warmup_steps(4)
        e.freeze()
        if draws is None:
            e.sample_steps(4)
        else:
            e.sample_steps(4, draws.data_ptr(), D, chains * D)
        e.synchronize()
        e.check()
        return e

    planes = torch.empty((4, n, D), dtype=torch.float64, device="cuda")
    shard = run(n, lo, planes)
    whole = run(C, 0)
    x = whole.positions()[lo:lo + n]
    assert np.array_equal(shard.positions(), x)
    assert np.array_equal(shard.logp(), whole.logp()[lo:lo + n])
    assert np.array_equal(shard.grad_evals(), whole.grad_evals()[lo:lo + n])
    assert np.array_equal(shard.step_sizes(), whole.step_sizes()[lo:lo + n])
    assert np.array_equal(planes[3].cpu().numpy(), x)                 # the launch's last draw plane = the positions
    assert np.allclose(shard.logp(), _logp_std_normal(x), rtol=1e-12, atol=0)
    assert torch.isfinite(planes).all() and not torch.equal(planes[0], planes[1])


def test_sample_device_contract_on_gpu():
    # python/tests/test_pyfunc.py:38-125 for the device entry point
    kw = dict(num_params=100, num_chains=4, seed=1234, min_warmup_iter=30, max_warmup_iter=30, min_sampling_iter=20,
              max_sampling_iter=20, save_inv_metric=True)
    a = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    b = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    c = wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "seed": 99})
    assert len(a) == 4 and all(x.shape == (20, 100) for x in a)
    for x, y in zip(a, b):
        assert np.array_equal(x, y) and x.warmup.stepsize == y.warmup.stepsize
        assert np.array_equal(x.warmup.inv_metric, y.warmup.inv_metric)
    assert not np.array_equal(a[0], c[0])
    with pytest.raises(ValueError, match="min_iter must be"):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, **{**kw, "min_sampling_iter": 100, "max_sampling_iter": 10})


def test_controller_statistics_match_oracle():
    """adapt.hpp:193-221 / sampler.hpp:132-145 monitors: device reductions against the device-order oracle, bit for bit
    (parity.check_monitors), below and above one run of 256 chains."""
    for model, D, C in (("std_normal", 100, 64), ("diag_normal", 1024, 48), ("diag_normal", 9000, 8),
                        ("std_normal", 40, 700)):
        dev, orc = parity.make_pair(model, D, C)
        pos = np.random.default_rng(1).normal(0, 2, size=(C, D))
        for x in (dev, orc):
            x.set_positions(pos)
            x.set_step_sizes(0.3)
            x.seed_chains(5, 0)
        for _ in range(8):
            dev.warmup_step()
            orc.warmup_step(8)
        parity.check_monitors(dev, orc, warm=True)
        dev.freeze()
        orc.freeze()
        for _ in range(6):
            dev.sample_step()
            orc.sample_step(8)
        parity.check_monitors(dev, orc, warm=False)


def test_sample_device_early_stopping_bounds():
    # python/tests/test_pyfunc.py:38-64
    kw = dict(num_params=50, num_chains=8, seed=11, min_warmup_iter=20, max_warmup_iter=200, min_sampling_iter=10,
              max_sampling_iter=300, save_warmup=True)
    out = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    n_w, n_s = out[0].warmup.warmup_draws.shape[0], out[0].shape[0]
    assert 20 <= n_w <= 200 and 10 <= n_s <= 300
    assert all(x.shape[0] == n_s and x.warmup.warmup_draws.shape[0] == n_w for x in out)
    loose = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw, step_size_converge_tol=1e6, mass_converge_tol=1e6,
                              rhat_converge_tol=1e6)
    assert loose[0].warmup.warmup_draws.shape[0] == 20 and loose[0].shape[0] == 10


def test_sampler_statistics_are_sane():
    """Not a parity test: the chains actually sample the target (std normal, D=100, many chains)."""
    D, C = 100, 4096
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C)
    e.init_positions(1, 0, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(1, 0)
    e.seed_chains(2, 0)
    for _ in range(150):
        e.warmup_step()
    e.freeze()
    for _ in range(20):
        e.sample_step()
    x = e.positions()
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1.0) < 0.02
    assert 0.2 < np.median(e.step_sizes()) < 1.5


@pytest.mark.parametrize("D,geometry", [(1024, None), (100, None), (1024, (2, 8)), (20000, None)])
def test_failed_extension_flag_on_gpu(D, geometry):
    """wn_engine_get_failed_extensions (the failure channel of device models: the counterpart of on_logp_exception,
    util.hpp:336-346) in the register kernels and the streaming kernels: chains that start where the log density
    overflows fail their first leaf at every step size and stay put, flagged after every transition; the others (small
    step: nothing fails) report nothing."""
    C = 67
    geo = {} if geometry is None else dict(waves_per_chain=geometry[0], elems_per_lane=geometry[1])
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C, wa.default_config(**geo))
    pos = np.full((C, D), 0.03)
    pos[5] = pos[66] = 1e200
    e.set_positions(pos)
    e.set_step_sizes(0.02)
    e.seed_chains(1, 0)
    want = [1 if c in (5, 66) else 0 for c in range(C)]
    e.warmup_steps(3)
    e.synchronize()
    # (warmup adapts the step: a healthy chain's extension may fail its energy or reversibility test now and then)
    assert e.failed_extensions()[[5, 66]].tolist() == [1, 1]
    e.set_step_sizes(0.02)    # back to the InitConfig's small step, frozen: nothing else fails
    e.set_positions(pos)
    e.freeze()
    e.sample_steps(2)
    e.synchronize()
    assert e.failed_extensions().tolist() == want
    assert np.array_equal(e.positions()[5], pos[5]) and np.all(e.depths()[[5, 66]] == 1)
