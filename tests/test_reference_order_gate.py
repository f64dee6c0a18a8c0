"""The REFERENCE-ORDER gate for every kernel family (VERDICT r05 "Hole B").

tests/parity.py compares the product with the oracle in DEVICE order (portable maths, lane-order sums) bit for bit;
that oracle mode is edited whenever the device changes.  This module compares the product with the oracle run the way the
REFERENCE runs -- libm exp/log, every product rounded, sums in Eigen 3.4's SSE2 redux order and in plain sequential
order (walnuts.hpp:192-201,339,379; util.hpp:220-223) -- one transition at a time from the device's own states with the
same counter-based streams: identical trees, no decision within 1e-12 of its threshold, selected positions within
1e-13 and log densities within 1e-10 relative (BASELINE.json's north star: <= 1e-10).  It is bench.py's same-run
parity gate (`bench.reference_order_gate`), run here on every kernel family in both device arithmetic modes.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpusim"))

POS_TOL, LOGP_TOL = 1e-13, 1e-10


def assert_gate_clean(g, where):
    assert g["tree_mismatches"] == 0, (where, g["orders"])
    for name, o in g["orders"].items():
        assert o["tree_mismatches"] == 0, (where, name)
        near = {k: v["near"] for k, v in o["near_ties_1e-12"].items()}
        decisions = {k: v["decisions"] for k, v in o["near_ties_1e-12"].items()}
        assert all(v == 0 for v in near.values()), (where, name, near)
        assert decisions["energy_error"] > 0 and decisions["uturn_sign"] > 0 and decisions["acceptance"] > 0, \
            (where, "vacuous gate", decisions)
        assert o["max_rel_diff"] <= POS_TOL, (where, name, o["max_rel_diff"])
        assert o["max_rel_diff_logp"] <= LOGP_TOL, (where, name, o["max_rel_diff_logp"])
        if g.get("phase") == "warmup":
            # Adam's state and the estimator's planes after the transition: same element-wise arithmetic on both sides up
            # to libm-vs-portable exp/pow in Adam and the summation order inside the energies it is fed
            assert o["max_rel_diff_adapt"] <= 1e-10, (where, name, o["max_rel_diff_adapt"])


# ---- CPU tier: the gate's own logic (incl. the warmup phase's state hand-over) on the workgroup emulation -------------
@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,D,geometry,phase", [
    ("std_normal", 96, (1, 2), "sampling"),
    ("funnel", 40, (1, 2), "sampling"),
    ("diag_normal", 300, (1, -1), "sampling"),      # streaming kernel, moving end held
    ("std_normal", 96, (1, 2), "warmup"),
    ("diag_normal", 130, (1, 4), "warmup"),
])
def test_gate_on_the_emulation(oracle, model, D, geometry, phase):
    import build as simbuild

    import bench

    sim = simbuild.build()
    g = bench.reference_order_gate(model, D, dict(waves_per_chain=geometry[0], elems_per_lane=geometry[1]), chains=3,
                                   transitions=3, adapt_iters=6, phase=phase, lib_path=sim)
    assert g["phase"] == phase and g["chains"] == 3
    assert_gate_clean(g, f"{model} D={D} {phase} (emulation)")


# ---- GPU tier: every kernel family, both device arithmetic modes -------------------------------------------------------
#  (tag, model, D, geometry or None = the engine's choice, chains, transitions, adapt, phase)
GPU_CASES = [
    ("headline (1,16) register kernel", "std_normal", 1024, None, 128, 6, 40, "sampling"),
    ("config #2 ill-conditioned", "ill_normal", 1024, None, 128, 6, 40, "sampling"),
    ("config #3 funnel D=128", "funnel", 128, None, 128, 6, 40, "sampling"),
    ("config #4: held streaming kernel", "diag_normal", 16384, None, 24, 4, 30, "sampling"),
    ("both ends streamed", "diag_normal", 20000, None, 16, 4, 20, "sampling"),
    ("two-pass held streaming (funnel)", "funnel", 16384, None, 16, 3, 20, "sampling"),
    ("halo streaming (rw1)", "rw1", 12000, None, 16, 3, 20, "sampling"),
    ("four wavefronts per chain", "funnel", 1000, (4, 4), 64, 4, 30, "sampling"),
    ("eight wavefronts per chain", "std_normal", 4096, (8, 8), 32, 4, 30, "sampling"),
    ("two wavefronts, 16 per lane (default at 2 048)", "diag_normal", 2048, None, 48, 4, 30, "sampling"),
    ("four wavefronts, 16 per lane (default at 4 096)", "diag_normal", 4096, None, 24, 3, 20, "sampling"),
    ("funnel at 4 096: eight wavefronts by its own hint", "funnel", 4096, None, 24, 3, 20, "sampling"),
    ("rw1 at 3 000: eight wavefronts, eight per lane by its own hint", "rw1", 3000, None, 16, 3, 20, "sampling"),
    ("four wavefronts, 16 per lane, adaptive transitions", "std_normal", 3000, None, 24, 3, 20, "warmup"),
    ("rw1 at 1 024", "rw1", 1024, None, 64, 4, 30, "sampling"),
    ("headline kernel, adaptive transitions", "std_normal", 1024, None, 64, 4, 30, "warmup"),
    ("config #4 kernel, adaptive transitions", "diag_normal", 16384, None, 16, 3, 20, "warmup"),
    ("funnel D=128, adaptive transitions", "funnel", 128, None, 64, 4, 30, "warmup"),
]


def _deep_low_density_gate(lib_path, chains, D, geometry):
    """Plain NUTS (no energy-error bound) from far-out starting points, trees up to 2^8 leaves: the device's span
    weights are LINEAR-domain values relative to a moving reference energy, clamped at e^-700 of it (wn_traj.h "span
    weights", wn_devmath.h dexp_weight) where the reference keeps log-domain weights of any size (walnuts.hpp:368-387,
    util.hpp:174-183).  The gate replays every transition on the reference-order oracle -- log_sum_exp, no clamp --: the
    Barker / Metropolis decisions must agree (identical trees, no acceptance within 1e-12 of its threshold)."""
    import bench

    kw = dict(max_hamiltonian_error=1e9, max_trajectory_doublings=8)
    if geometry is not None:
        kw.update(waves_per_chain=geometry[0], elems_per_lane=geometry[1])
    g = bench.reference_order_gate("std_normal", D, kw, chains=chains, transitions=4, adapt_iters=0, init_scale=30.0,
                                   step_size=0.04, lib_path=lib_path)
    assert_gate_clean(g, "deep trees far out")
    acc = g["orders"]["sequential"]["near_ties_1e-12"]["acceptance"]
    assert acc["decisions"] >= 4 * 40 * chains and acc["near"] == 0, acc   # (trees of ~2^6 leaves were built)


@pytest.mark.timeout(900)
def test_span_weight_clamp_against_log_domain_weights_on_the_emulation(oracle):
    import build as simbuild

    _deep_low_density_gate(simbuild.build(), 3, 40, (1, 2))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_span_weight_clamp_against_log_domain_weights_on_gpu(gpu):
    _deep_low_density_gate(None, 256, 1024, None)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("fma", [1, 0])
@pytest.mark.parametrize("tag,model,D,geometry,chains,transitions,adapt,phase", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_reference_order_gate_on_gpu(gpu, tag, model, D, geometry, chains, transitions, adapt, phase, fma):
    import bench

    kw = dict(fused_multiply_add=fma)
    if geometry is not None:
        kw.update(waves_per_chain=geometry[0], elems_per_lane=geometry[1])
    g = bench.reference_order_gate(model, D, kw, chains=chains, transitions=transitions, adapt_iters=adapt, phase=phase)
    if "held" in tag:
        assert g["kernel"]["streaming"] and g["kernel"]["held_tiles"] > 0, g["kernel"]
    if tag == "both ends streamed":
        assert g["kernel"]["streaming"] and g["kernel"]["held_tiles"] == 0, g["kernel"]
    assert_gate_clean(g, f"{tag} fma={fma}")
