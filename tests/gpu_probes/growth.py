"""GPU probe: per-transition drift curves of the drop-in entry point fed the reference's own random streams against
the oracle run in reference order (the cases of tests/test_gpu_parity.py::test_reference_streams_...); the output is
filed under profiles/ (the test prints the same numbers, which `pytest -q` swallows)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import parity  # noqa: E402

CASES = [("std_normal", 100, 4, 30, 30, True), ("funnel", 16, 6, 20, 15, True), ("diag_normal", 257, 4, 15, 10, True),
         ("std_normal", 1024, 4, 15, 10, True), ("std_normal", 1024, 4, 15, 10, False)]
print("# max over chains of ||theta_device - theta_reference_order||_inf / ||theta||_inf after each transition;")
print("# device = walnutpie_sample_device_reference_streams (the reference's mt19937_64 + libstdc++ streams fed from the")
print("# host), oracle = reference order (left-to-right sums, libm), seed 48.  One transition from identical inputs agrees")
print("# to ~1e-15; the two summation orders then drift apart chaotically (no decision flips: see bench.py parity_gate).")
for model, D, C, warm, samp, unit in CASES:
    metric = np.ones(D) if unit else None
    worst, growth = parity.check_reference_stream_run(model, D, C, seed=48, warmup=warm, sampling=samp, horizon=1,
                                                      init_inv_metric=metric, init_radius=1.0 if unit else 2.0)
    print(f"{model} D={D} C={C} {'unit initial metric' if unit else 'gradient-based initial masses (reference default)'}: "
          f"{warm} warmup + {samp} sampling transitions")
    print("   " + " ".join("%.1e" % g for g in growth))
