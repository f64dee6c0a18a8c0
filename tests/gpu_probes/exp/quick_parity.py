"""quick device-vs-oracle bit parity on a handful of cases (development probe; the real gate is tests/test_gpu_parity.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
lib = os.environ.get("WALNUTS_AMD_LIB")
cases = [("std_normal", 1024, 64, (2, 8), {}), ("std_normal", 1000, 48, (2, 8), dict(lds_vectors=1)),
         ("std_normal", 1024, 32, (2, 8), dict(lds_vectors=0)),
         ("diag_normal", 1024, 48, (2, 8), {}), ("funnel", 900, 32, (2, 8), {}), ("std_normal", 100, 64, None, {}),
         ("funnel", 128, 64, None, {}), ("std_normal", 1024, 32, (4, 4), {}), ("std_normal", 1024, 32, (1, 16), {}),
         ("std_normal", 64, 9, None, dict(max_trajectory_doublings=10, step=0.01, max_hamiltonian_error=50.0)),
         ("std_normal", 65, 9, None, dict(max_step_halvings=1, step=1.7)),
         ("diag_normal", 129, 9, None, dict(min_micro_steps=3))]
sel = sys.argv[1:] 
for i, (m, D, C, g, kw) in enumerate(cases):
    if sel and str(i) not in sel: continue
    try:
        parity.run_case(m, D, C, warmup=10, sampling=6, geometry=g, lib_path=lib, check_every=2, **kw)
        print("ok  ", i, m, D, C, g, kw, flush=True)
    except Exception as e:
        print("FAIL", i, m, D, C, g, kw, repr(e)[:300], flush=True)
