#!/usr/bin/env python3
"""GPU probe (library built with -DWN_COUNT_POOL: tests/gpu_probes/build_variant.sh pc "-DWN_COUNT_POOL"): what the span
pool of the headline kernel moves through LDS and through its HBM arena per transition -- counted IN the kernel (doubles
per lane per call site), to be set beside the WRITE_SIZE / FETCH_SIZE counters of rocprofv3.
   python tests/gpu_probes/pool_traffic.py [--lds-vectors N] [--chains C] [--dim D]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa

def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default

lib_path = os.path.join(ROOT, "tests/gpu_probes/libwalnuts_pc.so")
D, Cn, lds = arg("--dim", 1024), arg("--chains", 65536), arg("--lds-vectors", -1)
cfg = wa.default_config(lib_path, lds_vectors=lds)
e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, Cn, cfg, lib_path=lib_path)
e.init_positions(1, 0, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1, 0); e.seed_chains(2, 0)
for _ in range(13): e.warmup_steps(8)
e.freeze()
e.sample_steps(8); e.synchronize()
get = e.lib.wn_debug_pool_counts_std_normal
buf = (C.c_ulonglong * 5)()
get(buf)                                   # clear what warmup and the first launch counted
e.sample_steps(8); e.synchronize()
get(buf)
ls, ll, as_, al, tr = [int(v) for v in buf]
lanes = e.lanes
per = lambda n: n * lanes * 8 / max(tr, 1)   # bytes per chain-transition
print(f"# {Cn} chains x {D} dims, one launch of 8 sampling transitions, LDS pool vectors per chain: {e.lds_vectors}")
print(f"transitions counted {tr}")
print(f"span pool, bytes per chain-transition: LDS stores {per(ls):9.0f}  LDS loads {per(ll):9.0f}  "
      f"arena stores {per(as_):9.0f}  arena loads {per(al):9.0f}")
print(f"arena traffic per launch: stores {as_ * lanes * 8 / 1e9:.3f} GB, loads {al * lanes * 8 / 1e9:.3f} GB "
      f"(compulsory plane traffic per launch at this size: reads {(2 + 8) * Cn * D * 8 / 1e9:.2f} GB = theta, inv_mass once + "
      f"chol x 8; writes {(8 + 1) * Cn * D * 8 / 1e9:.2f} GB = 8 draw planes + theta)")
