"""GPU probe: a timeline of one workgroup's transitions (library built with -DWN_TIMELINE): shader-clock intervals
between the marks of the transition kernel, averaged per (mark -> next mark) edge."""
import ctypes as C, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
lib_path = os.path.join(ROOT, "tests/gpu_probes/libwalnuts_tl.so")
NAMES = ["idle", "prologue", "leapfrog", "energy", "restart", "reversible", "uturn", "combine", "push", "topmerge",
         "doubling", "epilogue", "loads_issued", "momentum", "tuned", "evaluated", "sel_loaded", "stored", "scalars"]
D, Cn = 1024, 65536
cfg = wa.default_config(lib_path, waves_per_chain=1, elems_per_lane=16, lds_vectors=3)
e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, Cn, cfg, lib_path=lib_path)
e.init_positions(1, 0, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1, 0); e.seed_chains(2, 0)
for _ in range(100): e.warmup_step()
WARM = "--warmup" in sys.argv   # profile an adaptive warmup transition instead of a sampling one
FUSED = int(sys.argv[sys.argv.index("--fused") + 1]) if "--fused" in sys.argv else 1   # transitions per launch
if WARM:
    e.synchronize()
    e.timing_reset()
    e.warmup_steps(FUSED); e.synchronize()
else:
    e.freeze()
    for _ in range(3): e.sample_step()
    e.synchronize()
    e.timing_reset()
    e.sample_steps(FUSED); e.synchronize()
print("launch ms", e.kernel_times_ms().mean())
get = e.lib.wn_debug_timeline_std_normal
N = 1536
buf = (C.c_ulonglong * N)()
get(buf, N)
rec = [(int(v) >> 6, int(v) & 63) for v in buf if v]
print("marks", len(rec))
edges = collections.defaultdict(list)
for (t0, k0), (t1, k1) in zip(rec, rec[1:]):
    edges[(k0, k1)].append(t1 - t0)
tot = rec[-1][0] - rec[0][0]
ntr = sum(1 for _, k in rec if k == 1)
print(f"total {tot} clock ticks over {ntr} transitions = {tot / max(ntr,1):.0f} per transition")
rows = sorted(edges.items(), key=lambda kv: -sum(kv[1]))
for (a, b), v in rows:
    print(f"  {NAMES[a]:>13s} -> {NAMES[b]:<13s} n={len(v):4d}  mean {np.mean(v):8.1f}  min {min(v):6d}  max {max(v):6d}  share {100*sum(v)/tot:5.1f} %")
if "--dump" in sys.argv:
    t0 = rec[0][0]
    for t, k in rec[:400]: print(t - t0, NAMES[k])
