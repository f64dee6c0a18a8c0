"""GPU probe: the timeline of one workgroup of the STREAMING kernels (library built with
`build_variant.sh tlm "-DWN_TIMELINE -DWN_TIMELINE_MARKS=192 -DWN_ONLY_MEM_NW=8"`): shader-clock intervals between the marks of the tree loop,
per (mark -> next mark) edge.  usage: timeline_mem.py [model] [dim] [chains] [waves per chain]"""
import ctypes as C, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
lib_path = os.path.join(ROOT, "tests/gpu_probes/libwalnuts_tlm.so")
NAMES = ["idle", "prologue", "leapfrog", "energy", "restart", "reversible", "uturn", "combine", "push", "topmerge",
         "doubling", "epilogue", "loads_issued", "momentum", "tuned", "evaluated", "sel_loaded", "stored", "scalars"]
model = sys.argv[1] if len(sys.argv) > 1 else "diag_normal"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
NW = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cfg = wa.default_config(lib_path, waves_per_chain=NW, elems_per_lane=-1, workgroups_per_cu=1)
params = (1.0 + (np.arange(D) % 16)) if model == "diag_normal" else None
e = wa.DeviceEngine(wa.model_id(model, lib_path), D, Cn, cfg, params=params, lib_path=lib_path)
e.init_positions(1, 0, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1, 0); e.seed_chains(2, 0)
for _ in range(100): e.warmup_step()
e.freeze()
for _ in range(3): e.sample_step()
e.synchronize()
e.timing_reset()
e.sample_steps(1); e.synchronize()
print("launch ms", e.kernel_times_ms().mean(), "held tiles", e.held_tiles, "(0: the marks' LDS left no room for the inverse mass -- fewer WN_TIMELINE_MARKS)")
get = getattr(e.lib, "wn_debug_timeline_" + model)
N = 192
buf = (C.c_ulonglong * N)()
get(buf, N)
rec = [(int(v) >> 6, int(v) & 63) for v in buf if v]
print("marks", len(rec))
edges = collections.defaultdict(list)
for (t0, k0), (t1, k1) in zip(rec, rec[1:]):
    edges[(k0, k1)].append(t1 - t0)
tot = rec[-1][0] - rec[0][0]
ntr = sum(1 for _, k in rec if k == 1)
nlf = sum(1 for _, k in rec if k == 2)
print(f"total {tot} clock ticks over {ntr} transitions, {nlf} leapfrog passes = {tot / max(nlf,1):.0f} per pass (100 MHz ticks: x 10 ns)")
rows = sorted(edges.items(), key=lambda kv: -sum(kv[1]))
for (a, b), v in rows:
    print(f"  {NAMES[a]:>13s} -> {NAMES[b]:<13s} n={len(v):4d}  mean {np.mean(v):8.1f}  min {min(v):6d}  max {max(v):6d}  share {100*sum(v)/tot:5.1f} %")
if "--dump" in sys.argv:
    t0 = rec[0][0]
    for t, k in rec[:300]: print(t - t0, NAMES[k])
