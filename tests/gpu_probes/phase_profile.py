"""GPU probe: per-phase shader-cycle breakdown of the transition kernel (library built with -DWN_PHASE_PROFILE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
lib_path = os.path.join(ROOT, "tests/gpu_probes/libwalnuts_prof.so")
NAMES = ["idle/fetch", "prologue", "leapfrog", "energy+accept", "restart save/restore", "reversible", "uturn", "combine",
         "push", "top merge", "doubling start", "epilogue"]
D, Cn = 1024, 65536
nw, epl, wg = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 8, 4)))
cfg = wa.default_config(lib_path, waves_per_chain=nw, elems_per_lane=epl, workgroups_per_cu=wg)
e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, Cn, cfg, lib_path=lib_path)
e.init_positions(1, 0, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1, 0); e.seed_chains(2, 0)
for _ in range(40): e.warmup_step()
e.freeze(); e.sample_step(); e.synchronize()
get = e.lib.wn_debug_phase_cycles_std_normal
buf = (C.c_ulonglong * 16)()
get(buf, 16)  # clear
g0 = e.total_grad_evals()
e.timing_reset()
for _ in range(5): e.sample_step()
e.synchronize()
ms = e.kernel_times_ms().mean()
g = (e.total_grad_evals() - g0) / 5
get(buf, 16)
tot = sum(buf[:12])
print(f"geometry nw={nw} epl={epl} wg/cu={wg}: {ms:.3f} ms/launch (profiled), {g:.0f} grad-evals/launch")
for n, v in zip(NAMES, buf[:12]):
    print(f"  {n:22s} {100.0 * v / tot:6.2f} %   {v / 5 / g:9.1f} wave-cycles per grad-eval")
