"""GPU probe: phase breakdown (shader cycles per chain-transition, library built with -DWN_PHASE_PROFILE) of
fixed-shape transitions: max_depth 1 (the fixed cost of a transition) and 4."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
lib_path = os.path.join(ROOT, "tests/gpu_probes/libwalnuts_prof.so")
NAMES = ["idle/fetch", "prologue", "leapfrog", "energy+accept", "restart", "reversible", "uturn", "combine",
         "push", "top merge", "doubling start", "epilogue"]
D, Cn = 1024, 65536
for md in (1, 4):
    cfg = wa.default_config(lib_path, max_trajectory_doublings=md, max_hamiltonian_error=1e9)
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, Cn, cfg, lib_path=lib_path)
    e.init_positions(1, 0, 1.0); e.set_step_sizes(1e-4); e.seed_chains(2, 0); e.freeze()
    for _ in range(2): e.sample_step()
    e.synchronize()
    get = e.lib.wn_debug_phase_cycles_std_normal
    buf = (C.c_ulonglong * 16)(); get(buf, 16)
    e.timing_reset()
    for _ in range(4): e.sample_step()
    e.synchronize()
    ms = e.kernel_times_ms().mean(); get(buf, 16)
    tot = sum(buf[:12])
    print(f"max_depth {md}: {ms:.3f} ms/launch (profiled); cycles per chain-transition: {tot / 4 / Cn:.0f}")
    for n, v in zip(NAMES, buf[:12]):
        print(f"  {n:16s} {100.0 * v / tot:6.2f} %   {v / 4 / Cn:9.0f} cycles")
    e.close()
