"""Would two independent halves of the chains on two streams hide each other's launch tails?  Two DeviceEngines of C/2
chains each (their own streams, counters, arenas; streams keyed by the GLOBAL chain id, so the draws are the single
engine's) launched alternately, against one engine of C chains.   two_groups.py <model> <C> <D> <adapt> [T] [rounds]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
import bench

model, C, D, adapt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
T = int(sys.argv[5]) if len(sys.argv) > 5 else 8
R = int(sys.argv[6]) if len(sys.argv) > 6 else 6
mid, params = bench.model_setup(model, D)

def make(first, count):
    e = wa.DeviceEngine(mid, D, count, wa.default_config(), params=params)
    e.init_positions(1234, first, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1234, first)
    e.seed_chains(1235, first)
    for i in range(0, adapt, T): e.warmup_steps(min(T, adapt - i))
    e.freeze()
    return e

def run(engines):
    for e in engines: e.sample_steps(T)
    for e in engines: e.synchronize()
    g0 = sum(e.total_grad_evals() for e in engines)
    t0 = time.perf_counter()
    for _ in range(R):
        for e in engines: e.sample_steps(T)
    for e in engines: e.synchronize()
    dt = time.perf_counter() - t0
    g = sum(e.total_grad_evals() for e in engines) - g0
    return dt / (R * T) * 1e3, g / dt

one = run([make(0, C)])
two = run([make(0, C // 2), make(C // 2, C - C // 2)])
four = run([make(i * (C // 4), C // 4) for i in range(4)])
print(f"{model} {C}x{D} T={T}: one engine {one[0]:.4f} ms/step {one[1]:.4e}/s | two halves on two streams {two[0]:.4f} {two[1]:.4e} "
      f"({(two[1] / one[1] - 1) * 100:+.1f} %) | four quarters {four[0]:.4f} {four[1]:.4e} ({(four[1] / one[1] - 1) * 100:+.1f} %)")
