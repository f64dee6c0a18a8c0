"""How uneven are the work items of a multi-transition launch, and are a chain's costs persistent?  Per-chain gradient
evaluations of consecutive launches (8 transitions each): correlation between launches, and list-scheduling makespans
(arrival order vs longest-first) against the mean load per workgroup slot."""
import os, sys, heapq
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT]
import walnuts_amd as wa
import bench

def run(model, C, D, adapt, T=8, launches=4):
    mid, params = bench.model_setup(model, D)
    e = wa.DeviceEngine(mid, D, C, wa.default_config(), params=params)
    e.init_positions(1234, 0, 2.0); e.init_masses_from_grad(1e-5); e.set_step_sizes(1.0); e.adapt_step(1234, 0); e.seed_chains(1235, 0)
    for i in range(0, adapt, T): e.warmup_steps(min(T, adapt - i))
    e.freeze()
    e.sample_steps(T); e.synchronize()
    g0 = e.grad_evals().copy(); costs = []
    for _ in range(launches):
        e.sample_steps(T); e.synchronize()
        g1 = e.grad_evals().copy(); costs.append((g1 - g0).astype(np.float64)); g0 = g1
    slots = e.workgroups
    def makespan(order, cost):
        h = [0.0] * slots; heapq.heapify(h)
        for c in order: heapq.heappush(h, heapq.heappop(h) + cost[c])
        return max(h)
    a, b = costs[-2], costs[-1]
    fixed = 0.25 * a.mean()   # per-transition overhead outside the leaves, roughly a quarter of an average item
    ca, cb = a + fixed, b + fixed
    print(f"{model} {C}x{D}: slots {slots}, items/slot {C/slots:.1f}, grad-evals per item mean {b.mean():.1f} sd {b.std():.1f} max {b.max():.0f}; "
          f"corr(launch n, n+1) {np.corrcoef(a, b)[0,1]:.3f}")
    ideal = cb.sum() / slots
    print(f"   mean load {ideal:.0f}; arrival-order makespan {makespan(range(C), cb)/ideal:.3f}x; longest-first by THIS launch's cost {makespan(np.argsort(-cb), cb)/ideal:.3f}x; "
          f"longest-first by the PREVIOUS launch's cost {makespan(np.argsort(-ca), cb)/ideal:.3f}x")
run("ill_normal", 4096, 1024, 300)
run("std_normal", 8192, 1024, 100)
run("std_normal", 65536, 1024, 100)
run("funnel", 16384, 128, 300)
