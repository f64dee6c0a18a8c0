"""GPU probe: fixed-shape transitions (tiny step: every leaf accepted at the first level, no U-turn, every tree runs to
max_depth) -> time per transition as a function of the number of leaves: the fixed cost of a transition (chain fetch,
loads, momentum refresh, stores) and the marginal cost of a leaf with its merges.
usage: cost_model.py [nw epl]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 0; epl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
D, C = 1024, 65536
rows = []
for md in (1, 2, 3, 4, 5, 6):
    cfg = wa.default_config(max_trajectory_doublings=md, max_hamiltonian_error=1e9, waves_per_chain=nw, elems_per_lane=epl)
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C, cfg)
    e.init_positions(1, 0, 1.0); e.set_step_sizes(1e-4); e.seed_chains(2, 0); e.freeze()
    for _ in range(3): e.sample_step()
    e.synchronize(); e.timing_reset()
    g0 = e.total_grad_evals()
    for _ in range(6): e.sample_step()
    e.synchronize()
    ms = float(np.mean(e.kernel_times_ms())); g = (e.total_grad_evals() - g0) / 6 / C
    rows.append((md, g, ms))
    print(f"max_depth {md}: {g:.1f} grad-evals per transition, {ms:.4f} ms per launch, {ms * 1e3 * 1024 / C:.2f} us per chain-transition (1024 resident chains)")
    e.close()
(d1, g1, t1), (d2, g2, t2) = rows[-2], rows[-1]
per_leaf = (t2 - t1) / (g2 - g1)
print(f"marginal cost per leaf (with its merges): {per_leaf * 1e3 * 1024 / C:.3f} us; fixed cost per transition: {(rows[0][2] - per_leaf * rows[0][1]) * 1e3 * 1024 / C:.2f} us")
