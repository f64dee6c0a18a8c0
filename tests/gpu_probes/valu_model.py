"""GPU probe: fixed-shape transitions (every leaf accepted at level 0, no U-turn) for a VALU cost model.
usage: valu_model.py <max_depth> [nw epl]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
import walnuts_amd as wa
md = int(sys.argv[1]); nw = int(sys.argv[2]) if len(sys.argv) > 2 else 0; epl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
D, C = 1024, 16384
cfg = wa.default_config(max_trajectory_doublings=md, max_hamiltonian_error=1e9, waves_per_chain=nw, elems_per_lane=epl)
e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, C, cfg)
e.init_positions(1, 0, 1.0); e.set_step_sizes(1e-4); e.seed_chains(2, 0); e.freeze()
for _ in range(3): e.sample_step()
e.synchronize()
print("depth", md, "grad-evals per transition", e.grad_evals().mean() / 3, "depths", np.bincount(e.depths()))
