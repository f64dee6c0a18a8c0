import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import parity
lib = sys.argv[1] if len(sys.argv) > 1 else None
for md in (1, 2, 5):
    dev, orc = parity.make_pair("std_normal", 100, 64, lib, max_trajectory_doublings=md)
    rng = np.random.default_rng(1234)
    pos = rng.normal(0.0, 2.0, size=(64, 100))
    for x in (dev, orc):
        x.set_positions(pos); x.set_step_sizes(0.5); x.seed_chains(1235, 3)
    dev.freeze(); orc.freeze()
    dev.sample_step(); orc.sample_step(); dev.synchronize()
    bad = np.where(np.any(dev.positions() != orc.positions(), axis=1))[0]
    print("max_depth", md, "bad chains", len(bad), bad[:10])
    for c in bad[:6]:
        print("  chain", c, "depth", dev.depths()[c], orc.depths()[c], "draws", dev.rng_draws()[c], orc.rng_draws()[c],
              "ngrad", dev.grad_evals()[c], orc.grad_evals()[c], "logp", dev.logp()[c], orc.logp()[c])
