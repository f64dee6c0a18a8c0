#!/usr/bin/env python3
"""GPU probe: a randomised bit-exact parity campaign -- device engine against the CPU oracle (device-order mode) on
random (model, dimension, chain count, geometry, pool tier sizes, step size, tree depth, halvings, micro steps, error
bound) configurations, a few adaptive warmup + sampling transitions each.
   python tests/gpu_probes/fuzz_parity.py [--cases N] [--seed S] [--seconds T]
Every case is printed with its configuration; a mismatch prints FAIL with the assertion text and the campaign goes on.
Exit code = number of failing cases."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np

import parity

GEOMETRIES = [(1, 2), (1, 4), (1, 8), (1, 16), (2, 2), (2, 4), (2, 8), (2, 16), (4, 2), (4, 4), (4, 8), (4, 16), (8, 2),
              (8, 4), (8, 8), (16, 4), (16, 8)]


def random_case(rng, geometries=GEOMETRIES, mem_waves=(2, 4, 8, 16), chain_counts=(1, 2, 3, 7, 16, 33, 64),
                streaming_share=0.15, streaming_dims=(3, 3000), models=("std_normal", "diag_normal", "funnel", "rw1")):
    model = rng.choice(list(models))
    streaming = rng.uniform() < streaming_share   # every model has streaming kernels (funnel / rw1: the two-pass form)
    if streaming:
        geometry = (int(rng.choice(list(mem_waves))), -1)
        D = int(rng.integers(*streaming_dims))
    else:
        geometry = geometries[int(rng.integers(len(geometries)))]
        cap = 64 * geometry[0] * geometry[1]
        D = int(rng.integers(max(2, cap // 4), cap + 1))
    kw = dict(warmup=int(rng.integers(0, 7)), sampling=int(rng.integers(1, 5)), geometry=geometry,
              seed=int(rng.integers(1, 2**31)), check_every=1)
    style = rng.integers(0, 6)
    if style == 0:      # deep trees
        kw.update(step=float(rng.uniform(0.01, 0.05)), max_trajectory_doublings=int(rng.integers(6, 10)))
    elif style == 1:    # halvings + reversibility
        kw.update(step=float(rng.uniform(1.5, 3.5)), max_trajectory_doublings=int(rng.integers(2, 6)),
                  max_step_halvings=int(rng.integers(1, 9)))
    elif style == 2:    # multi-step leaves
        kw.update(step=float(rng.uniform(0.3, 0.9)), min_micro_steps=int(rng.integers(2, 5)),
                  max_trajectory_doublings=int(rng.integers(2, 7)))
    elif style == 3:    # tight error bound: nearly every level rejected
        kw.update(step=float(rng.uniform(0.5, 1.2)), max_hamiltonian_error=float(10 ** rng.uniform(-4, -1)))
    elif style == 4:    # adaptive step from the init search
        kw.update(max_trajectory_doublings=int(rng.integers(1, 8)))
    else:
        kw.update(step=float(rng.uniform(0.1, 1.5)))
    if not streaming and rng.uniform() < 0.5:   # squeeze the on-chip pool: more traffic through the HBM arena
        kw.update(lds_vectors=int(rng.integers(0, 4)))
    if rng.uniform() < 0.2:
        kw.update(average_masses=True)
    kw.update(fused_multiply_add=int(rng.integers(0, 2)))   # both arithmetic modes
    kw.update(fused=int(rng.choice([1, 1, 2, 3, 5])))       # transitions per launch (wn_engine_*_steps)
    if rng.uniform() < 0.35:
        kw.update(lazy=True)   # nothing read between a phase's launches: the pending observation crosses launches
    C = int(rng.choice(list(chain_counts)))
    if C > 1 and rng.uniform() < 0.3:
        kw.update(chain_groups=2)                           # two independently launched blocks of chains
    return model, D, C, kw


def campaign(seed: int, seconds: float, cases: int = 10**9, verbose: bool = False, lib_path=None, **case_kw):
    """-> (cases run, failing descriptions, tally).  lib_path / case_kw: the CPU tier runs the same campaign against
    the workgroup emulation with its reduced geometry table (tests/test_sim_parity.py)."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    done, failed, tally = 0, [], {}
    while done < cases and time.time() - t0 < seconds:
        model, D, C, kw = random_case(rng, **case_kw)
        if lib_path is not None:
            kw["lib_path"] = lib_path
        desc = f"{model} D={D} C={C} " + " ".join(f"{k}={v}" for k, v in sorted(kw.items()) if k != "lib_path")
        try:
            dev, orc = parity.run_case(model, D, C, **kw)
            evals = int(dev.grad_evals().sum())
            key = (model, "streaming" if kw["geometry"][1] < 0 else "%dx%d" % kw["geometry"])
            n, e, nan = tally.get(key, (0, 0, 0))
            tally[key] = (n + 1, e + evals, nan + int(np.isnan(dev.adam()).any()) if kw["warmup"] else nan)
            if verbose:
                print(f"ok   {desc}  grad_evals={evals}")
        except (AssertionError, ValueError, RuntimeError) as e:
            txt = str(e).splitlines()[0][:200] if str(e) else traceback.format_exc().splitlines()[-1]
            if isinstance(e, ValueError) and ("unsupported" in txt or "exceeds" in txt or "pool" in txt
                                              or "no streaming" in txt):
                print(f"skip {desc}  ({txt})")
            else:
                failed.append(f"{desc}  {type(e).__name__}: {txt}")
                print(f"FAIL {failed[-1]}")
        done += 1
    return done, failed, tally


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--verbose", action="store_true", help="print every case, not only failures / skips + the tally")
    ap.add_argument("--held", action="store_true",
                    help="the streaming kernels that hold the moving end in registers only: one-pass models, 8 wavefronts "
                         "per chain, 1 000-16 384 dimensions (1-16 tiles per lane), few chains")
    ap.add_argument("--held-two-pass", action="store_true",
                    help="... the same for the two-pass models (funnel: sums; rw1: halo reads through lane shuffles and LDS)")
    a = ap.parse_args()
    t0 = time.time()
    case_kw = {}
    if a.held or a.held_two_pass:
        case_kw = dict(mem_waves=(8,), streaming_share=1.0, streaming_dims=(1000, 16385), chain_counts=(1, 2, 3, 5),
                       models=("funnel", "rw1") if a.held_two_pass else ("std_normal", "diag_normal"))
    done, failed, tally = campaign(a.seed, a.seconds, a.cases, a.verbose, **case_kw)
    bad = len(failed)
    print("# bit-exact cases by (model, waves x elements per lane): count, gradient evaluations compared, "
          "cases with a NaN-poisoned Adam state (identical on both sides)")
    for key in sorted(tally):
        print(f"#   {key[0]:12s} {key[1]:10s} {tally[key][0]:6d} {tally[key][1]:12d} {tally[key][2]:4d}")
    print(f"# seed {a.seed}: {done} cases in {time.time() - t0:.0f} s, {bad} failing")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
