#!/usr/bin/env python3
"""GPU probe: randomised bit-exact campaign of the device posterior summaries against the summary oracle -- random
numbers of ragged chains, dimensions (both the one-pass two-columns-per-lane moments and the column loops), chain
lengths, autocorrelations, ties and quantile probabilities.
   python tests/gpu_probes/fuzz_summaries.py [--seconds T] [--seed S]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np

import summary_parity as sp


def random_case(rng):
    C = int(rng.choice([1, 2, 3, 5, 17, 64, 300, 700]))
    D = int(rng.choice([1, 2, 3, 63, 64, 65, 66, 128, 130, 257]))
    short = rng.uniform() < 0.5          # <= 32 draws and even D -> the one-pass moments
    hi = 33 if short else int(rng.choice([40, 70, 130]))
    lens = [int(n) for n in rng.integers(3, hi, size=C)]
    phi = rng.choice([0.0, 0.5, 0.9, -0.6], size=D)
    chains = sp.ar_chains(rng, C, D, lens, phi)
    if rng.uniform() < 0.3:              # heavy ties: integers, zeros of both signs
        chains = [np.round(c) for c in chains]
        for c in chains:
            z = c == 0
            c[z] = np.where(rng.uniform(size=int(z.sum())) < 0.5, -0.0, 0.0)
    probs = tuple(sorted(set([float(p) for p in rng.uniform(0, 1, size=int(rng.integers(1, 9)))] + ([0.0, 1.0] if rng.uniform() < 0.3 else []))))
    total = sum(lens) * D
    return chains, probs, dict(C=C, D=D, max_len=max(lens), min_len=min(lens), probs=len(probs), full_acov=total < 400000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    done = bad = wide = 0
    while time.time() - t0 < a.seconds:
        chains, probs, info = random_case(rng)
        try:
            sp.check_all(chains, probs=probs, full_acov=info["full_acov"])
            wide += int(info["max_len"] <= 32 and info["D"] % 2 == 0)
        except AssertionError as e:
            bad += 1
            print("FAIL", info, str(e).splitlines()[0][:200])
        done += 1
    print(f"# seed {a.seed}: {done} cases ({wide} through the one-pass moments) in {time.time() - t0:.0f} s, {bad} failing")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
