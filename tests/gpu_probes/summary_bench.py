#!/usr/bin/env python3
"""Wall time of the device posterior summaries on synthetic draws resident in HBM.
   python tests/gpu_probes/summary_bench.py [--chains C] [--dim D] [--draws T] [--reps R]
Prints one JSON line; `gbps` = bytes of draws / time (one pass = C*T*D*8 bytes)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--draws", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--phi", type=float, default=0.0, help="AR(1) coefficient of the synthetic chains")
    a = ap.parse_args()
    import torch

    import walnuts_amd as wa
    from walnuts_amd import summary as ws

    C, D, T = a.chains, a.dim, a.draws
    g = torch.Generator(device="cuda").manual_seed(1)
    draws = torch.randn((C, T, D), generator=g, device="cuda", dtype=torch.float64)
    if a.phi != 0.0:
        s = (1 - a.phi ** 2) ** 0.5
        for t in range(1, T):
            draws[:, t] = a.phi * draws[:, t - 1] + s * draws[:, t]
    torch.cuda.synchronize()
    gb = C * T * D * 8 / 1e9
    out = {"chains": C, "dim": D, "draws_per_chain": T, "draws_gb": gb, "ms": {}, "passes_gbps": {}}
    fns = [("mean", lambda ch: ws.mean(ch)), ("sample_variance", lambda ch: ws.sample_variance(ch)),
           ("r_hat", lambda ch: ws.r_hat(ch)), ("effective_sample_size", lambda ch: ws.effective_sample_size(ch)),
           ("quantiles_5", lambda ch: ws.quantiles(ch, [0.05, 0.25, 0.5, 0.75, 0.95]))]
    for name, f in fns:
        best = 1e9
        for _ in range(a.reps):
            ch = wa.MarkovChains.from_device(draws.data_ptr(), C, T, D)   # fresh: no cached per-chain moments
            t0 = time.perf_counter()
            f(ch)
            best = min(best, time.perf_counter() - t0)
            ch.close()
        out["ms"][name] = best * 1e3
        out["passes_gbps"][name] = gb / best
    print(json.dumps(out))


if __name__ == "__main__":
    main()
