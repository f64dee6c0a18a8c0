#!/usr/bin/env python3
"""GPU probe: the drop-in call end to end at the headline size -- walnutpie_sample_device (draws streamed to the
caller's host buffer, pageable or page-locked) against walnutpie_sample_device_resident (draws kept in HBM, every
k-th to the host, summaries on the device).   python tests/gpu_probes/sample_device_e2e.py [chains] [dim] [W] [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import walnuts_amd as wa

C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S = int(sys.argv[4]) if len(sys.argv) > 4 else 32
kw = dict(num_params=D, num_chains=C, seed=7, min_warmup_iter=W, max_warmup_iter=W, min_sampling_iter=S,
          max_sampling_iter=S, refresh=0)
print(f"# {C} chains x {D} dims, {W} warmup + {S} sampling iterations; draws per iteration = {C * D * 8 / 2**20:.0f} MiB")
# warm the library / context once (first import pages the image in)
wa.walnuts_device(wa.MODEL_STD_NORMAL, num_params=8, num_chains=4, seed=1, min_warmup_iter=2, max_warmup_iter=2,
                  min_sampling_iter=2, max_sampling_iter=2)
ref = None
for label, env, extra in (("host buffer, pageable (default)", {"WALNUTS_AMD_PIN_OUTPUT": "0"}, {}),
                          ("host buffer, WALNUTS_AMD_PIN_OUTPUT=1 (registered beside the streams)", {"WALNUTS_AMD_PIN_OUTPUT": "1"}, {}),
                          ("resident, every 8th draw to the host", {}, dict(keep_on_device=True, thin=8)),
                          ("resident, no draw to the host", {}, dict(keep_on_device=True, thin=0)),
                          ("host buffer, two shards on the one device (devices=[0, 0])", {"WALNUTS_AMD_PIN_OUTPUT": "0"},
                           dict(devices=[0, 0]))):
    os.environ.update(env)
    t_alloc = time.perf_counter()
    t0 = time.perf_counter()
    res = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw, **extra)
    dt = time.perf_counter() - t0
    chains = None
    if isinstance(res, tuple):
        res, chains = res
    moved = sum(r.nbytes for r in res)
    line = (f"{label:70s} {dt:7.2f} s  = {C * (W + S) / dt:10.3e} transitions/s; {S * C / dt:10.3e} draws/s; "
            f"D2H {moved / 2**30:6.2f} GiB ({moved / dt / 1e9:5.1f} GB/s of the call's wall time)")
    if chains is not None:
        t1 = time.perf_counter()
        m = chains.mean(); rh = chains.r_hat(); ess = chains.effective_sample_size()
        line += f"; device summaries (mean, r_hat, ess) {time.perf_counter() - t1:5.2f} s, max r_hat {rh.max():.4f}"
        if ref is not None and extra.get("thin"):
            k = extra["thin"]
            same = all(np.array_equal(res[c], ref[c][::k]) for c in (0, 1, C // 2, C - 1))
            line += f"; thinned rows == rows 0,{k},.. of the streamed run: {same}"
        chains.close()
    else:
        ref = res
    print(line, flush=True)
