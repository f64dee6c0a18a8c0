"""GPU probe (not a test): the drop-in call with its 16 GiB of draws streamed to a fresh host buffer, first and second
call of a fresh process.  Switches: WALNUTS_AMD_NO_PREFAULT=1, WALNUTS_AMD_PREFAULT_SLICE_KB, WALNUTS_AMD_PREFAULT_THREADS.
   python3 tests/gpu_probes/d2h_first_call.py <label>      (profiles/r04/prefault_ab.txt)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import walnuts_amd as wa
C, D, W, S = 65536, 1024, 20, 32
kw = dict(num_params=D, num_chains=C, seed=7, min_warmup_iter=W, max_warmup_iter=W, min_sampling_iter=S, max_sampling_iter=S)
wa.walnuts_device(wa.MODEL_STD_NORMAL, num_params=8, num_chains=4, seed=1, min_warmup_iter=2, max_warmup_iter=2, min_sampling_iter=2, max_sampling_iter=2)
ts = []
for rep in range(2):
    t = time.perf_counter()
    r = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    ts.append(time.perf_counter() - t)
    del r
print(sys.argv[1], "first call %.2f s, second %.2f s" % tuple(ts), flush=True)
