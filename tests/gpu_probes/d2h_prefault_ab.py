"""GPU probe (not a test): interleaved A/B in ONE process of the host-buffer population helpers and the number of copy
streams of the draw sink (WALNUTS_AMD_NO_PREFAULT, WALNUTS_AMD_COPY_STREAMS); profiles/r04/prefault_ab.txt."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import walnuts_amd as wa
C, D, W, S = 65536, 1024, 20, 32
kw = dict(num_params=D, num_chains=C, seed=7, min_warmup_iter=W, max_warmup_iter=W, min_sampling_iter=S, max_sampling_iter=S)
wa.walnuts_device(wa.MODEL_STD_NORMAL, num_params=8, num_chains=4, seed=1, min_warmup_iter=2, max_warmup_iter=2, min_sampling_iter=2, max_sampling_iter=2)
ref = None
for rep in range(3):
    for nopf, lanes in (("1", "1"), ("0", "1"), ("0", "2"), ("0", "4"), ("0", "8"), ("1", "4")):
        os.environ["WALNUTS_AMD_NO_PREFAULT"] = nopf
        os.environ["WALNUTS_AMD_COPY_STREAMS"] = lanes
        t = time.perf_counter()
        r = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
        dt = time.perf_counter() - t
        same = ""
        if rep == 0:
            a = np.asarray(r[C - 1])
            if ref is None: ref = a.copy()
            same = " last chain equal: %s" % np.array_equal(a, ref)
        print(f"rep {rep} prefault {'off' if nopf == '1' else 'on '} copy streams {lanes}: {dt:.2f} s{same}", flush=True)
        del r
