import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import parity
for name, lib in (("default", None), ("nodpp", os.path.join(ROOT, "tests/gpu_probes/libwalnuts_nodpp.so"))):
    if lib and not os.path.exists(lib):
        continue
    for warm, samp in ((3, 0), (0, 3)):
        try:
            parity.run_case("std_normal", 100, 64, warmup=warm, sampling=samp, lib_path=lib, check_every=1, step=0.5)
            print(name, "warm", warm, "samp", samp, "OK")
        except AssertionError as e:
            print(name, "warm", warm, "samp", samp, "FAIL:", str(e)[:300])
