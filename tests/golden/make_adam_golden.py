"""Regenerates tests/golden/adam_reference.json from the REAL reference Adam.

Needs /root/reference (build container only): `make -C oracle ref` compiles
/root/reference/include/walnutpie/adam.hpp (Eigen-free) into
oracle/_ref/libadam_ref.so; this script feeds it acceptance sequences and stores
inputs + outputs (data only, no reference source text).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import wno  # noqa: E402

wno.build(force=True)
L = wno.adam_ref_lib()
assert L is not None, "oracle/_ref/libadam_ref.so missing (no reference tree?)"
rng = np.random.default_rng(20261001)
cases = []
param_sets = [
    dict(step_init=1.0, target=0.8, lr=0.05, b1=0.8, b2=0.9, eps=1e-4, decay=0.5),   # WarmupConfig defaults
    dict(step_init=0.39140625, target=0.8, lr=0.05, b1=0.8, b2=0.9, eps=1e-4, decay=0.5),
    dict(step_init=2.5, target=0.65, lr=0.1, b1=0.9, b2=0.999, eps=1e-8, decay=0.75),
    dict(step_init=0.01, target=0.9, lr=0.2, b1=0.5, b2=0.5, eps=1e-3, decay=0.95),  # examples.cpp decay
]
for ps in param_sets:
    for n, kind in ((5, "survey"), (64, "uniform"), (200, "beta")):
        if kind == "survey":
            alphas = np.array([0.9, 0.5, 0.99, 0.1, 0.8])
        elif kind == "uniform":
            alphas = rng.uniform(0.0, 1.0, n)
        else:
            alphas = rng.beta(4.0, 1.0, n)
        out = np.empty(n)
        L.adam_ref_run(ps["step_init"], ps["target"], ps["lr"], ps["b1"], ps["b2"], ps["eps"], ps["decay"],
                       wno._p(alphas), n, wno._p(out))
        cases.append(dict(params=ps, alphas=[float.hex(float(a)) for a in alphas],
                          steps=[float.hex(float(s)) for s in out]))
path = os.path.join(ROOT, "tests", "golden", "adam_reference.json")
json.dump(dict(source="reference include/walnutpie/adam.hpp compiled by oracle/Makefile (target ref)",
               cases=cases), open(path, "w"), indent=0)
print("wrote", path, len(cases), "cases")
