#!/usr/bin/env python3
"""Time sampling transitions of an arbitrary build of libwalnuts_hip.so through the stable part of the C ABI
(used to bisect performance between commits):
   abi_bench.py <lib.so> <model id> <chains> <dim> <adapt> <steps> [waves_per_chain elems_per_lane workgroups_per_cu]"""
import ctypes as C
import sys
import time

import torch  # noqa: F401  (one HIP runtime per process: torch's first)

lib = C.CDLL(sys.argv[1])
model, chains, dim, adapt, steps = (int(x) for x in sys.argv[2:7])
vp, err = C.c_void_p, C.c_void_p()
cfg = (C.c_char * 512)()
lib.wn_default_config(cfg)
if len(sys.argv) > 9:  # wn_config: 4 x int32, 9 x double, then waves_per_chain / elems_per_lane / workgroups_per_cu
    import struct
    struct.pack_into("iii", cfg, 88, *(int(x) for x in sys.argv[7:10]))
    if len(sys.argv) > 10:
        struct.pack_into("i", cfg, 100, int(sys.argv[10]))  # lds_vectors
eng = vp()


def call(fn, *a):
    fn.restype = C.c_int
    rc = fn(*a, C.byref(err))
    if rc:
        lib.walnutpie_get_error_message.restype = C.c_char_p
        raise RuntimeError(lib.walnutpie_get_error_message(err).decode())


params = (C.c_double * dim)(*[(1.0 + (d % 16)) ** 2 for d in range(dim)]) if model == 1 else None
call(lib.wn_engine_create, C.byref(eng), model, dim, params, C.c_size_t(chains), cfg)
call(lib.wn_engine_init_positions, eng, C.c_uint64(1234), C.c_uint32(0), C.c_double(2.0))
call(lib.wn_engine_init_masses_from_grad, eng, C.c_double(1e-5))
steps_arr = (C.c_double * chains)(*([1.0] * chains))
call(lib.wn_engine_set_step_sizes, eng, steps_arr)
call(lib.wn_engine_adapt_step, eng, C.c_uint64(1234), C.c_uint32(0))
call(lib.wn_engine_seed, eng, C.c_uint64(1235), C.c_uint32(0))
for _ in range(adapt):
    call(lib.wn_engine_warmup_step, eng, None, C.c_int64(0))
call(lib.wn_engine_freeze, eng)
for _ in range(5):
    call(lib.wn_engine_sample_step, eng, None, C.c_int64(0))
call(lib.wn_engine_synchronize, eng)
g0 = C.c_int64()
call(lib.wn_engine_total_grad_evals, eng, C.byref(g0))
t0 = time.perf_counter()
for _ in range(steps):
    call(lib.wn_engine_sample_step, eng, None, C.c_int64(0))
call(lib.wn_engine_synchronize, eng)
dt = time.perf_counter() - t0
g1 = C.c_int64()
call(lib.wn_engine_total_grad_evals, eng, C.byref(g1))
print(f"{sys.argv[1].split('/')[-1]}: {1e3 * dt / steps:.4f} ms/step  {(g1.value - g0.value) / dt:.4g} grad-evals/s")
