// wn_hip.h -- the one place the HIP runtime is pulled in.  Under WN_CPU_SIM (tests/cpusim, test
// infrastructure only -- never part of libwalnuts_hip.so) the same sources are compiled by g++ against
// a lock-step host emulation of a workgroup so that the host logic and the kernels' control flow can be
// exercised without a GPU.
#pragma once

#if defined(WN_CPU_SIM)
#include "wn_cpusim.h"
#else
#include <hip/hip_runtime.h>
#define WN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) double name[]
// LDS is addressed through address-space-3 pointers only, so every pool / scratch access is a
// ds_* instruction (no flat aperture tests)
#define WN_LDS __attribute__((address_space(3)))
typedef double v2f64 __attribute__((ext_vector_type(2)));
#endif
