// wn_hip.h -- the one place that decides what the kernels are compiled against: the gfx950 platform layer
// (wn_gfx950.h: HIP runtime + wavefront primitives) or, under WN_CPU_SIM, the lock-step host emulation of a
// workgroup that the CPU test tier uses to drive the real host code and the kernels' control flow without a GPU
// (tests/cpusim/wn_cpusim.h -- test infrastructure, never part of libwalnuts_hip.so).
#pragma once

#if defined(WN_CPU_SIM)
#include "wn_cpusim.h"
#else
#include "wn_gfx950.h"
#endif
