// wn_pregen.h -- the momentum's standard normals of transition t+1, generated while transition t runs.
//
// The momentum refresh (walnuts.hpp:528-529: rho = chol_mass * z, z ~ N(0, I)) is the one part of a transition that
// depends on nothing the chain computes: D normals keyed by (seed, chain, transition, element).  Inside the transition
// kernel it is ~1 750 of a chain's ~8 700 vector instructions and 11 % of its time (profiles/r02/timeline_sampling.txt),
// issued by the SIMD's ONLY wavefront -- the register kernels hold one chain per SIMD (512 registers), and one
// wavefront issues at most one vector instruction every ~5 cycles (tests/gpu_probes/fp64_latency.hip), so nearly half
// of the issue slots are idle.  This kernel needs ~40 registers and no LDS: launched on a second HIP stream it becomes
// resident BESIDE the persistent transition kernel (344 + 2 x 80 registers per SIMD lane fit the 512) and fills those
// slots with the next transition's normals, written to a [C][Dp] plane the transition kernel then streams in
// (kRngPregen).  Same Philox counters, same Box-Muller, same bits as the inline generator (wnd::stream_normal_pair).
#pragma once

#include "wn_devmath.h"
#include "wn_hip.h"
#include "wn_traj.h"

namespace wn {

constexpr int kPregenBlock = 256;

static __global__ __launch_bounds__(kPregenBlock) void momentum_pregen_kernel(int C, int Dp, uint64_t seed,
                                                                            uint32_t chain_offset, uint32_t transition,
                                                                            double* z_out /*[C][Dp]*/) {
  LaneTables tabs;
  tabs.load(static_cast<int>(threadIdx.x) & 63);
  const GatherTab tab{tabs};
  const int pairs = Dp / 2;  // a multiple of 64: every lane of a wavefront takes part in the table gathers
  const long long n = static_cast<long long>(C) * pairs;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const uint32_t chain = static_cast<uint32_t>(i / pairs);
    const uint32_t pair = static_cast<uint32_t>(i - static_cast<long long>(chain) * pairs);
    v2f64 z;
    double z0, z1;
    wnd::stream_normal_pair(seed, chain_offset + chain, transition, wnd::kStreamMomentum, pair, z0, z1, tab);
    z[0] = z0;
    z[1] = z1;
    stream_store(z, reinterpret_cast<v2f64*>(z_out + 2 * i));
  }
}

}  // namespace wn
