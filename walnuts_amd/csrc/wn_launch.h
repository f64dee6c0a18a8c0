// wn_launch.h -- launch geometry table and per-model kernel dispatch (host side).
#pragma once

#include "wn_hip.h"

#include <algorithm>
#include <stdexcept>

#include "wn_params.h"

namespace wn {

struct InitParams;

// NW wavefronts cooperate on one chain, each lane holds EPL elements of every vector:
// padded dimension Dp = 64*NW*EPL.  `mem`: the streaming backend (TrajMem) -- vectors in HBM, any dimension; epl is 0 then.
struct Geometry {
  int nw, epl;
  bool mem;
};

// X(NW) -- wavefronts per chain of the streaming kernels
#if defined(WN_SIM_GEOMETRIES)
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(1)
#elif defined(WN_FAST_BUILD)
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(4)
#else
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(2) X(4) X(8) X(16)
#endif
inline bool mem_geometry_exists(int nw) {
#define WN_X(NW) \
  if (nw == NW) return true;
  WN_FOR_EACH_MEM_GEOMETRY(WN_X)
#undef WN_X
  return false;
}
inline int default_mem_waves() {
  int best = 0;
#define WN_X(NW) \
  if (best == 0 || NW == 16) best = NW; /* measured: 16 waves per chain stream fastest (profiles/) */
  WN_FOR_EACH_MEM_GEOMETRY(WN_X)
#undef WN_X
  return best;
}
constexpr int kMaxRegisterDim = 8192;

// X(NW, EPL) -- the register kernels (TrajChip, wn_chip.h)
#if defined(WN_SIM_GEOMETRIES)  // tests/cpusim: small workgroups only
#define WN_FOR_EACH_GEOMETRY(X) X(1, 2) X(1, 4) X(2, 2)
#elif defined(WN_FAST_BUILD)
#define WN_FOR_EACH_GEOMETRY(X) X(1, 2) X(4, 4) X(2, 8) X(1, 16)
#else
#define WN_FOR_EACH_GEOMETRY(X)                                                            \
  X(1, 2) X(1, 4) X(1, 8) X(1, 16) X(2, 2) X(2, 4) X(2, 8) X(4, 2) X(4, 4) X(4, 8) X(8, 2) \
  X(8, 4) X(8, 8) X(16, 4) X(16, 8)
#endif

inline bool geometry_exists(int nw, int epl) {
#define WN_X(NW, EPL) \
  if (nw == NW && epl == EPL) return true;
  WN_FOR_EACH_GEOMETRY(WN_X)
#undef WN_X
  return false;
}

// elems_per_lane == -1 requests the streaming backend explicitly (it is the default above kMaxRegisterDim)
inline Geometry choose_geometry(int dim, int nw_req, int epl_req, bool light_model = true) {
  Geometry g{0, 0, false};
  if (epl_req < 0 || (dim > kMaxRegisterDim && epl_req == 0)) {
    g.mem = true;
    g.nw = nw_req > 0 ? nw_req : default_mem_waves();
    if (!mem_geometry_exists(g.nw)) throw std::invalid_argument("unsupported waves_per_chain for the streaming kernels");
    return g;
  }
  if (nw_req > 0 && epl_req > 0) {
    if (!geometry_exists(nw_req, epl_req) || 64 * nw_req * epl_req < dim)
      throw std::invalid_argument("unsupported waves_per_chain / elems_per_lane for this num_params");
    g.nw = nw_req;
    g.epl = epl_req;
    return g;
  }
  // measured on MI355X (profiles/): the wave-uniform tree logic is replicated in every wavefront of a chain and
  // every reduction of a multi-wavefront chain is an LDS exchange behind a barrier, so ONE wavefront per chain wins
  // as long as the vectors fit its registers (16 elements per lane = 1024 dimensions: 2.43 ms against 2.57 ms for
  // two wavefronts on the headline workload); beyond that, as few wavefronts as possible
  (void)light_model;
  static const int pref_all[][2] = {{1, 2}, {1, 4}, {1, 8}, {1, 16}, {2, 8}, {4, 8}, {8, 8}, {16, 8}};
  const int(*pref)[2] = pref_all;
  const int npref = 8;
  for (int i = 0; i < npref; ++i) {
    const int* p = pref[i];
    if (64 * p[0] * p[1] >= dim && geometry_exists(p[0], p[1])) {
      g.nw = p[0];
      g.epl = p[1];
      return g;
    }
  }
  // fast builds carry a reduced table: take the smallest entry that fits
  int best = 1 << 30;
#define WN_X(NW, EPL)                                      \
  if (64 * NW * EPL >= dim && 64 * NW * EPL < best) {      \
    best = 64 * NW * EPL;                                  \
    g = Geometry{NW, EPL, false};                          \
  }
  WN_FOR_EACH_GEOMETRY(WN_X)
#undef WN_X
  if (g.nw == 0) throw std::invalid_argument("num_params exceeds the register-resident kernels (max 8192)");
  return g;
}

inline int default_workgroups_per_cu(const Geometry& g, int waves_per_simd) {
  if (g.mem) return g.nw >= 16 ? 1 : 16 / g.nw;  // streaming: latency is hidden by resident waves
  return std::max(1, 4 * waves_per_simd / g.nw);  // fill the register budget the kernel is built for
}
inline int padded_dim(const Geometry& g, int dim) {
  const int lanes = 64 * g.nw;
  if (!g.mem) return lanes * g.epl;
  return ((dim + 2 * lanes - 1) / (2 * lanes)) * 2 * lanes;
}

// defined once per model in wn_kernels_<model>.hip
#define WN_DECLARE_MODEL(tag)                                                                              \
  void launch_transition_##tag(const Geometry&, int grid, size_t smem, hipStream_t, const Params&);        \
  void launch_init_##tag(const Geometry&, int grid, size_t smem, hipStream_t, const InitParams&);          \
  void prepare_##tag(const Geometry&, size_t smem);                                                        \
  int register_pool_##tag(const Geometry&);                                                                \
  int waves_per_simd_##tag(const Geometry&);
WN_DECLARE_MODEL(std_normal)
WN_DECLARE_MODEL(diag_normal)
WN_DECLARE_MODEL(funnel)
#undef WN_DECLARE_MODEL

inline void launch_transition(int model, const Geometry& g, int grid, size_t smem, hipStream_t s, const Params& p) {
  switch (model) {
    case kStdNormal: launch_transition_std_normal(g, grid, smem, s, p); break;
    case kDiagNormal: launch_transition_diag_normal(g, grid, smem, s, p); break;
    case kFunnel: launch_transition_funnel(g, grid, smem, s, p); break;
    default: throw std::invalid_argument("unknown device model id");
  }
}
inline void launch_init(int model, const Geometry& g, int grid, size_t smem, hipStream_t s, const InitParams& q) {
  switch (model) {
    case kStdNormal: launch_init_std_normal(g, grid, smem, s, q); break;
    case kDiagNormal: launch_init_diag_normal(g, grid, smem, s, q); break;
    case kFunnel: launch_init_funnel(g, grid, smem, s, q); break;
    default: throw std::invalid_argument("unknown device model id");
  }
}
// pool vectors the register kernel of this model / geometry keeps in VGPRs (0 for the streaming kernels)
inline int register_pool(int model, const Geometry& g) {
  switch (model) {
    case kStdNormal: return register_pool_std_normal(g);
    case kDiagNormal: return register_pool_diag_normal(g);
    case kFunnel: return register_pool_funnel(g);
    default: throw std::invalid_argument("unknown device model id");
  }
}
// wavefronts per SIMD the register kernel of this model / geometry is compiled for (its VGPR budget)
inline int waves_per_simd(int model, const Geometry& g) {
  switch (model) {
    case kStdNormal: return waves_per_simd_std_normal(g);
    case kDiagNormal: return waves_per_simd_diag_normal(g);
    case kFunnel: return waves_per_simd_funnel(g);
    default: throw std::invalid_argument("unknown device model id");
  }
}
inline void prepare_kernels(int model, const Geometry& g, size_t smem) {
  switch (model) {
    case kStdNormal: prepare_std_normal(g, smem); break;
    case kDiagNormal: prepare_diag_normal(g, smem); break;
    case kFunnel: prepare_funnel(g, smem); break;
    default: throw std::invalid_argument("unknown device model id");
  }
}

}  // namespace wn
