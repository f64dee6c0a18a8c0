// wn_launch.h -- launch geometry table and per-model kernel dispatch (host side).
#pragma once

#include "wn_hip.h"

#include <algorithm>
#include <stdexcept>
#include <string>

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
// (WN_ONLY_*: a device model compiled at run time -- walnuts_amd/models.py -- instantiates the ONE geometry its engine
// will launch, which the library names through wn_geometry_for(): seconds of hipcc instead of minutes)
#if defined(WN_ONLY_MEM_NW) && defined(WN_ONLY_MEM_NW_ALSO)  // (the default streaming choice depends on the model: both)
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(WN_ONLY_MEM_NW) X(WN_ONLY_MEM_NW_ALSO)
#elif defined(WN_ONLY_MEM_NW)
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(WN_ONLY_MEM_NW)
#elif defined(WN_ONLY_NW)
#define WN_FOR_EACH_MEM_GEOMETRY(X)
#elif defined(WN_SIM_GEOMETRIES)
#define WN_FOR_EACH_MEM_GEOMETRY(X) X(1) X(2) X(4)
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
// Streaming kernels of kHeldWaves wavefronts per chain keep the moving end of the trajectory in registers (wn_traj.h,
// TrajMem HOLD) for models whose gradient takes one pass, up to hold_tiles x 2 x 64 x kHeldWaves dimensions (16 384):
// measured on config #4 1.65e7 gradient evaluations per second against 1.05e7 for sixteen wavefronts streaming both
// ends (profiles/r05).  They are the default where they apply.
constexpr int kHeldWaves = 8;
// ... down to where the register kernels need sixteen wavefronts per chain ((16, 8): 4 097-8 192 dimensions at 128
// registers per wavefront, 400-600 bytes of them spilled, every reduction a 16-wavefront barrier): measured at 8 192
// chains of the diagonal normal, 8 192 dimensions 5.9 ms per step against 11.6, 6 000 dimensions 4.4 against 11.1 --
// while (8, 8) at 4 096 dimensions runs 1.08 ms against 3.13; rw1 (two passes, halo reads) 8 192 dimensions 49.4
// against 77.1, 5 000: 36.4 against 66.4; the funnel (two passes, sums) is the exception: 29.3 against (16, 8)'s 18.6
// at 8 192 dimensions (profiles/r05/cfg4_held_moving_end.md).  The model's own limit: ModelOps::register_dim_limit
// (mem_register_dim_limit<Model>(), wn_traj.h: 4 096, the funnel's kind 8 192).

// X(NW, EPL) -- the register kernels (TrajChip, wn_chip.h)
#if defined(WN_ONLY_NW)
#define WN_FOR_EACH_GEOMETRY(X) X(WN_ONLY_NW, WN_ONLY_EPL)
#elif defined(WN_ONLY_MEM_NW)
#define WN_FOR_EACH_GEOMETRY(X)
#elif defined(WN_SIM_GEOMETRIES)  // tests/cpusim: a cross-section (the headline's (1, 16) and its neighbours included)
#define WN_FOR_EACH_GEOMETRY(X) X(1, 2) X(1, 4) X(2, 2) X(1, 16) X(2, 8) X(4, 4) X(8, 8)
#elif defined(WN_FAST_BUILD)
#define WN_FOR_EACH_GEOMETRY(X) X(1, 2) X(4, 4) X(2, 8) X(1, 16) X(4, 8) X(8, 8) X(2, 16) X(4, 16)
#else
#define WN_FOR_EACH_GEOMETRY(X)                                                            \
  X(1, 2) X(1, 4) X(1, 8) X(1, 16) X(2, 2) X(2, 4) X(2, 8) X(2, 16) X(4, 2) X(4, 4) X(4, 8) X(4, 16) X(8, 2) \
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
// held_tiles: the model's ModelOps::hold_tiles(kHeldWaves) -- 0 for a model without such kernels (or not known yet);
// register_dim_limit: its ModelOps::register_dim_limit
inline Geometry choose_geometry(int dim, int nw_req, int epl_req, bool params_in_registers = false,
                                int preferred_epl = 0, int held_tiles = 0, int register_dim_limit = kMaxRegisterDim) {
  Geometry g{0, 0, false};
  const bool held = held_tiles > 0 && dim <= 2 * 64 * kHeldWaves * held_tiles && mem_geometry_exists(kHeldWaves);
  const int max_register_dim = (held && nw_req == 0) ? std::min(register_dim_limit, kMaxRegisterDim) : kMaxRegisterDim;
  if (epl_req < 0 || (dim > max_register_dim && epl_req == 0)) {
    g.mem = true;
    g.nw = nw_req > 0 ? nw_req : held ? kHeldWaves : default_mem_waves();
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
  // a model's own hint (wn_model_api.h: kPreferredElemsPerLane): the fewest wavefronts that hold the vectors at that
  // many elements per lane
  if (preferred_epl > 0) {
    for (int nw = 1; nw <= 16; nw *= 2)
      if (geometry_exists(nw, preferred_epl) && 64 * nw * preferred_epl >= dim) return Geometry{nw, preferred_epl, false};
  }
  // measured on MI355X (profiles/): the wave-uniform tree logic is replicated in every wavefront of a chain and
  // every reduction of a multi-wavefront chain is an LDS exchange behind a barrier, so ONE wavefront per chain wins
  // as long as the vectors fit its registers (16 elements per lane = 1024 dimensions: 2.43 ms against 2.57 ms for
  // two wavefronts on the headline workload); beyond that, as few wavefronts as possible.
  // (A model with a per-coordinate parameter vector holds 32 more registers at 16 elements per lane; while the
  // compiler used the accumulator file as it saw fit that made two wavefronts faster for such models.  With the
  // span's other end parked explicitly it no longer does -- config #2, 4 096 x 1 024 ill-conditioned normal:
  // 3.78e8 gradient evaluations per second with (1, 16) against 3.60e8 with (2, 8) -- so `params_in_registers`
  // does not change the choice any more.)
  (void)params_in_registers;
  // Beyond 1 024 dimensions (round 6): SIXTEEN elements per lane on two / four wavefronts instead of eight on four /
  // eight -- half as many copies of the wave-uniform tree logic, half as many wavefronts behind every reduction's
  // barrier, a full-width leaf per lane.  Measured, 8 192 chains (profiles/r06/mid_dimensions.txt): diagonal normal
  // 2 048 dims (2,16) 0.389 against (4,8) 0.402 ms per step, 3 000 dims (4,16) 0.924 against (8,8) 1.049, 4 096 dims
  // 0.966 against 1.087; funnel 2 048: 0.804 against 1.092, 4 096: 2.40 against 2.31 (the one case that loses, 4 %).
  static const int pref[][2] = {{1, 2}, {1, 4}, {1, 8}, {1, 16}, {2, 16}, {4, 16}, {8, 8}, {16, 8}};
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

// ---- model registry -----------------------------------------------------------------------------------
// One entry per device model, filled in by the model's own translation unit (WN_REGISTER_MODEL in
// wn_model_api.h) when the library is loaded.  Nothing else in the host code names a model: adding one is
// writing its header and a three-line .hip file ("Adding a device model").
struct ModelOps {
  int id;
  const char* name;
  bool uses_params;   // needs a parameter vector of num_params doubles
  bool elementwise;   // has streaming (num_params > 8192) kernels: an element-wise gradient, or a stated streaming form
  int (*preferred_epl)(int num_params);  // the model's geometry hint: elements per lane (0 = the default policy)
  void (*launch_transition)(const Geometry&, int grid, size_t smem, hipStream_t, const Params&);
  void (*launch_init)(const Geometry&, int grid, size_t smem, hipStream_t, const InitParams&);
  void (*prepare)(const Geometry&, size_t smem);
  int (*waves_per_simd)(const Geometry&);
  void (*host_params)(double* params, int num_params);  // validate / transform the parameter vector before upload
  void (*validate)(int num_params);
  int (*hold_tiles)(int nw);  // streaming kernels of nw wavefronts: tiles of the moving end held in registers (0: none)
  int register_dim_limit;     // the largest num_params the register kernels serve by default when the held streaming
                              // kernels exist (mem_register_dim_limit<Model>(), wn_traj.h)
};
constexpr int kMaxModels = 64;
inline const ModelOps** model_table() {
  static const ModelOps* table[kMaxModels] = {};
  return table;
}
// Registration runs in static initialisers (library load): nothing may throw there -- an exception would end the
// process in std::terminate inside dlopen with no usable message.  A bad WN_MODEL_ID of an out-of-tree model (out of
// range, or taken: the first entry stays) is recorded instead and reported by wn_engine_create as a config error.
inline std::string& registry_error() {
  static std::string msg;
  return msg;
}
// Everything a separately compiled model and the library must agree on: the layout of what crosses the boundary.
constexpr int kModelAbiVersion = 8;
struct ModelAbi {
  int version;
  unsigned sizeof_ops, sizeof_params, sizeof_geometry;
};
}  // namespace wn
// A device model compiled at RUN time (walnuts_amd/models.py: one hipcc -shared of its five-line translation unit
// against these headers) is a shared object of its own; when it is loaded, its registration lands in the LIBRARY's
// registry through this exported entry point (wn_engine.hip) -- the device counterpart of handing the reference a
// LOGP_CFUNC / numba cfunc (python/src/walnutpie/walnutpy.cpp:131-132, pyfunc.py:216).  -> 0, or -1 (wn_model_error()).
extern "C" __attribute__((visibility("default"))) int wn_plugin_register_model(const void* ops, const void* abi);
namespace wn {
inline bool register_model_here(const ModelOps* ops) {
  if (ops->id < 0 || ops->id >= kMaxModels) {
    registry_error() = std::string("device model '") + ops->name + "': WN_MODEL_ID " + std::to_string(ops->id) +
                       " is out of range (0 <= id < " + std::to_string(kMaxModels) + ")";
    return false;
  }
  if (model_table()[ops->id] != nullptr) {
    registry_error() = std::string("device model '") + ops->name + "': WN_MODEL_ID " + std::to_string(ops->id) +
                       " is already taken by '" + model_table()[ops->id]->name + "'";
    return false;
  }
  model_table()[ops->id] = ops;
  return true;
}
inline ModelAbi model_abi() {
  return ModelAbi{kModelAbiVersion, static_cast<unsigned>(sizeof(ModelOps)), static_cast<unsigned>(sizeof(Params)),
                  static_cast<unsigned>(sizeof(Geometry))};
}
#if defined(WN_MODEL_PLUGIN)
inline bool register_model(const ModelOps* ops) {  // (a static initialiser of the model's shared object)
  const ModelAbi abi = model_abi();
  return wn_plugin_register_model(ops, &abi) == 0;
}
#else
inline bool register_model(const ModelOps* ops) { return register_model_here(ops); }
#endif
inline const ModelOps& model_ops(int model) {
  if (model < 0 || model >= kMaxModels || model_table()[model] == nullptr)
    throw std::invalid_argument("unknown device model id");
  return *model_table()[model];
}
inline void launch_transition(int model, const Geometry& g, int grid, size_t smem, hipStream_t s, const Params& p) {
  model_ops(model).launch_transition(g, grid, smem, s, p);
}
inline void launch_init(int model, const Geometry& g, int grid, size_t smem, hipStream_t s, const InitParams& q) {
  model_ops(model).launch_init(g, grid, smem, s, q);
}
// wavefronts per SIMD the register kernel of this model / geometry is compiled for (its VGPR budget)
inline int waves_per_simd(int model, const Geometry& g) { return model_ops(model).waves_per_simd(g); }
inline void prepare_kernels(int model, const Geometry& g, size_t smem) { model_ops(model).prepare(g, smem); }

}  // namespace wn
