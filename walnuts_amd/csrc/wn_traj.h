// wn_traj.h -- the GPU-resident Walnuts transition for one chain per workgroup (gfx950).
//
// One workgroup of NW wavefronts owns one chain at a time.  Lane l of the
// workgroup (L = 64*NW lanes) owns the 16-byte element pairs (k*L + l),
// k = 0..EPL/2-1, of every D-vector of that chain; the moving trajectory end
// (theta, rho, grad), the inverse mass diagonal and the macro step's restart
// state stay in VGPRs for the whole transition, so a leapfrog micro step
// (walnuts.hpp:329-332) touches no memory at all.  The span bookkeeping of
// NUTS (SpanW, walnuts.hpp:34-131) is reduced to a pool of D-vector buffers,
// the first `pool_lds` of them in LDS and the rest in a per-workgroup HBM arena,
// addressed through wave-uniform indices.
//
// What is restated from the reference, with the recursion of build_span
// (walnuts.hpp:464-495) turned into a post-order loop over leaves that draws
// random numbers in exactly the reference's order:
//   transition_w   walnuts.hpp:520-563      Traj::run
//   macro_step     walnuts.hpp:307-345      Traj::macro_step
//   within_tolerance / reversible  :218-279 Traj::within_tolerance / reversible
//   uturn          walnuts.hpp:192-201      Traj::uturn_against
//   combine        walnuts.hpp:368-387      inline in Traj::run (Barker / Metropolis)
//   logp_momentum  util.hpp:220-223         Traj::energy
//   log_sum_exp    util.hpp:174-183         log_sum_exp
//   Adam           adam.hpp:70-93           Traj::adam_observe
//   MassEstimator / OnlineMoments / MinMicroStepsAdaptHandler
//                  adaptive_walnuts.hpp:54-94,127-157,234-251; online_moments.hpp:184-191
//
// All element-wise arithmetic keeps the reference's association order and the
// file is compiled with -ffp-contract=off, so element-wise results carry the
// reference's bits.  Sums over D run in a fixed order (per-lane partial in
// index order, xor butterfly 1..32 inside a wavefront, wavefronts left to
// right) that the CPU oracle can replay exactly.
#pragma once

#include "wn_hip.h"

#include "wn_devmath.h"
#include "wn_params.h"

namespace wn {

// ---- wave-uniform helpers ----------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
#if defined(WN_CPU_SIM)
  return wnsim::readfirstlane(v);
#endif
  const uint64_t u = wnd::as_u64(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
  return wnd::as_f64((static_cast<uint64_t>(hi) << 32) | lo);
}

// xor-butterfly sum over the 64 lanes, offsets 1,2,4,8,16,32: every lane ends with the same bits
// (a+b == b+a), and the CPU oracle replays exactly this association order.
#if defined(WN_CPU_SIM) || defined(WN_DISABLE_DPP)
__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double lane_value(double v, int src_lane) { return __shfl(v, src_lane, 64); }
#else
// gfx950: offsets 1,2 are quad permutes, 4 and 8 are row_half_mirror / row_mirror (the groups are already
// uniform there, so the mirrored lane holds the xor partner's value), 16 and 32 are v_permlane{16,32}_swap.
// All VALU: no LDS crossbar traffic (ds_bpermute) on the reduction path.
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
  const uint64_t u = wnd::as_u64(v);
  const int lo = static_cast<int>(u), hi = static_cast<int>(u >> 32);
  const int plo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  const int phi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return wnd::as_f64((static_cast<uint64_t>(static_cast<uint32_t>(phi)) << 32) | static_cast<uint32_t>(plo));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = v + dpp_partner<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  v = v + dpp_partner<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  v = v + dpp_partner<0x141>(v);  // row_half_mirror      : partner quad  (lane ^ 4)
  v = v + dpp_partner<0x140>(v);  // row_mirror           : partner octet (lane ^ 8)
  {
    const uint64_t u = wnd::as_u64(v);
    const uint32_t lo = static_cast<uint32_t>(u), hi = static_cast<uint32_t>(u >> 32);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = wnd::as_f64((static_cast<uint64_t>(b[0]) << 32) | a[0]) + wnd::as_f64((static_cast<uint64_t>(b[1]) << 32) | a[1]);
  }
  {
    const uint64_t u = wnd::as_u64(v);
    const uint32_t lo = static_cast<uint32_t>(u), hi = static_cast<uint32_t>(u >> 32);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = wnd::as_f64((static_cast<uint64_t>(b[0]) << 32) | a[0]) + wnd::as_f64((static_cast<uint64_t>(b[1]) << 32) | a[1]);
  }
  return v;
}
// value held by lane `src_lane` (wave-uniform index) as a scalar
__device__ __forceinline__ double lane_value(double v, int src_lane) {
  const uint64_t u = wnd::as_u64(v);
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(u), src_lane);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(u >> 32), src_lane);
  return wnd::as_f64((static_cast<uint64_t>(hi) << 32) | lo);
}
#endif

// util.hpp:174-183.  One of exp(x1-m), exp(x2-m) is exp(0) == 1 exactly, so only the other one is
// evaluated; the sum is commutative, hence the same bits as the two-exp form.
__device__ __forceinline__ double log_sum_exp(double x1, double x2) {
  const double m = fmax(x1, x2);
  if (x1 != x1 || x2 != x2) return __builtin_nan("");
  if (__builtin_isinf(m) || (x1 + x2) != (x1 + x2)) return fmax(x1, x2);
#if defined(WN_VARIANT_LSE2)
  return m + wnd::dlog(wnd::dexp(x1 - m) + wnd::dexp(x2 - m));
#else
  const double d = (x1 < x2) ? (x1 - m) : (x2 - m);
  return m + wnd::dlog(1.0 + wnd::dexp(d));
#endif
}

// ---- target densities (device form of the LogpGrad contract, concepts.hpp:258-262) ----
// eval():   writes grad for the lane's elements and returns the lane's partial of
//           the log-density sum; may reduce internally through cx.
// finish(): turns the reduced sum into logp.
struct StdNormalModel {  // examples/walnutpie_api.cpp:37-41
  static constexpr int kKind = kStdNormal;
  static constexpr bool kUsesParams = false;
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static double eval(Cx&, const double (&th)[EPL], double (&g)[EPL],
                                                const double (&)[EPL], Aux&) {
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      g[j] = -th[j];
      p += th[j] * th[j];
    }
    return p;
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return -0.5 * sum; }
};

struct DiagNormalModel {  // examples/examples.cpp:20-31, params = sigma_sq
  static constexpr int kKind = kDiagNormal;
  static constexpr bool kUsesParams = true;
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static double eval(Cx&, const double (&th)[EPL], double (&g)[EPL],
                                                const double (&s2)[EPL], Aux&) {
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      g[j] = -th[j] / s2[j];
      p += -0.5 * th[j] * th[j] / s2[j];
    }
    return p;
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return sum; }
};

struct FunnelModel {  // Neal's funnel, SURVEY.md §8d cfg3 (not in the reference)
  static constexpr int kKind = kFunnel;
  static constexpr bool kUsesParams = false;
  struct Aux {
    double v, S, hev;
  };
  template <int EPL, class Cx>
  __device__ __forceinline__ static double eval(Cx& cx, const double (&th)[EPL], double (&g)[EPL],
                                                const double (&)[EPL], Aux& aux) {
    const double v = cx.element0(th[0]);
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) sp += (cx.index(j) == 0) ? 0.0 : th[j] * th[j];
    const double S = cx.sum1(sp);
    const double ev = wnd::dexp(-v);
    const double hd = 0.5 * static_cast<double>(cx.dim() - 1);
    const double hev = 0.5 * ev;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      double gj = -(th[j] * ev);
      if (cx.index(j) == 0) gj = ((-v / 9.0) + hev * S) - hd;
      g[j] = cx.valid(j) ? gj : 0.0;
    }
    aux.v = v;
    aux.S = S;
    aux.hev = hev;
    return 0.0;
  }
  __device__ __forceinline__ static double finish(double, const Aux& a, int D) {
    const double hd = 0.5 * static_cast<double>(D - 1);
    return ((-(a.v * a.v) / 18.0) - a.hev * a.S) - hd * a.v;
  }
};

// ---- optional phase profiler (tests/gpu_probes only; compiled out of the product build) -------------
#if defined(WN_PHASE_PROFILE) && !defined(WN_CPU_SIM)
enum { kPhIdle = 0, kPhPrologue, kPhLeapfrog, kPhEnergy, kPhRestart, kPhReversible, kPhUturn, kPhCombine, kPhPush,
       kPhTopMerge, kPhDoublingStart, kPhEpilogue, kPhCount };
__device__ unsigned long long wn_phase_cycles[kPhCount];
#define WN_PHASE(k) this->phase_mark(k)
#define WN_PHASE_OUTER(k) t.phase_mark(k)
#else
#define WN_PHASE(k) ((void)0)
#define WN_PHASE_OUTER(k) ((void)0)
#endif

constexpr int kHot = -1;    // "this vector currently lives in the VGPR trajectory end"
constexpr int kStart = -2;  // "this vector is the macro step's restart state (= the previous leaf), in VGPRs"

template <class Model, int NW, int EPL, bool START_REGS>
struct Traj {
  static constexpr int L = 64 * NW;
  static constexpr int NP = EPL / 2;
  static_assert(EPL % 2 == 0, "lanes own 16-byte pairs");

  // per-wave scalar scratch in LDS
  struct Meta {
    double adam[6];
    double logsum[kMaxLevels];
    double lpsel[kMaxLevels];
    int in_th[kMaxLevels];
    int in_rh[kMaxLevels];
    int sel[kMaxLevels];
    double u[64];   // tree draws draw_base .. draw_base+63 of this transition
    double lu[64];  // their logarithms
#if defined(WN_PHASE_PROFILE) && !defined(WN_CPU_SIM)
    unsigned long long prof[16];
    unsigned long long prof_last;
    int prof_cur;
#endif
  };
#if defined(WN_PHASE_PROFILE) && !defined(WN_CPU_SIM)
  __device__ __forceinline__ void phase_mark(int k) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      meta->prof[meta->prof_cur] += t - meta->prof_last;
      meta->prof_last = t;
      meta->prof_cur = k;
    }
  }
  __device__ __forceinline__ void phase_begin() {
    if (lane == 0) {
      for (int i = 0; i < 16; ++i) meta->prof[i] = 0;
      meta->prof_last = __builtin_amdgcn_s_memtime();
      meta->prof_cur = kPhIdle;
    }
  }
  __device__ __forceinline__ void phase_end() {
    phase_mark(kPhIdle);
    if (lane == 0)
      for (int i = 0; i < kPhCount; ++i) atomicAdd(&wn_phase_cycles[i], meta->prof[i]);
  }
#endif
  static_assert(sizeof(Meta) <= kMetaDoubles * sizeof(double), "meta scratch too small");

  const Params& P;
  WN_LDS double* lds_pool;
  WN_LDS Meta* meta;
  WN_LDS double* red;  // [2][NW][2] cross-wave reduction scratch
  WN_LDS double* bcast;
  double* arena;
  int tid, lane, wave;
  int chain;
  int Dp;

  double th[EPL], rh[EPL], g[EPL], im[EPL], mp[EPL];
  double th0[EPL], rh0[EPL], g0[EPL];
  int start_buf[3];
  unsigned long long free_mask;
  int red_parity;
  long long n_grad;
  int n_draw;
  int draw_base;        // first tree-draw index held in meta->u / meta->lu (-1: none)
  int err;
  double step, max_error;
  double w_draw0, w_score0;  // estimator weights at entry (read once: another wave's lane 0 rewrites them at exit)
  int min_micro;
  typename Model::Aux aux;

  __device__ __forceinline__ Traj(const Params& p, WN_LDS double* pool, WN_LDS Meta* m, WN_LDS double* r,
                                  WN_LDS double* bc, double* ar)
      : P(p), lds_pool(pool), meta(m), red(r), bcast(bc), arena(ar) {
    tid = threadIdx.x;
    lane = tid & 63;
    wave = tid >> 6;
    Dp = p.dim_padded;
    red_parity = 0;
  }

  // ---- model context -----------------------------------------------------------
  __device__ __forceinline__ int index(int j) const { return ((j >> 1) * L + tid) * 2 + (j & 1); }
  __device__ __forceinline__ bool valid(int j) const { return index(j) < P.dim; }
  __device__ __forceinline__ int dim() const { return P.dim; }
  __device__ __forceinline__ double element0(double mine) {
    // element 0 is slot 0 of thread 0
    if (NW == 1) return __shfl(mine, 0, 64);
    if (tid == 0) bcast[0] = mine;
    __syncthreads();
    const double v = bcast[0];
    __syncthreads();
    return v;
  }

  // ---- reductions -----------------------------------------------------------------
  __device__ __forceinline__ void sum2(double& a, double& b) {
    a = wave_sum(a);
    b = wave_sum(b);
    if (NW > 1) {
      WN_LDS double* r = red + red_parity * (NW * 2);
      if (lane == 0) {
        r[wave * 2] = a;
        r[wave * 2 + 1] = b;
      }
      __syncthreads();
      double ta = r[0], tb = r[1];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        ta = ta + r[w * 2];
        tb = tb + r[w * 2 + 1];
      }
      a = ta;
      b = tb;
      red_parity ^= 1;
    }
    a = uni(a);
    b = uni(b);
  }
  __device__ __forceinline__ double sum1(double a) {
    double b = 0.0;
    sum2(a, b);
    return a;
  }

  // ---- vector buffers -------------------------------------------------------------
  __device__ __forceinline__ void vload(const double* base, double (&v)[EPL]) const {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const v2f64 t = *reinterpret_cast<const v2f64*>(base + (k * L + tid) * 2);
      v[2 * k] = t[0];
      v[2 * k + 1] = t[1];
    }
  }
  __device__ __forceinline__ void vstore(double* base, const double (&v)[EPL]) const {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f64 t;
      t[0] = v[2 * k];
      t[1] = v[2 * k + 1];
      *reinterpret_cast<v2f64*>(base + (k * L + tid) * 2) = t;
    }
  }
  __device__ __forceinline__ void lds_load(const WN_LDS double* base, double (&v)[EPL]) const {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const v2f64 t = *reinterpret_cast<const WN_LDS v2f64*>(base + (k * L + tid) * 2);
      v[2 * k] = t[0];
      v[2 * k + 1] = t[1];
    }
  }
  __device__ __forceinline__ void lds_store(WN_LDS double* base, const double (&v)[EPL]) const {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f64 t;
      t[0] = v[2 * k];
      t[1] = v[2 * k + 1];
      *reinterpret_cast<WN_LDS v2f64*>(base + (k * L + tid) * 2) = t;
    }
  }
  __device__ __forceinline__ void pool_load(int b, double (&v)[EPL]) const {
    if (b < P.pool_lds) {
      lds_load(lds_pool + b * Dp, v);
    } else {
      vload(arena + static_cast<long long>(b - P.pool_lds) * Dp, v);
    }
  }
  __device__ __forceinline__ void pool_store(int b, const double (&v)[EPL]) const {
    if (b < P.pool_lds) {
      lds_store(lds_pool + b * Dp, v);
    } else {
      vstore(arena + static_cast<long long>(b - P.pool_lds) * Dp, v);
    }
  }
  __device__ __forceinline__ int alloc() {
    if (free_mask == 0ull) {
      err = 1;
      return 0;
    }
    const int b = uni(__builtin_ctzll(free_mask));
    free_mask &= free_mask - 1ull;
    return b;
  }
  // long-lived vectors (accumulated span ends, parked states) take the highest free buffer so that the
  // LDS-resident low indices stay available for the short-lived span-stack entries
  __device__ __forceinline__ int alloc_cold() {
    if (free_mask == 0ull) {
      err = 1;
      return 0;
    }
    const int b = uni(63 - __builtin_clzll(free_mask));
    free_mask &= ~(1ull << b);
    return b;
  }
  __device__ __forceinline__ void release(int b) {
    if (b >= 0) free_mask |= (1ull << b);
  }
  __device__ __forceinline__ void release_unless(int b, int k0, int k1, int k2) {
    if (b >= 0 && b != k0 && b != k1 && b != k2) free_mask |= (1ull << b);
  }

  // ---- randomness (util.hpp:102,112 order; counter-based stream or host-fed variates) ----
  // The tree consumes wave-uniform scalars one at a time.  They are produced 64 at a time, lane j
  // computing draw number draw_base + j and its logarithm into the wave's LDS scratch, and handed out with
  // a broadcast LDS read.  (A v_readlane hand-out from registers was miscompiled by ROCm 7.2's backend:
  // after a refill the read used the stale register pair; caught by the bit-exact parity tests.)
  __device__ __forceinline__ void refill_draws(int base) {
    draw_base = base;
    const int j = base + lane;
    double u;
    if (P.rng_mode == kRngBuffer) {
      u = j < P.u_stride ? P.u_buf[static_cast<long long>(chain) * P.u_stride + j] : 0.5;
    } else {
      u = wnd::stream_uniform(P.seed, P.chain_offset + chain, P.transition, wnd::kStreamTree,
                              static_cast<uint32_t>(j));
    }
    meta->u[lane] = u;
    meta->lu[lane] = wnd::dlog(u);
  }
  __device__ __forceinline__ int next_draw_slot() {
    const int j = uni(n_draw);
    ++n_draw;
    if (draw_base < 0 || j - draw_base >= 64) refill_draws(j & ~63);
    return j - draw_base;
  }
#if defined(WN_VARIANT_NOCACHE)
  __device__ __forceinline__ double uniform01() {
    const int j = n_draw++;
    if (P.rng_mode == kRngBuffer) return uni(P.u_buf[static_cast<long long>(chain) * P.u_stride + j]);
    return uni(wnd::stream_uniform(P.seed, P.chain_offset + chain, P.transition, wnd::kStreamTree,
                                   static_cast<uint32_t>(j)));
  }
  __device__ __forceinline__ double log_uniform01() { return wnd::dlog(uniform01()); }
#else
  __device__ __forceinline__ double uniform01() { return uni(meta->u[next_draw_slot()]); }
  __device__ __forceinline__ double log_uniform01() { return uni(meta->lu[next_draw_slot()]); }
#endif

  // ---- Hamiltonian pieces ------------------------------------------------------------
  __device__ __forceinline__ double model_eval() {
    ++n_grad;
    return Model::eval(*this, th, g, mp, aux);
  }
  // joint log density of the moving end: logp_pos + logp_momentum (util.hpp:220-223)
  __device__ __forceinline__ void energy(double lp_partial, double& logp_pos, double& logp_joint) {
    double ke = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) ke += im[j] * (rh[j] * rh[j]);
    sum2(lp_partial, ke);
    // wave-uniform results go back to scalar registers: they live long and would otherwise hold VGPR pairs
    logp_pos = uni(Model::finish(lp_partial, aux, P.dim));
    logp_joint = uni(logp_pos + (-0.5 * ke));
  }
  // n leapfrog micro steps on the VGPR state (walnuts.hpp:328-333); returns the
  // last evaluation's log-density partial
  __device__ __forceinline__ double leapfrog(double h, int n) {
    const double half = 0.5 * h;
    double part = 0.0;
    for (int s = 0; s < n; ++s) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) rh[j] += half * g[j];
#pragma unroll
      for (int j = 0; j < EPL; ++j) th[j] += h * im[j] * rh[j];
      part = model_eval();
#pragma unroll
      for (int j = 0; j < EPL; ++j) rh[j] += half * g[j];
    }
    return part;
  }

  // adam.hpp:70-86
  __device__ __forceinline__ void adam_observe(double alpha) {
    WN_LDS double* a = meta->adam;
    double theta = a[0], m = a[1], v = a[2], t = a[3], b1p = a[4], b2p = a[5];
    t += 1;
    b1p *= P.adam_b1;
    b2p *= P.adam_b2;
    const double grad = P.adam_target - alpha;
    m = P.adam_b1 * m + (1 - P.adam_b1) * grad;
    v = P.adam_b2 * v + (1 - P.adam_b2) * grad * grad;
    const double m_hat = m / (1 - b1p);
    const double v_hat = v / (1 - b2p);
    const double lr_t = P.adam_lr / wnd::dpow_pos(t, P.adam_decay);
    const double denom = __builtin_sqrt(v_hat) + P.adam_eps;
    theta -= lr_t * m_hat / denom;
    if (lane == 0) {
      a[0] = theta; a[1] = m; a[2] = v; a[3] = t; a[4] = b1p; a[5] = b2p;
    }
  }

  // walnuts.hpp:218-235 on the VGPR state
  __device__ __forceinline__ bool within_tolerance(double h, int n, double logp_entry) {
    const double part = leapfrog(h, n);
    double lp, lj;
    energy(part, lp, lj);
    return fabs(lj - logp_entry) <= max_error;
  }

  // walnuts.hpp:254-279.  The accepted end state is parked in three pool buffers
  // while coarser reverse paths are tried from (theta', -rho', grad').
  __device__ __forceinline__ bool reversible(double h, int n, double logp_joint) {
    if (n == 1) return true;
    const int k0 = alloc(), k1 = alloc(), k2 = alloc();  // short-lived: LDS first
    pool_store(k0, th);
    pool_store(k1, rh);
    pool_store(k2, g);
    bool result = true;
    bool first = true;
    while (n >= 2 * min_micro) {
      if (!first) {
        pool_load(k0, th);
        pool_load(k2, g);
      }
      first = false;
      double keep[EPL];
      pool_load(k1, keep);
#pragma unroll
      for (int j = 0; j < EPL; ++j) rh[j] = -keep[j];
      n /= 2;
      h *= 2;
      if (within_tolerance(h, n, logp_joint)) {
        result = false;
        break;
      }
    }
    pool_load(k0, th);
    pool_load(k1, rh);
    pool_load(k2, g);
    release(k0);
    release(k1);
    release(k2);
    return result;
  }

  // walnuts.hpp:307-345.  In: VGPR state = span end, logp_start = its joint log
  // density.  Out (on success): VGPR state = new leaf.
  __device__ __forceinline__ bool macro_step(bool fwd, double logp_start, double& logp_pos, double& logp_joint) {
    WN_PHASE(kPhRestart);
    if (START_REGS) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        th0[j] = th[j];
        rh0[j] = rh[j];
        g0[j] = g[j];
      }
    } else {
      pool_store(start_buf[0], th);
      pool_store(start_buf[1], rh);
      pool_store(start_buf[2], g);
    }
    double h = fwd ? step : -step;
    int n = min_micro;
    for (int halvings = 0; halvings < P.max_halvings; ++halvings, n *= 2, h *= 0.5) {
      if (halvings > 0) {
        if (START_REGS) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            th[j] = th0[j];
            rh[j] = rh0[j];
            g[j] = g0[j];
          }
        } else {
          pool_load(start_buf[0], th);
          pool_load(start_buf[1], rh);
          pool_load(start_buf[2], g);
        }
      }
      WN_PHASE(kPhLeapfrog);
      const double part = leapfrog(h, n);
      WN_PHASE(kPhEnergy);
      energy(part, logp_pos, logp_joint);
      if (halvings == 0) {  // num_steps == min_micro_steps, walnuts.hpp:335-338
        if (P.warmup) adam_observe(wnd::dexp(-fabs(logp_start - logp_joint)));
      }
      if (fabs(logp_start - logp_joint) <= max_error) {
        WN_PHASE(kPhReversible);
        return reversible(h, n, logp_joint);
      }
      WN_PHASE(kPhRestart);
    }
    return false;
  }

  // walnuts.hpp:192-201: the VGPR state is the outer end of the newer span; (a, b) = (theta, rho) of
  // the far end it is tested against.
  __device__ __forceinline__ bool uturn_vectors(const double (&a)[EPL], const double (&b)[EPL], bool fwd) {
    double p_hot = 0.0, p_far = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const double diff = fwd ? (th[j] - a[j]) : (a[j] - th[j]);
      const double sd = im[j] * diff;
      p_hot += rh[j] * sd;
      p_far += b[j] * sd;
    }
    sum2(p_hot, p_far);
    return p_hot < 0 || p_far < 0;
  }
  __device__ __forceinline__ bool uturn_against(int bth, int brh, bool fwd) {
    if (START_REGS && bth == kStart) return uturn_vectors(th0, rh0, fwd);  // the previous leaf, still in VGPRs
    double a[EPL], b[EPL];
    pool_load(bth, a);
    pool_load(brh, b);
    return uturn_vectors(a, b, fwd);
  }

  __device__ __forceinline__ int materialize_theta() {
    const int b = alloc();
    pool_store(b, th);
    return b;
  }
  // give a symbolic vector (kHot = moving end, kStart = restart registers) a pool buffer
  __device__ __forceinline__ int materialize(int ref, bool rho) {
    if (ref >= 0) return ref;
    const int b = alloc();
    if (ref == kHot) {
      pool_store(b, rho ? rh : th);
    } else {
      pool_store(b, rho ? rh0 : th0);
    }
    return b;
  }

  // ------------------------------------------------------------------------------------
  // one MCMC transition (walnuts.hpp:520-563 wrapped as adaptive_walnuts.hpp:234-251 or
  // walnuts.hpp:682-692)
  // ------------------------------------------------------------------------------------
  __device__ void run(int chain_id) {
    WN_PHASE(kPhPrologue);
    chain = chain_id;
    err = 0;
    n_grad = 0;
    n_draw = 0;
    draw_base = -1;
    max_error = P.max_error;
    free_mask = (P.pool_total >= 64) ? ~0ull : ((1ull << P.pool_total) - 1ull);
    const long long row = static_cast<long long>(chain) * Dp;
    const bool warm = P.warmup != 0;

    vload(P.theta + row, th);
    if (Model::kUsesParams) vload(P.model_params, mp);

    // tuning parameters of this transition
    double chol[EPL];
    if (warm) {
      // adaptive_walnuts.hpp:235-236 with MassEstimator::inv_mass_estimate :89-94
      w_draw0 = uni(P.est_weight[2 * chain]);
      w_score0 = uni(P.est_weight[2 * chain + 1]);
      const double wd = w_draw0, ws = w_score0;
      double ds[EPL], ss[EPL];
      vload(P.est_draw_ssd + row, ds);
      vload(P.est_score_ssd + row, ss);
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        im[j] = __builtin_sqrt((ds[j] / wd) / (ss[j] / ws));
        chol[j] = __builtin_sqrt(1.0 / im[j]);
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) meta->adam[i] = P.adam[6 * chain + i];
      }
      step = uni(wnd::dexp(P.adam[6 * chain]));  // adam.hpp:93
      // adaptive_walnuts.hpp:152-157
      const double mean_micro = P.mm_state[2 * chain] / P.mm_state[2 * chain + 1];
      const long long est = static_cast<long long>(__builtin_round(mean_micro / P.macro_target));
      min_micro = uni(static_cast<int>(est > P.cfg_min_micro ? est : P.cfg_min_micro));
    } else {
      vload(P.inv_mass + row, im);
      vload(P.chol_mass + row, chol);  // 1/sqrt(inv_mass), walnuts.hpp:647, computed once at freeze
      step = uni(P.step_size[chain]);
      min_micro = uni(P.min_micro[chain]);
    }

    // momentum refresh rho = chol .* z (walnuts.hpp:528-529)
    if (P.rng_mode == kRngBuffer) {
      double z[EPL];
      vload(P.z_buf + row, z);
#pragma unroll
      for (int j = 0; j < EPL; ++j) rh[j] = chol[j] * z[j];
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        double z0, z1;
        const uint32_t pair = static_cast<uint32_t>(k * L + tid);
        wnd::stream_normal_pair(P.seed, P.chain_offset + chain, P.transition, wnd::kStreamMomentum, pair, z0, z1);
        rh[2 * k] = valid(2 * k) ? chol[2 * k] * z0 : 0.0;
        rh[2 * k + 1] = valid(2 * k + 1) ? chol[2 * k + 1] * z1 : 0.0;
      }
    }

    if (!START_REGS) {
      start_buf[0] = alloc_cold();
      start_buf[1] = alloc_cold();
      start_buf[2] = alloc_cold();
    }

    // initial point (walnuts.hpp:532-535)
    double lp_pos, lj;
    {
      const double part = model_eval();
      energy(part, lp_pos, lj);
    }
    int a_bk[3], a_fw[3];
    a_bk[0] = a_fw[0] = alloc_cold();
    a_bk[1] = a_fw[1] = alloc_cold();
    a_bk[2] = a_fw[2] = alloc_cold();
    pool_store(a_bk[0], th);
    pool_store(a_bk[1], rh);
    pool_store(a_bk[2], g);
    int a_sel = a_bk[0];
    double a_lj_bk = lj, a_lj_fw = lj, a_logsum = lj, a_lpsel = lp_pos;
    // The VGPR state equals one (initially both) of the accumulated span's ends.  An extended end is
    // written back to its pool buffers only when the walk turns around (`dirty`), not after every doubling.
    bool hot_is_fw = true, hot_is_bk = true, dirty = false;
    auto flush_hot_end = [&]() {
      int* endp = hot_is_fw ? a_fw : a_bk;
      const int* other = hot_is_fw ? a_bk : a_fw;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (endp[r] == other[r] || endp[r] == a_sel) endp[r] = alloc_cold();
      }
      pool_store(endp[0], th);
      pool_store(endp[1], rh);
      pool_store(endp[2], g);
      dirty = false;
    };

    int depth = 1;
    for (; depth <= P.max_depth; ++depth) {
      WN_PHASE(kPhDoublingStart);
      const bool fwd = uniform01() < 0.5;  // bernoulli(0.5), walnuts.hpp:552
      if (fwd ? !hot_is_fw : !hot_is_bk) {
        if (dirty) flush_hot_end();
        const int* e = fwd ? a_fw : a_bk;
        pool_load(e[0], th);
        pool_load(e[1], rh);
        pool_load(e[2], g);
      }
      double h_cur = fwd ? a_lj_fw : a_lj_bk;

      // ---- build_span(depth-1) as a post-order walk over 2^(depth-1) leaves ----
      const int nleaf = 1 << (depth - 1);
      int sp = 0;
      bool ok = true;
      int c_in_th = kHot, c_in_rh = kHot, c_sel = kHot;
      double c_logsum = 0.0, c_lpsel = 0.0;
      for (int i = 0; i < nleaf; ++i) {
        double leaf_lp, leaf_lj;
        if (!macro_step(fwd, h_cur, leaf_lp, leaf_lj)) {  // build_leaf, walnuts.hpp:420-442
          ok = false;
          break;
        }
        h_cur = leaf_lj;
        c_in_th = kHot;
        c_in_rh = kHot;
        c_sel = kHot;
        c_logsum = leaf_lj;
        c_lpsel = leaf_lp;
        for (int l = 0; (i >> l) & 1; ++l) {
          --sp;
          const int s_in_th = uni(meta->in_th[sp]), s_in_rh = uni(meta->in_rh[sp]), s_sel = uni(meta->sel[sp]);
          const double s_logsum = uni(meta->logsum[sp]), s_lpsel = uni(meta->lpsel[sp]);
          WN_PHASE(kPhUturn);
          if (uturn_against(s_in_th, s_in_rh, fwd)) {  // walnuts.hpp:490-492
            ok = false;
            break;
          }
          WN_PHASE(kPhCombine);
          // combine<Barker> (walnuts.hpp:370-386): old = s, new = c
          const double total = uni(log_sum_exp(s_logsum, c_logsum));
          const bool update = log_uniform01() < c_logsum - total;
          const int n_sel = update ? c_sel : s_sel;
          const double n_lpsel = update ? c_lpsel : s_lpsel;
          release_unless(s_sel, s_in_th, s_in_rh, n_sel);
          release_unless(c_in_th, s_in_th, s_in_rh, n_sel);
          release_unless(c_in_rh, s_in_th, s_in_rh, n_sel);
          release_unless(c_sel, s_in_th, s_in_rh, n_sel);
          c_in_th = s_in_th;
          c_in_rh = s_in_rh;
          c_sel = n_sel;
          c_lpsel = n_lpsel;
          c_logsum = total;
        }
        if (!ok) break;
        WN_PHASE(kPhPush);
        if (i + 1 < nleaf) {
          // the VGPR state is about to move on.  A lone leaf (even i) becomes the next macro step's
          // restart state, which is exactly where the next leaf's level-0 merge looks for it: nothing to
          // store.  Anything else gets pool buffers for its symbolic parts.
          if (START_REGS && c_in_th == kHot) {
            c_in_th = kStart;
            c_in_rh = kStart;
            c_sel = kStart;
          } else {
            const bool sel_is_inner = (c_sel == c_in_th);
            c_in_th = materialize(c_in_th, false);
            c_in_rh = materialize(c_in_rh, true);
            c_sel = sel_is_inner ? c_in_th : materialize(c_sel, false);
          }
          if (lane == 0) {
            meta->in_th[sp] = c_in_th;
            meta->in_rh[sp] = c_in_rh;
            meta->sel[sp] = c_sel;
            meta->logsum[sp] = c_logsum;
            meta->lpsel[sp] = c_lpsel;
          }
          ++sp;
        }
      }
      if (!ok) break;  // walnuts.hpp:543-545

      WN_PHASE(kPhTopMerge);
      // ---- merge into the accumulated span (walnuts.hpp:546-548) ----
      const bool turned = fwd ? uturn_against(a_bk[0], a_bk[1], true) : uturn_against(a_fw[0], a_fw[1], false);
      const double total = uni(log_sum_exp(a_logsum, c_logsum));
      const bool update = log_uniform01() < c_logsum - a_logsum;  // Metropolis
      // the new span's inner end is never read again
      release_unless(c_in_th, c_sel, -3, -3);
      release_unless(c_in_rh, -3, -3, -3);
      if (update) {
        c_sel = materialize(c_sel, false);
        release_unless(a_sel, a_bk[0], a_fw[0], c_sel);
        a_sel = c_sel;
        a_lpsel = c_lpsel;
      } else {
        release(c_sel);
      }
      // the extended end is now the VGPR state; it reaches the pool only if the walk turns around
      if (fwd) {
        a_lj_fw = h_cur;
        hot_is_fw = true;
        hot_is_bk = false;
      } else {
        a_lj_bk = h_cur;
        hot_is_bk = true;
        hot_is_fw = false;
      }
      dirty = true;
      a_logsum = total;
      if (turned) break;  // walnuts.hpp:549,556-558
    }

    WN_PHASE(kPhEpilogue);
    // ---- selected state out (walnuts.hpp:560-562) ----
    pool_load(a_sel, th);
    vstore(P.theta + row, th);
    if (P.draws_out != nullptr) {
      double* out = P.draws_out + static_cast<long long>(chain) * P.draws_stride;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        if (valid(j)) out[index(j)] = th[j];
      }
    }
    if (warm) {
      // adaptive_walnuts.hpp:247-248: observe (theta_sel, grad_sel).  grad_sel is a pure
      // function of theta_sel, so it is re-evaluated instead of being carried through the tree.
      const long long keep_grad = n_grad;
      (void)model_eval();
      n_grad = keep_grad;
      const double discount = 1.0 - 1.0 / (P.mass_init_count + static_cast<double>(P.warmup_iter));
      const double wd = discount * w_draw0 + 1;
      const double ws = discount * w_score0 + 1;
      double mean[EPL], ssd[EPL];
      vload(P.est_draw_mean + row, mean);
      vload(P.est_draw_ssd + row, ssd);
#pragma unroll
      for (int j = 0; j < EPL; ++j) {  // online_moments.hpp:184-191 (lazy delta => (y - mean_new)^2)
        mean[j] += (th[j] - mean[j]) / wd;
        ssd[j] = discount * ssd[j] + (th[j] - mean[j]) * (th[j] - mean[j]);
      }
      vstore(P.est_draw_mean + row, mean);
      vstore(P.est_draw_ssd + row, ssd);
      vload(P.est_score_mean + row, mean);
      vload(P.est_score_ssd + row, ssd);
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        mean[j] += (g[j] - mean[j]) / ws;
        ssd[j] = discount * ssd[j] + (g[j] - mean[j]) * (g[j] - mean[j]);
      }
      vstore(P.est_score_mean + row, mean);
      vstore(P.est_score_ssd + row, ssd);
      if (tid == 0) {
        P.est_weight[2 * chain] = wd;
        P.est_weight[2 * chain + 1] = ws;
        P.mm_state[2 * chain] += static_cast<double>(1ll << depth);  // observe(1 << depth)
        P.mm_state[2 * chain + 1] += 1.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) P.adam[6 * chain + i] = meta->adam[i];
      }
    }
    if (tid == 0) {
      P.logp_out[chain] = a_lpsel;
      P.depth_out[chain] = err ? -1 : depth;
      P.grad_evals[chain] += n_grad;
      P.rng_draws[chain] = n_draw;
    }
  }
};

// ---------------------------------------------------------------------------------------
// persistent kernel: workgroups pull chains from a shared counter (work per transition
// varies 5..200+ gradient evaluations)
// ---------------------------------------------------------------------------------------
template <class Model, int NW, int EPL, bool START_REGS>
__global__ __launch_bounds__(64 * NW) void transition_kernel(const Params P) {
  WN_DYN_SMEM(smem);
  using T = Traj<Model, NW, EPL, START_REGS>;
  // layout: [pool_lds * Dp] vectors | per-wave Meta | reduction scratch | broadcast word
  WN_LDS double* pool = (WN_LDS double*)smem;
  WN_LDS double* tail = pool + P.pool_lds * P.dim_padded;
  WN_LDS typename T::Meta* meta = (WN_LDS typename T::Meta*)(tail + (threadIdx.x >> 6) * kMetaDoubles);
  WN_LDS double* red = tail + NW * kMetaDoubles;
  WN_LDS double* bcast = red + 4 * NW;
  WN_LDS int* next_chain = (WN_LDS int*)(bcast + 1);
  double* arena = P.arena + static_cast<long long>(blockIdx.x) * P.arena_stride;

  T t(P, pool, meta, red, bcast, arena);
#if defined(WN_PHASE_PROFILE) && !defined(WN_CPU_SIM)
  t.phase_begin();
#endif
  for (;;) {
    WN_PHASE_OUTER(kPhIdle);
    int c;
    if (NW == 1) {
      int mine = 0;
      if (threadIdx.x == 0) mine = static_cast<int>(atomicAdd(P.work_counter, 1u));
      c = uni(mine);
    } else {
      if (threadIdx.x == 0) *next_chain = static_cast<int>(atomicAdd(P.work_counter, 1u));
      __syncthreads();
      c = uni(*next_chain);
      __syncthreads();
    }
    if (c >= P.num_chains) break;
    t.run(c);
  }
#if defined(WN_PHASE_PROFILE) && !defined(WN_CPU_SIM)
  t.phase_end();
#endif
}

inline size_t transition_smem_bytes(int nw, int pool_lds, int dim_padded) {
  return (static_cast<size_t>(pool_lds) * dim_padded + static_cast<size_t>(nw) * kMetaDoubles + 4 * nw + 2) *
         sizeof(double);
}

}  // namespace wn
