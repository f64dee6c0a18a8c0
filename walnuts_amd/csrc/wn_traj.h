// wn_traj.h -- the GPU-resident Walnuts transition for one chain per workgroup (gfx950).
//
// Layout of this file: device models; TrajBase (reductions, span-pool bookkeeping, random-number order, adaptation
// scalars, and the tree loop of the streaming backend); TrajMem (vectors streamed from HBM, any D); the persistent
// chain loop.  The register backend (vectors in VGPRs, D <= 8192) is TrajChip in wn_chip.h.
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
//   Adam           adam.hpp:70-93           Traj::adam_record / adam_flush (batched per transition)
//   MassEstimator / OnlineMoments / MinMicroStepsAdaptHandler
//                  adaptive_walnuts.hpp:54-94,127-157,234-251; online_moments.hpp:184-191
//
// All element-wise arithmetic keeps the reference's association order and the
// file is compiled with -ffp-contract=off, so element-wise results carry the
// reference's bits.  Sums over D run in a fixed order (per-lane partial in
// index order, xor butterfly with offsets 32,1,2,4,8,16 inside a wavefront,
// wavefronts left to right) that the CPU oracle can replay exactly.
#pragma once

#include "wn_hip.h"

#include <type_traits>

#include "wn_devmath.h"
#include "wn_models.h"
#include "wn_params.h"

namespace wn {

// reduction scratch in LDS: two parity halves of 4 doubles per wavefront
constexpr int kRedStride(int nw) { return 4 * nw; }
constexpr int kRedDoubles(int nw) { return 2 * kRedStride(nw); }
// cx.shift() scratch: the edge lanes of every wavefront publish one value per pair slot (at most 8 pairs per lane)
// (LDS tail of a workgroup: per-wave Meta | reduction scratch | broadcast word | next-chain words | shift scratch |
// streaming kernels: the inverse mass vector, when the engine parks it there)
// (two copies, used alternately: one barrier per exchange -- see TrajChip::shift)
constexpr int kShiftDoubles(int nw) { return 2 * 2 * 8 * nw; }

// The exp / log tables (wn_devmath.h) as one entry per lane of three VGPR pairs.  A wave-uniform argument looks its
// entries up with v_readlane (a few cycles, no memory), per-lane arguments with a lane gather.
struct LaneTables {
  double e2, rc, lc;
  __device__ __forceinline__ void load(int lane) {
    e2 = wnd::as_f64(wn_tab_exp2_bits[lane & 63]);
    rc = wnd::as_f64(wn_tab_rcp_bits[lane < 49 ? lane : 48]);
    lc = wnd::as_f64(wn_tab_logc_bits[lane < 49 ? lane : 48]);
  }
};
struct UniformTab {  // the argument is the same in every lane
  const LaneTables& t;
  __device__ __forceinline__ double exp2(int j) const { return lane_value(t.e2, j); }
  __device__ __forceinline__ double rcp(int i) const { return lane_value(t.rc, i); }
  __device__ __forceinline__ double logc(int i) const { return lane_value(t.lc, i); }
};
struct GatherTab {  // every lane has its own argument
  const LaneTables& t;
  __device__ __forceinline__ double exp2(int j) const { return __shfl(t.e2, j, 64); }
  __device__ __forceinline__ double rcp(int i) const { return __shfl(t.rc, i, 64); }
  __device__ __forceinline__ double logc(int i) const { return __shfl(t.lc, i, 64); }
};

// The same tables as 3 x 64 doubles in LDS (the streaming kernels with a held moving end, TrajMem HOLD): those kernels
// fill their registers in the leaf passes, values that live for the whole kernel are spilled there, and a reload from
// scratch waits -- vmcnt counts in order -- behind every store the pass has just issued.  An LDS read does not.
constexpr int kLdsTableDoubles = 3 * 64;
struct LdsTables {
  const WN_LDS double* t;  // [0, 64) exp2, [64, 128) rcp, [128, 192) logc (entries past 48 repeat entry 48)
  __device__ __forceinline__ double exp2(int j) const { return t[j]; }
  __device__ __forceinline__ double rcp(int i) const { return t[64 + i]; }
  __device__ __forceinline__ double logc(int i) const { return t[128 + i]; }
};
template <class S, class = void>
struct tables_in_lds : std::false_type {};
template <class S>
struct tables_in_lds<S, std::enable_if_t<S::kTablesInLds>> : std::true_type {};

// ---- optional timeline probe (tests/gpu_probes/timeline.py only; compiled out of the product build) -------------
#if defined(WN_TIMELINE)
// tests/gpu_probes only: (shader clock, mark id) pairs of workgroup 0's transitions, kept in LDS and copied out when
// the workgroup retires.  One s_memtime and one LDS store per mark.
enum { kPhIdle = 0, kPhPrologue, kPhLeapfrog, kPhEnergy, kPhRestart, kPhReversible, kPhUturn, kPhCombine, kPhPush,
       kPhTopMerge, kPhDoublingStart, kPhEpilogue, kPhLoadsIssued, kPhMomentum, kPhTuned, kPhEvaluated, kPhSelLoaded,
       kPhStored, kPhScalars, kPhCount };
__device__ unsigned long long wn_timeline[kTimelineMarks];
#define WN_PHASE(k) this->timeline_mark(k)
#define WN_PHASE_OUTER(k) t.timeline_mark(k)
#define WN_MARK(k) this->timeline_mark(k)
#else
#define WN_PHASE(k) ((void)0)
#define WN_PHASE_OUTER(k) ((void)0)
#endif
#if !defined(WN_MARK)
#define WN_MARK(k) ((void)0)  // marks only the timeline probe records
#endif

// Branch hints for the rare paths that are inlined into the leaf loops (refills, overflow into the HBM arena, step
// halvings, reversibility re-integrations): the register allocator weighs a block by its expected frequency.
#if defined(WN_NO_BRANCH_HINTS)
#define WN_LIKELY(x) (x)
#define WN_UNLIKELY(x) (x)
#else
#define WN_LIKELY(x) __builtin_expect(!!(x), 1)
#define WN_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif

constexpr int kHot = -1;    // "this vector is the moving trajectory end"
constexpr int kStart = -2;  // "this vector is the macro step's restart state (= the previous leaf)"

// components of the moving end / restart state that can be copied to and from pool buffers
enum Comp : int { kTh = 0, kRh = 1, kG = 2, kTh0 = 3, kRh0 = 4 };

// ---------------------------------------------------------------------------------------------------
// TrajBase: everything about a transition that does not touch vector elements -- span bookkeeping, the
// random-number order, step halving control, Adam, reductions.  `Self` supplies the vector operations:
//   begin_transition(row, warm) -> first log-density partial     finish_transition(a_sel, row, warm, depth)
//   leapfrog(h, n) -> partial      energy(partial, lp, lj)        reversible(h, n, lj)
//   macro_begin() / macro_retry() / macro_commit()
//   put(b, Comp) / get(b, Comp)    uturn_pool(bth, brh, fwd)      uturn_start(fwd)
// A backend with kZeroCopy keeps its vectors in pool buffers already: instead of put(b, c) into a buffer the
// base allocated it offers put_new(c) -> the buffer that now holds the vector (handed over, no copy).
// ---------------------------------------------------------------------------------------------------
template <class Self, class Model, int NW>
struct TrajBase {
  static constexpr int L = 64 * NW;

  // per-wave scalar scratch in LDS
  struct Meta {
    double adam[6];
#if defined(WN_TIMELINE)
    unsigned long long tl[kTimelineMarks];
#endif
  };
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
  unsigned long long free_mask;
  unsigned long long onchip_mask;  // pool buffers that never leave the chip (all of them for the legacy backends)
  int red_parity;
  long long n_grad;
  int n_draw;
  int draw_base;  // first tree-draw index held in draw_u (-1: none)
  double draw_u;
  // The span stack of the post-order walk (one entry per level: three buffer indices and two scalars, SpanW
  // walnuts.hpp:34-131 reduced to what a merge reads) lives in the LANES of three registers -- entry l's weight and
  // selected log density in lanes 2l, 2l+1 of a register pair, its buffer indices packed into lane l of a third --
  // written with v_writelane and read with v_readlane at a wave-uniform index: no LDS round trip and no exec masking
  // at a push, a pop hands the values to scalar registers directly.  (Every wavefront of a chain keeps its own copy.)
  double stk_d;
  int stk_i;
  static_assert(kMaxLevels <= 32, "two doubles per level in the 64 lanes of a register pair");
  __device__ __forceinline__ void stack_push(int sp, int in_th, int in_rh, int sel, double weight, double lpsel) {
    // (indices are -2 .. kMaxPool-1: seven bits each, offset by two; one lane id for the three selects)
    const int me = opaque_lane_id();
    const int packed = (in_th + 2) | ((in_rh + 2) << 7) | ((sel + 2) << 14);
    stk_i = me == sp ? packed : stk_i;
    stk_d = me == 2 * sp ? weight : (me == 2 * sp + 1 ? lpsel : stk_d);
  }
  __device__ __forceinline__ void stack_read(int sp, int& in_th, int& in_rh, int& sel, double& weight, double& lpsel) const {
    const int packed = lane_value(stk_i, sp);
    in_th = (packed & 127) - 2;
    in_rh = ((packed >> 7) & 127) - 2;
    sel = (packed >> 14) - 2;
    weight = lane_value(stk_d, 2 * sp);
    lpsel = lane_value(stk_d, 2 * sp + 1);
  }
  __device__ __forceinline__ void stack_buffers(int sp, int& in_th, int& in_rh) const {
    const int packed = lane_value(stk_i, sp);
    in_th = (packed & 127) - 2;
    in_rh = ((packed >> 7) & 127) - 2;
  }
  int err;
  // (`err` also carries kNoteExtensionFailed: the failure channel of device models, wn_params.h -- set where an
  // extension fails, in blocks that are cold already.  Anything finer -- a count of non-finite attempts kept in a
  // register, in LDS or in memory, a test inside or after the halving loop -- cost the headline kernel 2-8 %,
  // profiles/r04/headline_attempts.md.)
  double step, max_error;
  double w_draw0, w_score0;  // estimator weights at entry (read once: another wave's lane 0 rewrites them at exit)
  int min_micro;
  typename Model::Aux aux;
  LaneTables tabs;
  // (a backend may keep the tables elsewhere: make_uniform_tab / make_gather_tab of the derived class)
  __device__ __forceinline__ auto uniform_tab() const { return static_cast<const Self*>(this)->make_uniform_tab(); }
  __device__ __forceinline__ auto gather_tab() const { return static_cast<const Self*>(this)->make_gather_tab(); }
  __device__ __forceinline__ UniformTab make_uniform_tab() const { return UniformTab{tabs}; }
  __device__ __forceinline__ GatherTab make_gather_tab() const { return GatherTab{tabs}; }

  __device__ __forceinline__ Self& self() { return *static_cast<Self*>(this); }

  // Launch parameters read once or twice per transition (plane pointers, stream keys, adaptation constants) are
  // fetched from the kernel-argument segment where they are used instead of being held in scalar registers for
  // the whole kernel: the compiler otherwise hoists all ~100 dwords of Params to the kernel entry and spills
  // them to VGPR lanes (185-236 spilled SGPRs, 13 % of the VALU stream, before this).  The empty asm hides the
  // pointer's origin at every use so that the loads cannot be hoisted or merged across calls.  Only the
  // transition kernels, whose single kernel argument IS the Params struct, may call this.
  __device__ __forceinline__ auto& cold() const { return kernel_argument(P); }

  __device__ __forceinline__ TrajBase(const Params& p, WN_LDS double* pool, WN_LDS Meta* m, WN_LDS double* r,
                                      WN_LDS double* bc, double* ar)
      : P(p), lds_pool(pool), meta(m), red(r), bcast(bc), arena(ar) {
    wave = NW == 1 ? 0 : wave_in_workgroup();
    lane = opaque_lane_id();
    tid = (wave << 6) | lane;
    Dp = p.dim_padded;
    red_parity = 0;
    stk_d = 0.0;
    stk_i = 0;
    onchip_mask = ~0ull;
    if constexpr (!tables_in_lds<Self>::value) tabs.load(lane);
    adam_err = 0.0;
    adam_n = 0;
    fuse_t = 0;
    fetched = 0;
  }

#if defined(WN_TIMELINE)
  int tl_n = 0, tl_skip = 0;
  __device__ __forceinline__ void timeline_mark(int k) {
    if (blockIdx.x != 0 || k == kPhRestart || k == kPhReversible) return;
    if (k == kPhIdle) ++tl_skip;
    if (tl_skip <= 3) return;  // the workgroup's first chains run while the caches are cold
    const unsigned long long t = shader_clock();
    if (lane == 0 && wave == 0 && tl_n < kTimelineMarks) meta->tl[tl_n] = (t << 6) | static_cast<unsigned>(k);
    ++tl_n;
  }
  __device__ __forceinline__ void timeline_end() {
    if (blockIdx.x != 0 || wave != 0) return;
    for (int i = lane; i < kTimelineMarks; i += 64) wn_timeline[i] = i < tl_n ? meta->tl[i] : 0ull;
  }
#endif

  __device__ __forceinline__ int dim() const { return P.dim; }
  __device__ __forceinline__ double element0(double mine) {
    // element 0 is slot 0 of thread 0
    if (NW == 1) return lane_value(mine, 0);  // (v_readlane: a scalar, no LDS crossbar round trip)
    if (tid == 0) bcast[0] = mine;
    __syncthreads();
    const double v = bcast[0];
    __syncthreads();
    return v;
  }

  // The fetch of the workgroup's NEXT chain from the shared counter is issued here, once the current chain's own
  // loads have been consumed (memory operations retire in order: issued any earlier, the atomic's 1-2 us round trip
  // would stand in front of them), and is collected by persistent_loop after the transition.
  int fetched;
  // Which of the launch's back-to-back transitions of this chain is running (Params::fused): it offsets the stream
  // keys, the warmup iteration number and the draw row.
  int fuse_t;
  __device__ __forceinline__ uint32_t transition_now() const { return cold().transition + static_cast<uint32_t>(fuse_t); }
  __device__ __forceinline__ long long warmup_iter_now() const { return cold().warmup_iter + fuse_t; }
  __device__ __forceinline__ double* draw_row() const {
    const auto& Q = cold();
    return Q.draws_out + static_cast<long long>(chain) * Q.draws_stride + fuse_t * Q.draws_tstride;
  }
  __device__ __forceinline__ void prefetch_next_chain() {
    if (fuse_t + 1 < cold().fused) return;  // the chain stays for another transition
    fetched = 0;
    if (tid == 0) fetched = static_cast<int>((atomicAdd(P.work_counter, 1u) - P.work_base) + gridDim.x);
  }

  // Re-derive the lane identity behind an optimisation barrier.  Everything computed from it (addresses, padding
  // masks, the counter words of the random stream) is then rebuilt where it is used instead of being hoisted out
  // of the persistent chain loop to the kernel entry and held -- or spilled -- for the whole kernel.
  __device__ __forceinline__ void refresh_ids() {
    lane = opaque_lane_id();
    tid = (wave << 6) | lane;
  }

  // ---- reductions -----------------------------------------------------------------
  __device__ __forceinline__ void sum2(double& a, double& b) {
    const double packed = wave_sum_packed(a, b);  // a's sum in lanes 0-31, b's in lanes 32-63
    if (NW == 1) {
      a = uni(packed);
      b = lane_value(packed, 32);
      return;
    }
    if (NW > 1) {
      WN_LDS double* r = red + red_parity * kRedStride(NW);
      if (lane == 0) r[wave * 2] = packed;
      if (lane == 32) r[wave * 2 + 1] = packed;
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
  // four sums behind one exchange (a leaf's two energies and the two level-0 U-turn products): two packed
  // butterflies interleave, one barrier
  __device__ __forceinline__ void sum4(double& a, double& b, double& c, double& d) {
    const double p1 = wave_sum_packed(a, b), p2 = wave_sum_packed(c, d);
    if (NW == 1) {
      a = uni(p1);
      b = lane_value(p1, 32);
      c = uni(p2);
      d = lane_value(p2, 32);
      return;
    }
    if (NW > 1) {
      WN_LDS double* r = red + red_parity * kRedStride(NW);
      if (lane == 0) {
        r[wave * 4] = p1;
        r[wave * 4 + 2] = p2;
      }
      if (lane == 32) {
        r[wave * 4 + 1] = p1;
        r[wave * 4 + 3] = p2;
      }
      __syncthreads();
      double ta = r[0], tb = r[1], tc = r[2], td = r[3];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        ta = ta + r[w * 4];
        tb = tb + r[w * 4 + 1];
        tc = tc + r[w * 4 + 2];
        td = td + r[w * 4 + 3];
      }
      a = ta;
      b = tb;
      c = tc;
      d = td;
      red_parity ^= 1;
    }
    a = uni(a);
    b = uni(b);
    c = uni(c);
    d = uni(d);
  }
  __device__ __forceinline__ double sum1(double a) {
    double b = 0.0;
    sum2(a, b);
    return a;
  }

  // ---- span pool: wave-uniform buffer indices over a 64-bit free mask ---------------------
  // (no branch: an exhausted pool hands out buffer 0 and raises the error bit)
  __device__ __forceinline__ int alloc() {
    const unsigned long long m = free_mask;
    err |= static_cast<int>(m == 0ull);
    const int b = uni(m != 0ull ? __builtin_ctzll(m) : 0);
    free_mask = m & (m - 1ull);
    return b;
  }
  // alloc() when `want`, else `otherwise` (no branch either)
  __device__ __forceinline__ int alloc_if(bool want, int otherwise) {
    const unsigned long long m = free_mask;
    err |= static_cast<int>(want && m == 0ull);
    const int b = uni(m != 0ull ? __builtin_ctzll(m) : 0);
    free_mask = want ? (m & (m - 1ull)) : m;
    return want ? b : otherwise;
  }
  // long-lived vectors (accumulated span ends) take the highest free buffer so that the LDS-resident low
  // indices stay available for the short-lived span-stack entries
  __device__ __forceinline__ int alloc_cold() {
    if (WN_UNLIKELY(free_mask == 0ull)) {
      err |= 1;
      return 0;
    }
    // the highest free on-chip buffer (LDS / register pool); with none left, the lowest of the HBM arena
    const unsigned long long on = free_mask & onchip_mask;
    const int b = on != 0ull ? uni(63 - __builtin_clzll(on)) : uni(__builtin_ctzll(free_mask));
    free_mask &= ~(1ull << b);
    return b;
  }
  __device__ __forceinline__ void release(int b) {
    if (b >= 0) free_mask |= (1ull << b);
  }
  __device__ __forceinline__ void release_unless(int b, int k0, int k1, int k2) {
    if (b >= 0 && b != k0 && b != k1 && b != k2) free_mask |= (1ull << b);
  }
  // give a symbolic vector (kHot = moving end, kStart = restart state) a pool buffer
  __device__ __forceinline__ int materialize(int ref, bool rho) {
    if (ref >= 0) return ref;
    if constexpr (Self::kZeroCopy) {
      return self().put_new(ref == kHot ? (rho ? kRh : kTh) : (rho ? kRh0 : kTh0));
    } else {
      const int b = alloc();
      if (ref == kHot) {
        self().put(b, rho ? kRh : kTh);
      } else {
        self().put(b, rho ? kRh0 : kTh0);
      }
      return b;
    }
  }

  // ---- randomness (util.hpp:102,112 order; counter-based stream or host-fed variates) ----
  // The tree consumes wave-uniform scalars one at a time.  They are produced 64 at a time, lane j computing draw number
  // draw_base + j, kept in one VGPR pair and handed out with v_readlane.  (The acceptance tests compare u * weight
  // products -- "span weights" below --, so no logarithm of a draw is ever taken.)
  __device__ __forceinline__ void refill_draws(int base) {
    draw_base = base;
    const int j = base + lane;
    const auto& Q = cold();
    double u;
    if (WN_UNLIKELY(Q.rng_mode == kRngBuffer)) {
      // (no branch on the lane's index: a lane past the supplied variates reads the row's first one and discards it.
      // A per-lane branch in here would make everything that meets at its join -- draw_base with it -- divergent in
      // the compiler's eyes, and every test of draw_base in the tree loop a branch to be structurised)
      const bool have = j < Q.u_stride;
      const double fed_u = Q.u_buf[static_cast<long long>(chain) * Q.u_stride + (have ? j : 0)];
      u = have ? fed_u : 0.5;
    } else {
      u = wnd::stream_uniform(Q.seed, Q.chain_offset + chain, transition_now(), wnd::kStreamTree,
                              static_cast<uint32_t>(j));
    }
    draw_u = u;   // lane j holds draw number draw_base + j: handed out with v_readlane
  }
  __device__ __forceinline__ int next_draw_slot() {
    const int j = uni(n_draw);
    ++n_draw;
    if (WN_UNLIKELY(draw_base < 0 || j - draw_base >= kDrawCache)) refill_draws(j & ~(kDrawCache - 1));
    return j - draw_base;
  }
  // (the slot first, in a statement of its own: next_draw_slot() may refill the registers the read then uses, and the
  // order in which a call's arguments are evaluated is unspecified -- clang and g++ differ)
  __device__ __forceinline__ double uniform01() {
    const int slot = next_draw_slot();
    return lane_value(draw_u, slot);
  }
  // The register backend asks ONCE per doubling / leaf pair for the draws that stretch can consume (ensure_draws) and
  // then takes them without a test each (uniform01_ready): a scalar branch costs a lone wavefront 15-30 cycles
  // (tests/gpu_probes/branch_cost.hip), more than the lane read it guards.  Which index a draw has -- its counter in
  // the stream -- does not depend on where a block of 64 starts.
  __device__ __forceinline__ void ensure_draws(int need) {
    const int j = uni(n_draw);
    if (WN_UNLIKELY(draw_base < 0 || j + need > draw_base + kDrawCache)) refill_draws(j);
  }
  __device__ __forceinline__ double uniform01_ready() {
    const int j = uni(n_draw);
    ++n_draw;
    return lane_value(draw_u, j - draw_base);
  }

  // ---- span weights: combine (walnuts.hpp:368-387) in the linear domain ---------------------------------------------
  // The reference carries LOG weights: a leaf's is its joint log density, a merged span's the log_sum_exp of its halves
  // (util.hpp:174-183), and a merge moves its selection when log u < new - total (Barker, walnuts.hpp:493) or
  // log u < new - old (Metropolis, :547).  The device carries the weights themselves, relative to a per-transition
  // reference energy w_ref:
  //     leaf:  w = exp(logp_joint - w_ref)        merge:  total = w_old + w_new
  //     Barker:  u * total < w_new                Metropolis:  u * w_old < w_new
  // -- the same decisions up to rounding (the oracle audits every decision for near ties), at one exp per LEAF (~20
  // instructions, tables in VGPR lanes) instead of one log_sum_exp per MERGE (~65 around a scalar-memory look-up whose
  // latency stood in the open: 15 merges per headline transition, an eighth of its issue slots) and no logarithm of
  // the draws.
  // Range: w_ref starts at the initial point's energy, whose weight is exactly 1.  An accepted leaf lies within
  // max_error of its predecessor, so with the default limits (0.5, 5 doublings) no energy of a tree is further than 16
  // from w_ref.  For ANY limits: a leaf whose energy runs more than kWeightRebase ahead of w_ref moves the reference
  // there -- every live weight (the accumulated span's, the span stack's, the pair's even leaf) is scaled by
  // exp(old - new), the leaf weighs 1 -- so weights never overflow; a leaf far BELOW the reference bottoms out at
  // e^-700 (wnd::dexp_weight), beside a span of weight >= 1 that it is merged into sooner or later: selected with
  // probability < 2^-53 per draw either way, as in the log domain.
  static constexpr double kWeightRebase = 256.0;
  double w_ref;
  // weight of the leaf just built (energy lj); `sp` stack entries, a_w and pair_w are the weights alive beside it
  __device__ __forceinline__ double leaf_weight(double lj, int sp, double& a_w, double& pair_w) {
    const double x = lj - w_ref;
    if (WN_UNLIKELY(x > kWeightRebase)) {
      const double f = uni(wnd::dexp_weight(-x, uniform_tab()));  // exp(old reference - new reference)
      a_w = uni(a_w * f);
      pair_w = uni(pair_w * f);
      for (int s = 0; s < sp; ++s) set_lane(stk_d, uni(lane_value(stk_d, 2 * s) * f), 2 * s);
      w_ref = lj;
      return 1.0;
    }
    return uni(wnd::dexp_weight(x, uniform_tab()));
  }
  // the weights of a pair of leaves, even leaf first -- leaf_weight() for one after the other, behind ONE test for the
  // usual case that neither moves the reference (two independent chains of scalar maths side by side)
  __device__ __forceinline__ void pair_weights(double lj_even, double lj_odd, int sp, double& a_w, double& w_even,
                                               double& w_odd) {
    const double xe = lj_even - w_ref, xo = lj_odd - w_ref;
    if (WN_UNLIKELY(xe > kWeightRebase || xo > kWeightRebase)) {
      double none = 0.0;
      w_even = leaf_weight(lj_even, sp, a_w, none);
      w_odd = leaf_weight(lj_odd, sp, a_w, w_even);
      return;
    }
    w_even = uni(wnd::dexp_weight(xe, uniform_tab()));
    w_odd = uni(wnd::dexp_weight(xo, uniform_tab()));
  }
  // free-mask bit of a buffer index; a symbolic (negative) index has none
  __device__ __forceinline__ static unsigned long long pool_bit(int b) { return b >= 0 ? (1ull << b) : 0ull; }

  // adam.hpp:70-86, batched.  The reference updates Adam after every macro step (walnuts.hpp:335-338); nothing reads
  // its state before the NEXT transition (the step size is fixed at a transition's start, adaptive_walnuts.hpp:237),
  // and one update is ~300 instructions of wave-uniform scalar maths: exp, pow, four divisions and a square root, a
  // quarter of a warmup leaf's time.  So a macro step only RECORDS its energy error (lane i keeps the i-th), and the
  // flush evaluates everything that is a pure function of one observation in all lanes at once -- exp(-|error|), the
  // bias corrections' quotients, lr / t^decay, the square root -- around two short sequential passes for the
  // recurrences (m, v, the powers of beta; then theta), which keep the reference's order and therefore its bits.
  double adam_err;  // lane i: |energy error| of the i-th macro step since the last flush
  int adam_n;
  __device__ __forceinline__ void adam_record(double abs_error) {
    adam_err = (lane == adam_n) ? abs_error : adam_err;
    ++adam_n;
  }
  // The register holds 64 observations.  The flush is NOT part of adam_record: inlined into every macro-step
  // instantiation it sat in the middle of the leaf loops (three copies of ~400 instructions, and everything live
  // across them).  The tree loops call this after a leaf or a leaf pair instead -- one site each, off the leaf's path --,
  // which keeps room for the next two records; when the update runs does not change its arithmetic.
  __device__ __forceinline__ void adam_make_room() {
    if (WN_UNLIKELY(adam_n > 62)) adam_flush();
  }
  __device__ __forceinline__ void adam_flush() {
    const int n = adam_n;
    adam_n = 0;
    if (n == 0) return;
    WN_LDS double* a = meta->adam;
    const auto& Q = cold();
    double theta = a[0], m = a[1], v = a[2], t = a[3], b1p = a[4], b2p = a[5];
    const double alpha = wnd::dexp(-adam_err, gather_tab());  // walnuts.hpp:336 (lanes >= n: values nobody reads)
    const double grad = Q.adam_target - alpha;
    double m_i = 0.0, v_i = 0.0, b1p_i = 0.0, b2p_i = 0.0;    // lane i: the state after observation i
    for (int i = 0; i < n; ++i) {
      const double g = lane_value(grad, i);
      b1p *= Q.adam_b1;
      b2p *= Q.adam_b2;
      m = Q.adam_b1 * m + (1 - Q.adam_b1) * g;
      v = Q.adam_b2 * v + (1 - Q.adam_b2) * g * g;
      const bool mine = lane == i;
      m_i = mine ? m : m_i;
      v_i = mine ? v : v_i;
      b1p_i = mine ? b1p : b1p_i;
      b2p_i = mine ? b2p : b2p_i;
    }
    const double t_i = t + static_cast<double>(lane + 1);     // t += 1 per observation: small integers, exact
    const double m_hat = m_i / (1 - b1p_i);
    const double v_hat = v_i / (1 - b2p_i);
    const double lr_t = Q.adam_lr / wnd::dpow_pos(t_i, Q.adam_decay, gather_tab());
    const double denom = __builtin_sqrt(v_hat) + Q.adam_eps;
    const double delta = lr_t * m_hat / denom;
    for (int i = 0; i < n; ++i) theta -= lane_value(delta, i);
    t += static_cast<double>(n);
    if (lane == 0) {
      a[0] = theta; a[1] = m; a[2] = v; a[3] = t; a[4] = b1p; a[5] = b2p;
    }
  }

  // per-chain tuning parameters of this transition (adaptive_walnuts.hpp:235-245 / walnuts.hpp:686-689), in two
  // halves: request_tuning() only ISSUES the loads of the chain's scalars (they sit behind the kernel-argument fetch
  // of their array pointers: two dependent round trips), finish_tuning() turns them into wave-uniform values.  The
  // register backend requests them before the chain's planes and finishes after the momentum has been generated.
  double t_a, t_b, t_c, t_d;  // in flight between the two halves
  int t_i;
  // The running per-chain statistics a transition updates at its end (sampling: the lp_stats row and the
  // gradient-evaluation total; warmup: the min-micro handler's two sums).  A backend with kParkScalars requests them
  // with the tuning scalars and parks them in accumulator registers: read-modify-write at the end of the transition
  // otherwise stands behind an HBM round trip (~1 800 cycles of a ~73 000-cycle transition, tests/gpu_probes/timeline.py).
  long long t_g;
  ParkedDouble k_s0, k_s1, k_s2, k_ge;
  __device__ __forceinline__ void request_tuning(bool warm) {
    const auto& Q = cold();
    if (warm) {
      t_a = Q.est_weight[2 * chain];
      t_b = Q.est_weight[2 * chain + 1];
      t_c = Q.mm_state[2 * chain];
      t_d = Q.mm_state[2 * chain + 1];
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) meta->adam[i] = Q.adam[6 * chain + i];
      }
    } else {
      t_a = Q.step_size[chain];
      t_i = Q.min_micro[chain];
      if constexpr (Self::kParkScalars) {
        const double* w = Q.lp_stats + 3ll * chain;
        t_b = w[0];
        t_c = w[1];
        t_d = w[2];
      }
    }
    if constexpr (Self::kParkScalars) t_g = Q.grad_evals[chain];
  }
  __device__ __forceinline__ void finish_tuning(bool warm) {
    if (warm) {
      const auto& Q = cold();
      w_draw0 = uni(t_a);
      w_score0 = uni(t_b);
      step = uni(wnd::dexp(meta->adam[0], uniform_tab()));  // adam.hpp:93 (written by this wavefront's lane 0 above)
      // adaptive_walnuts.hpp:152-157
      const double mean_micro = t_c / t_d;
      const long long est = static_cast<long long>(__builtin_round(mean_micro / Q.macro_target));
      min_micro = uni(static_cast<int>(est > Q.cfg_min_micro ? est : Q.cfg_min_micro));
      if constexpr (Self::kParkScalars) {
        park(k_s0, t_c);
        park(k_s1, t_d);
      }
    } else {
      step = uni(t_a);
      min_micro = uni(t_i);
      if constexpr (Self::kParkScalars) {
        park(k_s0, t_b);
        park(k_s1, t_c);
        park(k_s2, t_d);
      }
    }
    if constexpr (Self::kParkScalars) park(k_ge, wnd::as_f64(static_cast<uint64_t>(t_g)));
  }
  __device__ __forceinline__ void load_tuning(bool warm) {
    request_tuning(warm);
    finish_tuning(warm);
  }
  // per-chain scalar results of this transition
  __device__ __forceinline__ void store_scalars(bool warm, int depth, double lpsel) {
    if (tid == 0) {
      const auto& Q = cold();
      if (warm) {
        if constexpr (Self::kDeferObservation) {
          // (the prologue applied the pending observation, if there was one: these ARE the weights in force)
          Q.est_weight[2 * chain] = w_draw0;
          Q.est_weight[2 * chain + 1] = w_score0;
        } else {
          const double discount = 1.0 - 1.0 / (Q.mass_init_count + static_cast<double>(warmup_iter_now()));
          Q.est_weight[2 * chain] = discount * w_draw0 + 1;
          Q.est_weight[2 * chain + 1] = discount * w_score0 + 1;
        }
        if constexpr (Self::kParkScalars) {
          Q.mm_state[2 * chain] = fetch(k_s0) + static_cast<double>(1ll << depth);  // observe(1 << depth)
          Q.mm_state[2 * chain + 1] = fetch(k_s1) + 1.0;
        } else {
          Q.mm_state[2 * chain] += static_cast<double>(1ll << depth);
          Q.mm_state[2 * chain + 1] += 1.0;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) Q.adam[6 * chain + i] = meta->adam[i];
      }
      if (!warm) {  // ChainWorker: logp_stats_.observe(lp), sampler.hpp:87-88 / online_moments.hpp:34-40
        double* w = Q.lp_stats + 3ll * chain;
        double w0, w1, w2;
        if constexpr (Self::kParkScalars) {
          w0 = fetch(k_s0);
          w1 = fetch(k_s1);
          w2 = fetch(k_s2);
        } else {
          w0 = w[0];
          w1 = w[1];
          w2 = w[2];
        }
        const double n = w0 + 1;
        const double delta = lpsel - w1;
        const double mean = w1 + delta / n;
        w[0] = n;
        w[1] = mean;
        w[2] = w2 + delta * (lpsel - mean);
      }
      // host-fed uniforms: a transition that consumed more than were supplied used the filler value
      if (WN_UNLIKELY(Q.rng_mode == kRngBuffer && n_draw > Q.u_stride)) err |= static_cast<int>(kErrVariatesExhausted);
      Q.failed_ext[chain] = (err >> 8) & 1;  // (kNoteExtensionFailed)
      err &= 0xff;
      // the per-transition report (depth -1) is overwritten by the next transition; the engine-wide word is not
      if (WN_UNLIKELY(err != 0)) atomicOr(Q.error_flags, static_cast<uint32_t>(err));
      Q.logp_out[chain] = lpsel;
      Q.depth_out[chain] = err ? -1 : depth;
      if constexpr (Self::kParkScalars) {
        Q.grad_evals[chain] = static_cast<long long>(wnd::as_u64(fetch(k_ge))) + n_grad;
      } else {
        Q.grad_evals[chain] += n_grad;
      }
      Q.rng_draws[chain] = n_draw;
    }
  }

  // walnuts.hpp:307-345.  In: moving end = span end, logp_start = its joint log density.
  // Out (on success): moving end = new leaf, restart state = previous leaf.
  __device__ __forceinline__ bool macro_step(bool fwd, double logp_start, double& logp_pos, double& logp_joint) {
    WN_PHASE(kPhRestart);
    self().macro_begin();
    double h = fwd ? step : -step;
    int n = min_micro;
    for (int halvings = 0; halvings < P.max_halvings; ++halvings, n *= 2, h *= 0.5) {
      if (halvings > 0) self().macro_retry();
      WN_PHASE(kPhLeapfrog);
      const double part = self().leapfrog(h, n);
      WN_PHASE(kPhEnergy);
      self().energy(part, logp_pos, logp_joint);
      if (halvings == 0) {  // num_steps == min_micro_steps, walnuts.hpp:335-338
        // Adam's state lives in wavefront 0's scratch (store_scalars reads it there): the others skip the update
        if (P.warmup && wave == 0) adam_record(fabs(logp_start - logp_joint));
      }
      if (fabs(logp_start - logp_joint) <= max_error) {
        WN_PHASE(kPhReversible);
        const bool rev = self().reversible(h, n, logp_joint);
        if (rev) self().macro_commit();
        return rev;
      }
      WN_PHASE(kPhRestart);
    }
    return false;
  }

  __device__ __forceinline__ bool uturn_against(int bth, int brh, bool fwd) {
    if (Self::kHasStartState && bth == kStart) return self().uturn_start(fwd);  // the previous leaf
    return self().uturn_pool(bth, brh, fwd);
  }

  // ------------------------------------------------------------------------------------
  // one MCMC transition (walnuts.hpp:520-563 wrapped as adaptive_walnuts.hpp:234-251 or
  // walnuts.hpp:682-692)
  // ------------------------------------------------------------------------------------
  __device__ __forceinline__ void run(int chain_id) {
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
    load_tuning(warm);
    WN_MARK(kPhTuned);

    // momentum refresh + initial point (walnuts.hpp:528-535)
    double lp_pos, lj;
    {
      const double part = self().begin_transition(row, warm);
      self().energy(part, lp_pos, lj);
    }
    prefetch_next_chain();
    int a_bk[3], a_fw[3];
    if constexpr (Self::kZeroCopy) {
      a_bk[0] = a_fw[0] = self().put_new(kTh);
      a_bk[1] = a_fw[1] = self().put_new(kRh);
      a_bk[2] = a_fw[2] = self().put_new(kG);
    } else {
      a_bk[0] = a_fw[0] = alloc_cold();
      a_bk[1] = a_fw[1] = alloc_cold();
      a_bk[2] = a_fw[2] = alloc_cold();
      self().put(a_bk[0], kTh);
      self().put(a_bk[1], kRh);
      self().put(a_bk[2], kG);
    }
    int a_sel = a_bk[0];
    double a_lj_bk = lj, a_lj_fw = lj, a_w = 1.0, a_lpsel = lp_pos;  // (a_w: the accumulated span's weight)
    w_ref = lj;
    // The moving end equals one (initially both) of the accumulated span's ends.  An extended end is
    // written back to its pool buffers only when the walk turns around (`dirty`), not after every doubling.
    bool hot_is_fw = true, hot_is_bk = true, dirty = false;
    auto flush_hot_end = [&]() {
      int* endp = hot_is_fw ? a_fw : a_bk;
      const int* other = hot_is_fw ? a_bk : a_fw;
      if constexpr (Self::kZeroCopy) {
        // the moving end's buffers become the span end; the old end's buffers go back unless something else
        // still names them
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (!(endp[r] == other[r] || endp[r] == a_sel)) release(endp[r]);
          endp[r] = self().put_new(r == 0 ? kTh : r == 1 ? kRh : kG);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (endp[r] == other[r] || endp[r] == a_sel) endp[r] = alloc_cold();
        }
        self().put(endp[0], kTh);
        self().put(endp[1], kRh);
        self().put(endp[2], kG);
      }
      dirty = false;
    };

    int depth = 1;
    for (; depth <= P.max_depth; ++depth) {
      WN_PHASE(kPhDoublingStart);
      const bool fwd = uniform01() < 0.5;  // bernoulli(0.5), walnuts.hpp:552
      if (fwd ? !hot_is_fw : !hot_is_bk) {
        if (dirty) flush_hot_end();
        const int* e = fwd ? a_fw : a_bk;
        self().get(e[0], kTh);
        self().get(e[1], kRh);
        self().get(e[2], kG);
        self().moving_end_replaced();
      }
      double h_cur = fwd ? a_lj_fw : a_lj_bk;

      // ---- build_span(depth-1) as a post-order walk over 2^(depth-1) leaves ----
      const int nleaf = 1 << (depth - 1);
      int sp = 0;
      bool ok = true;
      int c_in_th = kHot, c_in_rh = kHot, c_sel = kHot;
      double c_w = 0.0, c_lpsel = 0.0;
      for (int i = 0; i < nleaf; ++i) {
        double leaf_lp, leaf_lj;
        {
          // The U-turn tests this leaf's end state will face once it is built -- the merge cascade's levels above the
          // leaf pair (every set low bit of i beyond bit 0 pops one stack entry, see below) and, after the doubling's last
          // leaf, the accumulated span's other end -- are announced to the backend, which may take their sums along in
          // the leaf's own pass (TrajMem::expect_far_ends).
          // (slot 0: the accumulated span's other end after the doubling's last leaf; slots 1, 2: cascade levels 1, 2 --
          // fixed slots, so that the backend's copies stay in registers)
          int far_th[3] = {-1, -1, -1}, far_rh[3] = {-1, -1, -1};
          int nfar = 0;
          if (i + 1 == nleaf) {
            far_th[0] = fwd ? a_bk[0] : a_fw[0];
            far_rh[0] = fwd ? a_bk[1] : a_fw[1];
            nfar = 1;
          }
          if ((i & 3) == 3) {
            stack_buffers(sp - 2, far_th[1], far_rh[1]);
            nfar = 2;
            if ((i & 7) == 7) {
              stack_buffers(sp - 3, far_th[2], far_rh[2]);
              nfar = 3;
            }
          }
          self().expect_far_ends(nfar, far_th, far_rh);
        }
        if (!macro_step(fwd, h_cur, leaf_lp, leaf_lj)) {  // build_leaf, walnuts.hpp:420-442
          ok = false;
          break;
        }
        if (P.warmup && wave == 0) adam_make_room();
        h_cur = leaf_lj;
        c_in_th = kHot;
        c_in_rh = kHot;
        c_sel = kHot;
        {
          double none = 0.0;
          c_w = leaf_weight(leaf_lj, sp, a_w, none);
        }
        c_lpsel = leaf_lp;
        for (int l = 0; (i >> l) & 1; ++l) {
          --sp;
          int s_in_th, s_in_rh, s_sel;
          double s_w, s_lpsel;
          stack_read(sp, s_in_th, s_in_rh, s_sel, s_w, s_lpsel);
          WN_PHASE(kPhUturn);
          if (uturn_against(s_in_th, s_in_rh, fwd)) {  // walnuts.hpp:490-492
            ok = false;
            break;
          }
          WN_PHASE(kPhCombine);
          // combine<Barker> (walnuts.hpp:370-386): old = s, new = c
          const double total = uni(s_w + c_w);
          const bool update = uniform01() * total < c_w;
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
          c_w = total;
        }
        if (!ok) break;
        WN_PHASE(kPhPush);
        if (i + 1 < nleaf) {
          // the moving end is about to move on.  A lone leaf (even i) becomes the next macro step's
          // restart state, which is exactly where the next leaf's level-0 merge looks for it: nothing to
          // store.  Anything else gets pool buffers for its symbolic parts.
          if (Self::kHasStartState && c_in_th == kHot) {
            c_in_th = kStart;
            c_in_rh = kStart;
            c_sel = kStart;
          } else {
            const bool sel_is_inner = (c_sel == c_in_th);
            c_in_th = materialize(c_in_th, false);
            c_in_rh = materialize(c_in_rh, true);
            c_sel = sel_is_inner ? c_in_th : materialize(c_sel, false);
          }
          stack_push(sp, c_in_th, c_in_rh, c_sel, c_w, c_lpsel);
          ++sp;
        }
      }
      if (!ok) {  // walnuts.hpp:543-545
        err |= static_cast<int>(kNoteExtensionFailed);
        break;
      }

      WN_PHASE(kPhTopMerge);
      // ---- merge into the accumulated span (walnuts.hpp:546-548) ----
      const bool turned = fwd ? uturn_against(a_bk[0], a_bk[1], true) : uturn_against(a_fw[0], a_fw[1], false);
      const double total = uni(a_w + c_w);
      const bool update = uniform01() * a_w < c_w;  // Metropolis
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
      // the extended end is now the moving end; it reaches the pool only if the walk turns around
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
      a_w = total;
      if (turned) break;  // walnuts.hpp:549,556-558
    }

    WN_PHASE(kPhEpilogue);
    // ---- selected state out (walnuts.hpp:560-562), estimator update (adaptive_walnuts.hpp:247-248) ----
    self().finish_transition(a_sel, row, warm);
    WN_MARK(kPhSelLoaded);
    if (warm && wave == 0) adam_flush();
    store_scalars(warm, depth, a_lpsel);
    WN_MARK(kPhScalars);
  }
};

// ---------------------------------------------------------------------------------------------------
// TrajMem: the large-D backend.  Vectors do not fit on chip, so the moving end lives in a per-workgroup
// HBM scratch and every operation is a coalesced streaming pass over 16-byte pairs.  Two state sets
// (cur, alt) ping-pong: the first micro step of a macro step reads `cur` and writes `alt`, so the restart
// state (walnuts.hpp:324-326) is never copied and a halving retry costs nothing extra; a micro step is
// one fused pass reading theta, rho, grad, inv_mass and writing theta, rho, grad -- exactly the
// algorithmic 56*D bytes.  Only models whose gradient is element-wise are supported here.
// ---------------------------------------------------------------------------------------------------
// What the streaming backend needs to know about a model whose gradient is NOT element-wise (wn_model_api.h,
// "Streaming a model whose gradient is not element-wise"): how many sums over the coordinates its gradient depends on
// and whether a coordinate's gradient reads its neighbours.
template <class M, class = void>
struct is_streamable : std::false_type {};
template <class M>
struct is_streamable<M, std::enable_if_t<M::kStreamable>> : std::true_type {};
template <class M, bool Elementwise = M::kElementwise>
struct StreamTraits {
  static constexpr bool kTwoPass = false, kHasSums = false, kHalo = false;
  static constexpr int kSums = 1;
};
template <class M>
struct StreamTraits<M, false> {
  static_assert(is_streamable<M>::value, "this model has no streaming form (kStreamable)");
  static_assert(M::kStreamSums <= 2, "at most two sums over the coordinates");
  static constexpr bool kTwoPass = true, kHasSums = M::kStreamSums > 0, kHalo = M::kStreamHalo;
  static constexpr int kSums = M::kStreamSums > 0 ? M::kStreamSums : 1;
};

// The momentum refresh's standard normals (walnuts.hpp:528-531 through util.hpp:102: element pair k * L + tid of the
// kStreamMomentum stream) for `tiles` tiles of one lane, written to the lane's LDS slots.  NOT inlined: inside the
// transition kernel the generator's polynomial coefficients are hoisted to the kernel's entry, kept in registers for
// the whole kernel and -- where the leaf passes fill the register file -- spilled; every reload then waits, in order,
// behind the loads or stores in flight (measured: 9 000 cycles per tile for ~130 instructions).  A function of its own
// has its own registers.
template <int L>
__device__ __attribute__((noinline)) void momentum_normals_to_lds(WN_LDS double* slots, const WN_LDS double* tables,
                                                                  int tiles, int tid, uint64_t seed, uint32_t chain,
                                                                  uint32_t transition) {
  const LdsTables tab{tables};
#pragma unroll 2
  for (int k = 0; k < tiles; ++k) {
    double z0, z1;
    wnd::stream_normal_pair(seed, chain, transition, wnd::kStreamMomentum, static_cast<uint32_t>(k * L + tid), z0, z1, tab);
    v2f64 z;
    z[0] = z0;
    z[1] = z1;
    *reinterpret_cast<WN_LDS v2f64*>(slots + (k * L + tid) * 2) = z;
  }
}

template <class Model, int NW, bool FMA = false, int HOLD = 0>
struct TrajMem : TrajBase<TrajMem<Model, NW, FMA, HOLD>, Model, NW> {
  using Base = TrajBase<TrajMem<Model, NW, FMA, HOLD>, Model, NW>;
  using typename Base::Meta;
  using Base::P; using Base::arena; using Base::tid; using Base::chain; using Base::Dp; using Base::aux;
  using Base::n_grad; using Base::max_error; using Base::min_micro; using Base::w_draw0; using Base::w_score0;
  static constexpr int L = Base::L;
  static constexpr bool kHasStartState = true;
  static constexpr bool kZeroCopy = true;
  static constexpr bool kDeferObservation = false;  // (the estimator observes in this transition's own epilogue)
  static constexpr bool kParkScalars = false;
  // An element-wise gradient is recomputed from theta inside the one pass of a micro step.  Any other streamable model
  // (kTwoPass) takes two passes per micro step: its gradient at the new position needs sums over ALL of the new
  // position (funnel) and / or the neighbours' new values (rw1), which exist only once the first pass has finished.
  using ST = StreamTraits<Model>;
  static constexpr bool kTwoPass = ST::kTwoPass;
  typename Model::Aux auxs[4];  // kTwoPass: the model's by-products (sums) for the states in cur, alt, work, tmp
  // Round 5: HOLD > 0 -- the moving end's (theta, rho) also live in registers (2 * HOLD elements per lane and vector:
  // 128 registers at HOLD = 16, half of what a wavefront of an 8-wavefront workgroup may use; 8 wavefronts x 64 lanes x
  // 32 elements = 16 384 dimensions).  A micro step then reads NOTHING of its input from the memory system -- only the
  // new state goes out, because later U-turn tests, the selection and a turn-around name it as a pool buffer --, and
  // the steps of a multi-step leaf before the last one touch no memory at all.  `held`: the registers equal `cur`.
  static constexpr bool kHold = HOLD > 0;
  static constexpr int kHeld = kHold ? 2 * HOLD : 1;
  double hth[kHeld], hrh[kHeld];
  bool held;
  static constexpr bool kTablesInLds = kHold;  // (LdsTables above: behind the inverse mass)
  const WN_LDS double* tab_lds;
  // Halo models with a held moving end: a coordinate's neighbours are the lane's own other element, the adjacent
  // lane's (a lane shuffle) or -- for the first and last lane of a wavefront -- the adjacent wavefront's / tile's edge
  // element, which every wavefront publishes here before a pass reads positions: [tile][wavefront][lane 0's first |
  // lane 63's second element] (TrajChip::shift's scheme, tile by tile because a pass updates the position in place).
  // (Two copies, written alternately: a wavefront that publishes again has passed the barrier of the publication in
  // between, which every wavefront reaches only after its reads of the copy about to be overwritten.)
  WN_LDS double* edge_lds;
  WN_LDS double* edge_base;
  int edge_parity;
  double edge_vec;  // (lane k: tile k's left neighbour; lane HOLD + k: its right neighbour -- of the last publication)
  static constexpr int kEdgeDoubles = (HOLD > 0 ? HOLD : 1) * NW * 2;
  __device__ __forceinline__ void publish_edges() {
    if constexpr (kHold && ST::kHalo) {
      edge_parity ^= 1;
      edge_lds = edge_base + edge_parity * kEdgeDoubles;
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {
        if (k < tiles) {
          if (this->lane == 0) edge_lds[(k * NW + this->wave) * 2] = hth[2 * k];
          if (this->lane == 63) edge_lds[(k * NW + this->wave) * 2 + 1] = hth[2 * k + 1];
        }
      }
      __syncthreads();  // (also with one wavefront per chain: lane 0 reads what lane 63 wrote)
      // every edge value this wavefront's passes will want, fetched ONCE: lane k < HOLD takes tile k's left neighbour
      // (the element before the wavefront's first), lane HOLD + k its right neighbour; halo_held hands them out with
      // v_readlane -- a tile at a time they were two LDS round trips per tile and pass
      static_assert(2 * (HOLD > 0 ? HOLD : 1) <= 64, "one lane per tile and side");
      const int l = this->lane, w = this->wave;
      const int k = l < HOLD ? l : l - HOLD;
      const bool left = l < HOLD, first = w == 0, last = w == NW - 1;
      int idx;
      bool have;
      if (left) {
        have = !first || k > 0;
        idx = !first ? (k * NW + w - 1) * 2 + 1 : ((k > 0 ? k - 1 : 0) * NW + NW - 1) * 2 + 1;
      } else {
        have = !last || k + 1 < tiles;
        idx = !last ? (k * NW + w + 1) * 2 : ((k + 1 < HOLD ? k + 1 : k) * NW) * 2;
      }
      have = have && l < 2 * HOLD && k < tiles;
      edge_vec = have ? edge_lds[have ? idx : 0] : 0.0;
    }
  }
  // values at the coordinates before / after tile k's two, from the registers (halo()'s values: 0.0 beyond either end
  // of the padded vector); the tile's own elements must still be the ones that were published
  __device__ __forceinline__ void halo_held(int k, double (&prev)[2], double (&next)[2]) const {
    prev[0] = prev[1] = next[0] = next[1] = 0.0;
    if constexpr (kHold && ST::kHalo) {
      prev[1] = hth[2 * k];
      next[0] = hth[2 * k + 1];
      const double up = lane_below(hth[2 * k + 1]);
      const double dn = lane_above(hth[2 * k]);
      const double left_edge = lane_value(edge_vec, k), right_edge = lane_value(edge_vec, HOLD + k);
      prev[0] = this->lane == 0 ? left_edge : up;
      next[1] = this->lane == 63 ? right_edge : dn;
    }
  }
  __device__ __forceinline__ auto make_uniform_tab() const {
    if constexpr (kTablesInLds) {
      return LdsTables{tab_lds};
    } else {
      return UniformTab{this->tabs};
    }
  }
  __device__ __forceinline__ auto make_gather_tab() const {
    if constexpr (kTablesInLds) {
      return LdsTables{tab_lds};
    } else {
      return GatherTab{this->tabs};
    }
  }

  // The vector sets are pool buffers themselves (role slot r: 0-2 cur, 3-5 alt, 6-8 work, 9-11 tmp).  Handing a
  // vector to the span pool (put_new) or taking one from it (get) moves a buffer index, not 8*Dp bytes; a slot
  // whose buffer the pool also names is read-only (`own` bit clear) and gets a fresh buffer before it is written.
  double* cur[3];   // theta, rho, grad of the moving end
  double* alt[3];   // the other set: output of the running macro step / the previous leaf after commit
  double* work[3];  // reversibility re-integration (the reference's scratch vectors, walnuts.hpp:264-266)
  double* tmp[3];   // kTwoPass: ping-pong partner of the set a multi-step leaf is written to (never updated in place:
                    //   a neighbour's old value may still be wanted)
  int slot_buf[12];
  unsigned own;
  double* im_buf;   // warmup: this transition's inverse mass
  const double* im; // inverse mass row in force
  // Round 4: with room in LDS (one chain per CU at 16 wavefronts: 128 KB at 16 384 dimensions) the inverse mass is
  // parked there for the whole transition -- every micro step and every U-turn test then reads 8 bytes per element less
  // from the memory system, which is what bounds these kernels.  Each lane reads back exactly the elements it wrote.
  WN_LDS double* im_lds;
  // The far ends the merge cascade after the running leaf will test the leaf's end state against (walnuts.hpp:192-201
  // at every level of build_span, :490-492, and at the top, :546-549), announced by the tree loop BEFORE the leaf: a
  // single-step leaf accumulates their partial sums in its own pass, where the new (theta, rho) and the inverse mass
  // are in registers -- a later test then costs the far end's two vectors instead of five.
  static constexpr int kMaxPending = 3;
  const double* pend_th[kMaxPending];
  const double* pend_rh[kMaxPending];
  int pend_bth[kMaxPending], pend_brh[kMaxPending];
  double pend_hot[kMaxPending], pend_far[kMaxPending];
  int n_pend;
  int pend_mask;  // bit q: slot q is in use
  bool pend_valid;
  double ke_part;   // kinetic partial of the state produced by the last pass
  double ut_hot, ut_far;  // per-lane partials of the level-0 U-turn sums of the last forward pass
  bool ut_valid;          // ... valid: that pass was the whole macro step (one micro step)
  int tiles;        // pairs per lane

  __device__ __forceinline__ TrajMem(const Params& p, WN_LDS double* pool, WN_LDS Meta* m, WN_LDS double* r,
                                     WN_LDS double* bc, double* ar)
      : Base(p, pool, m, r, bc, ar) {
    // the inverse-mass scratch follows the pool buffers in this workgroup's arena slice
    im_buf = ar + static_cast<long long>(p.pool_total - p.pool_lds) * p.dim_padded;
    im = im_buf;
    own = 0u;
    held = false;
    ut_valid = false;
    ut_hot = ut_far = 0.0;
    im_lds = (p.im_in_lds & 1u) ? bc + 2 + kShiftDoubles(NW) : nullptr;
    tab_lds = nullptr;
    edge_lds = edge_base = nullptr;
    edge_parity = 0;
    edge_vec = 0.0;
    if constexpr (kTablesInLds) {
      // (a kernel with HOLD is launched only with the inverse mass in LDS: wn_kernels.inc, Params::im_in_lds bit 2)
      WN_LDS double* tl = bc + 2 + kShiftDoubles(NW) + p.dim_padded;
      if (this->wave == 0) {
        LaneTables t;
        t.load(this->lane);
        tl[this->lane] = t.e2;
        tl[64 + this->lane] = t.rc;
        tl[128 + this->lane] = t.lc;
      }
      tab_lds = tl;
      edge_base = tl + kLdsTableDoubles;
      edge_lds = edge_base;
      __syncthreads();
    }
    n_pend = 0;
    pend_mask = 0;
    pend_valid = false;
    for (int r = 0; r < 12; ++r) slot_buf[r] = -1;
    for (int i = 0; i < 3; ++i) cur[i] = alt[i] = work[i] = tmp[i] = nullptr;
    ke_part = 0.0;
    tiles = p.dim_padded / (2 * L);
  }

  __device__ __forceinline__ static double mad(double a, double b, double c) {  // a * b + c, fused or not
    if constexpr (FMA) return __builtin_fma(a, b, c);
    return a * b + c;
  }
  struct TileCx {  // model context of one 2-element tile
    int base, D;
    __device__ __forceinline__ static double mad(double a, double b, double c) { return TrajMem::mad(a, b, c); }
    __device__ __forceinline__ int index(int j) const { return base + j; }
    __device__ __forceinline__ bool valid(int j) const { return base + j < D; }
    __device__ __forceinline__ int dim() const { return D; }
  };
  __device__ __forceinline__ int pair_offset(int k) const { return (k * L + tid) * 2; }
  __device__ __forceinline__ static v2f64 ld(const double* p) { return *reinterpret_cast<const v2f64*>(p); }
  __device__ __forceinline__ static void st(double* p, double a, double b) {
    v2f64 t;
    t[0] = a;
    t[1] = b;
    *reinterpret_cast<v2f64*>(p) = t;
  }
  __device__ __forceinline__ double* pool_ptr(int b) const { return arena + static_cast<long long>(b) * Dp; }
  __device__ __forceinline__ v2f64 mass_at(int o) const {
    if (im_lds != nullptr) return *reinterpret_cast<const WN_LDS v2f64*>(im_lds + o);
    return ld(im + o);
  }
  // the tree loop's announcement (TrajBase::run): buffers of the far ends the coming tests will name
  __device__ __forceinline__ void expect_far_ends(int n, const int* bth, const int* brh) {
    n_pend = 0;
    pend_mask = 0;
    pend_valid = false;
    if constexpr (!kTwoPass) {
      if (P.im_in_lds & 2u) n = 0;  // (experiment switch WALNUTS_AMD_NO_FAR_END_SUMS: every test reads its five vectors)
      // (every index into the pend_* arrays is a compile-time constant after unrolling: they live in registers, not scratch)
#pragma unroll
      for (int k = 0; k < kMaxPending; ++k) {
        pend_bth[k] = -1;  // (an unused slot)
        if (k < n && bth[k] >= 0) {
          pend_bth[k] = bth[k];
          pend_brh[k] = brh[k];
          pend_th[k] = pool_ptr(bth[k]);
          pend_rh[k] = pool_ptr(brh[k]);
          n_pend = k + 1;
          pend_mask |= 1 << k;
        }
      }
    }
  }
  __device__ __forceinline__ void copy(double* dst, const double* src) const {
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t = ld(src + o);
      st(dst + o, t[0], t[1]);
    }
  }
  __device__ __forceinline__ double*& slot_ptr(int r) {
    return r < 3 ? cur[r] : r < 6 ? alt[r - 3] : r < 9 ? work[r - 6] : tmp[r - 9];
  }
  __device__ __forceinline__ void set_slot(int r, int b, bool owned) {
    slot_buf[r] = b;
    slot_ptr(r) = pool_ptr(b);
    own = owned ? (own | (1u << r)) : (own & ~(1u << r));
  }
  // before a pass writes slots r0..r0+2: buffers the span pool also names are left to it
  __device__ __forceinline__ void ensure_writable(int r0) {
#pragma unroll
    for (int r = r0; r < r0 + 3; ++r) {
      if (!((own >> r) & 1u)) set_slot(r, this->alloc(), true);
    }
  }
  static __device__ __forceinline__ int slot_of(Comp c) {
    return c == kTh ? 0 : c == kRh ? 1 : c == kG ? 2 : c == kTh0 ? 3 : 4;
  }
  // the span pool takes the vector: its buffer changes hands (no copy) unless the pool names that buffer already
  __device__ __forceinline__ int put_new(Comp c) {
    const int r = slot_of(c);
    if ((own >> r) & 1u) {
      own &= ~(1u << r);
      return slot_buf[r];
    }
    const int b = this->alloc();
    if (c != kG) copy(pool_ptr(b), slot_ptr(r));  // gradient buffers are bookkeeping only (never read: see leapfrog_sets)
    return b;
  }
  // the moving end becomes a vector of the span pool: read it in place
  __device__ __forceinline__ void get(int b, Comp c) {
    const int r = slot_of(c);
    if ((own >> r) & 1u) this->release(slot_buf[r]);
    set_slot(r, b, false);
    if (kHold && r < 2) held = false;
  }

  // ---- two-pass models -------------------------------------------------------------------------------------
  // values at the coordinates before / after the tile's two (0.0 beyond either end of the padded vector)
  __device__ __forceinline__ void halo(const double* v, int o, const v2f64& t, double (&prev)[2], double (&next)[2]) const {
    prev[0] = prev[1] = next[0] = next[1] = 0.0;
    if constexpr (ST::kHalo) {
      prev[0] = o > 0 ? v[o - 1] : 0.0;
      prev[1] = t[0];
      next[0] = t[1];
      next[1] = o + 2 < Dp ? v[o + 2] : 0.0;
    }
  }
  __device__ __forceinline__ void load_mp(int o, double (&mp2)[2]) const {
    mp2[0] = mp2[1] = 1.0;
    if (Model::kUsesParams) {
      const v2f64 p0 = ld(P.model_params + o);
      mp2[0] = p0[0];
      mp2[1] = p0[1];
    }
  }
  // lane partials of the model's sums -> the aux of that position (every wavefront ends with the same values)
  __device__ __forceinline__ void finish_sums(double (&sums)[ST::kSums], typename Model::Aux& out) {
    if constexpr (ST::kHasSums) {
      double a = sums[0], b = ST::kSums > 1 ? sums[ST::kSums - 1] : 0.0;
      this->sum2(a, b);
      sums[0] = a;
      if (ST::kSums > 1) sums[ST::kSums - 1] = b;
      Model::stream_aux(sums, P.dim, this->uniform_tab(), out);
    }
  }
  // the aux of a position that was not produced by a leapfrog pass (loaded, or taken over from the span pool)
  __device__ __forceinline__ void aux_of(const double* theta, typename Model::Aux& out) {
    if constexpr (kTwoPass && ST::kHasSums) {
      double sums[ST::kSums];
      for (int i = 0; i < ST::kSums; ++i) sums[i] = 0.0;
      for (int k = 0; k < tiles; ++k) {
        const int o = pair_offset(k);
        const v2f64 t0 = ld(theta + o);
        const double th2[2] = {t0[0], t0[1]};
        double mp2[2];
        load_mp(o, mp2);
        TileCx cx{o, P.dim};
        Model::stream_sums(cx, th2, mp2, sums);
      }
      finish_sums(sums, out);
    }
  }
  // the moving end was replaced by a span end from the pool (TrajBase::run, turn-around)
  __device__ __forceinline__ void moving_end_replaced() {
    if constexpr (kTwoPass) aux_of(cur[0], auxs[0]);
  }
  // (The role sets are named by TEMPLATE arguments from here on: with run-time set indices the slots are reached
  // through selected member addresses, the optimiser cannot split the object into registers any more, and everything
  // in it -- the held moving end included -- lives in scratch.)
  template <int RA, int RB>
  __device__ __forceinline__ void swap_sets() {  // role sets RA, RB (0 cur, 1 alt, 2 work, 3 tmp)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double* t = slot_ptr(3 * RA + i);
      slot_ptr(3 * RA + i) = slot_ptr(3 * RB + i);
      slot_ptr(3 * RB + i) = t;
      const int b = slot_buf[3 * RA + i];
      slot_buf[3 * RA + i] = slot_buf[3 * RB + i];
      slot_buf[3 * RB + i] = b;
    }
    constexpr unsigned ma = 7u << (3 * RA), mb = 7u << (3 * RB);
    const unsigned oa = (own & ma) >> (3 * RA), ob = (own & mb) >> (3 * RB);
    own = (own & ~(ma | mb)) | (ob << (3 * RA)) | (oa << (3 * RB));
    const typename Model::Aux t = auxs[RA];
    auxs[RA] = auxs[RB];
    auxs[RB] = t;
  }
  // One micro step of a two-pass model from role set FROM into role set TO.  Pass A: kick with the gradient at the old
  // position (its aux is known), drift, store, and take the model's sums of the NEW position; pass B: the gradient
  // there, second kick, log-density terms, kinetic energy.
  template <int FROM, int TO>
  __device__ __forceinline__ void two_pass_micro_step(bool neg, double h, double& part, double& ke) {
    const double half = 0.5 * h;
    ensure_writable(3 * TO);
    double* const in0 = slot_ptr(3 * FROM);
    double* const in1 = slot_ptr(3 * FROM + 1);
    double* const out0 = slot_ptr(3 * TO);
    double* const out1 = slot_ptr(3 * TO + 1);
    const typename Model::Aux a_in = auxs[FROM];
    double sums[ST::kSums];
    for (int i = 0; i < ST::kSums; ++i) sums[i] = 0.0;
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t0 = ld(in0 + o), r0 = ld(in1 + o), m0 = mass_at(o);
      double th2[2] = {t0[0], t0[1]}, rh2[2] = {neg ? -r0[0] : r0[0], neg ? -r0[1] : r0[1]};
      double g2[2], mp2[2], prev[2], next[2];
      load_mp(o, mp2);
      halo(in0, o, t0, prev, next);
      TileCx cx{o, P.dim};
      Model::stream_grad(cx, th2, prev, next, mp2, g2, a_in);
#pragma unroll
      for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) th2[j] = mad(h * m0[j], rh2[j], th2[j]);
      if (ST::kHasSums) Model::stream_sums(cx, th2, mp2, sums);
      st(out0 + o, th2[0], th2[1]);
      st(out1 + o, rh2[0], rh2[1]);
    }
    finish_sums(sums, auxs[TO]);
    if (ST::kHalo) __syncthreads();  // the neighbours' new positions are in memory
    const typename Model::Aux a_out = auxs[TO];
    part = 0.0;
    ke = 0.0;
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t1 = ld(out0 + o), r1 = ld(out1 + o), m0 = mass_at(o);
      const double th2[2] = {t1[0], t1[1]};
      double rh2[2] = {r1[0], r1[1]}, g2[2], mp2[2], prev[2], next[2];
      load_mp(o, mp2);
      halo(out0, o, t1, prev, next);
      TileCx cx{o, P.dim};
      Model::stream_grad(cx, th2, prev, next, mp2, g2, a_out);
      Model::stream_logp(cx, th2, prev, next, mp2, a_out, part);
#pragma unroll
      for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) ke = mad(m0[j], rh2[j] * rh2[j], ke);
      st(out1 + o, rh2[0], rh2[1]);
    }
    ++n_grad;
  }
  // n micro steps of a two-pass model from role set RS into role set RD (RD != RS; steps after the first ping-pong
  // between RD and tmp -- never in place: a neighbour's old value may still be wanted --, and the result ends in RD).
  template <int RS, int RD>
  __device__ __forceinline__ double leapfrog_two_pass(bool negate, double h, int n) {
    double part = 0.0, ke = 0.0;
    two_pass_micro_step<RS, RD>(negate, h, part, ke);
    bool in_tmp = false;
    for (int s = 1; s < n; ++s) {
      if (!in_tmp) {
        two_pass_micro_step<RD, 3>(false, h, part, ke);
      } else {
        two_pass_micro_step<3, RD>(false, h, part, ke);
      }
      in_tmp = !in_tmp;
    }
    if (in_tmp) swap_sets<RD, 3>();  // an even number of steps ended in tmp
    aux = auxs[RD];
    ke_part = ke;
    if (!negate) ut_valid = false;
    return part;
  }

  // n micro steps (walnuts.hpp:328-333): the first reads `src` (rho negated for the reversibility check)
  // and writes `dst`, the rest run in place on `dst`.  Returns the log-density partial and leaves the kinetic
  // partial of the final state in ke_part.
  __device__ __forceinline__ double leapfrog_sets(double* const* src, double* const* dst, bool negate, double h,
                                                  int n) {
    const double half = 0.5 * h;
    double part = 0.0, ke = 0.0;
    // A single-step leaf (n == 1, the usual case once the step size has adapted) has both ends of the two-leaf
    // span (previous leaf = input, new leaf = output) in registers: the level-0 U-turn sums (walnuts.hpp:192-201)
    // are accumulated here, in uturn_ptrs' order, and that pass over five vectors is skipped.
    const bool fuse = !negate && n == 1;
    const bool fwd = h > 0;
    double p_hot = 0.0, p_far = 0.0;
    const int np = fuse ? n_pend : 0;
#pragma unroll
    for (int q = 0; q < kMaxPending; ++q) pend_hot[q] = pend_far[q] = 0.0;
    for (int s = 0; s < n; ++s) {
      double* const* in = (s == 0) ? src : dst;
      const bool neg = negate && s == 0;
      part = 0.0;
      ke = 0.0;
      for (int k = 0; k < tiles; ++k) {
        const int o = pair_offset(k);
        const v2f64 t0 = ld(in[0] + o), r0 = ld(in[1] + o), m0 = mass_at(o);
        double th2[2] = {t0[0], t0[1]}, rh2[2] = {neg ? -r0[0] : r0[0], neg ? -r0[1] : r0[1]};
        double g2[2], mp2[2] = {1.0, 1.0};
        if (Model::kUsesParams) {
          const v2f64 p0 = ld(P.model_params + o);
          mp2[0] = p0[0];
          mp2[1] = p0[1];
        }
        TileCx cx{o, P.dim};
        // the gradient is a function of theta alone (element-wise model): recomputing it is cheaper than keeping a
        // third vector in HBM -- 16 bytes per element and pass less than the 56 the definition charges
        Model::grad(cx, th2, g2, mp2, aux);
#pragma unroll
        for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) th2[j] = mad(h * m0[j], rh2[j], th2[j]);
        Model::eval(cx, th2, g2, mp2, aux, part);
#pragma unroll
        for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) ke = mad(m0[j], rh2[j] * rh2[j], ke);
        if (fuse) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const double diff = fwd ? (th2[j] - t0[j]) : (t0[j] - th2[j]);
            const double sd = m0[j] * diff;
            p_hot = mad(rh2[j], sd, p_hot);
            p_far = mad(r0[j], sd, p_far);
          }
          // ... and the same products against the announced far ends (uturn_ptrs' expressions in uturn_ptrs' order)
#pragma unroll
          for (int q = 0; q < kMaxPending; ++q) {
            if (q < np && pend_bth[q] >= 0) {
              const v2f64 av = ld(pend_th[q] + o), bv = ld(pend_rh[q] + o);
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const double diff = fwd ? (th2[j] - av[j]) : (av[j] - th2[j]);
                const double sd = m0[j] * diff;
                pend_hot[q] = mad(rh2[j], sd, pend_hot[q]);
                pend_far[q] = mad(bv[j], sd, pend_far[q]);
              }
            }
          }
        }
        st(dst[0] + o, th2[0], th2[1]);
        st(dst[1] + o, rh2[0], rh2[1]);
      }
      ++n_grad;
    }
    ke_part = ke;
    if (!negate) {
      ut_valid = fuse;
      ut_hot = p_hot;
      ut_far = p_far;
      pend_valid = fuse;
    }
    return part;
  }
  // ---- HOLD: the moving end in registers -------------------------------------------------------------------
  // Addresses in these passes are (wave-uniform base + tile offset) + this lane's 16 bytes.  The tile's base is made
  // opaque in a scalar register pair: otherwise the optimiser re-associates it into one 64-bit vector address per
  // (vector, tile), hoists those out of every loop around the pass and spills them (measured: each access then sits
  // behind a scratch reload and its wait).
  // (formed anew -- two mbcnt instructions the optimiser does not merge -- by every pass that uses it: as ONE value
  // derived from the thread index it lives from the kernel's entry to its end, is spilled where the leaf passes fill
  // the register file, and comes back from scratch -- behind a wait for every access in flight -- before each load)
  __device__ __forceinline__ unsigned lane_bytes() const {
    return static_cast<unsigned>(this->wave * 64 + opaque_lane_id()) * 16u;
  }
  // (the tile's offset joins the LANE's 32-bit offset, not the base: sixteen tile bases per vector, formed ahead of
  // the asm statement that pins them, are sixteen scalar pairs the allocator parks in the lanes of a vector register
  // -- and that register in scratch, behind a wait before every access)
  __device__ __forceinline__ static v2f64 ld_tile(const double* base, int k, unsigned lb) {
    return load_pair_at(opaque_scalar_pointer(base), lb + static_cast<unsigned>(k) * (L * 16u));
  }
  __device__ __forceinline__ static void st_tile(double* base, int k, unsigned lb, double x, double y) {
    v2f64 t;
    t[0] = x;
    t[1] = y;
    store_pair_at(opaque_scalar_pointer(base), lb + static_cast<unsigned>(k) * (L * 16u), t);
  }
  __device__ __forceinline__ v2f64 mass_tile(int k, unsigned lb) const {  // (held kernels run with the mass in LDS)
    return *reinterpret_cast<const WN_LDS v2f64*>(reinterpret_cast<const WN_LDS char*>(im_lds) + k * (L * 16) + lb);
  }
  // the registers hold nothing any more: said in so many words, because the optimiser cannot tell that the next
  // transition's prologue overwrites them -- it would carry 128 live registers through the epilogue and the prologue
  // (where the normal generator and the planes' loads need them) and spill them around both
  __device__ __forceinline__ void drop_held() {
    if constexpr (kHold) {
#pragma unroll
      for (int i = 0; i < kHeld; ++i) hth[i] = hrh[i] = 0.0;
      held = false;
    }
  }
  __device__ __forceinline__ void ensure_held() {
    if constexpr (kHold) {
      if (!held) {
        const unsigned lb = lane_bytes();
#pragma unroll
        for (int k = 0; k < HOLD; ++k) {
          if (k < tiles) {
            const v2f64 t0 = ld_tile(cur[0], k, lb), r0 = ld_tile(cur[1], k, lb);
            hth[2 * k] = t0[0];
            hth[2 * k + 1] = t0[1];
            hrh[2 * k] = r0[0];
            hrh[2 * k + 1] = r0[1];
          }
        }
        held = true;
      }
    }
  }
  // What one tile's arithmetic reads from memory: the inverse mass (LDS), the model's parameters, the announced far
  // ends.  The passes below issue a tile's loads kPrefetch tiles AHEAD of its arithmetic (a ring of load sets, every
  // index a compile-time constant): with two wavefronts per SIMD nothing else hides a load's latency.
  struct TileLoads {
    v2f64 mp, av[kMaxPending], bv[kMaxPending];
  };
  template <int MASK>
  __device__ __forceinline__ void issue_tile(TileLoads& t, int k, unsigned lb) const {
    if (Model::kUsesParams) t.mp = ld_tile(P.model_params, k, lb);
#pragma unroll
    for (int q = 0; q < kMaxPending; ++q) {
      if ((MASK >> q) & 1) {
        t.av[q] = ld_tile(pend_th[q], k, lb);
        t.bv[q] = ld_tile(pend_rh[q], k, lb);
      }
    }
  }
  // Tiles in flight ahead of the arithmetic: what ~64 registers hold as a ring of PD + 1 load sets, 4 registers per
  // 16-byte load (of a wavefront's 256: 128 hold the moving end, ~20 the pass's sums, ~40 the tree's state and a tile's
  // temporaries).  The inverse mass comes from LDS: one tile ahead is enough for it.
  static constexpr int prefetch_tiles(int mask) {
    const int loads = (Model::kUsesParams ? 1 : 0) + 2 * ((mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1));
    if (loads == 0) return 1;
    const int sets = 16 / loads;
    return sets < 2 ? 1 : sets > 5 ? 4 : sets - 1;
  }
  // One micro step on the registers.  MASK: the announced far ends whose sums ride along (FUSE: with the level-0 sums
  // against the step's own input) -- a template parameter, so that a tile is straight-line code; STORE: the new state
  // goes out to `alt`.
  template <int MASK, bool FUSE, bool STORE>
  __device__ __forceinline__ void held_step(double h, double& part, double& ke, double& p_hot, double& p_far) {
    const double half = 0.5 * h;
    const unsigned lb = lane_bytes();
    constexpr int PD = prefetch_tiles(MASK);
    TileLoads ring[PD + 1];
    v2f64 mass[2];
    part = 0.0;
    ke = 0.0;
    mass[0] = mass_tile(0, lb);
#pragma unroll
    for (int k = 0; k < PD; ++k) {
      if (k < tiles) issue_tile<MASK>(ring[k % (PD + 1)], k, lb);
    }
#pragma unroll
    for (int k = 0; k < HOLD; ++k) {
      if (k + PD < HOLD) {
        if (k + PD < tiles) issue_tile<MASK>(ring[(k + PD) % (PD + 1)], k + PD, lb);
      }
      if (k + 1 < HOLD) {
        if (k + 1 < tiles) mass[(k + 1) & 1] = mass_tile(k + 1, lb);
      }
      if (k < tiles) {
        const TileLoads& in = ring[k % (PD + 1)];
        const int o = pair_offset(k);
        const v2f64 m0 = mass[k & 1];
        const double t0[2] = {hth[2 * k], hth[2 * k + 1]}, r0[2] = {hrh[2 * k], hrh[2 * k + 1]};
        double th2[2] = {t0[0], t0[1]}, rh2[2] = {r0[0], r0[1]};
        double g2[2], mp2[2] = {1.0, 1.0};
        if (Model::kUsesParams) {
          mp2[0] = in.mp[0];
          mp2[1] = in.mp[1];
        }
        TileCx cx{o, P.dim};
        Model::grad(cx, th2, g2, mp2, aux);
#pragma unroll
        for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) th2[j] = mad(h * m0[j], rh2[j], th2[j]);
        Model::eval(cx, th2, g2, mp2, aux, part);
#pragma unroll
        for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) ke = mad(m0[j], rh2[j] * rh2[j], ke);
        if (FUSE) {
          // (the forward expressions; a backward step's sums are their negatives, bit for bit: rounding is symmetric)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const double sd = m0[j] * (th2[j] - t0[j]);
            p_hot = mad(rh2[j], sd, p_hot);
            p_far = mad(r0[j], sd, p_far);
          }
#pragma unroll
          for (int q = 0; q < kMaxPending; ++q) {
            if ((MASK >> q) & 1) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const double sd = m0[j] * (th2[j] - in.av[q][j]);
                pend_hot[q] = mad(rh2[j], sd, pend_hot[q]);
                pend_far[q] = mad(in.bv[q][j], sd, pend_far[q]);
              }
            }
          }
        }
        hth[2 * k] = th2[0];
        hth[2 * k + 1] = th2[1];
        hrh[2 * k] = rh2[0];
        hrh[2 * k + 1] = rh2[1];
      }
    }
    // The new state goes out AFTER the last tile's arithmetic, from the registers it stays in: loads and stores share
    // one in-order counter (vmcnt), so a store issued between two tiles' loads would put its acknowledgement -- an HBM
    // write's round trip -- into the wait for the next tile's operands.
    if (STORE) {
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {
        if (k < tiles) {
          st_tile(alt[0], k, lb, hth[2 * k], hth[2 * k + 1]);
          st_tile(alt[1], k, lb, hrh[2 * k], hrh[2 * k + 1]);
        }
      }
    }
    ++n_grad;
  }
  // leapfrog_two_pass(cur -> alt, not negated) on the registers: both passes of every micro step read the state where
  // it is, the model's sums are reduced between them, and the result goes out to `alt` once, after the last step --
  // 16 bytes per element and macro step where the streamed form moves 72 per micro step.
  __device__ __forceinline__ double leapfrog_two_pass_held(double h, int n) {
    const double half = 0.5 * h;
    const unsigned lb = lane_bytes();
    double part = 0.0, ke = 0.0;
    ensure_held();
    held = false;  // (the candidate: `cur` again once macro_commit has swapped the sets)
    typename Model::Aux a_in = auxs[0], a_out = auxs[0];
    for (int s = 0; s < n; ++s) {
      double sums[ST::kSums];
#pragma unroll
      for (int i = 0; i < ST::kSums; ++i) sums[i] = 0.0;
      v2f64 mass[2];
      publish_edges();
      mass[0] = mass_tile(0, lb);
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {  // pass A: kick with the gradient at the old position, drift, the new position's sums
        if (k + 1 < HOLD) {
          if (k + 1 < tiles) mass[(k + 1) & 1] = mass_tile(k + 1, lb);
        }
        if (k < tiles) {
          const v2f64 m0 = mass[k & 1];
          double th2[2] = {hth[2 * k], hth[2 * k + 1]}, rh2[2] = {hrh[2 * k], hrh[2 * k + 1]};
          double g2[2], mp2[2] = {1.0, 1.0}, prev[2], next[2];
          halo_held(k, prev, next);
          if (Model::kUsesParams) {
            const v2f64 p0 = ld_tile(P.model_params, k, lb);
            mp2[0] = p0[0];
            mp2[1] = p0[1];
          }
          TileCx cx{pair_offset(k), P.dim};
          Model::stream_grad(cx, th2, prev, next, mp2, g2, a_in);
#pragma unroll
          for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) th2[j] = mad(h * m0[j], rh2[j], th2[j]);
          if (ST::kHasSums) Model::stream_sums(cx, th2, mp2, sums);
          hth[2 * k] = th2[0];
          hth[2 * k + 1] = th2[1];
          hrh[2 * k] = rh2[0];
          hrh[2 * k + 1] = rh2[1];
        }
      }
      finish_sums(sums, a_out);
      publish_edges();  // (the new position's)
      part = 0.0;
      ke = 0.0;
      mass[0] = mass_tile(0, lb);
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {  // pass B: the gradient there, second kick, log-density terms, kinetic energy
        if (k + 1 < HOLD) {
          if (k + 1 < tiles) mass[(k + 1) & 1] = mass_tile(k + 1, lb);
        }
        if (k < tiles) {
          const v2f64 m0 = mass[k & 1];
          const double th2[2] = {hth[2 * k], hth[2 * k + 1]};
          double rh2[2] = {hrh[2 * k], hrh[2 * k + 1]}, g2[2], mp2[2] = {1.0, 1.0}, prev[2], next[2];
          halo_held(k, prev, next);
          if (Model::kUsesParams) {
            const v2f64 p0 = ld_tile(P.model_params, k, lb);
            mp2[0] = p0[0];
            mp2[1] = p0[1];
          }
          TileCx cx{pair_offset(k), P.dim};
          Model::stream_grad(cx, th2, prev, next, mp2, g2, a_out);
          Model::stream_logp(cx, th2, prev, next, mp2, a_out, part);
#pragma unroll
          for (int j = 0; j < 2; ++j) rh2[j] = mad(half, g2[j], rh2[j]);
#pragma unroll
          for (int j = 0; j < 2; ++j) ke = mad(m0[j], rh2[j] * rh2[j], ke);
          hrh[2 * k] = rh2[0];
          hrh[2 * k + 1] = rh2[1];
        }
      }
      ++n_grad;
      a_in = a_out;
    }
#pragma unroll
    for (int k = 0; k < HOLD; ++k) {
      if (k < tiles) {
        st_tile(alt[0], k, lb, hth[2 * k], hth[2 * k + 1]);
        st_tile(alt[1], k, lb, hrh[2 * k], hrh[2 * k + 1]);
      }
    }
    auxs[1] = a_out;
    aux = a_out;
    ke_part = ke;
    ut_valid = false;
    return part;
  }
  // leapfrog_sets(cur, alt, false, h, n) with the input taken from -- and every intermediate state kept in -- the
  // registers: the same expressions in the same order, element by element
  __device__ __forceinline__ double leapfrog_held(double h, int n) {
    double part = 0.0, ke = 0.0, p_hot = 0.0, p_far = 0.0;
    const bool fuse = n == 1;
    const bool fwd = h > 0;
#pragma unroll
    for (int q = 0; q < kMaxPending; ++q) pend_hot[q] = pend_far[q] = 0.0;
    ensure_held();
    held = false;  // (the registers are about to hold the candidate: `cur` again once macro_commit has swapped the sets)
    if (fuse) {
      switch (pend_mask) {
        case 0: held_step<0, true, true>(h, part, ke, p_hot, p_far); break;
        case 1: held_step<1, true, true>(h, part, ke, p_hot, p_far); break;
        case 2: held_step<2, true, true>(h, part, ke, p_hot, p_far); break;
        case 3: held_step<3, true, true>(h, part, ke, p_hot, p_far); break;
        case 4: held_step<4, true, true>(h, part, ke, p_hot, p_far); break;
        case 5: held_step<5, true, true>(h, part, ke, p_hot, p_far); break;
        case 6: held_step<6, true, true>(h, part, ke, p_hot, p_far); break;
        default: held_step<7, true, true>(h, part, ke, p_hot, p_far); break;
      }
    } else {
      for (int s = 1; s < n; ++s) held_step<0, false, false>(h, part, ke, p_hot, p_far);
      held_step<0, false, true>(h, part, ke, p_hot, p_far);
    }
    ke_part = ke;
    ut_valid = fuse;
    ut_hot = fwd ? p_hot : -p_hot;
    ut_far = fwd ? p_far : -p_far;
#pragma unroll
    for (int q = 0; q < kMaxPending; ++q) {
      pend_hot[q] = fwd ? pend_hot[q] : -pend_hot[q];
      pend_far[q] = fwd ? pend_far[q] : -pend_far[q];
    }
    pend_valid = fuse;
    return part;
  }
  __device__ __forceinline__ double leapfrog(double h, int n) {
    if constexpr (kTwoPass) {
      if constexpr (kHold) {
        ensure_writable(3);
        return leapfrog_two_pass_held(h, n);
      }
      return leapfrog_two_pass<0, 1>(false, h, n);
    } else {
      ensure_writable(3);
      if constexpr (kHold) return leapfrog_held(h, n);
      return leapfrog_sets(cur, alt, false, h, n);
    }
  }
  __device__ __forceinline__ void energy(double lp_partial, double& logp_pos, double& logp_joint) {
    double ke = ke_part;
    this->sum2(lp_partial, ke);
    logp_pos = uni(Model::finish(lp_partial, aux, P.dim));
    logp_joint = uni(logp_pos + (-0.5 * ke));
  }
  __device__ __forceinline__ void macro_begin() {}
  __device__ __forceinline__ void macro_retry() {}
  __device__ __forceinline__ void macro_commit() {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double* t = cur[i];
      cur[i] = alt[i];
      alt[i] = t;
      const int b = slot_buf[i];
      slot_buf[i] = slot_buf[3 + i];
      slot_buf[3 + i] = b;
    }
    own = (own & ~0x3fu) | ((own & 0x7u) << 3) | ((own >> 3) & 0x7u);
    if constexpr (kHold) held = true;  // (leapfrog_held left the accepted candidate in the registers)
    if constexpr (kTwoPass) {
      const typename Model::Aux t = auxs[0];
      auxs[0] = auxs[1];
      auxs[1] = t;
    }
  }
  // walnuts.hpp:254-279: coarser reverse paths from (theta', -rho', grad') = the candidate in `alt`
  __device__ __forceinline__ bool reversible(double h, int n, double logp_joint) {
    if (n == 1) return true;
    while (n >= 2 * min_micro) {
      n /= 2;
      h *= 2;
      double part;
      if constexpr (kTwoPass) {
        part = leapfrog_two_pass<1, 2>(true, h, n);
      } else {
        ensure_writable(6);
        part = leapfrog_sets(alt, work, true, h, n);
      }
      double lp, lj;
      energy(part, lp, lj);
      if (fabs(lj - logp_joint) <= max_error) return false;
    }
    return true;
  }

  // walnuts.hpp:192-201 against the far end (a, b) = (theta, rho)
  __device__ __forceinline__ bool uturn_ptrs(const double* a, const double* b, bool fwd) {
    double p_hot = 0.0, p_far = 0.0;
    if constexpr (kHold) {
      ensure_held();
      const unsigned lb = lane_bytes();
      constexpr int PD = 4;
      v2f64 rm[PD + 1], ra[PD + 1], rb[PD + 1];
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        if (k < tiles) {
          rm[k] = mass_tile(k, lb);
          ra[k] = ld_tile(a, k, lb);
          rb[k] = ld_tile(b, k, lb);
        }
      }
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {
        if (k + PD < HOLD) {
          if (k + PD < tiles) {
            rm[(k + PD) % (PD + 1)] = mass_tile(k + PD, lb);
            ra[(k + PD) % (PD + 1)] = ld_tile(a, k + PD, lb);
            rb[(k + PD) % (PD + 1)] = ld_tile(b, k + PD, lb);
          }
        }
        if (k < tiles) {
          const v2f64 m = rm[k % (PD + 1)], av = ra[k % (PD + 1)], bv = rb[k % (PD + 1)];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const double sd = m[j] * (hth[2 * k + j] - av[j]);
            p_hot = mad(hrh[2 * k + j], sd, p_hot);
            p_far = mad(bv[j], sd, p_far);
          }
        }
      }
      this->sum2(p_hot, p_far);
      return fwd ? (p_hot < 0 || p_far < 0) : (p_hot > 0 || p_far > 0);
    }
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t = ld(cur[0] + o), r = ld(cur[1] + o), m = mass_at(o), av = ld(a + o), bv = ld(b + o);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const double diff = fwd ? (t[j] - av[j]) : (av[j] - t[j]);
        const double sd = m[j] * diff;
        p_hot = mad(r[j], sd, p_hot);
        p_far = mad(bv[j], sd, p_far);
      }
    }
    this->sum2(p_hot, p_far);
    return p_hot < 0 || p_far < 0;
  }
  __device__ __forceinline__ bool uturn_start(bool fwd) {
    if (ut_valid) {  // the accepted macro step was one micro step: its pass already holds the partial sums
      double p_hot = ut_hot, p_far = ut_far;
      this->sum2(p_hot, p_far);
      return p_hot < 0 || p_far < 0;
    }
    return uturn_ptrs(alt[0], alt[1], fwd);
  }
  __device__ __forceinline__ bool uturn_pool(int bth, int brh, bool fwd) {
    if (pend_valid) {  // the accepted single-step leaf's pass took this far end's sums along
      bool have = false;
      double p_hot = 0.0, p_far = 0.0;
#pragma unroll
      for (int q = 0; q < kMaxPending; ++q) {
        if (q < n_pend && pend_bth[q] >= 0 && pend_bth[q] == bth && pend_brh[q] == brh) {
          have = true;
          p_hot = pend_hot[q];
          p_far = pend_far[q];
        }
      }
      if (have) {
        this->sum2(p_hot, p_far);
        return p_hot < 0 || p_far < 0;
      }
    }
    return uturn_ptrs(pool_ptr(bth), pool_ptr(brh), fwd);
  }

  // begin_transition_held's sweep over the tiles, four at a time: the inverse mass (WARM: the two estimator planes it
  // is computed from) and the model's parameters of kAhead chunks are in flight while a chunk is worked on; each tile
  // takes its normals from its LDS slot and leaves the inverse mass there.
  template <bool WARM>
  __device__ __forceinline__ void initial_state_sweep(long long row, unsigned lb, double& part, double& ke,
                                                      double (&sums)[ST::kSums]) {
    constexpr int kChunk = 4, kChunks = (HOLD + kChunk - 1) / kChunk;
    constexpr int kAhead = WARM ? 2 : 4;  // (three planes' loads per tile in warmup, two otherwise)
    const wnd::SharedDivisor wd(w_draw0), ws(w_score0);  // (wn_devmath.h: the same quotients as `/`)
    const double* plane_a = WARM ? P.est_draw_ssd + row : P.inv_mass + row;
    struct Chunk {
      v2f64 pa[kChunk], pb[kChunk], mp[kChunk];
    };
    Chunk ring[kAhead];
    auto issue = [&](Chunk& ch, int c) {
#pragma unroll
      for (int i = 0; i < kChunk; ++i) {
        const int k = c * kChunk + i;
        if (k < HOLD && k < tiles) {
          ch.pa[i] = ld_tile(plane_a, k, lb);
          if (WARM) ch.pb[i] = ld_tile(P.est_score_ssd + row, k, lb);
          if (Model::kUsesParams) ch.mp[i] = ld_tile(P.model_params, k, lb);
        }
      }
    };
#pragma unroll
    for (int c = 0; c < kAhead - 1 && c < kChunks; ++c) issue(ring[c % kAhead], c);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (c + kAhead - 1 < kChunks) issue(ring[(c + kAhead - 1) % kAhead], c + kAhead - 1);
      const Chunk& ch = ring[c % kAhead];
#pragma unroll
      for (int i = 0; i < kChunk; ++i) {
        const int k = c * kChunk + i;
        if (k < HOLD && k < tiles) {
          const int o = pair_offset(k);
          const v2f64 z = mass_tile(k, lb);  // (this tile's normals, parked in its slot)
          v2f64 m0 = ch.pa[i];
          if (WARM) {  // adaptive_walnuts.hpp:235-236, :89-94
#pragma unroll
            for (int j = 0; j < 2; ++j) m0[j] = __builtin_sqrt((ch.pa[i][j] / wd) / (ch.pb[i][j] / ws));
          }
          *reinterpret_cast<WN_LDS v2f64*>(im_lds + o) = m0;
          double th2[2] = {hth[2 * k], hth[2 * k + 1]}, g2[2], mp2[2] = {1.0, 1.0}, ch2[2];
          if (Model::kUsesParams) {
            mp2[0] = ch.mp[i][0];
            mp2[1] = ch.mp[i][1];
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            ch2[j] = WARM ? __builtin_sqrt(1.0 / m0[j])    // adaptive_walnuts.hpp:89-94
                          : 1.0 / __builtin_sqrt(m0[j]);   // walnuts.hpp:647 (= the chol_mass plane)
          }
          double rh2[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) rh2[j] = (o + j < P.dim) ? ch2[j] * z[j] : 0.0;
          TileCx cx{o, P.dim};
          if constexpr (kTwoPass) {
            (void)g2;
            if (ST::kHasSums) Model::stream_sums(cx, th2, mp2, sums);  // (the position's aux first: the log-density terms follow it)
          } else {
            Model::eval(cx, th2, g2, mp2, aux, part);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) ke = mad(m0[j], rh2[j] * rh2[j], ke);
          hrh[2 * k] = rh2[0];
          hrh[2 * k + 1] = rh2[1];
        }
      }
    }
  }
  // begin_transition for a kernel with a held moving end: the same expressions element by element, arranged for two
  // wavefronts per SIMD.  The runtime loop below it takes a tile at a time -- three loads, their round trip, the
  // normal generator, two stores and their acknowledgement (measured at 16 tiles per lane: 210 us per transition, as
  // much as fifteen leapfrog passes).  Here (1) every tile of the position is requested at once, straight into the
  // registers that will hold it; (2) the momentum's normals are drawn while those loads are in flight -- in a loop
  // that is NOT unrolled, with the register file to itself (unrolled beside 128 held registers the generator's
  // constants are spilled, and every reload waits, in order, behind the loads in flight) -- and parked in the LDS
  // slots of the inverse mass, which is not there yet; (3) the inverse mass (warmup: the two estimator planes it is
  // computed from) arrives in chunks of four tiles, several chunks ahead of the arithmetic (initial_state_sweep above),
  // and each tile reads its normals from its slot before the mass takes it.  The Cholesky factor of a sampling transition is computed, 1 / sqrt(inverse mass) as freeze_kernel
  // (wn_elementwise.h) wrote its plane -- the same operations, the same bits -- instead of being loaded.  The initial
  // state goes out to `cur` from the registers after the last tile.
  __device__ __forceinline__ double begin_transition_held(long long row, bool warm) {
    const bool fed = P.rng_mode == kRngBuffer;
    if (fed) {
#pragma nounroll
      for (int k = 0; k < tiles; ++k) {
        const int o = pair_offset(k);
        *reinterpret_cast<WN_LDS v2f64*>(im_lds + o) = ld(P.z_buf + row + o);
      }
    } else {
      momentum_normals_to_lds<L>(im_lds, tab_lds, tiles, tid, P.seed, P.chain_offset + chain, this->transition_now());
    }
    WN_MARK(kPhMomentum);
    const unsigned lb = lane_bytes();
#pragma unroll
    for (int k = 0; k < HOLD; ++k) {
      if (k < tiles) {
        const v2f64 t0 = ld_tile(P.theta + row, k, lb);
        hth[2 * k] = t0[0];
        hth[2 * k + 1] = t0[1];
      }
    }
    double part = 0.0, ke = 0.0;
    double sums[ST::kSums];
#pragma unroll
    for (int i = 0; i < ST::kSums; ++i) sums[i] = 0.0;
    if (warm) {
      initial_state_sweep<true>(row, lb, part, ke, sums);
    } else {
      initial_state_sweep<false>(row, lb, part, ke, sums);
    }
    if constexpr (kTwoPass) {
      finish_sums(sums, auxs[0]);
      aux = auxs[0];
      publish_edges();
#pragma unroll
      for (int k = 0; k < HOLD; ++k) {
        if (k < tiles) {
          const double th2[2] = {hth[2 * k], hth[2 * k + 1]};
          double mp2[2] = {1.0, 1.0}, prev[2], next[2];
          halo_held(k, prev, next);
          if (Model::kUsesParams) {
            const v2f64 p0 = ld_tile(P.model_params, k, lb);
            mp2[0] = p0[0];
            mp2[1] = p0[1];
          }
          TileCx cx{pair_offset(k), P.dim};
          Model::stream_logp(cx, th2, prev, next, mp2, auxs[0], part);
        }
      }
    }
    WN_MARK(kPhEvaluated);
#pragma unroll
    for (int k = 0; k < HOLD; ++k) {
      if (k < tiles) {
        st_tile(cur[0], k, lb, hth[2 * k], hth[2 * k + 1]);
        st_tile(cur[1], k, lb, hrh[2 * k], hrh[2 * k + 1]);
      }
    }
    WN_MARK(kPhStored);
    held = true;
    ++n_grad;
    ke_part = ke;
    return part;
  }

  __device__ __forceinline__ double begin_transition(long long row, bool warm) {
    const wnd::SharedDivisor wd(w_draw0), ws(w_score0);  // (wn_devmath.h: the same quotients as `/`)
    im = warm ? im_buf : P.inv_mass + row;
    n_pend = 0;
    pend_mask = 0;
    pend_valid = false;
    drop_held();
    own = 0u;  // the base has just marked every pool buffer free
    for (int r = 0; r < 12; ++r) slot_buf[r] = -1;
    ensure_writable(0);
    if constexpr (kHold) {
      if (im_lds != nullptr) return begin_transition_held(row, warm);
    }
    double part = 0.0, ke = 0.0;
    double sums[ST::kSums];
    for (int i = 0; i < ST::kSums; ++i) sums[i] = 0.0;
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t0 = ld(P.theta + row + o);
      double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2] = {1.0, 1.0}, m2[2], ch2[2], z2[2];
      if (Model::kUsesParams) {
        const v2f64 p0 = ld(P.model_params + o);
        mp2[0] = p0[0];
        mp2[1] = p0[1];
      }
      if (warm) {  // adaptive_walnuts.hpp:235-236, :89-94
        const v2f64 ds = ld(P.est_draw_ssd + row + o), ss = ld(P.est_score_ssd + row + o);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          m2[j] = __builtin_sqrt((ds[j] / wd) / (ss[j] / ws));
          ch2[j] = __builtin_sqrt(1.0 / m2[j]);
        }
        if (im_lds == nullptr) st(im_buf + o, m2[0], m2[1]);
      } else {
        const v2f64 m0 = ld(P.inv_mass + row + o), c0 = ld(P.chol_mass + row + o);
        m2[0] = m0[0]; m2[1] = m0[1];
        ch2[0] = c0[0]; ch2[1] = c0[1];
      }
      if (im_lds != nullptr) {
        v2f64 mm;
        mm[0] = m2[0];
        mm[1] = m2[1];
        *reinterpret_cast<WN_LDS v2f64*>(im_lds + o) = mm;
      }
      if (P.rng_mode == kRngBuffer) {
        const v2f64 z0 = ld(P.z_buf + row + o);
        z2[0] = z0[0];
        z2[1] = z0[1];
      } else {
        wnd::stream_normal_pair(P.seed, P.chain_offset + chain, this->transition_now(), wnd::kStreamMomentum,
                                static_cast<uint32_t>(k * L + tid), z2[0], z2[1], this->gather_tab());
      }
      double rh2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) rh2[j] = (o + j < P.dim) ? ch2[j] * z2[j] : 0.0;
      TileCx cx{o, P.dim};
      if constexpr (kTwoPass) {
        // the position's sums for its aux; without sums the log-density terms can be taken right here (neighbours
        // from the position plane), with sums they may depend on the aux and get a pass of their own below
        if (ST::kHasSums) {
          Model::stream_sums(cx, th2, mp2, sums);
        } else {
          double prev[2], next[2];
          halo(P.theta + row, o, t0, prev, next);
          Model::stream_logp(cx, th2, prev, next, mp2, auxs[0], part);
        }
        (void)g2;
      } else {
        Model::eval(cx, th2, g2, mp2, aux, part);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) ke = mad(m2[j], rh2[j] * rh2[j], ke);
      st(cur[0] + o, th2[0], th2[1]);
      st(cur[1] + o, rh2[0], rh2[1]);
    }
    ++n_grad;
    ke_part = ke;
    if constexpr (kTwoPass) {
      finish_sums(sums, auxs[0]);
      aux = auxs[0];
      if (ST::kHasSums) {
        for (int k = 0; k < tiles; ++k) {
          const int o = pair_offset(k);
          const v2f64 t0 = ld(P.theta + row + o);
          const double th2[2] = {t0[0], t0[1]};
          double mp2[2], prev[2], next[2];
          load_mp(o, mp2);
          halo(P.theta + row, o, t0, prev, next);
          TileCx cx{o, P.dim};
          Model::stream_logp(cx, th2, prev, next, mp2, auxs[0], part);
        }
      }
    }
    return part;
  }

  __device__ __forceinline__ void finish_transition(int a_sel, long long row, bool warm) {
    drop_held();
    const double* sel = pool_ptr(a_sel);
    typename Model::Aux aux_sel{};
    if constexpr (kTwoPass) {
      if (warm) aux_of(sel, aux_sel);  // the estimator wants the gradient at the selected position
    }
    const double discount = 1.0 - 1.0 / (P.mass_init_count + static_cast<double>(this->warmup_iter_now()));
    const wnd::SharedDivisor wd(discount * w_draw0 + 1), ws(discount * w_score0 + 1);
    double* const out = P.draws_out != nullptr ? this->draw_row() : nullptr;
    // a draw row on a 16-byte boundary takes whole pairs (streamed: written once, read by nobody here); the pair that
    // straddles the end of an odd-length row, or a row at an odd offset, goes element by element
    const bool pair_rows = (reinterpret_cast<unsigned long long>(out) & 15ull) == 0ull;
    if constexpr (kHold) {
      // sampling: every tile of the selected state is in flight before the first store (a tile at a time the copy
      // is sixteen round trips back to back)
      if (!warm) {
        const unsigned lb = lane_bytes();
        v2f64 all[HOLD];  // (the registers that held the moving end: it has just been dropped)
#pragma unroll
        for (int k = 0; k < HOLD; ++k) {
          if (k < tiles) all[k] = ld_tile(sel, k, lb);
        }
#pragma unroll
        for (int k = 0; k < HOLD; ++k) {
          if (k < tiles) {
            const int o = pair_offset(k);
            const v2f64 t0 = all[k];
            st_tile(P.theta + row, k, lb, t0[0], t0[1]);
            if (out != nullptr) {
              if (pair_rows && o + 1 < P.dim) {
                stream_store(t0, reinterpret_cast<v2f64*>(out + o));
              } else {
                if (o < P.dim) stream_store(t0[0], &out[o]);
                if (o + 1 < P.dim) stream_store(t0[1], &out[o + 1]);
              }
            }
          }
        }
        return;
      }
    }
    for (int k = 0; k < tiles; ++k) {
      const int o = pair_offset(k);
      const v2f64 t0 = ld(sel + o);
      st(P.theta + row + o, t0[0], t0[1]);
      if (out != nullptr) {
        if (pair_rows && o + 1 < P.dim) {
          stream_store(t0, reinterpret_cast<v2f64*>(out + o));
        } else {
          if (o < P.dim) stream_store(t0[0], &out[o]);
          if (o + 1 < P.dim) stream_store(t0[1], &out[o + 1]);
        }
      }
      if (warm) {  // adaptive_walnuts.hpp:247-248, online_moments.hpp:184-191
        double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2] = {1.0, 1.0};
        if (Model::kUsesParams) {
          const v2f64 p0 = ld(P.model_params + o);
          mp2[0] = p0[0];
          mp2[1] = p0[1];
        }
        TileCx cx{o, P.dim};
        if constexpr (kTwoPass) {
          double prev[2], next[2];
          halo(sel, o, t0, prev, next);
          Model::stream_grad(cx, th2, prev, next, mp2, g2, aux_sel);
        } else {
          double unused = 0.0;
          Model::eval(cx, th2, g2, mp2, aux, unused);
        }
        v2f64 mean = ld(P.est_draw_mean + row + o), ssd = ld(P.est_draw_ssd + row + o);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          mean[j] += (th2[j] - mean[j]) / wd;
          ssd[j] = discount * ssd[j] + (th2[j] - mean[j]) * (th2[j] - mean[j]);
        }
        st(P.est_draw_mean + row + o, mean[0], mean[1]);
        st(P.est_draw_ssd + row + o, ssd[0], ssd[1]);
        mean = ld(P.est_score_mean + row + o);
        ssd = ld(P.est_score_ssd + row + o);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          mean[j] += (g2[j] - mean[j]) / ws;
          ssd[j] = discount * ssd[j] + (g2[j] - mean[j]) * (g2[j] - mean[j]);
        }
        st(P.est_score_mean + row + o, mean[0], mean[1]);
        st(P.est_score_ssd + row + o, ssd[0], ssd[1]);
      }
    }
  }
};

// TrajMem: cur 3 + alt 3 + work 3 + tmp 3 are pool buffers (the host adds kMemRoleVectors to the pool), + inverse mass 1
constexpr int kMemRoleVectors = 12;
constexpr int kMemScratchVectors = 1;

// ---------------------------------------------------------------------------------------
// persistent kernels: workgroups pull chains from a shared counter (work per transition
// varies 5..200+ gradient evaluations)
// ---------------------------------------------------------------------------------------
template <class T, int NW>
__device__ __forceinline__ void persistent_loop(const Params& P) {
  WN_DYN_SMEM(smem);
  // layout: [lds_state * Dp] state vectors | [pool_lds * Dp] pool vectors | per-wave Meta | reduction scratch |
  // broadcast word
  WN_LDS double* pool = (WN_LDS double*)smem;
  WN_LDS double* tail = pool + P.pool_lds * P.dim_padded;
  WN_LDS typename T::Meta* meta = (WN_LDS typename T::Meta*)(tail + (NW == 1 ? 0 : wave_in_workgroup()) * kMetaDoubles);
  WN_LDS double* red = tail + NW * kMetaDoubles;
  WN_LDS double* bcast = red + kRedDoubles(NW);
  WN_LDS int* next_chain = (WN_LDS int*)(bcast + 1);  // (the shift scratch of TrajChip follows at bcast + 2)
  double* arena = P.arena + static_cast<long long>(blockIdx.x) * P.arena_stride;

  T t(P, pool, meta, red, bcast, arena);
  // The workgroup's first chain is its own index; the following ones come from the shared counter, and the fetch
  // for chain n+1 is issued while chain n is being processed, so that its round trip (a device-scope atomic, 1-2 us
  // under load) overlaps the tree instead of standing between two transitions.
  // (P.chain_begin: the launch covers one chain group, chains [chain_begin, num_chains))
  int c = opaque_scalar_add(static_cast<int>(blockIdx.x), kernel_argument(P).chain_begin);
  int slot = 0;
  // (unsigned: a chain index that went negative -- counter and base out of step -- ends the loop instead of indexing
  // rows in front of the planes)
  while (WN_LIKELY(static_cast<unsigned>(c) < static_cast<unsigned>(P.num_chains))) {
    WN_PHASE_OUTER(kPhIdle);
    // (Params::fused transitions of the chain back to back; the last one issues t.prefetch_next_chain() on the way)
    int k = 0;
    do {
      // what the chain's transition wrote (planes by every lane, scalars by thread 0) is read back by the next one:
      // program order within a wavefront, a workgroup barrier between wavefronts
      if (NW > 1 && k > 0) __syncthreads();
      t.fuse_t = k;
      t.run(c);
    } while (++k < kernel_argument(P).fused);
    if (NW == 1) {
      c = uni(t.fetched);
    } else {
      // two alternating words: a wavefront that races ahead to the next hand-over writes the other one
      if (t.wave == 0 && opaque_lane_id() == 0) next_chain[slot] = t.fetched;
      __syncthreads();
      c = uni(next_chain[slot]);
      slot ^= 1;
    }
  }
#if defined(WN_TIMELINE)
  t.timeline_end();
#endif
}

// Tiles (pairs per lane) of the moving end a streaming kernel keeps in registers (TrajMem, HOLD); 0 = it keeps none.
// Eight wavefronts per chain, one chain per CU: 256 registers each, 128 of them for 16 tiles of (theta, rho) = 16 384
// dimensions.  (Sixteen wavefronts with 8 tiles have 64 registers left for everything else: measured, the operands of
// a tile then arrive one round trip at a time and part of the held state is spilled.)
constexpr int kMemHoldTiles = 16;
template <class Model>
constexpr int mem_hold_tiles(int nw) {
  if constexpr (!(Model::kElementwise || is_streamable<Model>::value)) {
    return 0;  // (no streaming kernels at all)
  } else {
#if defined(WN_SIM_GEOMETRIES)
    return kMemHoldTiles;  // (tests/cpusim: every geometry it builds)
#else
    return nw == 8 ? kMemHoldTiles : 0;
#endif
  }
}

// num_params up to which the register kernels stay the default for a model that has held streaming kernels
// (wn_launch.h, beside kHeldWaves: measured): two passes with sums only (the funnel) keep (16, 8)
template <class Model>
constexpr int mem_register_dim_limit() {
  if constexpr (!(Model::kElementwise || is_streamable<Model>::value)) {
    return 8192;
  } else {
    return (StreamTraits<Model>::kTwoPass && !StreamTraits<Model>::kHalo) ? 8192 : 4096;
  }
}

template <class Model, int NW, bool FMA, int HOLD = 0>
__global__ __launch_bounds__(64 * NW) void transition_kernel_mem(const Params P) {
  persistent_loop<TrajMem<Model, NW, FMA, HOLD>, NW>(P);
}

inline size_t transition_smem_bytes(int nw, int lds_vectors, int dim_padded) {
  return (static_cast<size_t>(lds_vectors) * dim_padded + static_cast<size_t>(nw) * kMetaDoubles + kRedDoubles(nw) + 2 +
          kShiftDoubles(nw)) *
         sizeof(double);
}

}  // namespace wn
