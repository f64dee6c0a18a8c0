// wn_chip.h -- TrajChip: the register backend of the Walnuts transition whose span pool stays on the chip.
//
// Same tree, same random-number order and the same arithmetic as wn_traj.h's TrajBase::run (walnuts.hpp:520-563
// wrapped as adaptive_walnuts.hpp:234-251 / walnuts.hpp:682-692), laid out for what the MI355X measurements of
// round 1 said bounds the headline workload (65 536 chains x 1 024 dims): 10 of the 13 GB the kernel moved per
// launch were span-pool vectors going to the per-workgroup HBM arena and register spills, not the 2.7 GB the
// chains' state needs.
//
//  * Ping-pong trajectory end.  The moving end is TWO register sets.  A macro step (walnuts.hpp:307-345) reads
//    set A and writes set 1-A, so the restart state the reference copies (walnuts.hpp:324-326) is simply the set
//    that was not written: a halving retry costs nothing and no leaf copies its predecessor.  Leaves are built in
//    pairs (even leaf: set 0 -> 1, odd leaf: set 1 -> 0), which makes the set index a compile-time constant
//    everywhere; the one single-leaf extension of a transition (the first doubling) copies once.
//  * Level-0 merges never touch memory: both leaves of the pair are in registers when the odd leaf is done, and the
//    two U-turn products (walnuts.hpp:192-201) ride in the SAME reduction as the leaf's two energies (sum4).
//  * Two-tier span pool behind the wave-uniform buffer indices: LDS vectors, then -- only when a deep tree needs
//    more than the chip holds -- the HBM arena.  With 4 chains per CU at D = 1024 that is 4 vectors in LDS (+ the
//    span's other end in registers) against a worst case of 12 live pool vectors at the default five doublings.
//    (A third tier, pool vectors in VGPR banks indexed with s_set_gpr_idx, was built and measured in round 2: any
//    bank made the allocator spill at these sizes.  Removed; the measurements are in DESIGN.md section 5.)
//  * The "other" end of the accumulated span is one (theta, rho[, grad]) triple that is swapped with the moving
//    end when the walk turns around; nothing is written while the walk keeps its direction.  For models whose
//    gradient is recomputed it is parked in accumulator registers (AGPRs), which vector arithmetic cannot read and
//    the register allocator therefore leaves alone.
//  * The first doubling (a single leaf) is peeled out of the doubling loop: the two shapes share one generic lambda
//    but not their register assignments -- in one loop the allocator re-homed whole vectors on every doubling.
#pragma once

#include <type_traits>

#include "wn_traj.h"

namespace wn {

// Probe build only (tests/gpu_probes/pool_traffic.py, -DWN_COUNT_POOL): how many doubles per lane the span pool moved
// through each tier since the counters were last read -- [0] LDS stores, [1] LDS loads, [2] arena (HBM) stores,
// [3] arena loads, [4] transitions.  Compiled out of the product.
#if defined(WN_COUNT_POOL)
__device__ unsigned long long wn_pool_counts[5];
#define WN_COUNT(k, n) (pool_count[k] += (n))
#else
#define WN_COUNT(k, n) ((void)0)
#endif

// Geometry table of the on-chip kernels: waves per SIMD the register budget is cut for.  One vector costs 2*EPL VGPRs
// per lane; the moving end's two sets, the inverse mass and the two operands of a pool-side U-turn test must fit.
template <class Model, int EPL>
constexpr int chip_waves_per_simd() {
  // a model that keeps its gradient vectors (two more per set) gets the next larger register budget
  if (!Model::kCheapGrad) return EPL >= 8 ? 1 : EPL == 4 ? 2 : 3;
  return EPL >= 16 ? 1 : EPL == 8 ? 2 : EPL == 4 ? 3 : 4;
}

// WARM: the kernel of the adaptive warmup transitions (Adam, mass estimator) or of the frozen sampler's -- two
// instantiations, because the sampler's, freed of the adaptation code, needs fewer registers (measured: +3 % on the
// one-wavefront headline kernel, +8 % on the two-wavefront one)
// FMA: the multiply-adds of the integrator are fused (one rounding), as an FMA-target build of the reference fuses
// them; false: every product is rounded before it is added, the bits of the reference's x86-64 -O3 builds
template <class Model, int NW, int EPL, bool WARM = false, bool FMA = false>
struct TrajChip : TrajBase<TrajChip<Model, NW, EPL, WARM, FMA>, Model, NW> {
  using Base = TrajBase<TrajChip<Model, NW, EPL, WARM, FMA>, Model, NW>;
  using typename Base::Meta;
  using Base::P; using Base::lds_pool; using Base::arena; using Base::tid; using Base::lane; using Base::wave;
  using Base::chain; using Base::aux; using Base::n_grad; using Base::n_draw; using Base::draw_base; using Base::err;
  using Base::max_error; using Base::min_micro; using Base::step; using Base::free_mask; using Base::onchip_mask;
  using Base::w_draw0; using Base::w_score0; using Base::meta; using Base::bcast; using Base::w_ref;
  static constexpr int L = Base::L;
  static constexpr int NP = EPL / 2;
  static constexpr int kDp = L * EPL;  // padded dimension: a compile-time constant of the geometry
  static constexpr bool kNoGrad = Model::kCheapGrad;  // the gradient is recomputed from theta at each use
  // ... and, for an element-wise model, so are the log density's terms: they are taken ONCE, for the state a macro step
  // ends in, in the same loop as the kinetic energy's (energy_partials) -- two accumulation chains side by side where
  // each alone is sixteen dependent multiply-adds of a single wavefront, every one waiting for its predecessor
  static constexpr bool kLateLogp = kNoGrad && Model::kElementwise;
  static constexpr bool kHasStartState = true;
  static constexpr bool kZeroCopy = false;
  // The transition's initial point is loaded into -- and its result left in -- set 1: the first doubling's single leaf
  // then runs 1 -> 0 and ends where every later leaf pair starts (the moving end is set 0 at pair boundaries) without
  // the copy of two vectors that a 0 -> 1 leaf needed.
  static constexpr int kI = 1;
  // values that live long and are read rarely sit in accumulator registers -- in the kernels built for one or two
  // wavefronts per SIMD; the others name no accumulator register (wn_gfx950.h: PlainDouble)
  static constexpr bool kPark = chip_waves_per_simd<Model, EPL>() <= 2;
  static constexpr bool kParkScalars = kPark;
  using Parked = std::conditional_t<kPark, ParkedDouble, PlainDouble>;
  static_assert(EPL % 2 == 0, "lanes own 16-byte pairs");

  double th[2][EPL], rh[2][EPL], g[2][EPL];  // the two sets of the moving end (g is dead when kNoGrad)
  double im[EPL], mp[EPL];                    // inverse mass diagonal, model parameters
  // The accumulated span's other end.  When the gradient is recomputed from theta (kNoGrad) the pair (theta, rho)
  // has registers of its own instead of two pool buffers -- accumulator registers, parked there explicitly
  // (ParkedDouble, wn_gfx950.h): the top-level U-turn test fetches it, turning around exchanges it with the moving
  // end, and the pool's LDS vectors all serve the span stack.
  static constexpr bool kOtherRegs = kNoGrad;
  static constexpr int kOther = -4;           // "this vector is the other end's theta" (kOtherRegs)
  Parked oth[EPL], orh[EPL];
  int n_lds;                                  // pool buffers [0, n_lds) live in LDS, the rest in the HBM arena
  int shift_parity = 0;                       // which copy of the shift scratch the last exchange used
#if defined(WN_COUNT_POOL)
  int pool_count[4];
#endif

  __device__ __forceinline__ TrajChip(const Params& p, WN_LDS double* pool, WN_LDS Meta* m, WN_LDS double* r,
                                      WN_LDS double* bc, double* ar)
      : Base(p, pool, m, r, bc, ar) {
    n_lds = p.pool_lds;
    onchip_mask = n_lds >= 64 ? ~0ull : ((1ull << n_lds) - 1ull);
  }

  __device__ __forceinline__ static constexpr bool is_warmup() { return WARM; }
  // a * b + c, fused or not (what Model::eval sees as cx.mad)
  __device__ __forceinline__ static double mad(double a, double b, double c) {
    if constexpr (FMA) return __builtin_fma(a, b, c);
    return a * b + c;
  }

  // ---- model context (what Model::eval sees) ------------------------------------------------------
  __device__ __forceinline__ int index(int j) const { return ((j >> 1) * L + tid) * 2 + (j & 1); }
  __device__ __forceinline__ bool valid(int j) const { return index(j) < P.dim; }
  template <int S>
  __device__ __forceinline__ double G(int j) const {
    return kNoGrad ? Model::grad_elem(th[S][j], mp[j]) : g[S][j];
  }

  // Neighbours in coordinate order, for models whose gradient couples adjacent coordinates (wn_model_api.h):
  // prev[j] = v at coordinate index(j) - 1, next[j] = v at index(j) + 1, 0.0 beyond either end of the padded vector.
  // Slot pairs are consecutive coordinates, so half of the neighbours are the lane's own; the others are the
  // adjacent lane's (a lane shuffle), the adjacent wavefront's edge lane (through LDS) or the adjacent pair row's.
  __device__ __forceinline__ void shift(const double (&v)[EPL], double (&prev)[EPL], double (&next)[EPL]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      prev[2 * k + 1] = v[2 * k];
      next[2 * k] = v[2 * k + 1];
    }
    // [pair slot][wavefront][first lane's even | last lane's odd], two copies used alternately: a wavefront that writes
    // a copy again has passed the barrier of the exchange in between, which every wavefront reaches only after its
    // reads of that copy -- one barrier per exchange instead of two
    shift_parity ^= 1;
    WN_LDS double* sh = this->bcast + 2 + shift_parity * (2 * 8 * NW);
    if (NW > 1) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (lane == 0) sh[(k * NW + wave) * 2] = v[2 * k];
        if (lane == 63) sh[(k * NW + wave) * 2 + 1] = v[2 * k + 1];
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      // pair m - 1's odd element / pair m + 1's even element, m = k * L + tid
      const double up = lane_below(v[2 * k + 1]);
      const double dn = lane_above(v[2 * k]);
      double left_edge, right_edge;  // what lane 0 / lane 63 of this wavefront take instead
      if (NW == 1) {
        left_edge = k > 0 ? __shfl(v[2 * (k > 0 ? k - 1 : 0) + 1], 63, 64) : 0.0;
        right_edge = k + 1 < NP ? __shfl(v[2 * (k + 1 < NP ? k + 1 : k)], 0, 64) : 0.0;
      } else {
        const bool first = wave == 0, last = wave == NW - 1;
        left_edge = !first ? sh[(k * NW + wave - 1) * 2 + 1] : (k > 0 ? sh[((k - 1) * NW + NW - 1) * 2 + 1] : 0.0);
        right_edge = !last ? sh[(k * NW + wave + 1) * 2] : (k + 1 < NP ? sh[((k + 1) * NW) * 2] : 0.0);
      }
      prev[2 * k] = lane == 0 ? left_edge : up;
      next[2 * k + 1] = lane == 63 ? right_edge : dn;
    }
  }

  // ---- vector buffers -----------------------------------------------------------------------------
  // the lane's byte offset inside a pair row, rebuilt at every vector operation (behind an optimisation barrier): left
  // alone, the compiler keeps one 64-bit per-lane offset per pair-row group alive for the whole kernel
  __device__ __forceinline__ const char* lane_base(const double* base) const {
    unsigned long long lo = static_cast<unsigned long long>(static_cast<uint32_t>(tid)) * 16ull;
    asm volatile("" : "+v"(lo));
    return reinterpret_cast<const char*>(base) + lo;
  }
  __device__ __forceinline__ void vload(const double* base, double (&v)[EPL]) const {
    const char* lb = lane_base(base);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const v2f64 t = *reinterpret_cast<const v2f64*>(lb + k * (16 * L));
      v[2 * k] = t[0];
      v[2 * k + 1] = t[1];
    }
  }
  __device__ __forceinline__ void vstore(double* base, const double (&v)[EPL]) const {
    const char* lb = lane_base(base);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f64 t;
      t[0] = v[2 * k];
      t[1] = v[2 * k + 1];
      *reinterpret_cast<v2f64*>(const_cast<char*>(lb) + k * (16 * L)) = t;
    }
  }
  // the chain's own planes are read once and written once per transition: streamed past the L2 (nt) so that they
  // do not evict the arena vectors a deep tree spills there
  __device__ __forceinline__ void vload_stream(const double* base, double (&v)[EPL]) const {
    const char* lb = lane_base(base);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const v2f64 t = stream_load(reinterpret_cast<const v2f64*>(lb + k * (16 * L)));
      v[2 * k] = t[0];
      v[2 * k + 1] = t[1];
    }
  }
  __device__ __forceinline__ void vstore_stream(double* base, const double (&v)[EPL]) const {
    const char* lb = lane_base(base);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f64 t;
      t[0] = v[2 * k];
      t[1] = v[2 * k + 1];
      stream_store(t, reinterpret_cast<v2f64*>(const_cast<char*>(lb) + k * (16 * L)));
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
  __device__ __forceinline__ void pool_load(int b, double (&v)[EPL]) {
    if (WN_LIKELY(b < n_lds)) {
      WN_COUNT(1, EPL);
      lds_load(lds_pool + b * kDp, v);
      return;
    }
    WN_COUNT(3, EPL);
    vload(arena + static_cast<long long>(b - n_lds) * kDp, v);
  }
  __device__ __forceinline__ void pool_store(int b, const double (&v)[EPL]) {
    if (WN_LIKELY(b < n_lds)) {
      WN_COUNT(0, EPL);
      lds_store(lds_pool + b * kDp, v);
      return;
    }
    WN_COUNT(2, EPL);
    vstore(arena + static_cast<long long>(b - n_lds) * kDp, v);
  }

  // ---- Hamiltonian pieces -------------------------------------------------------------------------
  template <int S>
  __device__ __forceinline__ double model_eval() {
    ++n_grad;
    double part = 0.0;
    if constexpr (!kLateLogp) Model::eval(*this, th[S], g[S], mp, aux, part);
    return part;
  }
  // what Model::eval sees when it is handed ONE element (kLateLogp)
  struct ElemCx {
    const TrajChip& t;
    int j;
    __device__ __forceinline__ static double mad(double a, double b, double c) { return TrajChip::mad(a, b, c); }
    __device__ __forceinline__ int index(int) const { return t.index(j); }
    __device__ __forceinline__ bool valid(int) const { return t.valid(j); }
    __device__ __forceinline__ int dim() const { return t.dim(); }
  };
  // the lane's partial sums of the log density (`part`: in for a model that accumulated it during the step, out) and of
  // the kinetic energy (util.hpp:220-223 before the -0.5) of set S, both in index order
  template <int S>
  __device__ __forceinline__ void energy_partials(double& part, double& ke) {
    if constexpr (kLateLogp) {
      part = 0.0;
      ke = 0.0;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const double t1[1] = {th[S][j]}, m1[1] = {mp[j]};
        double g1[1];
        ElemCx cx{*this, j};
        Model::template eval<1>(cx, t1, g1, m1, aux, part);
        ke = mad(im[j], rh[S][j] * rh[S][j], ke);
      }
    } else {
      ke = kinetic_partial<S>();
    }
  }
  // kinetic partial of set S (util.hpp:220-223 before the -0.5)
  template <int S>
  __device__ __forceinline__ double kinetic_partial() const {
    double ke = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) ke = mad(im[j], rh[S][j] * rh[S][j], ke);
    return ke;
  }
  // one leapfrog micro step (walnuts.hpp:329-332) reading set A and writing set B (A == B: in place);
  // returns the new state's log-density partial
  template <int A, int B>
  __device__ __forceinline__ double micro_step(double h, double half) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) rh[B][j] = mad(half, G<A>(j), rh[A][j]);
#pragma unroll
    for (int j = 0; j < EPL; ++j) th[B][j] = mad(h * im[j], rh[B][j], th[A][j]);
    const double part = model_eval<B>();
#pragma unroll
    for (int j = 0; j < EPL; ++j) rh[B][j] = mad(half, G<B>(j), rh[B][j]);
    return part;
  }
  // n micro steps in place on set S (walnuts.hpp:328-333)
  template <int S>
  __device__ __forceinline__ double leapfrog_inplace(double h, int n, double part) {
    const double half = 0.5 * h;
    for (int s = 0; s < n; ++s) part = micro_step<S, S>(h, half);
    return part;
  }
  __device__ __forceinline__ void finish_energy(double lp_sum, double ke_sum, double& logp_pos, double& logp_joint) {
    // wave-uniform results go back to scalar registers: they live long and would otherwise hold VGPR pairs
    logp_pos = uni(Model::finish(lp_sum, aux, P.dim));
    logp_joint = uni(logp_pos + (-0.5 * ke_sum));
  }
  // One wavefront per chain: the energies and the test |H0 - H1| <= max_error (walnuts.hpp:339, :234) from the packed
  // butterfly (the log-density sum in lanes 0-31, the kinetic sum in lanes 32-63).  The test is taken on the joint
  // energy where the arithmetic left it -- its scalar copy is made beside the branch, not in front of the compare.
  __device__ __forceinline__ bool energies_within(double packed, double logp_start, double& logp_pos, double& logp_joint) {
    const double lp_sum = uni(packed), ke_sum = lane_value(packed, 32);
    const double lp_v = Model::finish(lp_sum, aux, P.dim);
    const double lj_v = lp_v + (-0.5 * ke_sum);
    // (the same value in every lane, which the compiler cannot always see -- a model's finish() may read lane-held
    // by-products --: the lane mask of the compare makes the branch a scalar one)
    const bool ok = either_half(fabs(logp_start - lj_v) <= max_error);
    logp_pos = uni(lp_v);
    logp_joint = uni(lj_v);
    return ok;
  }

  // walnuts.hpp:218-235 in place on set S
  template <int S>
  __device__ __forceinline__ bool within_tolerance(double h, int n, double logp_entry) {
    double part = leapfrog_inplace<S>(h, n, 0.0);
    double ke;
    energy_partials<S>(part, ke);
    this->sum2(part, ke);
    double lp, lj;
    finish_energy(part, ke, lp, lj);
    return fabs(lj - logp_entry) <= max_error;
  }

  // walnuts.hpp:254-279.  The candidate (set S) is parked in pool buffers while coarser reverse paths are tried
  // from (theta', -rho', grad').
  template <int S>
  __device__ __forceinline__ bool reversible(double h, int n, double logp_joint) {
    if (WN_LIKELY(n == 1)) return true;
    const int k0 = this->alloc(), k1 = this->alloc(), k2 = kNoGrad ? -1 : this->alloc();
    pool_store(k0, th[S]);
    pool_store(k1, rh[S]);
    if (!kNoGrad) pool_store(k2, g[S]);
    bool result = true;
    bool first = true;
    while (n >= 2 * min_micro) {
      if (!first) {
        pool_load(k0, th[S]);
        if (!kNoGrad) pool_load(k2, g[S]);
      }
      first = false;
      double keep[EPL];
      pool_load(k1, keep);
#pragma unroll
      for (int j = 0; j < EPL; ++j) rh[S][j] = -keep[j];
      n /= 2;
      h *= 2;
      if (within_tolerance<S>(h, n, logp_joint)) {
        result = false;
        break;
      }
    }
    pool_load(k0, th[S]);
    pool_load(k1, rh[S]);
    if (!kNoGrad) pool_load(k2, g[S]);
    drain_loads();  // (rare path; nothing stays in flight into the set's registers, see wn_gfx950.h)
    this->release(k0);
    this->release(k1);
    this->release(k2);
    return result;
  }

  // walnuts.hpp:192-201 partial sums: `H` is the outer end of the newer span, (a, b) = (theta, rho) of the far end.
  // The products are taken with theta_H - theta_far whatever the direction.  Walking backwards the reference's
  // difference is theta_far - theta_H = -(theta_H - theta_far) exactly, and negating every term negates every partial
  // sum and every butterfly stage exactly (rounding to nearest is symmetric): the reference's two sums are MINUS
  // these, so its tests `sum < 0` read `sum > 0` here (turned_sign) -- no sign flip per element.
  template <int H>
  __device__ __forceinline__ void uturn_partials(const double (&a)[EPL], const double (&b)[EPL], double& p_hot,
                                                 double& p_far) const {
    p_hot = 0.0;
    p_far = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const double sd = im[j] * (th[H][j] - a[j]);
      p_hot = mad(rh[H][j], sd, p_hot);
      p_far = mad(b[j], sd, p_far);
    }
  }
  // rho_fw . sd < 0 || rho_bk . sd < 0 (walnuts.hpp:199-200) from the two sums as uturn_partials takes them
  __device__ __forceinline__ static bool turned_sign(double p_hot, double p_far, bool fwd) {
    return fwd ? (p_hot < 0 || p_far < 0) : (p_hot > 0 || p_far > 0);
  }
  // CH consecutive slots (a whole number of pairs) of a pool vector, starting at slot j0
  template <int CH>
  __device__ __forceinline__ void pool_load_slots(int b, int j0, double (&v)[CH]) {
    const int k0 = j0 / 2;
    WN_COUNT(b < n_lds ? 1 : 3, CH);
    if (WN_LIKELY(b < n_lds)) {
      const WN_LDS double* base = lds_pool + b * kDp;
#pragma unroll
      for (int k = 0; k < CH / 2; ++k) {
        const v2f64 t = *reinterpret_cast<const WN_LDS v2f64*>(base + ((k0 + k) * L + tid) * 2);
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
      }
      return;
    }
    const double* base = arena + static_cast<long long>(b - n_lds) * kDp;
    const char* lb = lane_base(base);
#pragma unroll
    for (int k = 0; k < CH / 2; ++k) {
      const v2f64 t = *reinterpret_cast<const v2f64*>(lb + (k0 + k) * (16 * L));
      v[2 * k] = t[0];
      v[2 * k + 1] = t[1];
    }
  }
  // the moving end (always set 0 when a pool-resident span is merged) against a span end kept in the pool.  With 16
  // elements per lane the two operands are taken half a vector at a time: the kernel is at its register limit there,
  // and 64 more live registers mean as many moves to and from the accumulator file.  Both operands in LDS -- the usual
  // case -- is decided by ONE test in front of the loads.
  template <bool LDS_ONLY, int CH>
  __device__ __forceinline__ void pool_operands(int b, int j0, double (&v)[CH]) {
    if constexpr (LDS_ONLY) {
      WN_COUNT(1, CH);
      const WN_LDS double* base = lds_pool + b * kDp;
#pragma unroll
      for (int k = 0; k < CH / 2; ++k) {
        const v2f64 t = *reinterpret_cast<const WN_LDS v2f64*>(base + ((j0 / 2 + k) * L + tid) * 2);
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
      }
    } else {
      pool_load_slots<CH>(b, j0, v);
    }
  }
  template <bool LDS_ONLY>
  __device__ __forceinline__ void uturn_pool_partials(int bth, int brh, double& p_hot, double& p_far) {
    constexpr int CH = EPL >= 16 ? EPL / 2 : EPL;
    p_hot = 0.0;
    p_far = 0.0;
#pragma unroll
    for (int h = 0; h < EPL / CH; ++h) {
      double a[CH], b[CH];
      pool_operands<LDS_ONLY, CH>(bth, h * CH, a);
      pool_operands<LDS_ONLY, CH>(brh, h * CH, b);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int jj = h * CH + j;
        const double sd = im[jj] * (th[0][jj] - a[j]);
        p_hot = mad(rh[0][jj], sd, p_hot);
        p_far = mad(b[j], sd, p_far);
      }
    }
  }
  __device__ __forceinline__ bool uturn_pool(int bth, int brh, bool fwd) {
    double p_hot, p_far;
    if (WN_LIKELY(bth < n_lds && brh < n_lds)) {
      uturn_pool_partials<true>(bth, brh, p_hot, p_far);
    } else {
      uturn_pool_partials<false>(bth, brh, p_hot, p_far);
    }
    return turned_packed(p_hot, p_far, fwd);
  }
  // the two sums' reduction and the test, one wavefront per chain: read off the compare's lane mask
  __device__ __forceinline__ bool turned_packed(double p_hot, double p_far, bool fwd) {
    if constexpr (NW == 1) {
      const double packed = wave_sum_packed(p_hot, p_far);
      return either_half(fwd ? packed < 0 : packed > 0);
    } else {
      this->sum2(p_hot, p_far);
      return turned_sign(p_hot, p_far, fwd);
    }
  }

  // One attempt of a macro step (the body of walnuts.hpp:323-343's loop): n micro steps of size h from set A into set
  // 1-A, the new state's energies, the test |H0 - H1| <= max_error.  `want_turn`: also the U-turn test of the two-leaf
  // span (previous leaf = set A, new leaf = set 1-A), whose two sums share the energy reduction.  ONE: n == 1 is known.
  template <int A, bool ONE, bool want_turn>
  __device__ __forceinline__ bool leaf_attempt(double h, int n, bool fwd, double logp_start, double& logp_pos,
                                               double& logp_joint, bool first, bool& turn_now) {
    constexpr int B = 1 - A;
    WN_PHASE(kPhLeapfrog);
    h = opaque_uniform(h);
    double part = micro_step<A, B>(h, 0.5 * h);
    if constexpr (!ONE) part = leapfrog_inplace<B>(h, n - 1, part);
    double ke;
    energy_partials<B>(part, ke);
    double p_hot = 0.0, p_far = 0.0;
    WN_PHASE(kPhEnergy);
    bool within;
    if constexpr (NW == 1) {
      double packed_turn = 0.0;
      if (want_turn) {
        uturn_partials<B>(th[A], rh[A], p_hot, p_far);
        packed_turn = wave_sum_packed(p_hot, p_far);
      }
      within = energies_within(wave_sum_packed(part, ke), logp_start, logp_pos, logp_joint);
      turn_now = want_turn && either_half(fwd ? packed_turn < 0 : packed_turn > 0);
    } else {
      if (want_turn) {
        uturn_partials<B>(th[A], rh[A], p_hot, p_far);
        this->sum4(part, ke, p_hot, p_far);
      } else {
        this->sum2(part, ke);
      }
      finish_energy(part, ke, logp_pos, logp_joint);
      within = fabs(logp_start - logp_joint) <= max_error;
      turn_now = turned_sign(p_hot, p_far, fwd);
    }
    if (first) {  // num_steps == min_micro_steps, walnuts.hpp:335-338
      // Adam's state lives in wavefront 0's scratch (store_scalars reads it there): the others skip the update
      if (is_warmup() && wave == 0) this->adam_record(fabs(logp_start - logp_joint));
    }
    WN_PHASE(kPhRestart);
    return within;
  }

  // walnuts.hpp:307-345 from set A into set 1-A.
  // Control flow is laid out for the usual leaf -- one micro step, accepted at the first step size --: a scalar branch
  // costs a lone wavefront 15-30 cycles (tests/gpu_probes/branch_cost.hip), and a loop with two ways out makes the
  // compiler test its exit flags again behind it.  So the usual leaf is straight-line code behind ONE test (n == 1)
  // and leaves through ONE more (accepted); everything else -- more micro steps, halvings, the reversibility check --
  // is the general loop below it, which the usual leaf never enters.
  template <int A, bool want_turn>
  __device__ __forceinline__ bool macro_step(bool fwd, double logp_start, double& logp_pos, double& logp_joint,
                                             bool& turned) {
    constexpr int B = 1 - A;
    double h = fwd ? step : -step;
    int n = min_micro;
    int halvings = 0;
    bool turn_now = false;
    if (WN_LIKELY(n == 1)) {
      if (WN_LIKELY((leaf_attempt<A, true, want_turn>(h, 1, fwd, logp_start, logp_pos, logp_joint, true, turn_now)))) {
        turned = turn_now;
        return true;  // (one micro step is reversible by definition, walnuts.hpp:261-263)
      }
      halvings = 1;
      if (halvings >= P.max_halvings) return false;
      n = 2;
      h *= 0.5;
    }
    for (;;) {
      // (the restart state does not change from one attempt to the next, and left to itself the optimiser computes
      // what this loop derives from it -- the gradient -theta, sixteen negated copies -- in front of the loop, i.e. on
      // the usual leaf's path into the next macro step)
#pragma unroll
      for (int j = 0; j < EPL; ++j) launder(th[A][j]);
      if (leaf_attempt<A, false, want_turn>(h, n, fwd, logp_start, logp_pos, logp_joint, halvings == 0, turn_now)) {
        turned = turn_now;
        WN_PHASE(kPhReversible);
        return reversible<B>(h, n, logp_joint);
      }
      if (++halvings >= P.max_halvings) return false;
      n *= 2;
      h *= 0.5;
    }
  }

  // give a symbolic vector (kHot = set 0, kStart = set 1) a pool buffer
  __device__ __forceinline__ int materialize(int ref, bool rho) {
    if (ref >= 0) return ref;
    const int b = this->alloc();
    if (ref == kHot) {
      pool_store(b, rho ? rh[0] : th[0]);
    } else {
      pool_store(b, rho ? rh[1] : th[1]);
    }
    return b;
  }

  // ------------------------------------------------------------------------------------
  // one MCMC transition (walnuts.hpp:520-563 wrapped as adaptive_walnuts.hpp:234-251 or walnuts.hpp:682-692)
  // ------------------------------------------------------------------------------------
  __device__ void run(int chain_id) {
    WN_PHASE(kPhPrologue);
    this->refresh_ids();
#if defined(WN_COUNT_POOL)
    pool_count[0] = pool_count[1] = pool_count[2] = pool_count[3] = 0;
#endif
    chain = chain_id;
    err = 0;
    n_grad = 0;
    n_draw = 0;
    draw_base = -1;
    max_error = P.max_error;
    free_mask = (P.pool_total >= 64) ? ~0ull : ((1ull << P.pool_total) - 1ull);
    const long long row = static_cast<long long>(chain) * kDp;
    const bool warm = is_warmup();
    if constexpr (WARM) {
      if (WN_UNLIKELY(this->cold().est_mode == 2)) {
        observe_only(row);
        return;
      }
    }

    // momentum refresh + initial point (walnuts.hpp:528-535), into set 0
    double lp_pos, lj;
    {
      double part = begin_transition(row, warm);
      double ke;
      energy_partials<kI>(part, ke);
      this->sum2(part, ke);
      finish_energy(part, ke, lp_pos, lj);
    }
    WN_MARK(kPhEvaluated);
    this->prefetch_next_chain();
    // The accumulated span (walnuts.hpp:34-131) is: the moving end (set 0), the other end parked in the pool,
    // the selected position and three scalars.  Both ends are the initial point to begin with.
    int o_th = -1, o_rh = -1, o_g = -1;
    int a_sel;
    if (kOtherRegs) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        park(oth[j], th[kI][j]);
        park(orh[j], rh[kI][j]);
      }
      a_sel = kOther;
    } else {
      o_th = this->alloc_cold();
      o_rh = this->alloc_cold();
      o_g = kNoGrad ? -1 : this->alloc_cold();
      pool_store(o_th, th[kI]);
      pool_store(o_rh, rh[kI]);
      if (!kNoGrad) pool_store(o_g, g[kI]);
      a_sel = o_th;
    }
    // (a_w: the accumulated span's weight -- wn_traj.h, "span weights"; the initial point weighs exactly 1)
    double lj_hot = lj, lj_other = lj, a_w = 1.0, a_lpsel = lp_pos;
    w_ref = lj;
    bool hot_fw = true;

    // One doubling (walnuts.hpp:541-558); returns whether the tree keeps growing.  The first doubling is a single
    // leaf, every later one a loop over leaf pairs: two instantiations, so that neither carries the other's
    // register shuffles.
    int depth = 1;
    auto doubling = [&](auto first_tag) -> bool {
      constexpr bool kFirst = decltype(first_tag)::value;
      WN_PHASE(kPhDoublingStart);
      // (draws are asked for in stretches, wn_traj.h ensure_draws: here this one and, for the single leaf of the first
      // doubling, the Metropolis draw of its merge)
      this->ensure_draws(2);
      const bool fwd = this->uniform01_ready() < 0.5;  // bernoulli(0.5), walnuts.hpp:552
      if (kFirst) {
        hot_fw = fwd;
      } else if (fwd != hot_fw) {
        // the walk turns around: the moving end and the parked end change places
        if (kOtherRegs) {
          if (a_sel == kOther) {  // the selected position was the other end's: it gets a buffer of its own
            a_sel = this->alloc_cold();
            double t[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) t[j] = fetch(oth[j]);
            pool_store(a_sel, t);
          }
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            const double t0 = th[0][j], t1 = rh[0][j];
            th[0][j] = fetch(oth[j]);
            rh[0][j] = fetch(orh[j]);
            park(oth[j], t0);
            park(orh[j], t1);
          }
        } else {
          double a[EPL], b[EPL];
          pool_load(o_th, a);
          pool_load(o_rh, b);
          if (o_th == a_sel) o_th = this->alloc_cold();  // the selected position keeps its buffer
          pool_store(o_th, th[0]);
          pool_store(o_rh, rh[0]);
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            th[0][j] = a[j];
            rh[0][j] = b[j];
          }
        }
        if (!kNoGrad) {
          double a[EPL];
          pool_load(o_g, a);
          pool_store(o_g, g[0]);
#pragma unroll
          for (int j = 0; j < EPL; ++j) g[0][j] = a[j];
        }
        const double t = lj_hot;
        lj_hot = lj_other;
        lj_other = t;
        hot_fw = fwd;
      }
      double h_cur = lj_hot;

      // ---- build_span(depth-1) (walnuts.hpp:464-495) as a post-order walk over 2^(depth-1) leaves, two at a time ----
      const int nleaf = kFirst ? 1 : 1 << (depth - 1);
      int sp = 0;
      bool ok = true;
      bool top_turned = false;
      int c_in_th = kHot, c_in_rh = kHot, c_sel = kHot;
      double c_w = 0.0, c_lpsel = 0.0;
      if (kFirst) {
        // a single leaf, from the initial point in set 1 into set 0: its U-turn test against the span's other end (= the
        // initial point, still in set 1 when the leaf is done) rides in the leaf's reduction
        double leaf_lp, leaf_lj;
        ok = macro_step<kI, true>(fwd, h_cur, leaf_lp, leaf_lj, top_turned);
        if (WN_LIKELY(ok)) {
          h_cur = leaf_lj;
          double none = 0.0;
          c_w = this->leaf_weight(leaf_lj, 0, a_w, none);
          c_lpsel = leaf_lp;
        }
      } else {
        for (int i = 0; i < nleaf; i += 2) {
          double e_lp, e_lj, leaf_lp, leaf_lj;
          bool pair_turned = false, unused = false;
          if (WN_UNLIKELY((!macro_step<0, false>(fwd, h_cur, e_lp, e_lj, unused)))) {  // build_leaf, walnuts.hpp:420-442
            ok = false;
            break;
          }
          if (WN_UNLIKELY((!macro_step<1, true>(fwd, e_lj, leaf_lp, leaf_lj, pair_turned)))) {
            ok = false;
            break;
          }
          // the draws this pair's merges can take: its own, one per stack level, the doubling's Metropolis draw
          this->ensure_draws(kMaxLevels + 2);
          // the two leaves' weights (wn_traj.h, "span weights"), in the order the leaves were built.  (Had the odd
          // leaf failed, the extension -- and with it the transition's tree -- would have ended: the even leaf's
          // weight is not missed.)
          double e_w, leaf_w;
          this->pair_weights(e_lj, leaf_lj, sp, a_w, e_w, leaf_w);
          if (is_warmup() && wave == 0) this->adam_make_room();
          h_cur = leaf_lj;
          // level-0 merge, combine<Barker> (walnuts.hpp:370-386): old = the even leaf (set 1), new = the odd leaf (set 0)
          WN_PHASE(kPhCombine);
          {
            if (WN_UNLIKELY(pair_turned)) {  // walnuts.hpp:490-492
              ok = false;
              break;
            }
            const double total = uni(e_w + leaf_w);
            const bool update = this->uniform01_ready() * total < leaf_w;
            c_in_th = kStart;
            c_in_rh = kStart;
            c_sel = update ? kHot : kStart;
            c_lpsel = update ? leaf_lp : e_lp;
            c_w = total;
          }
          for (int l = 1; ((i + 1) >> l) & 1; ++l) {
            --sp;
            int s_in_th, s_in_rh, s_sel;
            double s_w, s_lpsel;
            this->stack_read(sp, s_in_th, s_in_rh, s_sel, s_w, s_lpsel);
            WN_PHASE(kPhUturn);
            if (WN_UNLIKELY(uturn_pool(s_in_th, s_in_rh, fwd))) {  // walnuts.hpp:490-492
              ok = false;
              break;
            }
            WN_PHASE(kPhCombine);
            const double total = uni(s_w + c_w);
            const bool update = this->uniform01_ready() * total < c_w;
            const int n_sel = update ? c_sel : s_sel;
            const double n_lpsel = update ? c_lpsel : s_lpsel;
            // The merged span keeps the older span's inner end and one of the two selections; what goes back to the
            // pool: the newer span's inner momentum, its inner position unless that is the selection kept, and the
            // selection not kept -- the older span's stays if it is that span's inner end.  (An index of the newer
            // span can be symbolic -- a leaf still in its register set -- and has no bit.)
            free_mask |= Base::pool_bit(c_in_rh) | (c_in_th != n_sel ? Base::pool_bit(c_in_th) : 0ull) |
                         (update ? (s_sel != s_in_th ? (1ull << s_sel) : 0ull) : Base::pool_bit(c_sel));
            c_in_th = s_in_th;
            c_in_rh = s_in_rh;
            c_sel = n_sel;
            c_lpsel = n_lpsel;
            c_w = total;
          }
          if (WN_UNLIKELY(!ok)) break;
          WN_PHASE(kPhPush);
          if (i + 2 < nleaf) {
            // the next pair overwrites both sets: whatever is still symbolic gets pool buffers
            if (((i + 1) & 2) == 0) {
              // no merge above level 0 (the cascade's first test is bit 1 of i + 1): the span is the pair itself, its
              // inner end the even leaf (set 1), its selection one of the two leaves.  Buffers in the order inner
              // position, inner momentum, selection; one tier test for all of them (the lowest free index comes first,
              // LDS buffers are the low indices).
              const bool sel_hot = c_sel == kHot;
              const int b0 = this->alloc(), b1 = this->alloc();
              const int b2 = this->alloc_if(sel_hot, b0);
              if (WN_LIKELY((b1 > b2 ? b1 : b2) < n_lds)) {
                WN_COUNT(0, (sel_hot ? 3 : 2) * EPL);
                lds_store(lds_pool + b0 * kDp, th[1]);
                lds_store(lds_pool + b1 * kDp, rh[1]);
                if (sel_hot) lds_store(lds_pool + b2 * kDp, th[0]);
              } else {
                pool_store(b0, th[1]);
                pool_store(b1, rh[1]);
                if (sel_hot) pool_store(b2, th[0]);
              }
              c_in_th = b0;
              c_in_rh = b1;
              c_sel = b2;
            } else if (c_sel < 0) {
              // after a cascade the ends are the older span's pool buffers; the selection may still be a leaf of the pair
              c_sel = materialize(c_sel, false);
            }
            this->stack_push(sp, c_in_th, c_in_rh, c_sel, c_w, c_lpsel);
            ++sp;
          }
        }
      }
      if (WN_UNLIKELY(!ok)) {  // walnuts.hpp:543-545
        err |= static_cast<int>(kNoteExtensionFailed);
        return false;
      }

      // ---- merge into the accumulated span (walnuts.hpp:546-548) ----
      WN_PHASE(kPhTopMerge);
      bool turned;
      if (kFirst) {
        turned = top_turned;
      } else {
        if (kOtherRegs) {
          double p_hot, p_far;
          double a[EPL], b[EPL];
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            a[j] = fetch(oth[j]);
            b[j] = fetch(orh[j]);
          }
          uturn_partials<0>(a, b, p_hot, p_far);
          turned = turned_packed(p_hot, p_far, fwd);
        } else {
          turned = uturn_pool(o_th, o_rh, fwd);
        }
      }
      const bool update = this->uniform01_ready() * a_w < c_w;  // Metropolis
      // the new span's inner end is never read again
      this->release_unless(c_in_th, c_sel, -3, -3);
      this->release_unless(c_in_rh, -3, -3, -3);
      if (update) {
        c_sel = materialize(c_sel, false);
        if (a_sel != o_th) this->release(a_sel);  // (release ignores the symbolic kOther)
        a_sel = c_sel;
        a_lpsel = c_lpsel;
      } else {
        this->release(c_sel);
      }
      lj_hot = h_cur;
      a_w = uni(a_w + c_w);
      return !turned;  // walnuts.hpp:549,556-558
    };
    if (depth <= P.max_depth && doubling(std::true_type{})) {
      for (depth = 2; depth <= P.max_depth; ++depth) {
        if (!doubling(std::false_type{})) break;
      }
    }

    // ---- selected state out (walnuts.hpp:560-562), estimator update (adaptive_walnuts.hpp:247-248) ----
    WN_PHASE(kPhEpilogue);
    this->refresh_ids();
    finish_transition(a_sel, row, warm);  // (runs the batched Adam update behind its plane requests)
    WN_MARK(kPhStored);
    this->store_scalars(warm, depth, a_lpsel);
    WN_MARK(kPhScalars);
#if defined(WN_COUNT_POOL)
    if (tid == 0) {
      for (int k = 0; k < 4; ++k) atomicAdd(&wn_pool_counts[k], static_cast<unsigned long long>(pool_count[k]));
      atomicAdd(&wn_pool_counts[4], 1ull);
    }
#endif
  }

  // The mass estimator's observation of a warmup transition's result (adaptive_walnuts.hpp:247-248) is applied by the
  // NEXT transition's prologue (kDeferObservation), not by its own epilogue: what it observes -- the selected position
  // and the gradient there -- is the next transition's initial point, which the prologue holds and evaluates anyway;
  // the four estimator planes it reads arrive behind the momentum generator like every other prologue load (in the
  // epilogue they stood in the open: ~4 700 of a warmup transition's 63 000 cycles, tests/gpu_probes/timeline.py),
  // the two sums of squared deviations are read once per transition instead of twice, and a model whose gradient is
  // carried (not recomputed) saves its re-evaluation.  Same operations in the same order, so the same bits.  Between
  // two launches the engine remembers that an observation is pending (Params::est_mode) and applies it before anything
  // else looks at the estimator (wn_engine: flush_pending_observation -> est_mode 2 -> observe_only()).
  static constexpr bool kDeferObservation = true;
  // online_moments.hpp:184-191 (the lazy delta: (y - mean_new)^2) for both moments, in place; -> the new weights
  __device__ __forceinline__ void observe(double (&mean)[EPL], double (&ssd)[EPL], double (&smean)[EPL], double (&sssd)[EPL],
                                          long long iteration) {
    const auto& Q = this->cold();
    const double discount = 1.0 - 1.0 / (Q.mass_init_count + static_cast<double>(iteration));
    const wnd::SharedDivisor wd(discount * w_draw0 + 1);
    const wnd::SharedDivisor ws(discount * w_score0 + 1);
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      mean[j] += (th[kI][j] - mean[j]) / wd;
      ssd[j] = discount * ssd[j] + (th[kI][j] - mean[j]) * (th[kI][j] - mean[j]);
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      smean[j] += (G<kI>(j) - smean[j]) / ws;
      sssd[j] = discount * sssd[j] + (G<kI>(j) - smean[j]) * (G<kI>(j) - smean[j]);
    }
    w_draw0 = wd.b;
    w_score0 = ws.b;
  }

  // load the chain, refresh the momentum (walnuts.hpp:528-529), evaluate the initial point (:532)
  __device__ __forceinline__ double begin_transition(long long row, bool warm) {
    const auto& Q = this->cold();
    // Order matters: the chain's scalars and planes are REQUESTED first, then the momentum's standard normals are
    // generated (pure arithmetic: Philox + Box-Muller, ~1 000 VALU instructions per wavefront at 16 elements per
    // lane), and only then is anything loaded looked at -- the round trips to HBM hide behind the generator.
    this->request_tuning(warm);
    // (a launch runs Params::fused transitions of the chain back to back: after the first one the position is the
    // selected state the previous epilogue left in set 1, and the frozen inverse mass is still in its registers)
    const bool first_of_launch = this->fuse_t == 0;
    if (first_of_launch) vload_stream(Q.theta + row, th[kI]);
    if (Model::kUsesParams) vload(Q.model_params, mp);
    double ds[EPL], ss[EPL];  // warmup: the estimator's two sums of squared deviations; sampling: ds = cholesky_mass
    double dm[EPL], sm[EPL];  // warmup with an observation pending: the two means
    const bool pending = warm && (!first_of_launch || Q.est_mode == 1);
    if (warm) {
      vload_stream(Q.est_draw_ssd + row, ds);
      vload_stream(Q.est_score_ssd + row, ss);
      if (pending) {
        vload_stream(Q.est_draw_mean + row, dm);
        vload_stream(Q.est_score_mean + row, sm);
      }
    } else {
      if (first_of_launch) vload_stream(Q.inv_mass + row, im);
      vload_stream(Q.chol_mass + row, ds);  // 1/sqrt(inv_mass), walnuts.hpp:647, stored once by freeze_kernel
    }
    WN_MARK(kPhLoadsIssued);
    const bool fed = Q.rng_mode == kRngBuffer;
    if (WN_UNLIKELY(fed)) {
      vload_stream(Q.z_buf + row, rh[kI]);
    } else {
      const uint64_t seed = Q.seed;
      const uint32_t key_chain = Q.chain_offset + chain, key_tr = this->transition_now();
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const uint32_t pair = static_cast<uint32_t>(k * L + tid);
        wnd::stream_normal_pair(seed, key_chain, key_tr, wnd::kStreamMomentum, pair, rh[kI][2 * k], rh[kI][2 * k + 1],
                                this->gather_tab());
      }
    }
    WN_MARK(kPhMomentum);
    this->finish_tuning(warm);
    WN_MARK(kPhTuned);
    // the initial point (walnuts.hpp:532): its gradient is also what a pending observation observes
    const double part = model_eval<kI>();
    if (pending) {
      observe(dm, ds, sm, ss, this->warmup_iter_now() - 1);
      vstore_stream(Q.est_draw_mean + row, dm);
      vstore_stream(Q.est_draw_ssd + row, ds);
      vstore_stream(Q.est_score_mean + row, sm);
      vstore_stream(Q.est_score_ssd + row, ss);
    }
    // rho = cholesky_mass * z (walnuts.hpp:528-529), padding slots zero
    const wnd::SharedDivisor wd0(warm ? w_draw0 : 1.0), ws0(warm ? w_score0 : 1.0);  // (wnd::SharedDivisor: same quotients)
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      double chol;
      if (warm) {
        // adaptive_walnuts.hpp:235-236 with MassEstimator::inv_mass_estimate :89-94
        // (wnd::sqrt_normal<true>: sqrt's bits for every operand that is normal when it is finite and positive)
        im[j] = wnd::sqrt_normal<true>((ds[j] / wd0) / (ss[j] / ws0));
        chol = wnd::sqrt_normal<true>(1.0 / im[j]);
      } else {
        // Streaming the plane costs 8 KB of the 48 KB a 1024-dimensional chain moves per transition; re-evaluating
        // 1 / sqrt(im) (a division and a square root per element) costs ~2 000 cycles of the ~85 000 a transition took
        // when this was measured (round 2): 2.22 ms / 3.7 GB streamed against 2.27 ms / 3.15 GB recomputed.
        chol = ds[j];
      }
      const double r = chol * rh[kI][j];
      rh[kI][j] = (fed || valid(j)) ? r : 0.0;
    }
    return part;
  }

  // Params::est_mode == 2: nothing but the pending observation of this chain (the engine's flush before a read of the
  // estimator, a freeze, or anything that replaces the positions).  The observed state is the position plane's row.
  __device__ __forceinline__ void observe_only(long long row) {
    const auto& Q = this->cold();
    this->request_tuning(true);
    vload_stream(Q.theta + row, th[kI]);
    if (Model::kUsesParams) vload(Q.model_params, mp);
    double ds[EPL], ss[EPL], dm[EPL], sm[EPL];
    vload_stream(Q.est_draw_ssd + row, ds);
    vload_stream(Q.est_score_ssd + row, ss);
    vload_stream(Q.est_draw_mean + row, dm);
    vload_stream(Q.est_score_mean + row, sm);
    this->finish_tuning(true);
    (void)model_eval<kI>();
    observe(dm, ds, sm, ss, Q.warmup_iter - 1);
    vstore_stream(Q.est_draw_mean + row, dm);
    vstore_stream(Q.est_draw_ssd + row, ds);
    vstore_stream(Q.est_score_mean + row, sm);
    vstore_stream(Q.est_score_ssd + row, ss);
    if (NW > 1) __syncthreads();  // (every wavefront has read the old weights before thread 0 replaces them)
    if (tid == 0) {
      Q.est_weight[2 * chain] = w_draw0;
      Q.est_weight[2 * chain + 1] = w_score0;
    }
    this->prefetch_next_chain();
  }

  __device__ __forceinline__ void finish_transition(int a_sel, long long row, bool warm) {
    const auto& Q = this->cold();
    // warmup: the batched Adam update (adam.hpp:70-86 for every macro step of this transition, wave-uniform arithmetic);
    // the estimator's observation of the selected state waits for the next prologue (kDeferObservation)
    if (warm && wave == 0) this->adam_flush();
    if (kOtherRegs && a_sel == kOther) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) th[kI][j] = fetch(oth[j]);
    } else {
      pool_load(a_sel, th[kI]);
    }
    WN_MARK(kPhSelLoaded);
    // the position plane is read again by the NEXT launch only (its first prologue, or the engine's flush of the pending
    // observation): the launch's last transition of the chain writes it
    if (this->fuse_t + 1 >= Q.fused) vstore_stream(Q.theta + row, th[kI]);
    double* draws = Q.draws_out;
    if (WN_LIKELY(draws != nullptr)) {
      double* out = this->draw_row();
      // an unpadded row on a 16-byte boundary takes the pair stores; anything else goes element by element
      if (WN_LIKELY(P.dim == kDp && ((reinterpret_cast<unsigned long long>(out) & 15ull) == 0ull))) {
        vstore_stream(out, th[kI]);
      } else {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          if (valid(j)) stream_store(th[kI][j], &out[index(j)]);
        }
      }
    }
  }
};

template <class Model, int NW, int EPL, bool WARM, bool FMA>
__global__ __launch_bounds__(64 * NW, (chip_waves_per_simd<Model, EPL>())) void transition_kernel_chip(const Params P) {
  persistent_loop<TrajChip<Model, NW, EPL, WARM, FMA>, NW>(P);
}

}  // namespace wn
