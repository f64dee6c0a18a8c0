// wn_summary.hip -- posterior summaries over device-resident draws: the reference's
// include/walnutpie/summary.hpp (mean :370-378, sample_variance :396-405, quantiles :483-514, autocovariance
// :529-545, r_hat :593-619, effective_sample_size :663-749, monte_carlo_standard_error :764-768) for ragged
// collections of chains that never leave HBM.
//
// Everything here is HBM-bound streaming over the [chain][draw][dim] block the sampler wrote: a wavefront's lanes
// are 64 consecutive dimensions of one draw (512-byte coalesced rows), every reduction runs in the order the
// reference's loops run (over draws inside a chain, then over chains), so results do not depend on the launch
// geometry.  Two places restate the reference's method for the device:
//   * autocovariance: the direct sum  acov[t] = (1/N) sum_n (y[n]-ybar)(y[n+t]-ybar)  that the reference's
//     zero-padded FFT evaluates (summary.hpp:55-73), in blocks of kLagBlock lags held in registers.  The ESS
//     (Geyer's initial monotone sequence, :712-729) only reads lags up to where the sequence stops, so lag blocks
//     are computed on demand until every dimension has stopped -- typically one or two passes over the draws.
//   * quantiles: instead of sorting every column (:505-506), a radix select on the order-preserving 64-bit image
//     of the doubles, 4 bits per pass, for all requested order statistics of all dimensions at once: 16 passes
//     over the draws, LDS-private histograms (lane = dimension: no intra-wave conflicts), exact.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "wn_host.h"

namespace wns {

constexpr int kBlock = wn::kSummaryBlock;  // (launch sizes are constants of the platform layer: wn_hip.h)
constexpr int kWaves = kBlock / 64;
constexpr int kLagBlock = 16;     // lags per pass of the autocovariance kernel (accumulators + ring in VGPRs)
constexpr int kChainBlock = 256;  // chains per run of the two-stage sums over chains
// the ESS computes its lag table for this many chains at a time (a multiple of kChainBlock): the [chains][16][D]
// workspace stays at 1 GB for 1 024 dimensions however many chains there are
constexpr int kLagSlabChains = wn::kSummaryLagSlabChains;
constexpr int kRows = 8;          // draws a lane loads ahead in the radix-select pass
constexpr int kMaxTargets = 16;
// radix select: draws left per (target, dimension) at which the rest is finished in a gathered list instead of more
// passes over all draws
constexpr int kCandidateCap = wn::kSummaryCandidateCap;

struct View {
  const double* x;        // draws
  const long long* off;   // [C] offset (doubles) of the chain's first draw
  const int* len;         // [C] draws in the chain
  const long long* row0;  // [C] index of the chain's first draw in the stacked (unified) order
  int C, D;
};

// wave w of the grid -> (chain, 64-column tile); lane -> column
struct Slot {
  int c, d;
  bool ok;
};
static __device__ __forceinline__ Slot slot_of(const View& v) {
  const int tiles = (v.D + 63) / 64;
  const long long wave = static_cast<long long>(blockIdx.x) * kWaves + threadIdx.x / 64;
  Slot s;
  s.c = static_cast<int>(wave / tiles);
  s.d = static_cast<int>(wave % tiles) * 64 + static_cast<int>(threadIdx.x % 64);
  s.ok = s.c < v.C && s.d < v.D;
  return s;
}
static int slot_blocks(int C, int D) {
  const long long waves = static_cast<long long>(C) * ((D + 63) / 64);
  return static_cast<int>((waves + kWaves - 1) / kWaves);
}

// detail::col_means (:19-22) and detail::sample_variance (:93-99) of every chain; also the chain's column sums
static __global__ void chain_moments_kernel(View v, double* csum, double* cmean, double* cvar) {
  const Slot s = slot_of(v);
  if (!s.ok) return;
  const double* p = v.x + v.off[s.c] + s.d;
  const int n = v.len[s.c];
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += p[static_cast<long long>(i) * v.D];
  const double mean = sum / static_cast<double>(n);
  double q = 0.0;
  for (int i = 0; i < n; ++i) {
    const double r = p[static_cast<long long>(i) * v.D] - mean;
    q += r * r;
  }
  const long long o = static_cast<long long>(s.c) * v.D + s.d;
  csum[o] = sum;
  cmean[o] = mean;
  cvar[o] = q / static_cast<double>(n - 1);
}
// per chain: sum over draws of (x - mu)^2 about the GLOBAL mean (:400-403)
static __global__ void chain_sqdev_kernel(View v, const double* mu, double* csq) {
  const Slot s = slot_of(v);
  if (!s.ok) return;
  const double* p = v.x + v.off[s.c] + s.d;
  const int n = v.len[s.c];
  const double m = mu[s.d];
  double q = 0.0;
  for (int i = 0; i < n; ++i) {
    const double r = p[static_cast<long long>(i) * v.D] - m;
    q += r * r;
  }
  csq[static_cast<long long>(s.c) * v.D + s.d] = q;
}
// The same two kernels for the common shape -- few draws per chain, an even number of dimensions, 16-byte aligned
// rows: a lane owns TWO adjacent columns (16-byte loads: a wavefront's load is 1 KB of one draw) and keeps the chain's
// at most NMAX draws in registers.  All loads are issued before the first is used (NMAX KB in flight per wavefront),
// and the second loop of the sample variance reads registers, not memory: ONE pass over the draws where the column
// loops above make two.  Same operations in the same order per column, hence the same bits.
constexpr int kRegDraws = 32;
struct Slot2 {
  int c, d;
  bool ok;
};
static __device__ __forceinline__ Slot2 slot2_of(const View& v) {
  const int tiles = (v.D + 127) / 128;
  const long long wave = static_cast<long long>(blockIdx.x) * kWaves + threadIdx.x / 64;
  Slot2 s;
  s.c = static_cast<int>(wave / tiles);
  s.d = static_cast<int>(wave % tiles) * 128 + 2 * static_cast<int>(threadIdx.x % 64);
  s.ok = s.c < v.C && s.d < v.D;
  return s;
}
static int slot2_blocks(int C, int D) {
  const long long waves = static_cast<long long>(C) * ((D + 127) / 128);
  return static_cast<int>((waves + kWaves - 1) / kWaves);
}
template <int NMAX>
static __global__ __launch_bounds__(kBlock) void chain_moments_wide_kernel(View v, double* csum, double* cmean, double* cvar) {
  const Slot2 s = slot2_of(v);
  if (!s.ok) return;
  const double* p = v.x + v.off[s.c] + s.d;
  const int n = v.len[s.c];
  v2f64 r[NMAX];
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n) r[i] = wn::stream_load(reinterpret_cast<const v2f64*>(p + static_cast<long long>(i) * v.D));
  double sum0 = 0.0, sum1 = 0.0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n) {
      sum0 += r[i][0];
      sum1 += r[i][1];
    }
  const double mean0 = sum0 / static_cast<double>(n), mean1 = sum1 / static_cast<double>(n);
  double q0 = 0.0, q1 = 0.0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n) {
      const double a = r[i][0] - mean0, b = r[i][1] - mean1;
      q0 += a * a;
      q1 += b * b;
    }
  const long long o = static_cast<long long>(s.c) * v.D + s.d;
  *reinterpret_cast<v2f64*>(csum + o) = v2f64{sum0, sum1};
  *reinterpret_cast<v2f64*>(cmean + o) = v2f64{mean0, mean1};
  *reinterpret_cast<v2f64*>(cvar + o) =
      v2f64{q0 / static_cast<double>(n - 1), q1 / static_cast<double>(n - 1)};
}
template <int NMAX>
static __global__ __launch_bounds__(kBlock) void chain_sqdev_wide_kernel(View v, const double* mu, double* csq) {
  const Slot2 s = slot2_of(v);
  if (!s.ok) return;
  const double* p = v.x + v.off[s.c] + s.d;
  const int n = v.len[s.c];
  v2f64 r[NMAX];
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n) r[i] = wn::stream_load(reinterpret_cast<const v2f64*>(p + static_cast<long long>(i) * v.D));
  const double m0 = mu[s.d], m1 = mu[s.d + 1];
  double q0 = 0.0, q1 = 0.0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i)
    if (i < n) {
      const double a = r[i][0] - m0, b = r[i][1] - m1;
      q0 += a * a;
      q1 += b * b;
    }
  *reinterpret_cast<v2f64*>(csq + static_cast<long long>(s.c) * v.D + s.d) = v2f64{q0, q1};
}
// Sums over chains (rows of a [C][width] matrix with row stride `ld`), in two deterministic stages: runs of
// kChainBlock consecutive chains are summed left to right, then the run totals are summed left to right.  Up to
// kChainBlock chains this IS the reference's left-to-right loop over chains; beyond, it keeps 65 536-chain
// reductions from serialising on one thread per column.  With `mu` the terms are (a - mu)^2.
static __global__ void block_sum_kernel(const double* a, int C, long long ld, int width, const double* mu,
                                        double* partial /*[blocks][width]*/) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int nb = (C + kChainBlock - 1) / kChainBlock;
  if (i >= static_cast<long long>(nb) * width) return;
  const int b = static_cast<int>(i / width), w = static_cast<int>(i % width);
  const int c_hi = (b + 1) * kChainBlock, c1 = c_hi < C ? c_hi : C;
  double s = 0.0;
  if (mu != nullptr) {
    const double m = mu[w];
    for (int c = b * kChainBlock; c < c1; ++c) {
      const double r = a[static_cast<long long>(c) * ld + w] - m;
      s += r * r;
    }
  } else {
    for (int c = b * kChainBlock; c < c1; ++c) s += a[static_cast<long long>(c) * ld + w];
  }
  partial[i] = s;
}
static __global__ void final_sum_kernel(const double* partial, int nb, long long ld, int width, double denom,
                                        double* out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= width) return;
  double s = 0.0;
  for (int b = 0; b < nb; ++b) s += partial[static_cast<long long>(b) * ld + w];
  out[w] = s / denom;
}

// kLagBlock lags [t0, t0 + kLagBlock) of every chain's autocovariance, per column.  The thread streams its chain's
// column once: y[n] = x[n] - ybar and a ring of the kLagBlock values y[n + t0 .. n + t0 + kLagBlock - 1] stay in
// registers, so each draw costs one new load (+ one re-read of x[n], an L2 hit) for kLagBlock multiply-adds, each
// lag accumulated over n ascending.  Output: blk[c][j][d] (block mode, for the ESS) or the reference's stacked
// [num_draws][D] table (full mode: row = first draw of the chain + lag).
static __device__ __forceinline__ void chain_lag_block(const double* p, long long D, int n_c, double ybar, int t0,
                                                       double (&acc)[kLagBlock]) {
  double ring[kLagBlock];
#pragma unroll
  for (int j = 0; j < kLagBlock; ++j) {
    acc[j] = 0.0;
    ring[j] = (t0 + j < n_c) ? p[static_cast<long long>(t0 + j) * D] - ybar : 0.0;  // y[t0 + j]
  }
  // lag t0 + j pairs y[n] with y[n + t0 + j] for n < n_c - t0 - j: the longest run is n < n_c - t0
  const int runs = n_c - t0;
  for (int n0 = 0; n0 < runs; n0 += kLagBlock) {
#pragma unroll
    for (int i = 0; i < kLagBlock; ++i) {
      const int n = n0 + i;
      if (n < runs) {
        // y[n]: for the first block it is the ring's oldest entry (lag 0 pairs y[n] with itself); later blocks read
        // the draw again
        const double y = (t0 == 0) ? ring[i % kLagBlock] : p[static_cast<long long>(n) * D] - ybar;
#pragma unroll
        for (int j = 0; j < kLagBlock; ++j) {
          // ring[(i + j) % kLagBlock] holds y[n + t0 + j]
          if (n + t0 + j < n_c) acc[j] += y * ring[(i + j) % kLagBlock];
        }
        const int nxt = n + t0 + kLagBlock;  // the slot of y[n + t0] is free now: refill with y[n + t0 + kLagBlock]
        ring[i % kLagBlock] = (nxt < n_c) ? p[static_cast<long long>(nxt) * D] - ybar : 0.0;
      }
    }
  }
}
static __global__ void acov_block_kernel(View v, int c_base, int c_count, const double* cmean, int t0,
                                         int max_lag /*exclusive*/, double* blk /*[c_count][kLagBlock][D]*/, double* full) {
  Slot s = slot_of(v);
  if (s.c >= c_count || s.d >= v.D) return;
  const int c_local = s.c;
  s.c += c_base;
  const long long D = v.D;
  const int n_c = v.len[s.c];
  double acc[kLagBlock];
  chain_lag_block(v.x + v.off[s.c] + s.d, D, n_c, cmean[static_cast<long long>(s.c) * D + s.d], t0, acc);
#pragma unroll
  for (int j = 0; j < kLagBlock; ++j) {
    const int t = t0 + j;
    if (t >= max_lag) continue;
    const double val = acc[j] / static_cast<double>(n_c);  // biased estimate, :70-71
    if (blk != nullptr) blk[(static_cast<long long>(c_local) * kLagBlock + j) * D + s.d] = val;
    if (full != nullptr && t < n_c) full[(v.row0[s.c] + t) * D + s.d] = val;
  }
}
// Geyer's initial positive + monotone sequence on paired lags and the ESS (:706-745), resumable: a dimension whose
// sequence needs a lag that has not been computed yet parks its state and raises *need_more.
static __global__ void geyer_kernel(int D, int min_len, int avail, int first, const double* macov, const double* W,
                                    const double* var_plus, double n_total, double tau_floor, int* st_t,
                                    double* st_even, double* st_odd, int* st_done, double* rho, double* ess,
                                    int* need_more) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  if (!first && st_done[d]) return;
  const double w = W[d], vp = var_plus[d];
  int t;
  double even, odd;
  auto R = [&](int i) -> double& { return rho[static_cast<long long>(i) * D + d]; };
  if (first) {
    for (int i = 0; i < min_len; ++i) R(i) = 0.0;
    even = 1.0;
    R(0) = even;
    odd = 1.0 - (w - macov[static_cast<long long>(1) * D + d]) / vp;
    R(1) = odd;
    t = 1;
    st_done[d] = 0;
  } else {
    t = st_t[d];
    even = st_even[d];
    odd = st_odd[d];
  }
  const int bound = min_len - 4;
  while (t < bound && (even + odd) > 0.0) {
    if (t + 2 >= avail) {
      st_t[d] = t;
      st_even[d] = even;
      st_odd[d] = odd;
      *need_more = 1;
      return;
    }
    even = 1.0 - (w - macov[static_cast<long long>(t + 1) * D + d]) / vp;
    odd = 1.0 - (w - macov[static_cast<long long>(t + 2) * D + d]) / vp;
    if ((even + odd) >= 0.0) {
      R(t + 1) = even;
      R(t + 2) = odd;
    }
    if (R(t + 1) + R(t + 2) > R(t - 1) + R(t)) {
      R(t + 1) = (R(t - 1) + R(t)) / 2.0;
      R(t + 2) = R(t + 1);
    }
    t += 2;
  }
  const int max_t = t;
  if (even > 0.0) R(max_t + 1) = even;  // antithetic-tail correction
  double head = 0.0;
  for (int i = 0; i < max_t; ++i) head += R(i);
  double tau = -1.0 + 2.0 * head + R(max_t + 1);
  tau = (tau < tau_floor) ? tau_floor : tau;  // std::max(tau_hat, 1 / log10(N_total))
  ess[d] = n_total / tau;
  st_done[d] = 1;
}

// ---- quantiles: radix select ---------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned long long order_key(double x) {
  unsigned long long u;
  __builtin_memcpy(&u, &x, sizeof(u));
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending doubles <-> ascending unsigned keys
}
static double key_value(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double x;
  std::memcpy(&x, &u, sizeof(x));
  return x;
}
// One 4-bit pass: per (dimension, target) a 16-bin histogram of the digit at `shift` over the draws whose higher
// bits equal the target's prefix so far.  Lane = dimension, so a wavefront's LDS atomics never collide; the layout
// [target][bin][lane] keeps them on distinct banks.  In the first pass all targets share one histogram.
static __global__ void radix_hist_kernel(View v, int T, int shift, int first, const unsigned long long* prefix,
                                         unsigned long long* ghist, int chains_per_block) {
  WN_DYN_SMEM(smem_raw);
  unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);
  const int tiles = (v.D + 63) / 64;
  const int tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  const int d = tile * 64 + lane;
  const int Teff = first ? 1 : T;
  for (int i = threadIdx.x; i < Teff * 16 * 64; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  if (d < v.D) {
    // the bits above this pass's digit, per target: an element counts for target t iff its own high bits equal them
    // (~0 never matches: the top bits of a shifted key are zero)
    const int hs = shift + 4;
    unsigned long long pre_hi[kMaxTargets];
#pragma unroll
    for (int t = 0; t < kMaxTargets; ++t)
      pre_hi[t] = (!first && t < T) ? prefix[static_cast<long long>(t) * v.D + d] >> hs : ~0ull;
    const int c_hi = (chunk + 1) * chains_per_block, c_end = c_hi < v.C ? c_hi : v.C;
    for (int c = chunk * chains_per_block + wave; c < c_end; c += kWaves) {
      const double* p = v.x + v.off[c] + d;
      const int n = v.len[c];
      // kRows loads in flight per lane before any of them is consumed: the LDS atomics below would otherwise
      // serialise the row loop on HBM latency
      for (int i0 = 0; i0 < n; i0 += kRows) {
        double x[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) x[r] = (i0 + r < n) ? p[static_cast<long long>(i0 + r) * v.D] : 0.0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          if (i0 + r >= n) break;
          const unsigned long long k = order_key(x[r]);
          const unsigned dig = static_cast<unsigned>(k >> shift) & 15u;
          if (first) {
            atomicAdd(&hist[dig * 64 + lane], 1u);
          } else {
            const unsigned long long k_hi = k >> hs;
#pragma unroll
            for (int t = 0; t < kMaxTargets; ++t) {
              if (k_hi == pre_hi[t]) atomicAdd(&hist[(t * 16 + dig) * 64 + lane], 1u);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Teff * 16 * 64; i += blockDim.x) {
    const unsigned cnt = hist[i];
    const int dd = tile * 64 + i % 64;
    if (cnt != 0u && dd < v.D) {
      const int tb = i / 64;  // t * 16 + bin
      atomicAdd(&ghist[static_cast<long long>(dd) * (T * 16) + tb], static_cast<unsigned long long>(cnt));
    }
  }
}
// choose the bin that holds the target's rank; extend its prefix; make the rank relative to the bin
static __global__ void radix_pick_kernel(int D, int T, int shift, int first, const unsigned long long* ghist,
                                         unsigned long long* prefix, unsigned long long* rank,
                                         unsigned long long* match /*draws left in the chosen bin*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * D) return;
  const int t = i / D, d = i % D;
  const unsigned long long* h = ghist + static_cast<long long>(d) * (T * 16) + (first ? 0 : t * 16);
  unsigned long long r = rank[i], below = 0ull;
  int bin = 15;
  for (int b = 0; b < 16; ++b) {
    const unsigned long long cnt = h[b];
    if (r < below + cnt) {
      bin = b;
      break;
    }
    below += cnt;
  }
  prefix[i] |= static_cast<unsigned long long>(bin) << shift;
  rank[i] = r - below;
  match[i] = h[bin];
}

// Once few draws are left in every target's bin, the remaining passes over ALL draws are replaced by one pass that
// copies the keys still in the running (high bits above `shift` equal to the target's prefix) to a short list per
// (target, dimension) ...
static __global__ void radix_collect_kernel(View v, int T, int shift, const unsigned long long* prefix, int cap,
                                            unsigned long long* cand /*[T*D][cap]*/, unsigned* cand_n /*[T*D]*/,
                                            int chains_per_block) {
  const int tiles = (v.D + 63) / 64;
  const int tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  const int d = tile * 64 + lane;
  if (d >= v.D) return;
  unsigned long long pre_hi[kMaxTargets];
#pragma unroll
  for (int t = 0; t < kMaxTargets; ++t) pre_hi[t] = t < T ? prefix[static_cast<long long>(t) * v.D + d] >> shift : ~0ull;
  const int c_hi = (chunk + 1) * chains_per_block, c_end = c_hi < v.C ? c_hi : v.C;
  for (int c = chunk * chains_per_block + wave; c < c_end; c += kWaves) {
    const double* p = v.x + v.off[c] + d;
    const int n = v.len[c];
    for (int i0 = 0; i0 < n; i0 += kRows) {
      double x[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) x[r] = (i0 + r < n) ? p[static_cast<long long>(i0 + r) * v.D] : 0.0;
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (i0 + r >= n) break;
        const unsigned long long k = order_key(x[r]);
        const unsigned long long k_hi = k >> shift;
#pragma unroll
        for (int t = 0; t < kMaxTargets; ++t) {
          if (k_hi == pre_hi[t]) {
            const long long slot = static_cast<long long>(t) * v.D + d;
            const unsigned pos = atomicAdd(&cand_n[slot], 1u);
            if (pos < static_cast<unsigned>(cap)) cand[slot * cap + pos] = k;
          }
        }
      }
    }
  }
}
// ... and the order statistic is finished inside the list, one bit at a time (the list's order does not matter).
// One 64-thread block per (target, dimension).
static __global__ void radix_finish_kernel(int TD, int shift, int cap, const unsigned long long* cand,
                                           const unsigned* cand_n, unsigned long long* prefix, unsigned long long* rank) {
  __shared__ unsigned cnt0;
  const int slot = blockIdx.x;
  if (slot >= TD) return;
  const unsigned long long* list = cand + static_cast<long long>(slot) * cap;
  const int n = static_cast<int>(cand_n[slot]);
  unsigned long long pre = prefix[slot], r = rank[slot];
  for (int b = shift - 1; b >= 0; --b) {
    if (threadIdx.x == 0) cnt0 = 0u;
    __syncthreads();
    unsigned mine = 0u;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long k = list[i];
      if ((k >> (b + 1)) == (pre >> (b + 1)) && ((k >> b) & 1ull) == 0ull) ++mine;
    }
    if (mine) atomicAdd(&cnt0, mine);
    __syncthreads();
    const unsigned long long zeros = cnt0;
    if (r >= zeros) {
      pre |= 1ull << b;
      r -= zeros;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    prefix[slot] = pre;
    rank[slot] = r;
  }
}

}  // namespace wns

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct wn_chains {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  size_t C = 0, D = 0;
  long long N = 0;
  int max_len = 0, min_len = 0;
  const double* x = nullptr;  // borrowed (view) or owned.p
  DevBuf<double> owned;
  DevBuf<long long> off, row0;
  DevBuf<int> len;
  std::vector<int> h_len;
  // lazily computed per-chain moments
  bool have_moments = false;
  DevBuf<double> csum, cmean, cvar;
  DevBuf<double> partial;  // run totals of the two-stage sums over chains
  DevBuf<double> lag_blk;  // [slab][kLagBlock][D] workspace of the ESS (kept between calls)

  wns::View view() const {
    return wns::View{x, off.p, len.p, row0.p, static_cast<int>(C), static_cast<int>(D)};
  }
  void use() const { HIP_OK(hipSetDevice(device)); }
  template <class T>
  void up(DevBuf<T>& b, const std::vector<T>& h) {
    b.alloc(h.size());
    if (!h.empty()) HIP_OK(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
  }
  template <class T>
  void down(const T* dev, T* host, size_t n) {
    if (n) HIP_OK(hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
  }
  void finish_setup(const std::vector<long long>& h_off) {
    std::vector<long long> h_row0(C);
    N = 0;
    max_len = 0;
    min_len = h_len.empty() ? 0 : h_len[0];
    offsets_even = true;
    for (size_t c = 0; c < C; ++c) {
      if (h_off[c] & 1) offsets_even = false;
      h_row0[c] = N;
      N += h_len[c];
      max_len = std::max(max_len, h_len[c]);
      min_len = std::min(min_len, h_len[c]);
    }
    up(off, h_off);
    up(row0, h_row0);
    up(len, h_len);
    HIP_OK(hipStreamSynchronize(stream));
  }
  // the register-resident two-columns-per-lane kernels apply: short chains, even D, 16-byte aligned rows
  bool offsets_even = true;
  bool wide_ok() const {
    return max_len <= wns::kRegDraws && D % 2 == 0 && offsets_even && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  }
  void ensure_moments() {
    if (have_moments) return;
    use();
    csum.alloc(C * D);
    cmean.alloc(C * D);
    cvar.alloc(C * D);
    if (wide_ok()) {
      hipLaunchKernelGGL(wns::chain_moments_wide_kernel<wns::kRegDraws>,
                         dim3(wns::slot2_blocks(static_cast<int>(C), static_cast<int>(D))), dim3(wns::kBlock), 0, stream,
                         view(), csum.p, cmean.p, cvar.p);
    } else {
      hipLaunchKernelGGL(wns::chain_moments_kernel, dim3(wns::slot_blocks(static_cast<int>(C), static_cast<int>(D))),
                         dim3(wns::kBlock), 0, stream, view(), csum.p, cmean.p, cvar.p);
    }
    HIP_OK(hipGetLastError());
    have_moments = true;
  }
  ~wn_chains() {
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

int col_blocks(size_t n) { return static_cast<int>((n + wns::kBlock - 1) / wns::kBlock); }

void check_sizes(size_t num_chains, size_t dims) {
  if (num_chains < 1) throw std::invalid_argument("require at least one chain");  // summary.hpp:133-137
  if (dims < 1) throw std::invalid_argument("dims must be in {1, 2, ... }");
  if (num_chains > 0x7fffffffull || dims > 0x7fffffffull) throw std::invalid_argument("too many chains or dimensions");
}

// out[w] = (sum over chains of a[c][w], or of (a[c][w] - mu[w])^2) / denom, chains grouped as block_sum_kernel says
void chain_sum(wn_chains* ch, const double* a, long long ld, int width, const double* mu, double denom, double* d_out) {
  const int C = static_cast<int>(ch->C), nb = (C + wns::kChainBlock - 1) / wns::kChainBlock;
  if (ch->partial.n < static_cast<size_t>(nb) * width) ch->partial.alloc(static_cast<size_t>(nb) * width);
  hipLaunchKernelGGL(wns::block_sum_kernel, dim3(col_blocks(static_cast<size_t>(nb) * width)), dim3(wns::kBlock), 0,
                     ch->stream, a, C, ld, width, mu, ch->partial.p);
  hipLaunchKernelGGL(wns::final_sum_kernel, dim3(col_blocks(width)), dim3(wns::kBlock), 0, ch->stream, ch->partial.p, nb,
                     static_cast<long long>(width), width, denom, d_out);
  HIP_OK(hipGetLastError());
}
void device_mean(wn_chains* ch, double* d_out /*device [D]*/) {  // :370-378
  ch->ensure_moments();
  chain_sum(ch, ch->csum.p, static_cast<long long>(ch->D), static_cast<int>(ch->D), nullptr, static_cast<double>(ch->N),
            d_out);
}
void device_sample_variance(wn_chains* ch, double* d_out /*device [D]*/) {  // :396-405
  DevBuf<double> mu, csq;
  mu.alloc(ch->D);
  csq.alloc(ch->C * ch->D);
  device_mean(ch, mu.p);
  if (ch->wide_ok()) {
    hipLaunchKernelGGL(wns::chain_sqdev_wide_kernel<wns::kRegDraws>,
                       dim3(wns::slot2_blocks(static_cast<int>(ch->C), static_cast<int>(ch->D))), dim3(wns::kBlock), 0,
                       ch->stream, ch->view(), mu.p, csq.p);
  } else {
    hipLaunchKernelGGL(wns::chain_sqdev_kernel, dim3(wns::slot_blocks(static_cast<int>(ch->C), static_cast<int>(ch->D))),
                       dim3(wns::kBlock), 0, ch->stream, ch->view(), mu.p, csq.p);
  }
  chain_sum(ch, csq.p, static_cast<long long>(ch->D), static_cast<int>(ch->D), nullptr, static_cast<double>(ch->N - 1),
            d_out);
  HIP_OK(hipStreamSynchronize(ch->stream));  // mu/csq are released on return
}
// detail::col_means / detail::sample_variance(draws) over the rows of a [C][D] per-chain matrix (:19-22, :101-105)
void rows_mean(wn_chains* ch, const double* a, double* d_mean) {
  chain_sum(ch, a, static_cast<long long>(ch->D), static_cast<int>(ch->D), nullptr, static_cast<double>(ch->C), d_mean);
}
void rows_sample_variance(wn_chains* ch, const double* a, double* d_var) {
  DevBuf<double> mu;
  mu.alloc(ch->D);
  rows_mean(ch, a, mu.p);
  chain_sum(ch, a, static_cast<long long>(ch->D), static_cast<int>(ch->D), mu.p,
            static_cast<double>(static_cast<long long>(ch->C) - 1), d_var);
  HIP_OK(hipStreamSynchronize(ch->stream));
}
void host_effective_sample_size(wn_chains* ch, double* out) {
  if (ch->N < 3) throw std::invalid_argument("chains must have at least 3 draws");  // :665-667
  // the reference indexes rho_hat_t(1) and rho_hat_t(max_t + 1) unconditionally (:705,:738): with fewer than three
  // draws in the shortest chain that is out of bounds there; an error here
  if (ch->min_len < 3) throw std::invalid_argument("each chain must have at least 3 draws");
  ch->ensure_moments();
  const int C = static_cast<int>(ch->C), D = static_cast<int>(ch->D), min_len = ch->min_len;
  DevBuf<double> W, between, var_plus, macov, even, odd, rho, ess;
  DevBuf<double>& blk = ch->lag_blk;
  DevBuf<int> st_t, st_done, need;
  W.alloc(D);
  var_plus.alloc(D);
  rows_mean(ch, ch->cvar.p, W.p);
  std::vector<double> h_w(D), h_vp(D);
  ch->down(W.p, h_w.data(), D);
  h_vp = h_w;
  if (C > 1) {  // var_plus = W + sample_variance(chain_means), :682-686
    between.alloc(D);
    rows_sample_variance(ch, ch->cmean.p, between.p);
    std::vector<double> h_b(D);
    ch->down(between.p, h_b.data(), D);
    for (int d = 0; d < D; ++d) h_vp[d] += h_b[d];
  }
  HIP_OK(hipMemcpyAsync(var_plus.p, h_vp.data(), D * sizeof(double), hipMemcpyHostToDevice, ch->stream));
  const int slab = std::min(C, wns::kLagSlabChains);
  const int runs_of_chains = (C + wns::kChainBlock - 1) / wns::kChainBlock;
  if (blk.n < static_cast<size_t>(slab) * wns::kLagBlock * D) blk.alloc(static_cast<size_t>(slab) * wns::kLagBlock * D);
  if (ch->partial.n < static_cast<size_t>(runs_of_chains) * wns::kLagBlock * D)
    ch->partial.alloc(static_cast<size_t>(runs_of_chains) * wns::kLagBlock * D);
  macov.alloc(static_cast<size_t>(min_len) * D);
  rho.alloc(static_cast<size_t>(min_len) * D);
  even.alloc(D);
  odd.alloc(D);
  ess.alloc(D);
  st_t.alloc(D);
  st_done.alloc(D);
  need.alloc(1);
  const double tau_floor = 1.0 / std::log10(static_cast<double>(ch->N));  // :741-742
  int avail = 0;
  bool first = true;
  while (true) {
    const int t0 = avail, nl = std::min(wns::kLagBlock, min_len - t0);
    // mean_acov_at_lag (:696-704) for the block's lags: slab by slab, the per-chain lag values go to the workspace
    // and the runs of kChainBlock chains inside the slab are summed; then the run totals, left to right
    const int width = wns::kLagBlock * D;
    for (int c0 = 0; c0 < C; c0 += slab) {
      const int cn = std::min(slab, C - c0), nb = (cn + wns::kChainBlock - 1) / wns::kChainBlock;
      hipLaunchKernelGGL(wns::acov_block_kernel, dim3(wns::slot_blocks(cn, D)), dim3(wns::kBlock), 0, ch->stream,
                         ch->view(), c0, cn, ch->cmean.p, t0, min_len, blk.p, static_cast<double*>(nullptr));
      hipLaunchKernelGGL(wns::block_sum_kernel, dim3(col_blocks(static_cast<size_t>(nb) * width)), dim3(wns::kBlock), 0,
                         ch->stream, blk.p, cn, static_cast<long long>(width), width, static_cast<const double*>(nullptr),
                         ch->partial.p + static_cast<size_t>(c0 / wns::kChainBlock) * width);
    }
    hipLaunchKernelGGL(wns::final_sum_kernel, dim3(col_blocks(static_cast<size_t>(nl) * D)), dim3(wns::kBlock), 0,
                       ch->stream, ch->partial.p, runs_of_chains, static_cast<long long>(width), nl * D,
                       static_cast<double>(C), macov.p + static_cast<size_t>(t0) * D);
    avail += nl;
    HIP_OK(hipMemsetAsync(need.p, 0, sizeof(int), ch->stream));
    hipLaunchKernelGGL(wns::geyer_kernel, dim3(col_blocks(D)), dim3(wns::kBlock), 0, ch->stream, D, min_len, avail,
                       first ? 1 : 0, macov.p, W.p, var_plus.p, static_cast<double>(ch->N), tau_floor, st_t.p, even.p,
                       odd.p, st_done.p, rho.p, ess.p, need.p);
    HIP_OK(hipGetLastError());
    first = false;
    int more = 0;
    ch->down(need.p, &more, 1);
    if (!more) break;
    if (avail >= min_len) throw std::runtime_error("autocovariance lags exhausted before the sequence stopped");
  }
  ch->down(ess.p, out, D);
}

}  // namespace

extern "C" {

int wn_chains_view(wn_chains** out, const double* draws_dev, size_t num_chains, size_t max_len, size_t dims,
                   int64_t chain_stride, const int64_t* lengths, int device, void* stream, WalnutpyError** err) {
  return guarded(err, [&] {
    check_sizes(num_chains, dims);
    if (draws_dev == nullptr) throw std::invalid_argument("draws must not be null");
    if (chain_stride < static_cast<int64_t>(max_len * dims)) throw std::invalid_argument("chain_stride is smaller than max_len * dims");
    auto ch = std::make_unique<wn_chains>();
    ch->device = device;
    ch->use();
    if (stream != nullptr) {
      ch->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
      HIP_OK(hipStreamCreateWithFlags(&ch->stream, hipStreamNonBlocking));
      ch->own_stream = true;
    }
    ch->C = num_chains;
    ch->D = dims;
    ch->x = draws_dev;
    std::vector<long long> h_off(num_chains);
    ch->h_len.resize(num_chains);
    for (size_t c = 0; c < num_chains; ++c) {
      const int64_t l = lengths ? lengths[c] : static_cast<int64_t>(max_len);
      if (l < 1) throw std::invalid_argument("each chain must have at least one draw");  // :139-150
      if (l > static_cast<int64_t>(max_len)) throw std::invalid_argument("chain length exceeds max_len");
      ch->h_len[c] = static_cast<int>(l);
      h_off[c] = static_cast<long long>(c) * chain_stride;
    }
    ch->finish_setup(h_off);
    *out = ch.release();
  });
}

int wn_chains_adopt(wn_chains** out, double* draws_dev, size_t num_chains, size_t max_len, size_t dims,
                    int64_t chain_stride, const int64_t* lengths, int device, WalnutpyError** err) {
  wn_chains* ch = nullptr;
  const int rc = wn_chains_view(&ch, draws_dev, num_chains, max_len, dims, chain_stride, lengths, device, nullptr, err);
  if (rc != 0) return rc;
  ch->owned.p = draws_dev;  // freed with the handle
  ch->owned.n = num_chains * static_cast<size_t>(chain_stride);
  *out = ch;
  return 0;
}

int wn_chains_upload(wn_chains** out, const double* draws_host, size_t dims, const int64_t* sizes, size_t num_chains,
                     int device, WalnutpyError** err) {
  return guarded(err, [&] {
    check_sizes(num_chains, dims);
    auto ch = std::make_unique<wn_chains>();
    ch->device = device;
    ch->use();
    HIP_OK(hipStreamCreateWithFlags(&ch->stream, hipStreamNonBlocking));
    ch->own_stream = true;
    ch->C = num_chains;
    ch->D = dims;
    std::vector<long long> h_off(num_chains);
    ch->h_len.resize(num_chains);
    long long total = 0;
    for (size_t c = 0; c < num_chains; ++c) {
      if (sizes[c] < 1) throw std::invalid_argument("each chain must have at least one draw");
      if (sizes[c] > 0x7fffffffll) throw std::invalid_argument("chain too long");
      ch->h_len[c] = static_cast<int>(sizes[c]);
      h_off[c] = total * static_cast<long long>(dims);
      total += sizes[c];
    }
    ch->owned.alloc(static_cast<size_t>(total) * dims);
    HIP_OK(hipMemcpyAsync(ch->owned.p, draws_host, static_cast<size_t>(total) * dims * sizeof(double),
                          hipMemcpyHostToDevice, ch->stream));
    ch->x = ch->owned.p;
    ch->finish_setup(h_off);
    *out = ch.release();
  });
}

void wn_chains_destroy(wn_chains* ch) { delete ch; }
size_t wn_chains_num_chains(const wn_chains* ch) { return ch->C; }
size_t wn_chains_dims(const wn_chains* ch) { return ch->D; }
size_t wn_chains_num_draws(const wn_chains* ch) { return static_cast<size_t>(ch->N); }
size_t wn_chains_min_chain_size(const wn_chains* ch) { return static_cast<size_t>(ch->min_len); }
const double* wn_chains_device_draws(const wn_chains* ch) { return ch->x; }
int wn_chains_device(const wn_chains* ch) { return ch->device; }

int wn_summary_mean(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    ch->use();
    DevBuf<double> d;
    d.alloc(ch->D);
    device_mean(ch, d.p);
    ch->down(d.p, out, ch->D);
  });
}
int wn_summary_sample_variance(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    ch->use();
    DevBuf<double> d;
    d.alloc(ch->D);
    device_sample_variance(ch, d.p);
    ch->down(d.p, out, ch->D);
  });
}
int wn_summary_sample_standard_deviation(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    ch->use();
    DevBuf<double> d;
    d.alloc(ch->D);
    device_sample_variance(ch, d.p);
    ch->down(d.p, out, ch->D);
    for (size_t i = 0; i < ch->D; ++i) out[i] = std::sqrt(out[i]);  // :423-426
  });
}
int wn_summary_r_hat(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (ch->C < 2) throw std::invalid_argument("require at least two chains to compute R-hat");  // :595-597
    if (ch->min_len < 3) throw std::invalid_argument("each chain must have at least 3 draws");    // :598-603
    ch->use();
    ch->ensure_moments();
    const int D = static_cast<int>(ch->D);
    DevBuf<double> var_mu, mean_sig;
    var_mu.alloc(D);
    mean_sig.alloc(D);
    rows_sample_variance(ch, ch->cmean.p, var_mu.p);
    rows_mean(ch, ch->cvar.p, mean_sig.p);
    std::vector<double> a(D), b(D);
    ch->down(var_mu.p, a.data(), D);
    ch->down(mean_sig.p, b.data(), D);
    for (int d = 0; d < D; ++d) out[d] = std::sqrt(1.0 + a[d] / b[d]);  // :616-618
  });
}
int wn_summary_autocovariance(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    ch->use();
    ch->ensure_moments();
    const int C = static_cast<int>(ch->C), D = static_cast<int>(ch->D);
    DevBuf<double> full;
    full.alloc(static_cast<size_t>(ch->N) * D);
    for (int t0 = 0; t0 < ch->max_len; t0 += wns::kLagBlock)
      hipLaunchKernelGGL(wns::acov_block_kernel, dim3(wns::slot_blocks(C, D)), dim3(wns::kBlock), 0, ch->stream,
                         ch->view(), 0, C, ch->cmean.p, t0, ch->max_len, static_cast<double*>(nullptr), full.p);
    HIP_OK(hipGetLastError());
    ch->down(full.p, out, static_cast<size_t>(ch->N) * D);
  });
}
int wn_summary_effective_sample_size(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    ch->use();
    host_effective_sample_size(ch, out);
  });
}
int wn_summary_monte_carlo_standard_error(wn_chains* ch, double* out, WalnutpyError** err) {
  return guarded(err, [&] {  // :764-768
    ch->use();
    std::vector<double> ess(ch->D);
    host_effective_sample_size(ch, ess.data());
    DevBuf<double> d;
    d.alloc(ch->D);
    device_sample_variance(ch, d.p);
    ch->down(d.p, out, ch->D);
    for (size_t i = 0; i < ch->D; ++i) out[i] = std::sqrt(out[i]) / std::sqrt(ess[i]);
  });
}
int wn_summary_quantiles(wn_chains* ch, const double* probs, size_t num_probs, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    for (size_t k = 0; k < num_probs; ++k)
      if (!(probs[k] >= 0) || !(probs[k] <= 1)) throw std::invalid_argument("probs must be in [0, 1]");  // :485-496
    if (num_probs == 0) return;
    ch->use();
    const int C = static_cast<int>(ch->C), D = static_cast<int>(ch->D);
    const long long N = ch->N;
    const double n_minus_1 = static_cast<double>(N - 1);
    // the order statistics to find: sorted[lo_k] and sorted[hi_k] for every probability (:507-511)
    std::vector<unsigned long long> ranks;
    std::vector<double> frac(num_probs);
    for (size_t k = 0; k < num_probs; ++k) {
      const double h = probs[k] * n_minus_1;
      const long long lo = static_cast<long long>(std::floor(h));
      const long long hi = std::min(lo + 1, N - 1);
      frac[k] = h - static_cast<double>(lo);
      ranks.push_back(static_cast<unsigned long long>(lo));
      ranks.push_back(static_cast<unsigned long long>(hi));
    }
    std::vector<double> stat(ranks.size() * static_cast<size_t>(D));  // [target][D]
    const int tiles = (D + 63) / 64;
    // enough blocks to fill the chip, each a contiguous run of chains
    const int want_chunks = std::max(1, 2048 / tiles);
    const int chains_per_block = std::max(wns::kWaves, (C + want_chunks - 1) / want_chunks);
    const int chunks = (C + chains_per_block - 1) / chains_per_block;
    for (size_t g0 = 0; g0 < ranks.size(); g0 += wns::kMaxTargets) {
      const int T = static_cast<int>(std::min<size_t>(wns::kMaxTargets, ranks.size() - g0));
      DevBuf<unsigned long long> prefix, rank, ghist;
      prefix.alloc(static_cast<size_t>(T) * D);
      rank.alloc(static_cast<size_t>(T) * D);
      ghist.alloc(static_cast<size_t>(D) * T * 16);
      std::vector<unsigned long long> h_rank(static_cast<size_t>(T) * D);
      for (int t = 0; t < T; ++t)
        for (int d = 0; d < D; ++d) h_rank[static_cast<size_t>(t) * D + d] = ranks[g0 + t];
      HIP_OK(hipMemcpyAsync(rank.p, h_rank.data(), h_rank.size() * sizeof(unsigned long long), hipMemcpyHostToDevice,
                            ch->stream));
      HIP_OK(hipMemsetAsync(prefix.p, 0, prefix.n * sizeof(unsigned long long), ch->stream));
      const size_t smem = static_cast<size_t>(T) * 16 * 64 * sizeof(unsigned);
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(wns::radix_hist_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
      DevBuf<unsigned long long> match, cand;
      DevBuf<unsigned> cand_n;
      match.alloc(static_cast<size_t>(T) * D);
      std::vector<unsigned long long> h_match(static_cast<size_t>(T) * D);
      for (int pass = 0; pass < 16; ++pass) {
        const int shift = 60 - 4 * pass, first = pass == 0 ? 1 : 0;
        HIP_OK(hipMemsetAsync(ghist.p, 0, ghist.n * sizeof(unsigned long long), ch->stream));
        hipLaunchKernelGGL(wns::radix_hist_kernel, dim3(chunks * tiles), dim3(wns::kBlock), smem, ch->stream, ch->view(),
                           T, shift, first, prefix.p, ghist.p, chains_per_block);
        hipLaunchKernelGGL(wns::radix_pick_kernel, dim3(col_blocks(static_cast<size_t>(T) * D)), dim3(wns::kBlock), 0,
                           ch->stream, D, T, shift, first, ghist.p, prefix.p, rank.p, match.p);
        if (pass >= 2 && pass < 15) {
          // few draws left in every bin?  then gather them once and finish there instead of passing over all the
          // draws another (15 - pass) times
          ch->down(match.p, h_match.data(), h_match.size());
          const unsigned long long most = *std::max_element(h_match.begin(), h_match.end());
          if (most <= static_cast<unsigned long long>(wns::kCandidateCap)) {
            cand.alloc(static_cast<size_t>(T) * D * wns::kCandidateCap);
            cand_n.alloc(static_cast<size_t>(T) * D);
            HIP_OK(hipMemsetAsync(cand_n.p, 0, cand_n.n * sizeof(unsigned), ch->stream));
            hipLaunchKernelGGL(wns::radix_collect_kernel, dim3(chunks * tiles), dim3(wns::kBlock), 0, ch->stream,
                               ch->view(), T, shift, prefix.p, wns::kCandidateCap, cand.p, cand_n.p, chains_per_block);
            hipLaunchKernelGGL(wns::radix_finish_kernel, dim3(T * D), dim3(64), 0, ch->stream, T * D, shift,
                               wns::kCandidateCap, cand.p, cand_n.p, prefix.p, rank.p);
            break;
          }
        }
      }
      HIP_OK(hipGetLastError());
      std::vector<unsigned long long> h_key(static_cast<size_t>(T) * D);
      ch->down(prefix.p, h_key.data(), h_key.size());
      for (int t = 0; t < T; ++t)
        for (int d = 0; d < D; ++d) stat[(g0 + t) * D + d] = wns::key_value(h_key[static_cast<size_t>(t) * D + d]);
    }
    for (size_t k = 0; k < num_probs; ++k)
      for (int d = 0; d < D; ++d) {
        const double lo = stat[(2 * k) * D + d], hi = stat[(2 * k + 1) * D + d];
        out[k * D + d] = lo + frac[k] * (hi - lo);  // :511
      }
  });
}

}  // extern "C"
