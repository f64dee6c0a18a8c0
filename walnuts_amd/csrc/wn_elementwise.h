// wn_elementwise.h -- element-wise kernels around the transition kernel (included by wn_engine.hip only).
#pragma once

#include "wn_devmath.h"
#include "wn_hip.h"

namespace wn {

// AdaptiveWalnuts construction (adaptive_walnuts.hpp:205-223): estimator planes from the
// init mass (:54-62), Adam on log step (adam.hpp:48-62), min-micro handler (:127-132)
static __global__ void begin_warmup_kernel(int C, int Dp, double count, const double* mass, double* draw_mean,
                                    double* draw_ssd, double* score_mean, double* score_ssd, double* est_weight,
                                    const double* step_init, double* adam, double* mm_state) {
  const long long n = static_cast<long long>(C) * Dp;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const double m = mass[i];
    draw_mean[i] = 0.0;
    score_mean[i] = 0.0;
    score_ssd[i] = count * m;
    draw_ssd[i] = count * (1.0 / m);
    if (i < C) {
      est_weight[2 * i] = count;
      est_weight[2 * i + 1] = count;
      adam[6 * i + 0] = wnd::dlog(step_init[i]);
      adam[6 * i + 1] = 0.0;
      adam[6 * i + 2] = 0.0;
      adam[6 * i + 3] = 0.0;
      adam[6 * i + 4] = 1.0;
      adam[6 * i + 5] = 1.0;
      mm_state[2 * i] = 2.0;
      mm_state[2 * i + 1] = 1.0;
    }
  }
}

// AdaptiveWalnuts::sampler() (adaptive_walnuts.hpp:263-271)
static __global__ void freeze_kernel(int C, int Dp, const double* draw_ssd, const double* score_ssd,
                              const double* est_weight, const double* adam, const double* mm_state,
                              double macro_target, int cfg_min_micro, double* inv_mass, double* chol_mass,
                              double* step_size, int* min_micro) {
  const long long n = static_cast<long long>(C) * Dp;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long c = i / Dp;
    const wnd::SharedDivisor wd(est_weight[2 * c]), ws(est_weight[2 * c + 1]);  // (as the warmup transitions divide)
    const double im = __builtin_sqrt((draw_ssd[i] / wd) / (score_ssd[i] / ws));
    inv_mass[i] = im;
    chol_mass[i] = 1.0 / __builtin_sqrt(im);  // walnuts.hpp:647
    if (i < C) {
      step_size[i] = wnd::dexp(adam[6 * i]);
      const double mean_micro = mm_state[2 * i] / mm_state[2 * i + 1];
      const long long est = static_cast<long long>(__builtin_round(mean_micro / macro_target));
      min_micro[i] = static_cast<int>(est > cfg_min_micro ? est : cfg_min_micro);
    }
  }
}

// AdaptiveWalnuts::inv_mass() during warmup (adaptive_walnuts.hpp:89-94, :297-299): the estimate the NEXT warmup
// transition will integrate with, written to the (otherwise idle before freeze) inverse-mass plane
static __global__ void inv_mass_estimate_kernel(int C, int Dp, const double* draw_ssd, const double* score_ssd,
                                                const double* est_weight, double* inv_mass) {
  const long long n = static_cast<long long>(C) * Dp;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long c = i / Dp;
    inv_mass[i] = __builtin_sqrt((draw_ssd[i] / wnd::SharedDivisor(est_weight[2 * c])) / (score_ssd[i] / wnd::SharedDivisor(est_weight[2 * c + 1])));
  }
}

// ---- cross-chain monitors (the reference's controller loops) --------------------------------------
// Deterministic two-stage sums over chains, independent of the launch geometry: stage 1 adds every RUN of kMonitorRun
// consecutive chains left to right (one thread per run), stage 2 (one thread) adds the run totals left to right.  Up
// to kMonitorRun chains that IS the left-to-right sum; beyond, the test suite's CPU restatement groups the same way (its
// chain_sum), so the statistics are compared bit for bit.
constexpr int kMonitorRun = 256;
inline int monitor_runs(int n) { return (n + kMonitorRun - 1) / kMonitorRun; }

template <int K, class F>
static __device__ void run_partial_sums(int n, F f, double* partial /*[runs][K]*/) {
  const int runs = (n + kMonitorRun - 1) / kMonitorRun;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < runs; r += gridDim.x * blockDim.x) {
    double acc[K];
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    const int lo = r * kMonitorRun, hi = lo + kMonitorRun < n ? lo + kMonitorRun : n;
    for (int i = lo; i < hi; ++i) f(i, acc);
    for (int k = 0; k < K; ++k) partial[r * K + k] = acc[k];
  }
}
template <int K>
static __global__ void finish_sums_kernel(const double* partial, int blocks, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int b = 0; b < blocks; ++b) s += partial[b * K + k];
      out[k] = s;
    }
  }
}
// sampling monitor, sampler.hpp:132-145: sums of the per-chain lp means and sample variances
static __global__ void lp_sums_kernel(int C, const double* lp_stats, double* partial) {
  run_partial_sums<2>(C, [&](int c, double* acc) {
    const double n = lp_stats[3 * c], mean = lp_stats[3 * c + 1], m2 = lp_stats[3 * c + 2];
    acc[0] += mean;
    acc[1] += n > 1 ? m2 / (n - 1) : __builtin_nan("");  // WelfordAccumulator::sample_variance
  }, partial);
}
static __global__ void lp_sqdev_kernel(int C, const double* lp_stats, double mu, double* partial) {
  run_partial_sums<1>(C, [&](int c, double* acc) {
    const double d = lp_stats[3 * c + 1] - mu;
    acc[0] += d * d;
  }, partial);
}
// warmup monitor, adapt.hpp:193-221.  log step per chain from Adam's theta; log mass = -log(inv_mass).
static __global__ void log_step_sum_kernel(int C, const double* adam, double* partial) {
  run_partial_sums<1>(C, [&](int c, double* acc) { acc[0] += wnd::dlog(wnd::dexp(adam[6 * c])); }, partial);
}
// column sums over chains of log mass: thread per column, chains in order (coalesced rows)
static __global__ void log_mass_colsum_kernel(int C, int D, int Dp, const double* draw_ssd, const double* score_ssd,
                                              const double* est_weight, double* colsum) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  double s = 0.0;
  for (int c = 0; c < C; ++c) {
    const long long i = static_cast<long long>(c) * Dp + d;
    const double im = __builtin_sqrt((draw_ssd[i] / wnd::SharedDivisor(est_weight[2 * c])) / (score_ssd[i] / wnd::SharedDivisor(est_weight[2 * c + 1])));
    s += -wnd::dlog(im);
  }
  colsum[d] = s;
}
// InitConfigBuilder::masses(..., average_masses = true), config.hpp:371-380: every chain's mass becomes the
// geometric mean over chains.  Thread per column; chains summed in order.
static __global__ void mass_log_colsum_kernel(int C, int D, int Dp, const double* mass, double* colsum) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  double s = 0.0;
  for (int c = 0; c < C; ++c) s += wnd::dlog(mass[static_cast<long long>(c) * Dp + d]);
  colsum[d] = wnd::dexp(s / static_cast<double>(C));
}
static __global__ void mass_broadcast_kernel(int C, int D, int Dp, const double* geom_mean, double* mass) {
  const long long n = static_cast<long long>(C) * Dp;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int d = static_cast<int>(i % Dp);
    if (d < D) mass[i] = geom_mean[d];
  }
}
// per-chain l2_rel_diff(mass_m, geom_mean_mass) (util.hpp:379-382) and rel diff of the step; block per chain
static __global__ void warmup_spread_kernel(int C, int D, int Dp, const double* draw_ssd, const double* score_ssd,
                                            const double* est_weight, const double* adam, const double* colsum,
                                            double n_chains /*chains behind colsum: all ranks'*/,
                                            double mean_log_step, double* rel_mass, double* rel_step) {
  __shared__ double sh[256];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const long long i = static_cast<long long>(c) * Dp + d;
    const double im = __builtin_sqrt((draw_ssd[i] / wnd::SharedDivisor(est_weight[2 * c])) / (score_ssd[i] / wnd::SharedDivisor(est_weight[2 * c + 1])));
    const double mass = wnd::dexp(-wnd::dlog(im));                       // snap.mass, adapt.hpp:141
    const double gm = wnd::dexp(colsum[d] / n_chains);                   // geom_mean_mass, adapt.hpp:203-205
    const double r = (mass - gm) / gm;
    acc += r * r;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rel_mass[c] = __builtin_sqrt(sh[0]);
    const double gms = wnd::dexp(mean_log_step);
    rel_step[c] = (wnd::dexp(wnd::dlog(wnd::dexp(adam[6 * c]))) - gms) / gms;  // adapt.hpp:213-215
  }
}
static __global__ void max2_kernel(int C, const double* a, const double* b, double* out) {
  __shared__ double sa[256], sb[256];
  double ma = 0.0, mb = 0.0;  // std::fmax from 0.0, adapt.hpp:208-216
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    ma = fmax(ma, a[i]);
    mb = fmax(mb, b[i]);
  }
  sa[threadIdx.x] = ma;
  sb[threadIdx.x] = mb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) {
      sa[threadIdx.x] = fmax(sa[threadIdx.x], sa[threadIdx.x + s]);
      sb[threadIdx.x] = fmax(sb[threadIdx.x], sb[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sa[0];
    out[1] = sb[0];
  }
}

static __global__ void fill_kernel(double* p, long long n, double v) {
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    p[i] = v;
}
static __global__ void sum_i64_kernel(const int64_t* v, int n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc += static_cast<unsigned long long>(v[i]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace wn
