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
    const double wd = est_weight[2 * c], ws = est_weight[2 * c + 1];
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

static __global__ void sum_i64_kernel(const int64_t* v, int n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc += static_cast<unsigned long long>(v[i]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace wn
