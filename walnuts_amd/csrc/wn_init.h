// wn_init.h -- per-chain initialisation kernels (InitConfigBuilder, config.hpp:195-484):
//   positions(rng, scale)          config.hpp:258-268   -> counter stream kStreamInitPos
//   masses(logp_grad, smoothing)   config.hpp:360-370   mass = (1-s)*|grad| + s
//   adapt_step_build(rng, logp)    config.hpp:470-476 + util.hpp:242-303 (leapfrog_error, adapt_step)
// and the small element-wise kernels that turn an InitConfig into AdaptiveWalnuts state
// (adaptive_walnuts.hpp:54-62,205-223) and freeze it into a WalnutsSampler (:263-271).
#pragma once

#include "wn_traj.h"

namespace wn {

struct InitParams {
  int32_t num_chains, dim, dim_padded;
  int32_t do_positions, do_masses, do_step;
  double* theta;      // [C][Dp]
  double* mass;       // [C][Dp] (padding 1.0)
  double* step_init;  // [C]
  int64_t* grad_evals;
  const double* model_params;
  const double* z_buf;  // nullable [C][Dp]: host-generated normals for the step search (exact libstdc++ stream)
  double scale, smoothing;
  uint64_t pos_seed, step_seed;
  uint32_t pos_chain_offset, step_chain_offset;
};

template <class Model, int NW, int EPL>
__global__ __launch_bounds__(64 * NW) void init_kernel(const InitParams Q) {
  WN_DYN_SMEM(smem);
  using T = Traj<Model, NW, EPL, true>;
  Params P{};  // only the fields the model context and reductions read
  P.num_chains = Q.num_chains;
  P.dim = Q.dim;
  P.dim_padded = Q.dim_padded;
  P.model_params = Q.model_params;
  WN_LDS double* base = (WN_LDS double*)smem;
  WN_LDS typename T::Meta* meta = (WN_LDS typename T::Meta*)(base + (threadIdx.x >> 6) * kMetaDoubles);
  WN_LDS double* red = base + NW * kMetaDoubles;
  WN_LDS double* bcast = red + 4 * NW;
  T t(P, base, meta, red, bcast, nullptr);
  constexpr int L = T::L;
  constexpr int NP = T::NP;

  for (int chain = blockIdx.x; chain < Q.num_chains; chain += gridDim.x) {
    const long long row = static_cast<long long>(chain) * Q.dim_padded;
    t.n_grad = 0;
    if (Model::kUsesParams) t.vload(Q.model_params, t.mp);
    if (Q.do_positions) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        double z0, z1;
        wnd::stream_normal_pair(Q.pos_seed, Q.pos_chain_offset + chain, 0u, wnd::kStreamInitPos,
                                static_cast<uint32_t>(k * L + t.tid), z0, z1);
        t.th[2 * k] = t.valid(2 * k) ? z0 * Q.scale : 0.0;
        t.th[2 * k + 1] = t.valid(2 * k + 1) ? z1 * Q.scale : 0.0;
      }
      t.vstore(Q.theta + row, t.th);
    } else {
      t.vload(Q.theta + row, t.th);
    }
    double mass[EPL];
    if (Q.do_masses) {
      (void)t.model_eval();
#pragma unroll
      for (int j = 0; j < EPL; ++j) mass[j] = t.valid(j) ? (1 - Q.smoothing) * fabs(t.g[j]) + Q.smoothing : 1.0;
      t.vstore(Q.mass + row, mass);
    } else {
      t.vload(Q.mass + row, mass);
    }
    if (Q.do_step) {
      // util.hpp:285-303
      double rho0[EPL], th_keep[EPL];
#pragma unroll
      for (int j = 0; j < EPL; ++j) t.im[j] = 1.0 / mass[j];
      if (Q.z_buf != nullptr) {
        double z[EPL];
        t.vload(Q.z_buf + row, z);
#pragma unroll
        for (int j = 0; j < EPL; ++j) rho0[j] = t.valid(j) ? z[j] * __builtin_sqrt(mass[j]) : 0.0;
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          double z0, z1;
          wnd::stream_normal_pair(Q.step_seed, Q.step_chain_offset + chain, 0u, wnd::kStreamInitStep,
                                  static_cast<uint32_t>(k * L + t.tid), z0, z1);
          rho0[2 * k] = t.valid(2 * k) ? z0 * __builtin_sqrt(mass[2 * k]) : 0.0;
          rho0[2 * k + 1] = t.valid(2 * k + 1) ? z1 * __builtin_sqrt(mass[2 * k + 1]) : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < EPL; ++j) th_keep[j] = t.th[j];
      double step = uni(Q.step_init[chain]);
      const double log09 = wnd::dlog(0.9), log06 = wnd::dlog(0.6), rt = __builtin_sqrt(0.5);
      // util.hpp:242-259
      auto leapfrog_error = [&](double h) -> double {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          t.th[j] = th_keep[j];
          t.rh[j] = rho0[j];
        }
        double part = t.model_eval();
        double lp, lj0, lj1;
        t.energy(part, lp, lj0);
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.rh[j] = t.rh[j] + 0.5 * h * t.g[j];
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.th[j] = t.th[j] + h * (t.im[j] * t.rh[j]);
        part = t.model_eval();
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.rh[j] = t.rh[j] + 0.5 * h * t.g[j];
        t.energy(part, lp, lj1);
        return lj1 - lj0;
      };
      while (leapfrog_error(step) > log09) step *= 2;
      while (leapfrog_error(step) < log06) step *= rt;
      if (t.tid == 0) Q.step_init[chain] = step;
    }
    if (t.tid == 0) Q.grad_evals[chain] += t.n_grad;
  }
}

}  // namespace wn
