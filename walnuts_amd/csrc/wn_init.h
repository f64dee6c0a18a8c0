// wn_init.h -- per-chain initialisation kernels (InitConfigBuilder, config.hpp:195-484):
//   positions(rng, scale)          config.hpp:258-268   -> counter stream kStreamInitPos
//   masses(logp_grad, smoothing)   config.hpp:360-370   mass = (1-s)*|grad| + s
//   adapt_step_build(rng, logp)    config.hpp:470-476 + util.hpp:242-303 (leapfrog_error, adapt_step)
// and the small element-wise kernels that turn an InitConfig into AdaptiveWalnuts state
// (adaptive_walnuts.hpp:54-62,205-223) and freeze it into a WalnutsSampler (:263-271).
#pragma once

#include "wn_chip.h"

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
  double* scratch;      // streaming kernels: one Dp-vector per workgroup at scratch + blockIdx*scratch_stride
  int64_t scratch_stride;
  double scale, smoothing;
  uint64_t pos_seed, step_seed;
  uint32_t pos_chain_offset, step_chain_offset;
};

template <class Model, int NW, int EPL>
__global__ __launch_bounds__(64 * NW) void init_kernel(const InitParams Q) {
  WN_DYN_SMEM(smem);
  using T = TrajChip<Model, NW, EPL>;  // the register kernels' vector helpers (unfused arithmetic); set 0 is the working state
  Params P{};  // only the fields the model context and reductions read
  P.num_chains = Q.num_chains;
  P.dim = Q.dim;
  P.dim_padded = Q.dim_padded;
  P.model_params = Q.model_params;
  WN_LDS double* base = (WN_LDS double*)smem;
  WN_LDS typename T::Meta* meta = (WN_LDS typename T::Meta*)(base + wave_in_workgroup() * kMetaDoubles);
  WN_LDS double* red = base + NW * kMetaDoubles;
  WN_LDS double* bcast = red + kRedDoubles(NW);
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
        t.th[0][2 * k] = t.valid(2 * k) ? z0 * Q.scale : 0.0;
        t.th[0][2 * k + 1] = t.valid(2 * k + 1) ? z1 * Q.scale : 0.0;
      }
      t.vstore(Q.theta + row, t.th[0]);
    } else {
      t.vload(Q.theta + row, t.th[0]);
    }
    double mass[EPL];
    if (Q.do_masses) {
      (void)t.template model_eval<0>();
#pragma unroll
      for (int j = 0; j < EPL; ++j) mass[j] = t.valid(j) ? (1 - Q.smoothing) * fabs(t.template G<0>(j)) + Q.smoothing : 1.0;
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
      for (int j = 0; j < EPL; ++j) th_keep[j] = t.th[0][j];
      double step = uni(Q.step_init[chain]);
      const double log09 = wnd::dlog(0.9), log06 = wnd::dlog(0.6), rt = __builtin_sqrt(0.5);
      // util.hpp:242-259
      auto leapfrog_error = [&](double h) -> double {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          t.th[0][j] = th_keep[j];
          t.rh[0][j] = rho0[j];
        }
        double part = t.template model_eval<0>();
        double lp, lj0, lj1;
        double ke;
        t.template energy_partials<0>(part, ke);
        t.sum2(part, ke);
        t.finish_energy(part, ke, lp, lj0);
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.rh[0][j] = t.rh[0][j] + 0.5 * h * t.template G<0>(j);
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.th[0][j] = t.th[0][j] + h * (t.im[j] * t.rh[0][j]);
        part = t.template model_eval<0>();
#pragma unroll
        for (int j = 0; j < EPL; ++j) t.rh[0][j] = t.rh[0][j] + 0.5 * h * t.template G<0>(j);
        t.template energy_partials<0>(part, ke);
        t.sum2(part, ke);
        t.finish_energy(part, ke, lp, lj1);
        return lj1 - lj0;
      };
      while (leapfrog_error(step) > log09) step *= 2;
      while (leapfrog_error(step) < log06) step *= rt;
      if (t.tid == 0) Q.step_init[chain] = step;
    }
    if (t.tid == 0) Q.grad_evals[chain] += t.n_grad;
  }
}

// The same three InitConfigBuilder steps for the streaming backend (vectors in HBM).  leapfrog_error
// (util.hpp:242-259) is one fused pass per probe: both energies of the probe come out of the same sweep.
template <class Model, int NW>
__global__ __launch_bounds__(64 * NW) void init_kernel_mem(const InitParams Q) {
  WN_DYN_SMEM(smem);
  using T = TrajMem<Model, NW>;
  Params P{};
  P.num_chains = Q.num_chains;
  P.dim = Q.dim;
  P.dim_padded = Q.dim_padded;
  P.model_params = Q.model_params;
  WN_LDS double* base = (WN_LDS double*)smem;
  WN_LDS typename T::Meta* meta = (WN_LDS typename T::Meta*)(base + wave_in_workgroup() * kMetaDoubles);
  WN_LDS double* red = base + NW * kMetaDoubles;
  WN_LDS double* bcast = red + kRedDoubles(NW);
  double* rho0 = Q.scratch + static_cast<long long>(blockIdx.x) * Q.scratch_stride;
  T t(P, base, meta, red, bcast, rho0);
  constexpr int L = T::L;
  const int tiles = Q.dim_padded / (2 * L);
  typename Model::Aux aux;

  auto load_mp = [&](int o, double (&mp2)[2]) {
    mp2[0] = mp2[1] = 1.0;
    if (Model::kUsesParams) {
      const v2f64 p0 = T::ld(Q.model_params + o);
      mp2[0] = p0[0];
      mp2[1] = p0[1];
    }
  };

  for (int chain = blockIdx.x; chain < Q.num_chains; chain += gridDim.x) {
    const long long row = static_cast<long long>(chain) * Q.dim_padded;
    long long n_grad = 0;
    if (Q.do_positions) {
      for (int k = 0; k < tiles; ++k) {
        const int o = t.pair_offset(k);
        double z0, z1;
        wnd::stream_normal_pair(Q.pos_seed, Q.pos_chain_offset + chain, 0u, wnd::kStreamInitPos,
                                static_cast<uint32_t>(k * L + t.tid), z0, z1);
        T::st(Q.theta + row + o, o < Q.dim ? z0 * Q.scale : 0.0, o + 1 < Q.dim ? z1 * Q.scale : 0.0);
      }
    }
    if (Q.do_masses) {
      typename Model::Aux aux_here{};
      if constexpr (T::kTwoPass) t.aux_of(Q.theta + row, aux_here);
      for (int k = 0; k < tiles; ++k) {
        const int o = t.pair_offset(k);
        const v2f64 t0 = T::ld(Q.theta + row + o);
        double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2];
        load_mp(o, mp2);
        typename T::TileCx cx{o, Q.dim};
        if constexpr (T::kTwoPass) {
          double prev[2], next[2];
          t.halo(Q.theta + row, o, t0, prev, next);
          Model::stream_grad(cx, th2, prev, next, mp2, g2, aux_here);
        } else {
          double unused = 0.0;
          Model::eval(cx, th2, g2, mp2, aux, unused);
        }
        T::st(Q.mass + row + o, o < Q.dim ? (1 - Q.smoothing) * fabs(g2[0]) + Q.smoothing : 1.0,
              o + 1 < Q.dim ? (1 - Q.smoothing) * fabs(g2[1]) + Q.smoothing : 1.0);
      }
      ++n_grad;
    }
    if (Q.do_step) {
      // momentum rho0 = z .* sqrt(mass) and the energy at the start point (util.hpp:248-249, 289-293)
      double lp0 = 0.0, ke0 = 0.0;
      typename Model::Aux aux0{};
      if constexpr (T::kTwoPass) t.aux_of(Q.theta + row, aux0);
      for (int k = 0; k < tiles; ++k) {
        const int o = t.pair_offset(k);
        const v2f64 t0 = T::ld(Q.theta + row + o), m0 = T::ld(Q.mass + row + o);
        double z2[2];
        if (Q.z_buf != nullptr) {
          const v2f64 zz = T::ld(Q.z_buf + row + o);
          z2[0] = zz[0];
          z2[1] = zz[1];
        } else {
          wnd::stream_normal_pair(Q.step_seed, Q.step_chain_offset + chain, 0u, wnd::kStreamInitStep,
                                  static_cast<uint32_t>(k * L + t.tid), z2[0], z2[1]);
        }
        double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2], r2[2];
        load_mp(o, mp2);
        typename T::TileCx cx{o, Q.dim};
        if constexpr (T::kTwoPass) {
          double prev[2], next[2];
          t.halo(Q.theta + row, o, t0, prev, next);
          Model::stream_logp(cx, th2, prev, next, mp2, aux0, lp0);
          (void)g2;
        } else {
          Model::eval(cx, th2, g2, mp2, aux, lp0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          r2[j] = (o + j < Q.dim) ? z2[j] * __builtin_sqrt(m0[j]) : 0.0;
          ke0 += (1.0 / m0[j]) * (r2[j] * r2[j]);
        }
        T::st(rho0 + o, r2[0], r2[1]);
      }
      t.sum2(lp0, ke0);
      if constexpr (T::kTwoPass) aux = aux0;
      const double lj0 = Model::finish(lp0, aux, Q.dim) + (-0.5 * ke0);
      double step = uni(Q.step_init[chain]);
      const double log09 = wnd::dlog(0.9), log06 = wnd::dlog(0.6), rt = __builtin_sqrt(0.5);
      auto leapfrog_error = [&](double h) -> double {
        double lp1 = 0.0, ke1 = 0.0;
        if constexpr (T::kTwoPass) {
          // the probe step in two passes (the gradient at the new position needs its sums / its neighbours): the
          // new position and the half-kicked momentum go to two scratch vectors of this workgroup
          double* th_new = rho0 + Q.dim_padded;
          double* r_new = rho0 + 2 * static_cast<long long>(Q.dim_padded);
          double sums[T::ST::kSums];
          for (int i = 0; i < T::ST::kSums; ++i) sums[i] = 0.0;
          for (int k = 0; k < tiles; ++k) {
            const int o = t.pair_offset(k);
            const v2f64 t0 = T::ld(Q.theta + row + o), m0 = T::ld(Q.mass + row + o), r0 = T::ld(rho0 + o);
            double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2], r2[2] = {r0[0], r0[1]}, prev[2], next[2];
            load_mp(o, mp2);
            t.halo(Q.theta + row, o, t0, prev, next);
            typename T::TileCx cx{o, Q.dim};
            Model::stream_grad(cx, th2, prev, next, mp2, g2, aux0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              r2[j] = r2[j] + 0.5 * h * g2[j];
              th2[j] = th2[j] + h * ((1.0 / m0[j]) * r2[j]);
            }
            if (T::ST::kHasSums) Model::stream_sums(cx, th2, mp2, sums);
            T::st(th_new + o, th2[0], th2[1]);
            T::st(r_new + o, r2[0], r2[1]);
          }
          typename Model::Aux aux1{};
          t.finish_sums(sums, aux1);
          __syncthreads();
          for (int k = 0; k < tiles; ++k) {
            const int o = t.pair_offset(k);
            const v2f64 t1 = T::ld(th_new + o), m0 = T::ld(Q.mass + row + o), r1 = T::ld(r_new + o);
            const double th2[2] = {t1[0], t1[1]};
            double g2[2], mp2[2], r2[2] = {r1[0], r1[1]}, prev[2], next[2];
            load_mp(o, mp2);
            t.halo(th_new, o, t1, prev, next);
            typename T::TileCx cx{o, Q.dim};
            Model::stream_grad(cx, th2, prev, next, mp2, g2, aux1);
            Model::stream_logp(cx, th2, prev, next, mp2, aux1, lp1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              r2[j] = r2[j] + 0.5 * h * g2[j];
              ke1 += (1.0 / m0[j]) * (r2[j] * r2[j]);
            }
          }
          n_grad += 2;
          t.sum2(lp1, ke1);
          __syncthreads();  // the scratch vectors are free for the next probe
          return (Model::finish(lp1, aux1, Q.dim) + (-0.5 * ke1)) - lj0;
        } else {
        for (int k = 0; k < tiles; ++k) {
          const int o = t.pair_offset(k);
          const v2f64 t0 = T::ld(Q.theta + row + o), m0 = T::ld(Q.mass + row + o), r0 = T::ld(rho0 + o);
          double th2[2] = {t0[0], t0[1]}, g2[2], mp2[2], r2[2] = {r0[0], r0[1]}, im2[2];
          load_mp(o, mp2);
          typename T::TileCx cx{o, Q.dim};
          double unused = 0.0;
          Model::eval(cx, th2, g2, mp2, aux, unused);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            im2[j] = 1.0 / m0[j];
            r2[j] = r2[j] + 0.5 * h * g2[j];
            th2[j] = th2[j] + h * (im2[j] * r2[j]);
          }
          Model::eval(cx, th2, g2, mp2, aux, lp1);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            r2[j] = r2[j] + 0.5 * h * g2[j];
            ke1 += im2[j] * (r2[j] * r2[j]);
          }
        }
        n_grad += 2;
        t.sum2(lp1, ke1);
        return (Model::finish(lp1, aux, Q.dim) + (-0.5 * ke1)) - lj0;
        }
      };
      while (leapfrog_error(step) > log09) step *= 2;
      while (leapfrog_error(step) < log06) step *= rt;
      if (t.tid == 0) Q.step_init[chain] = step;
    }
    if (t.tid == 0) Q.grad_evals[chain] += n_grad;
  }
}

}  // namespace wn
