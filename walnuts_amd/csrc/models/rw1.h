// models/rw1.h -- a device model added through the public model interface (wn_model_api.h), as a user would:
// the reference's AR(1) "random walk" example density, examples/examples.cpp:34-49,
//     p(y) = normal(y | 0, Sigma),  Sigma[i, j] = rho^|i - j|,  rho = 0.99,
// whose gradient couples NEIGHBOURING coordinates: grad[n] = -w[n] + rho * w[n + 1], w[n] = (y[n] - rho * y[n - 1]) / (1 - rho^2).
// Nothing about it is element-wise or cheap, so it exercises the general path of the kernels: the gradient vector is
// kept (two more register vectors per trajectory-end set), parked and reloaded with the span ends, and every
// evaluation needs a cross-lane exchange (cx.shift).
#pragma once

#include "../wn_model_api.h"

namespace wn {

struct Rw1Model {
  static constexpr bool kUsesParams = false;
  static constexpr bool kElementwise = false;
  static constexpr bool kGradIsNegTheta = false;
  static constexpr bool kCheapGrad = false;
  // geometry hint (optional): every evaluation exchanges neighbours across lanes and keeps four vectors per set.
  // Measured in round 6 (16 384 / 8 192 chains, ms per step; after the DPP neighbour shifts of round 5): one wavefront
  // per chain -- the default policy -- wins up to 1 024 dimensions (512: 0.75 against 0.87 at four per lane, 768: 1.69
  // against 1.95, 1 024: 1.80 against 2.01); four elements per lane on eight wavefronts up to 2 048 (1 500: 2.66
  // against 2.79 for (2,16), 2 048: 2.86 against 3.01); eight per lane on eight wavefronts up to 4 096 (3 000: 5.64
  // against 7.29 for (16,4), 4 096: 5.86 against 7.60 and 7.93 for (4,16)).  Beyond: the held streaming kernels.
  static constexpr int preferred_elems_per_lane(int num_params) {
    return num_params <= 1024 ? 0 : num_params <= 2048 ? 4 : num_params <= 4096 ? 8 : 0;
  }
  __device__ __forceinline__ static double grad_elem(double, double) { return 0.0; }
  struct Aux {};

  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx& cx, const double (&y)[EPL], double (&g)[EPL], const double (&)[EPL],
                                              Aux&, double& acc) {
    constexpr double rho = 0.99;
    const double sigma_sq = 1.0 - rho * rho;       // examples.cpp:37-38
    const double inv_sigma_sq = 1.0 / sigma_sq;
    double prev[EPL], next[EPL];
    cx.shift(y, prev, next);                        // y[n - 1] and y[n + 1] of every coordinate this lane owns
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const int n = cx.index(j);
      double ta, tb, gj;                            // the log-density term is ta * tb
      if (n == 0) {
        ta = tb = y[j];                             // logp = -0.5 * y[0] * y[0]; grad[0] -= y[0]
        gj = -y[j];
      } else {
        const double r = y[j] - rho * prev[j];      // examples.cpp:43-46
        const double w = r * inv_sigma_sq;
        ta = r;
        tb = w;
        gj = -w;
      }
      if (n + 1 < cx.dim()) {                       // grad[n] += rho * w[n + 1] (examples.cpp:47)
        const double r1 = next[j] - rho * y[j];
        gj = gj + rho * (r1 * inv_sigma_sq);
      }
      const bool in = cx.valid(j);
      g[j] = in ? gj : 0.0;
      acc = Cx::mad(in ? ta : 0.0, in ? tb : 0.0, acc);  // (fused when the engine runs with fused multiply-adds)
    }
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return -0.5 * sum; }

  // ---- streaming form (num_params > 8192): no sums over coordinates, but every coordinate needs its two neighbours
  // (kStreamHalo: the kernel hands them over as prev / next); the same expressions as eval() above
  static constexpr bool kStreamable = true;
  static constexpr int kStreamSums = 0;
  static constexpr bool kStreamHalo = true;
  template <class Cx>
  __device__ __forceinline__ static void stream_sums(Cx&, const double (&)[2], const double (&)[2], double (&)[1]) {}
  template <class Tab>
  __device__ __forceinline__ static void stream_aux(const double (&)[1], int, const Tab&, Aux&) {}
  template <class Cx>
  __device__ __forceinline__ static void stream_grad(Cx& cx, const double (&y)[2], const double (&prev)[2],
                                                     const double (&next)[2], const double (&)[2], double (&g)[2],
                                                     const Aux&) {
    constexpr double rho = 0.99;
    const double inv_sigma_sq = 1.0 / (1.0 - rho * rho);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = cx.index(j);
      double gj = n == 0 ? -y[j] : -((y[j] - rho * prev[j]) * inv_sigma_sq);
      if (n + 1 < cx.dim()) gj = gj + rho * ((next[j] - rho * y[j]) * inv_sigma_sq);
      g[j] = cx.valid(j) ? gj : 0.0;
    }
  }
  template <class Cx>
  __device__ __forceinline__ static void stream_logp(Cx& cx, const double (&y)[2], const double (&prev)[2],
                                                     const double (&)[2], const double (&)[2], const Aux&, double& acc) {
    constexpr double rho = 0.99;
    const double inv_sigma_sq = 1.0 / (1.0 - rho * rho);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = cx.index(j);
      double ta, tb;
      if (n == 0) {
        ta = tb = y[j];
      } else {
        const double r = y[j] - rho * prev[j];
        ta = r;
        tb = r * inv_sigma_sq;
      }
      const bool in = cx.valid(j);
      acc = Cx::mad(in ? ta : 0.0, in ? tb : 0.0, acc);
    }
  }
};

}  // namespace wn
