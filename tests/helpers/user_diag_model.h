// TEST FIXTURE: a user's device model for the run-time route (walnuts_amd/models.py), written against the public
// interface only (walnuts_amd/csrc/wn_model_api.h).  Independent normals with per-coordinate variances -- the density
// of examples/examples.cpp:20-31 --, so that the oracle's diagonal-normal model can check it bit for bit: the same
// expressions in the same order, multiplying by 1/sigma^2 (inverted once on the host) as the built-in model does.
#pragma once
#include <stdexcept>

#include "wn_model_api.h"
namespace user {
struct MyDiagNormal {
  static constexpr bool kUsesParams = true;  // mp = sigma_sq (handed over), 1 / sigma_sq (on the device)
  static constexpr bool kElementwise = true;
  static constexpr bool kGradIsNegTheta = false;
  static constexpr bool kCheapGrad = true;
  __device__ __forceinline__ static double grad_elem(double th, double rs2) { return -th * rs2; }
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx&, const double (&th)[EPL], double (&g)[EPL], const double (&rs2)[EPL],
                                              Aux&, double& acc) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      g[j] = -th[j] * rs2[j];
      acc = Cx::mad(-0.5 * th[j] * th[j], rs2[j], acc);
    }
  }
  template <int EPL, class Cx>
  __device__ __forceinline__ static void grad(Cx&, const double (&th)[EPL], double (&g)[EPL], const double (&rs2)[EPL],
                                              Aux&) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) g[j] = -th[j] * rs2[j];
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return sum; }
  static void host_params(double* sigma_sq, int num_params) {
    for (int i = 0; i < num_params; ++i) {
      if (!(sigma_sq[i] > 0)) throw std::invalid_argument("sigma_sq must be positive");
      sigma_sq[i] = 1.0 / sigma_sq[i];
    }
  }
};
}  // namespace user
