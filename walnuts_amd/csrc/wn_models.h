// wn_models.h -- the built-in device models: what a target density looks like to the kernels.
//
// A device model is the LogpGrad contract of the reference (concepts.hpp:258-262, C form LOGP_CFUNC
// walnutpy.cpp:131-132: f(theta) -> (logp, grad)) as a struct of static device functions that the transition
// kernels are instantiated with, so that the gradient is compiled INTO the trajectory loop.  The contract is spelled
// out in wn_model_api.h; models of your own go into their own header next to a five-line .hip file
// ("Adding a device model") -- nothing in this file or in the kernels has to change.
#pragma once

#include <cmath>
#include <stdexcept>

#include "wn_devmath.h"
#include "wn_hip.h"
#include "wn_params.h"

namespace wn {

// ---- target densities (device form of the LogpGrad contract, concepts.hpp:258-262) ----
// eval():   writes grad for the lane's elements and ADDS the lane's log-density terms, in index order,
//           to `acc` (the running per-lane partial); may reduce internally through cx.
// finish(): turns the reduced sum into logp.
struct StdNormalModel {  // examples/walnutpie_api.cpp:37-41
  static constexpr int kKind = kStdNormal;
  static constexpr bool kUsesParams = false;
  static constexpr bool kElementwise = true;
  // grad = -theta: the register kernels carry no gradient vector at all (a sign modifier on theta at every use)
  static constexpr bool kGradIsNegTheta = true;
  static constexpr bool kCheapGrad = true;
  __device__ __forceinline__ static double grad_elem(double th, double) { return -th; }
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx&, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&)[EPL], Aux&, double& acc) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      g[j] = -th[j];
      acc = Cx::mad(th[j], th[j], acc);
    }
  }
  // element-wise models: the gradient alone, the same expression eval() uses (so the same bits)
  template <int EPL, class Cx>
  __device__ __forceinline__ static void grad(Cx&, const double (&th)[EPL], double (&g)[EPL], const double (&)[EPL],
                                              Aux&) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) g[j] = -th[j];
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return -0.5 * sum; }
};

// The device receives 1/sigma_sq (rounded once on the host, wn_engine_create) and multiplies where the reference's
// example divides: (-0.5 x x) * (1/s2) and -x * (1/s2) -- within an ulp of the quotients, a third of the instructions.
struct DiagNormalModel {  // examples/examples.cpp:20-31, params = 1 / sigma_sq
  static constexpr int kKind = kDiagNormal;
  static constexpr bool kUsesParams = true;
  static constexpr bool kElementwise = true;
  static constexpr bool kGradIsNegTheta = false;
  // one multiply per element: cheaper to recompute at each use than to keep, park and reload a gradient vector
  static constexpr bool kCheapGrad = true;
  __device__ __forceinline__ static double grad_elem(double th, double rs2) { return -th * rs2; }
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx&, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&rs2)[EPL], Aux&, double& acc) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      g[j] = -th[j] * rs2[j];
      acc = Cx::mad(-0.5 * th[j] * th[j], rs2[j], acc);
    }
  }
  template <int EPL, class Cx>
  __device__ __forceinline__ static void grad(Cx&, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&rs2)[EPL], Aux&) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) g[j] = -th[j] * rs2[j];
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return sum; }
  // host side, before the parameter vector is uploaded: sigma_sq -> 1 / sigma_sq, rounded once (an fp64 division costs
  // about ten multiply-adds on the device; DESIGN.md "differs on purpose")
  static void host_params(double* sigma_sq, int num_params) {
    for (int i = 0; i < num_params; ++i) {
      if (!(sigma_sq[i] > 0) || !std::isfinite(sigma_sq[i])) throw std::invalid_argument("sigma_sq must be positive and finite");
      sigma_sq[i] = 1.0 / sigma_sq[i];
    }
  }
};

// The device multiplies by the (once-rounded) reciprocals 1/18 and 1/9 where the definition divides: within an ulp of
// the quotients, one instruction instead of the ~25 of a correctly rounded fp64 division, at every gradient evaluation
// and every energy (as the diagonal normal does with 1/sigma^2; the oracle's device-order mode does the same, its
// reference-order mode divides).
struct FunnelModel {  // Neal's funnel, SURVEY.md §8d cfg3 (not in the reference)
  static constexpr double kInv18 = 1.0 / 18.0, kInv9 = 1.0 / 9.0;
  static constexpr int kKind = kFunnel;
  static constexpr bool kUsesParams = false;
  static constexpr bool kElementwise = false;  // the gradient needs sum(x^2): register backend only
  static constexpr bool kGradIsNegTheta = false;
  static constexpr bool kCheapGrad = false;
  // geometry hint: with the gradient carried (three vectors per set) four wavefronts at sixteen elements per lane lose
  // 4 % to eight at eight between 2 049 and 4 096 dimensions (profiles/r06/mid_dimensions.txt); up to 2 048 the default
  // policy's (2,16) wins by 37 %
  static constexpr int preferred_elems_per_lane(int num_params) { return (num_params > 2048 && num_params <= 4096) ? 8 : 0; }
  __device__ __forceinline__ static double grad_elem(double, double) { return 0.0; }
  struct Aux {
    double v, S, hev, ev;
  };
  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx& cx, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&)[EPL], Aux& aux, double&) {
    const double v = cx.element0(th[0]);
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const double x = (cx.index(j) == 0) ? 0.0 : th[j];
      sp = Cx::mad(x, x, sp);
    }
    const double S = cx.sum1(sp);
    const double ev = wnd::dexp(-v, cx.uniform_tab());
    const double hd = 0.5 * static_cast<double>(cx.dim() - 1);
    const double hev = 0.5 * ev;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      double gj = -(th[j] * ev);
      if (cx.index(j) == 0) gj = ((-v * kInv9) + hev * S) - hd;
      g[j] = cx.valid(j) ? gj : 0.0;
    }
    aux.v = v;
    aux.S = S;
    aux.hev = hev;
    aux.ev = ev;
  }
  // ---- streaming form (num_params > 8192, wn_model_api.h "Streaming a model whose gradient is not element-wise"):
  // the gradient needs two sums over the coordinates -- sum_{i>=1} x_i^2 and x_0 itself (a sum whose other terms are
  // exact zeros) --, taken in one pass over the vector; the same expressions as eval() above, hence the same bits
  static constexpr bool kStreamable = true;
  static constexpr int kStreamSums = 2;
  static constexpr bool kStreamHalo = false;
  template <class Cx>
  __device__ __forceinline__ static void stream_sums(Cx& cx, const double (&th)[2], const double (&)[2],
                                                     double (&sums)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool first = cx.index(j) == 0;
      const double x = first ? 0.0 : th[j];
      sums[0] = Cx::mad(x, x, sums[0]);
      sums[1] += first ? th[j] : 0.0;
    }
  }
  template <class Tab>
  __device__ __forceinline__ static void stream_aux(const double (&sums)[2], int, const Tab& tab, Aux& aux) {
    aux.S = sums[0];
    aux.v = sums[1];
    aux.ev = wnd::dexp(-aux.v, tab);
    aux.hev = 0.5 * aux.ev;
  }
  template <class Cx>
  __device__ __forceinline__ static void stream_grad(Cx& cx, const double (&th)[2], const double (&)[2],
                                                     const double (&)[2], const double (&)[2], double (&g)[2],
                                                     const Aux& aux) {
    const double hd = 0.5 * static_cast<double>(cx.dim() - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double gj = -(th[j] * aux.ev);
      if (cx.index(j) == 0) gj = ((-aux.v * kInv9) + aux.hev * aux.S) - hd;
      g[j] = cx.valid(j) ? gj : 0.0;
    }
  }
  template <class Cx>
  __device__ __forceinline__ static void stream_logp(Cx&, const double (&)[2], const double (&)[2], const double (&)[2],
                                                     const double (&)[2], const Aux&, double&) {}  // finish() has it all
  __device__ __forceinline__ static double finish(double, const Aux& a, int D) {
    const double hd = 0.5 * static_cast<double>(D - 1);
    return ((-(a.v * a.v) * kInv18) - a.hev * a.S) - hd * a.v;
  }
  static void validate(int num_params) {
    if (num_params < 2) throw std::invalid_argument("funnel needs num_params >= 2");
  }
};

}  // namespace wn
