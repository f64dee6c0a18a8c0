// wn_model_api.h -- the interface a device model implements.
//
// The reference's model contract is `logp_grad(theta) -> (logp, grad)` (LogpGrad, concepts.hpp:258-262; C form
// LOGP_CFUNC, python/src/walnutpie/walnutpy.cpp:131-132).  A host function cannot be called from a GPU-resident
// trajectory, so here the model is a struct of static device functions that the kernels are instantiated with.  One
// workgroup of L = 64 * NW lanes evaluates one chain; lane `tid` holds EPL coordinates of every vector, slot j of it
// being coordinate cx.index(j) (pairs of consecutive coordinates, pair m = k * L + tid).
//
//   struct MyModel {
//     static constexpr bool kUsesParams;      // a parameter vector of num_params doubles exists (arrives as `mp`,
//                                             //   padded with 1.0); wn_engine_create requires it then
//     static constexpr bool kElementwise;     // grad[i] depends on theta[i] (and mp[i]) only: the streaming kernels (the
//                                             //   default above 4 096 parameters) then take ONE pass per micro step; needs grad()
//                                             //   below (other models: the optional streaming form further down)
//     static constexpr bool kCheapGrad;       // grad_elem() is one or two operations: the kernels then store no
//                                             //   gradient vector at all and call grad_elem() at each use
//     static constexpr bool kGradIsNegTheta;  // (informational) grad == -theta
//     static double grad_elem(double theta_i, double mp_i);     // used only when kCheapGrad
//     struct Aux { ... };                     // wave-uniform by-products of eval() that finish() wants
//
//     // Write the gradient of the lane's coordinates to g and ADD the lane's log-density terms, in slot order, to
//     // `acc`.  The kernels sum `acc` over the chain in a fixed order (lane partials, butterfly, wavefronts left to
//     // right) and hand the total to finish().  Padding slots (!cx.valid(j)) must leave g[j] = 0 and add nothing.
//     template <int EPL, class Cx>
//     static void eval(Cx& cx, const double (&theta)[EPL], double (&g)[EPL], const double (&mp)[EPL], Aux&, double& acc);
//     template <int EPL, class Cx>            // kElementwise only: the gradient alone, same expression as in eval()
//     static void grad(Cx& cx, const double (&theta)[EPL], double (&g)[EPL], const double (&mp)[EPL], Aux&);
//     static double finish(double sum, const Aux&, int num_params);   // -> logp
//
//     // optional -- "Streaming a model whose gradient is not element-wise": above 8 192 parameters (4 096 by default
//     // where the kernels that hold the trajectory's moving end in registers apply) the span pool lives in HBM and
//     // vectors are processed two coordinates at a time, so eval() (which sees the lane's whole share at once) cannot
//     // run.  A model states what its gradient needs beyond the coordinate itself, and gets two passes per micro step:
//     static constexpr bool kStreamable = true;
//     static constexpr int kStreamSums;       // 0..2 sums over ALL coordinates (funnel: sum x_i^2 and x_0)
//     static constexpr bool kStreamHalo;      // the gradient reads the neighbouring coordinates (rw1)
//     template <class Cx> static void stream_sums(Cx&, const double (&theta)[2], const double (&mp)[2], double (&sums)[N]);
//     template <class Tab> static void stream_aux(const double (&sums)[N], int num_params, const Tab&, Aux&);
//     template <class Cx> static void stream_grad(Cx&, const double (&theta)[2], const double (&prev)[2],
//                                                 const double (&next)[2], const double (&mp)[2], double (&g)[2], const Aux&);
//     template <class Cx> static void stream_logp(Cx&, theta, prev, next, mp, const Aux&, double& acc);  // adds the terms
//     // (prev[j] / next[j]: the values at coordinate index(j) - 1 / + 1, 0.0 beyond the ends; the same expressions as
//     // in eval() give the same bits.  wn_models.h: FunnelModel, models/rw1.h)
//
//     // optional, host side (wn_engine_create): check / transform the parameter vector before it is uploaded;
//     // check num_params.  Throw std::invalid_argument to reject (-> error type `config`).
//     static void host_params(double* params, int num_params);
//     static void validate(int num_params);
//     // optional: geometry hint -- elements per lane (2, 4, 8 or 16); the engine then takes the fewest wavefronts per
//     // chain that hold num_params at that width instead of its default policy (one wavefront up to 1 024
//     // dimensions).  A model whose evaluation exchanges data across lanes or keeps many live vectors may prefer
//     // more, narrower wavefronts (models/rw1.h); an explicit waves_per_chain / elems_per_lane request still wins.
//     static constexpr int kPreferredElemsPerLane;
//     // ... or, where the best width depends on the dimension, the same hint as a function (0 = the default policy there):
//     static constexpr int preferred_elems_per_lane(int num_params);
//   };
//
// What `cx` offers (all of it collective: every lane of the chain's workgroup must make the same calls):
//   cx.index(j), cx.valid(j), cx.dim()        coordinate of slot j; whether it is < num_params; num_params
//   cx.sum1(x)                                sum of x over all lanes of the chain (fixed order), wave-uniform
//   cx.element0(x)                            the value slot 0 of lane 0 holds (coordinate 0), wave-uniform
//   cx.shift(v, prev, next)                   prev[j] = v at coordinate index(j) - 1, next[j] = v at index(j) + 1
//                                             (0.0 beyond either end of the padded vector)
//   cx.uniform_tab()                          tables for wnd::dexp / wnd::dlog of a wave-uniform argument
//   Cx::mad(a, b, c)                          a * b + c: one fused multiply-add when the engine was created with
//                                             wn_config::fused_multiply_add, a rounded product plus an add otherwise
// Arithmetic: the library is compiled with -ffp-contract=off; what you write is what is evaluated (Cx::mad is the one
// place where the engine's arithmetic mode shows), so a CPU restatement of the same expressions reproduces the device
// bit for bit (that is how the parity tests work).
//
// Registration is a five-line translation unit, wn_kernels_<name>.hip, that the Makefile picks up by its name:
//   #include "models/my_model.h"
//   #define WN_MODEL_ID 4                 // 0-3 are taken (std_normal, diag_normal, funnel, rw1); < 64
//   #define WN_MODEL_TAG my_model         // wn_model_id("my_model") finds it at run time
//   #define WN_MODEL_TYPE wn::MyModel
//   #include "wn_kernels.inc"
// models/rw1.h is a complete example (the reference's AR(1) density with a neighbour-coupled gradient).
#pragma once

#include "wn_devmath.h"
#include "wn_hip.h"
