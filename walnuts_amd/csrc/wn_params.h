// wn_params.h -- plain-data launch parameters shared by the host side (wn_capi.cpp) and the kernels.
#pragma once

#include <stdint.h>

namespace wn {

constexpr int kMaxLevels = 16;   // span-stack levels => max_trajectory_doublings <= 17
constexpr int kMaxPool = 64;     // vector buffers per resident chain (free mask is 64 bits)
constexpr int kDrawCache = 64;    // tree draws (and their logs) produced per refill, one per lane
#if defined(WN_TIMELINE)
#if defined(WN_TIMELINE_MARKS)  // (the streaming probe: every wavefront's scratch carries the array, beside 128 KB of inverse mass)
constexpr int kTimelineMarks = WN_TIMELINE_MARKS;
#else
constexpr int kTimelineMarks = 1536;
#endif
constexpr int kMetaDoubles = 128 + kTimelineMarks + 8;
#else
constexpr int kMetaDoubles = 128;
#endif

enum ModelKind : int32_t { kStdNormal = 0, kDiagNormal = 1, kFunnel = 2 };
enum RngMode : int32_t { kRngPhilox = 0, kRngBuffer = 1 };

// Every [C][Dp] plane is chain-major: one chain's vector is contiguous, rows are
// padded to Dp = 64*NW*EPL doubles so that lane l of the chain's workgroup owns
// the 16-byte pairs (k*L + l), k = 0..EPL/2-1 (L = 64*NW lanes per chain).
struct Params {
  // geometry
  int32_t num_chains;  // one past the last chain of this launch (a launch covers [chain_begin, num_chains): the whole engine,
                       // or one of its chain groups)
  int32_t dim;        // D
  int32_t dim_padded; // Dp
  int32_t warmup;     // 1: AdaptiveWalnuts transition, 0: WalnutsSampler transition
  // per-chain planes [C][Dp]
  double* theta;
  double* inv_mass;        // sampling phase: frozen inverse mass diagonal
  double* chol_mass;       // sampling phase: 1/sqrt(inv_mass) (WalnutsSampler::cholesky_mass_, walnuts.hpp:647)
  double* est_draw_mean;   // MassEstimator state (adaptive_walnuts.hpp:25-105)
  double* est_draw_ssd;
  double* est_score_mean;
  double* est_score_ssd;
  // per-chain scalars
  double* step_size;   // [C] sampling: frozen step; warmup: exp(adam.theta) is used instead
  int32_t* min_micro;  // [C] sampling: frozen
  double* adam;        // [C][6] theta, m, v, t, b1pow, b2pow
  double* est_weight;  // [C][2] draw, score
  double* mm_state;    // [C][2] total, count (MinMicroStepsAdaptHandler)
  double* logp_out;    // [C]
  int32_t* depth_out;  // [C]
  int64_t* grad_evals; // [C] running totals
  int32_t* rng_draws;  // [C] scalar draws of the last transition
  int32_t* failed_ext; // [C] 1: an extension of the chain's last transition failed (kNoteExtensionFailed)
  double* lp_stats;    // [C][3] WelfordAccumulator (count, mean, M2) of the sampling log densities
  // output of this transition
  double* draws_out;   // nullable; chain c's row of the launch's k-th transition at draws_out + c*draws_stride + k*draws_tstride
  int64_t draws_stride;
  int64_t draws_tstride;
  // model
  const double* model_params; // DIAG_NORMAL: sigma_sq [Dp] (padding = 1.0)
  // configuration (SamplingConfig / WarmupConfig)
  int32_t max_depth;
  int32_t max_halvings;
  int32_t cfg_min_micro;
  int32_t fma;  // 1: the integrator's multiply-adds are fused (wn_config::fused_multiply_add)
  double max_error;
  double mass_init_count;
  double macro_target;
  double adam_target, adam_lr, adam_b1, adam_b2, adam_eps, adam_decay;
  // randomness
  uint64_t seed;
  uint32_t chain_offset;
  uint32_t transition;
  int32_t rng_mode;
  int32_t u_stride;      // kRngBuffer: uniforms per chain
  const double* z_buf;   // kRngBuffer: [C][Dp] standard normals
  const double* u_buf;   // kRngBuffer: [C][u_stride] canonical uniforms
  int64_t warmup_iter;   // AdaptiveWalnuts::iteration_
  // scratch
  double* arena;         // [slots][pool_global][Dp]
  int64_t arena_stride;  // doubles per slot
  int32_t pool_lds;      // vector buffers living in LDS
  int32_t pool_total;    // LDS + arena buffers
  int32_t fused;         // transitions per launch (>= 1): a workgroup runs them back to back on the chain it fetched
  uint32_t* work_counter;  // chains fetched so far by all launches of this engine (mod 2^32; never reset)
  uint32_t work_base;      // its value when this launch starts
  uint32_t im_in_lds;      // streaming kernels: bit 0 = the chain's inverse mass is parked in LDS for the whole
                           // transition; bit 1 = (experiment switch) no far-end sums in the leaf's pass; bit 2 = the
                           // vectors fit the registers a kernel with a held moving end has for them (TrajMem, HOLD)
  int32_t chain_begin;    // first chain of this launch
  int32_t est_mode;       // register kernels, warmup (TrajChip::kDeferObservation): 1 = the mass estimator has not seen the
                          // previous launch's last transition yet (this launch's first prologue applies the observation);
                          // 2 = apply that observation and do nothing else (the engine's flush); 0 = nothing pending
  uint32_t* error_flags;  // OR of kErr* bits of every chain and transition since wn_engine_check last read (and cleared) it
};

enum : uint32_t {
  kErrPoolExhausted = 1u,     // a chain needed more span-pool vectors than the engine holds
  kErrVariatesExhausted = 2u, // host-fed uniforms (wn_engine_set_variates) ran out inside a transition
  // not an error, a note kept beside the error bits (one register) until the transition's scalars are stored: an
  // extension of this transition FAILED -- a leaf's energy error exceeded the bound at every step size, or its
  // reversibility check failed (walnuts.hpp:344,:271-274,:543-545)
  kNoteExtensionFailed = 256u
};

}  // namespace wn
