/* wn_oracle.h -- C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * A CPU restatement of the reference's Walnuts/NUTS leapfrog trajectory path
 * (flatironinstitute/walnuts, include/walnutpie/{walnuts,util,adam,
 * online_moments,adaptive_walnuts,api,config}.hpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (walnuts_amd/, include/walnuts_hip.h) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - logp_momentum, leapfrog_error, log_sum_exp, init masses: pinned by the
 *     reference's own known-answer tests (tests/util_test.cpp:102-160,
 *     236-266,385-476; tests/config_test.cpp:383-398) -> tests/test_oracle_kat.py
 *   - Adam: pinned against the REAL reference header (adam.hpp is Eigen-free
 *     and is compiled as oracle/_ref/libadam_ref.so by oracle/Makefile).
 *   - macro_step / reversible / uturn / combine / build_span / transition_w /
 *     OnlineMoments / MassEstimator: the reference ships no test vectors and
 *     cannot be built here (Eigen 3.4 is fetched at configure time,
 *     CMakeLists.txt:28-41, and is absent).  PARITY UNPINNED by reference
 *     vectors for these; they are pinned by (i) the regression values the
 *     survey obtained from the unmodified reference headers (SURVEY.md §8c:
 *     config #1 end state) and (ii) closed-form Gaussian leapfrog checks.
 */
#ifndef WN_ORACLE_H
#define WN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { WNO_MODEL_STD_NORMAL = 0, WNO_MODEL_DIAG_NORMAL = 1, WNO_MODEL_FUNNEL = 2, WNO_MODEL_RW1 = 3 };
enum { WNO_MATH_LIBM = 0, WNO_MATH_PORTABLE = 1 };
enum { WNO_RNG_STD_MT64 = 0, WNO_RNG_STD_MT32 = 1, WNO_RNG_PHILOX = 2 };

/* reduce_lanes = WNO_REDUCE_EIGEN_SSE2: sums in the order of Eigen 3.4's vectorised redux with 2-lane packets (the
 * reference's default x86-64 build); restated from Eigen's published algorithm, see Reducer in wn_oracle.cpp */
enum { WNO_REDUCE_EIGEN_SSE2 = -2 };

typedef struct wno_config {
  /* SamplingConfig, reference defaults config.hpp:947-953 */
  int32_t max_trajectory_doublings; /* 5 */
  int32_t max_step_halvings;        /* 5 */
  int32_t min_micro_steps;          /* 1 */
  double max_hamiltonian_error;     /* 0.5 */
  /* WarmupConfig, reference defaults config.hpp:626-640 */
  double mass_init_count;          /* 4 */
  double max_macro_steps_target;   /* 15 */
  double step_accept_rate_target;  /* 0.8 */
  double step_learning_rate;       /* 0.05 */
  double step_gradient_decay;      /* 0.8 */
  double step_sq_gradient_decay;   /* 0.9 */
  double step_stabilization;       /* 1e-4 */
  double step_learn_rate_decay;    /* 0.5 */
  /* how the oracle executes */
  int32_t math_mode;    /* WNO_MATH_* */
  int32_t reduce_lanes; /* 0: left-to-right sums; L>0: the device order with L lanes; WNO_REDUCE_EIGEN_SSE2 */
  int32_t rng_mode;     /* WNO_RNG_* */
  int32_t fma;          /* 1: device arithmetic with fused multiply-adds (wn_config::fused_multiply_add): the
                           transition's leapfrog updates, kinetic and U-turn sums and the models' log-density sums
                           use std::fma exactly where the kernels do; 0: every product rounded */
} wno_config;

void wno_default_config(wno_config* cfg);

typedef struct wno_engine wno_engine;

/* model params: DIAG_NORMAL -> sigma_sq[D]; others none */
wno_engine* wno_create(int model, int dim, const double* params, size_t num_chains,
                       const wno_config* cfg);
void wno_destroy(wno_engine* e);
/* switch the generator family used by the NEXT init/seed call (the reference's
 * example seeds its init engine as mt19937 but its chains as mt19937_64:
 * examples/walnutpie_api.cpp:48-49,78) */
void wno_set_rng_mode(wno_engine* e, int mode);

/* ---- initialisation (InitConfigBuilder, config.hpp:195-484) ------------- */
void wno_set_positions(wno_engine* e, const double* pos /*[C*D]*/);
void wno_set_masses(wno_engine* e, const double* mass /*[C*D]*/);
void wno_set_step_sizes(wno_engine* e, const double* steps /*[C]*/);
/* positions(rng, scale): one shared std engine seeded seed_seq{s0,s1} walks the
 * chains in order (config.hpp:258-268, walnutpy.cpp:187-189); PHILOX mode uses
 * stream WNO_STREAM_INIT_POS keyed by (seed=s0, chain). */
void wno_init_positions(wno_engine* e, uint64_t s0, uint64_t s1, double scale);
/* masses(logp_grad, smoothing, average) config.hpp:360-382 */
void wno_init_masses_from_grad(wno_engine* e, double smoothing, int average);
/* adapt_step_build(rng, logp_grad) config.hpp:470-476 + util.hpp:285-303 */
void wno_adapt_step(wno_engine* e, uint64_t s0, uint64_t s1);
/* per-chain engines: seed_seq{seed, m+1} (api.hpp:46-51) or Philox key=seed,
 * chain ids chain_offset+m */
void wno_seed_chains(wno_engine* e, uint64_t seed, uint32_t chain_offset);

/* host-supplied variates for the NEXT transition only: normals [C*D], canonical uniforms [C*u_per_chain]
 * consumed in order, bernoulli = (u < 0.5) as libstdc++ does (mirrors wn_engine_set_variates) */
void wno_set_variates(wno_engine* e, const double* normals, const double* uniforms, size_t u_per_chain);

/* ---- stepping ----------------------------------------------------------- */
void wno_warmup_step(wno_engine* e, int num_threads); /* AdaptiveWalnuts::operator() for all chains */
void wno_freeze(wno_engine* e);                       /* AdaptiveWalnuts::sampler() */
void wno_sample_step(wno_engine* e, int num_threads); /* WalnutsSampler::operator() for all chains */

/* frozen sampler parameters handed in as they are: inverse mass [C*D], step [C], min micro steps [C] */
void wno_set_sampler_state(wno_engine* e, const double* inv_mass, const double* step, const int64_t* min_micro);
/* adaptation state handed in as it is: Adam [C*6] (theta,m,v,t,b1pow,b2pow), the estimator's planes [C*D] and weights
 * [C*2], the min-micro-steps value in force [C], warmup transitions done so far */
void wno_set_adapt_state(wno_engine* e, const double* adam, const double* draw_mean, const double* draw_ssd,
                         const double* score_mean, const double* score_ssd, const double* weights,
                         const int64_t* min_micro, uint64_t iteration);
/* counter-based streams are keyed by the transition index: make the next transition number `t` */
void wno_set_transition_index(wno_engine* e, uint32_t t);
/* near-tie audit: tolerance relative to the compared magnitudes (0 = off); totals over
 * all chains: out[0..2] = near ties (energy error walnuts.hpp:339, U-turn signs :199-200, acceptance :379),
 * out[3..5] = decisions taken */
void wno_set_tie_tolerance(wno_engine* e, double tol);
void wno_get_near_ties(wno_engine* e, int64_t* out, int reset);
/* WNO_MATH_PORTABLE: how often a transition moved the reference energy of its span weights (all chains, so far) */
int64_t wno_get_weight_rebases(const wno_engine* e);

/* ---- state -------------------------------------------------------------- */
void wno_get_positions(const wno_engine* e, double* out /*[C*D]*/);
void wno_get_grad_select(const wno_engine* e, double* out /*[C*D]*/);
void wno_get_logp(const wno_engine* e, double* out /*[C]*/);
void wno_get_step_sizes(const wno_engine* e, double* out /*[C]*/);
void wno_get_masses(const wno_engine* e, double* out /*[C*D]*/);
void wno_get_inv_mass(const wno_engine* e, double* out /*[C*D]*/);
void wno_get_min_micro(const wno_engine* e, int64_t* out /*[C]*/);
void wno_get_depths(const wno_engine* e, int32_t* out /*[C]*/);
void wno_get_grad_evals(const wno_engine* e, int64_t* out /*[C]*/);
void wno_get_rng_draws(const wno_engine* e, int64_t* out /*[C]*/); /* scalar tree draws of last transition */
void wno_get_estimator(const wno_engine* e, double* draw_mean, double* draw_ssd, double* score_mean,
                       double* score_ssd, double* weights /*[C*2]*/);
void wno_get_adam(const wno_engine* e, double* out /*[C*6]: theta,m,v,t,b1pow,b2pow*/);
int64_t wno_iteration(const wno_engine* e);
/* the reference's controller statistics: R-hat of the log density over the sampling draws so far
 * (sampler.hpp:132-145) and the warmup spread of step size / mass across chains (adapt.hpp:193-221) */
double wno_rhat(const wno_engine* e);
void wno_warmup_spread(wno_engine* e, double* max_rel_step, double* max_rel_mass);

/* the controllers' two helpers by themselves, reference order: l2_rel_diff (util.hpp:380-383) and the bias-adjusted
 * sample variance (util.hpp:401-404) -- the functions wno_warmup_spread / wno_rhat are built on */
double wno_l2_rel_diff(size_t n, const double* a, const double* b);
/* device arithmetic: a / w the way the kernels divide a plane by the estimator's weight (wn_devmath.h SharedDivisor) */
double wno_div_shared(double a, double w);
double wno_variance(size_t n, const double* xs);

/* ---- per-macro-step trace of the LAST transition of one chain ------------ */
/* record layout (doubles): [dir, level, n_micro, step, H_start, H_end, accepted,
 * reversible, logp_pos_end] ; returns number of records written (<= max_rec) */
#define WNO_TRACE_FIELDS 9
void wno_enable_trace(wno_engine* e, int on);
size_t wno_get_trace(const wno_engine* e, size_t chain, double* out, size_t max_rec);

/* ---- function-level entry points (KATs / unit parity) -------------------- */
double wno_logp_momentum(size_t n, const double* rho, const double* inv_mass, int reduce_lanes);
double wno_log_sum_exp(double a, double b, int math_mode);
/* sum of x[0..n) in the order `reduce_lanes` names (0, L > 0, WNO_REDUCE_EIGEN_SSE2) */
double wno_reduce_sum(size_t n, const double* x, int reduce_lanes);
int wno_model_logp_grad(int model, int dim, const double* params, const double* x, double* logp,
                        double* grad, int math_mode, int reduce_lanes);
double wno_leapfrog_error(int model, int dim, const double* params, const double* theta,
                          const double* rho, const double* inv_m, double step, int math_mode,
                          int reduce_lanes);
int wno_uturn(size_t n, int forward, const double* th_in, const double* rho_in, const double* th_out,
              const double* rho_out, const double* inv_mass, int reduce_lanes);
/* Adam (adam.hpp:48-93): feed alphas, get step size after each */
void wno_adam_run(double step_init, double target, double lr, double b1, double b2, double eps,
                  double decay, const double* alphas, size_t n, double* steps_out, int math_mode);
/* OnlineMoments (online_moments.hpp:151-230): one discounted observation in place */
void wno_online_moments_observe(size_t n, double discount, double* weight, double* mean, double* ssd,
                                const double* y);
/* one macro step (walnuts.hpp:307-345) from (theta,rho,grad,logp_joint); outputs
 * next state; returns 1 on success.  alpha_out = value handed to the adapter. */
int wno_macro_step(int model, int dim, const double* params, const wno_config* cfg, int forward,
                   double step, int min_micro, const double* inv_mass, const double* theta,
                   const double* rho, const double* grad, double logp_joint, double* theta_out,
                   double* rho_out, double* grad_out, double* logp_pos_out, double* logp_joint_out,
                   double* alpha_out, int64_t* grad_evals_out);
/* Philox stream accessors (for checking the device generator) */
double wno_stream_uniform(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream,
                          uint32_t index);
void wno_stream_normals(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, size_t n,
                        double* out);
double wno_math_exp(double x);
double wno_math_log(double x);
double wno_math_exp_weight(double x);  /* exp(max(x, -700)): the device's span-weight exponential */

#ifdef __cplusplus
}
#endif
#endif
