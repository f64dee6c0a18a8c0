/* walnuts_hip.h -- C ABI of the MI355X-native many-chain Walnuts/NUTS leapfrog engine.
 *
 * This is the drop-in boundary for ONE path of flatironinstitute/walnuts: the
 * per-chain trajectory loop (detail::transition_w and everything below it,
 * include/walnutpie/walnuts.hpp:520-563) together with the per-chain warmup
 * adaptation wrapped around it (AdaptiveWalnuts, adaptive_walnuts.hpp:182-363),
 * run for C chains at once on one GPU.  Plain pointers and sizes only.
 *
 * Two layers:
 *   1. walnutpie_sample_device(): the sibling of the reference's
 *      walnutpie_sample_cfunc() (python/src/walnutpie/walnutpy.cpp:134-222) with
 *      the host callback (LOGP_CFUNC, void* data) replaced by a built-in device
 *      model id + parameter vector.  Same trailing argument list, same output
 *      buffer layout, same error object and accessors.
 *   2. wn_engine_*(): the batched equivalents of the C++ objects behind it --
 *      walnutpie::AdaptiveWalnuts::operator() (adaptive_walnuts.hpp:234-251),
 *      ::sampler() (:263-271) and walnutpie::WalnutsSampler::operator()
 *      (walnuts.hpp:682-692) -- each call advancing ALL chains by one transition.
 *
 * A host LOGP_CFUNC cannot be called from a GPU-resident trajectory and this library has
 * no CPU path.  walnutpie_sample_cfunc / walnutpie_sample_bridgestan are exported with the
 * reference's signatures so that the library can stand in for libwalnutpy at LOAD time (the
 * reference's python/src/walnutpie/_ffi.py binds every symbol at import), but calling one
 * fails with a config error naming walnutpie_sample_device; callers with host models keep
 * the reference library for those calls.
 *
 * Every function returns 0 on success and -1 on failure; on failure *err (when
 * err != NULL) receives a WalnutpyError to be freed with walnutpie_destroy_error,
 * exactly as in python/src/walnutpie/errors.hpp:10-72.
 */
#ifndef WALNUTS_HIP_H
#define WALNUTS_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WALNUTS_HIP_EXPORT __attribute__((visibility("default")))


/* ---- errors: python/src/walnutpie/errors.hpp:10-24, walnutpy.cpp:371-389 ---- */
typedef enum { generic = 0, config = 1, interrupt = 2 } WalnutpyErrorType;
typedef struct WalnutpyError WalnutpyError;
WALNUTS_HIP_EXPORT const char* walnutpie_get_error_message(const WalnutpyError* err);
WALNUTS_HIP_EXPORT WalnutpyErrorType walnutpie_get_error_type(const WalnutpyError* err);
WALNUTS_HIP_EXPORT void walnutpie_destroy_error(WalnutpyError* err);

/* progress callback: python/src/walnutpie/handlers.hpp:15 */
typedef void (*PRINT_CALLBACK)(const char* msg, size_t len, bool bad);
/* BridgeStan's print callback, as walnutpie_sample_bridgestan takes it (thirdparty/bridgestan/bridgestan.h:400,
 * python/src/walnutpie/walnutpy.cpp:227-229) */
typedef void (*STREAM_CALLBACK)(const char* data, size_t size);

/* ---- the reference's host-model entry points: present, never sampling (see above) -------------
 * walnutpy.cpp:131-149 (LOGP_CFUNC model), :227-245 (BridgeStan model), :224-225 (the separator of
 * the per-chain init strings).  Both samplers return -1 with error type `config`. */
typedef int (*WN_LOGP_CFUNC)(size_t size, const double* theta, double* grad, double* lp, void* data);
#define WN_REFERENCE_SAMPLING_PARAMS                                                                              \
  size_t num_chains, unsigned int seed, unsigned int id, double init_radius, const double *init_inv_metric,      \
      int min_warmup_iter, int max_warmup_iter, int min_sampling_iter, int max_sampling_iter,                    \
      int max_trajectory_doublings, int max_step_halvings, int min_micro_steps, double max_hamiltonian_error,    \
      double step_size_converge_tol, double mass_converge_tol, double rhat_converge_tol, double mass_init_count, \
      double mass_additive_smoothing, double max_macro_steps_target, double step_size_init,                      \
      double step_accept_rate_target, double step_learning_rate, double step_gradient_decay,                     \
      double step_sq_gradient_decay, double step_stabilization, double step_learn_rate_decay, bool save_warmup,  \
      double *out, size_t out_size, int *final_lengths, double *stepsize_out, double *inv_metric_out,            \
      int refresh, PRINT_CALLBACK print, WalnutpyError **err
WALNUTS_HIP_EXPORT int walnutpie_sample_cfunc(WN_LOGP_CFUNC logp_c, void* data, int num_params, const double* inits,
                                              WN_REFERENCE_SAMPLING_PARAMS);
WALNUTS_HIP_EXPORT int walnutpie_sample_bridgestan(const char* bs_dll, const char* json_data,
                                                   STREAM_CALLBACK callback, unsigned int model_seed,
                                                   const char* inits, WN_REFERENCE_SAMPLING_PARAMS);
WALNUTS_HIP_EXPORT char walnutpie_separator_char(void);

/* ---- built-in device models (the LogpGrad contract, concepts.hpp:258-262) ---- */
typedef enum {
  WN_MODEL_STD_NORMAL = 0,  /* examples/walnutpie_api.cpp:37-41; no parameters            */
  WN_MODEL_DIAG_NORMAL = 1, /* examples/examples.cpp:20-31; params = sigma_sq[num_params]  */
  WN_MODEL_FUNNEL = 2,      /* Neal's funnel, x0 ~ N(0,9), x_i ~ N(0,e^x0); no parameters  */
  WN_MODEL_RW1 = 3          /* examples/examples.cpp:34-49, AR(1) covariance rho^|i-j|, rho = 0.99; added through the
                               public model interface (csrc/models/rw1.h); further models: wn_model_id()        */
} wn_model;

/* device-resident chains of draws for the posterior summaries at the end of this header */
typedef struct wn_chains wn_chains;

/* ---- replaces walnutpie_sample_cfunc (walnutpy.cpp:134-149) -------------------
 * Random numbers: initial positions (inits == NULL; config.hpp:258-268), the step-size search's momenta
 * (util.hpp:285-303) and the chains' trajectories all come from the counter-based generator on the device, keyed by
 * `seed` (positions, search), seed + id + num_chains (chains; walnutpy.cpp:82) and the chain id: same seed, same
 * output, whatever the launch geometry.  Arithmetic mode: wn_default_config()'s (fused multiply-adds unless
 * WALNUTS_AMD_FMA=0 is in the environment; see wn_config::fused_multiply_add). */
WALNUTS_HIP_EXPORT int walnutpie_sample_device(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    WalnutpyError** err);

/* The same call with ALL random numbers taken from the reference's own streams, generated on the host: initial
 * positions from mt19937_64(seed_seq{seed, 1}) (walnutpy.cpp:187-189), the step-size search's normals from
 * mt19937_64(seed_seq{seed, 2}) (:75-80), the chains' from mt19937_64(seed_seq{seed + id + num_chains, m + 1})
 * (api.hpp:46-51; see wn_engine_seed_reference_streams), and with every product rounded (fused_multiply_add = 0, the
 * reference's x86-64 element-wise bits) whatever the environment says: slower, for parity runs against the reference
 * at equal seed. */
WALNUTS_HIP_EXPORT int walnutpie_sample_device_reference_streams(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    WalnutpyError** err);

/* walnutpie_sample_device for runs whose draws should not cross PCIe (65 536 chains x 1 024 parameters are 512 MiB per
 * iteration: ~10 ms of PCIe against ~1.6 ms of compute).  The SAMPLING draws stay in HBM as one [C][S][D] block
 * (S = max_sampling_iter; it has to fit) owned by *chains_out -- a wn_chains for the wn_summary_* functions below (mean,
 * variance, quantiles, R-hat, ESS, MCSE on the device), to be released with wn_chains_destroy.  `out` receives only
 * every `thin`-th sampling draw (iterations 1, 1 + thin, ...; thin = 0: none, `out` may then be NULL unless
 * save_warmup).  Layout: out[C][max_warmup_iter * save_warmup + ceil(S / thin)][D] is the CAPACITY; as in
 * walnutpie_sample_device and the reference (handlers.hpp:73-89: a chain writes its rows sequentially) the thinned
 * sampling rows follow the warmup rows ACTUALLY WRITTEN: chain c's thinned draw k is row final_lengths[c] + k.
 * final_lengths[C + c] is the number of sampling draws chain c holds in *chains_out (all of them), of which the rows
 * 0, thin, 2 thin, ... are the ones in `out`.  Everything else as walnutpie_sample_device. */
WALNUTS_HIP_EXPORT int walnutpie_sample_device_resident(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    int thin, wn_chains** chains_out, WalnutpyError** err);

/* walnutpie_sample_device over SEVERAL devices of the node ("one process, one driver thread +
 * stream per GPU").  devices[num_devices]: HIP ordinals; shard s -- a contiguous block of the global chain ids, sizes
 * differing by at most one -- runs on devices[s] with its own host thread, engine and stream and writes its own slice
 * of out[C][T][D], final_lengths, stepsize_out and inv_metric_out: no exchange on the data path.  The random streams are
 * keyed by global chain id and the controllers' statistics (warmup spread adapt.hpp:193-221, R-hat sampler.hpp:139-145)
 * are reduced over all shards, so every shard stops at the same iteration and the output equals the one-device call's
 * for the same arguments (bit for bit with min_*_iter == max_*_iter; the stopping statistics are summed shard by shard
 * otherwise).  An ordinal may repeat: {0, 0} runs two half-size engines on two streams of one device, each filling the
 * other's launch tail (+12 % / +23 % on BASELINE configs #2 / #3, profiles/r03/two_groups.txt). */
WALNUTS_HIP_EXPORT int walnutpie_sample_device_multi(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    const int* devices, int num_devices, WalnutpyError** err);

/* ... with the sampling draws kept on the devices (walnutpie_sample_device_resident's contract for `out`, `thin` and
 * `chains_out`): every shard keeps its draws on its own device while it samples; at the end the shards' blocks --
 * contiguous slabs of the chain-major [C][S][D] layout -- are gathered on devices[0] with one peer-to-peer copy per
 * shard (hipMemcpyPeerAsync, each on a stream of its own so that the inbound copies run side by side, each over the
 * xGMI link of its source; peer access is enabled per pair first, and a pair without a peer path is a `generic` error
 * that names it) and handed back as ONE wn_chains there.  Unmeasured on more than one physical device. */
WALNUTS_HIP_EXPORT int walnutpie_sample_device_multi_resident(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    const int* devices, int num_devices, int thin, wn_chains** chains_out, WalnutpyError** err);

/* ... the ALL-GATHER of the draws (BASELINE.json's north star: "only an ... all-gather of draws over xGMI"), for a C/C++
 * caller (the reference's callers: examples/walnutpie_api.cpp:78-79, walnutpy.cpp:82) without Python or RCCL: as
 * _multi_resident, but EVERY listed device ends with the whole [C][S][D] block -- chains_out is an array of
 * num_devices handles, chains_out[d] living on devices[d].  All-pairs direct copies: each device pulls every other
 * shard's slab over the link of that pair (xGMI is point-to-point: N-1 inbound copies per device side by side, no
 * ring), its own shard by a local copy.  Unmeasured on more than one physical device. */
WALNUTS_HIP_EXPORT int walnutpie_sample_device_multi_allgather(
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    const int* devices, int num_devices, int thin, wn_chains** chains_out, WalnutpyError** err);

/* ---- batched engine ------------------------------------------------------------- */
typedef struct wn_engine wn_engine;

/* SamplingConfig (config.hpp:885-1059, defaults :947-953) + WarmupConfig
 * (config.hpp:513-850, defaults :626-640) fields that reach the per-chain path,
 * plus launch geometry knobs (0 = choose automatically). */
typedef struct wn_config {
  int32_t max_trajectory_doublings; /* 5 */
  int32_t max_step_halvings;        /* 5 */
  int32_t min_micro_steps;          /* 1 */
  int32_t device;                   /* HIP device ordinal */
  double max_hamiltonian_error;     /* 0.5 */
  double mass_init_count;           /* 4 */
  double max_macro_steps_target;    /* 15 */
  double step_accept_rate_target;   /* 0.8 */
  double step_learning_rate;        /* 0.05 */
  double step_gradient_decay;       /* 0.8 */
  double step_sq_gradient_decay;    /* 0.9 */
  double step_stabilization;        /* 1e-4 */
  double step_learn_rate_decay;     /* 0.5 */
  int32_t waves_per_chain;          /* NW: wavefronts cooperating on one chain */
  int32_t elems_per_lane;           /* EPL: vector elements held per lane; -1 = streaming kernels (the span pool in
                                       HBM: the default above 8192 parameters -- above 4096 for a model whose
                                       gradient is element-wise, wn_engine_held_tiles) */
  int32_t workgroups_per_cu;        /* resident chains per compute unit */
  int32_t lds_vectors;              /* span-pool vectors kept in LDS (-1: as many as fit) */
  int32_t reserved_cus;             /* compute units the persistent grid leaves free (e.g. for RCCL kernels that
                                       all-gather the previous iteration's draws while this one runs); 0 */
  int32_t fused_multiply_add;       /* 1 (default; WALNUTS_AMD_FMA=0 in the environment makes it 0): the multiply-adds of
                                       the integrator (walnuts.hpp:228-231,329-332), of logp_momentum (util.hpp:222),
                                       of the U-turn products (walnuts.hpp:196-200) and of the built-in models' log
                                       densities are single fused operations, as in a build of the reference for an
                                       FMA target (aarch64, x86-64 -march=haswell and later).  0: every product is
                                       rounded before it is added -- the element-wise bits of the reference built
                                       as its CMake files build it on x86-64 (-O3, SSE2).  Both meet the <= 1e-10
                                       bar against the reference order; the fused kernels are ~10 % faster. */
  int32_t chain_groups;             /* 0 (default): chosen by the engine -- 2 when there are more chains than resident
                                       workgroups, else 1; up to 4; WALNUTS_AMD_CHAIN_GROUPS in the environment
                                       overrides.  With g > 1 the chains are launched as g independent contiguous
                                       blocks on g streams, each filling the others' launch tails (2 groups: +13 % /
                                       +26 % on BASELINE configs #2 / #3, +2 % on the headline; 3 and 4 measured no
                                       better -- profiles/r04/ab_chain_groups.txt); results do not depend on it.
                                       An engine fed host variates (reference_streams) keeps its groups in lock step;
                                       one moved to a caller's stream (wn_engine_set_stream) runs as one group. */
} wn_config;

WALNUTS_HIP_EXPORT void wn_default_config(wn_config* cfg);
/* Device models are compiled into the library and entered into a registry by their translation units
 * (walnuts_amd/csrc/wn_model_api.h; INTEGRATION.md "Adding a device model").  -> the id registered under `name`
 * ("std_normal" 0, "diag_normal" 1, "funnel" 2, "rw1" 3, ...), or -1. */
WALNUTS_HIP_EXPORT int wn_model_id(const char* name);
/* Version of the counter-based random streams of this build (the map (seed, chain, transition, index) -> variate,
 * csrc/wn_devmath.h): results at a fixed seed are reproducible within one version only.  A caller that stores draws
 * or resumes a run records it beside the seed; walnuts_amd writes it into every result and bench line.
 * (The reference's counterpart is implicit: mt19937_64 + libstdc++'s distributions, util.hpp:78-162.) */
WALNUTS_HIP_EXPORT int wn_stream_version(void);
/* The code-generation flags the library was compiled with (csrc/Makefile CODEGEN_FLAGS, a space-separated string):
 * device models compiled at run time are built with exactly these, so that a plugin and the library cannot drift. */
WALNUTS_HIP_EXPORT const char* wn_build_flags(void);
/* ... and the compiler it was built with (the "HIP version:" line of `hipcc --version`; "" if the build did not say) */
WALNUTS_HIP_EXPORT const char* wn_build_compiler(void);
/* Device models compiled at RUN time -- the device counterpart of handing the reference a host callable
 * (LOGP_CFUNC / a numba cfunc: python/src/walnutpie/walnutpy.cpp:131-132, pyfunc.py:216).  The model's five-line
 * translation unit is compiled against the installed headers (walnuts_amd/csrc) into a shared object of its own
 * (walnuts_amd.build_device_model: one hipcc -shared with -DWN_MODEL_PLUGIN and the ONE launch geometry
 * wn_geometry_for() names); loading that object (dlopen / ctypes.CDLL) runs its static initialiser, which enters the
 * model into THIS library's registry through wn_plugin_register_model.  wn_model_id(name) then resolves it and every
 * entry point takes the id.  A failed registration (id taken, headers of another build) is kept in wn_model_error()
 * and reported by the next wn_engine_create as a `config` error. */
WALNUTS_HIP_EXPORT int wn_plugin_register_model(const void* model_ops, const void* model_abi);
WALNUTS_HIP_EXPORT const char* wn_model_error(void);
WALNUTS_HIP_EXPORT void wn_model_clear_error(void);
WALNUTS_HIP_EXPORT int wn_geometry_for(int num_params, int waves_per_chain, int elems_per_lane,
                                       int preferred_elems_per_lane, int* waves_out, int* elems_per_lane_out,
                                       int* streaming_out, WalnutpyError** err);
/* wn_geometry_for answers for a model WITHOUT held streaming kernels.  The engine's choice also depends on the model
 * (one-pass gradients stream from 4 097 parameters on the kernels that hold the moving end in registers, the funnel's
 * kind from 8 193): wn_geometry_for_model is the ONE geometry wn_engine_create picks for a REGISTERED model;
 * wn_geometry_candidates lists every geometry it may pick over all model traits, as triples (waves per chain, elements
 * per lane, streaming) -- what a model compiled at run time, not registered yet, instantiates (at most 3). */
WALNUTS_HIP_EXPORT int wn_geometry_for_model(int model, int num_params, int waves_per_chain, int elems_per_lane,
                                             int* waves_out, int* elems_per_lane_out, int* streaming_out,
                                             WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_geometry_candidates(int num_params, int waves_per_chain, int elems_per_lane,
                                              int preferred_elems_per_lane, int* out_triples, int max_triples,
                                              int* count, WalnutpyError** err);

/* model_params: host pointer (copied). */
WALNUTS_HIP_EXPORT int wn_engine_create(wn_engine** out, int model, int num_params, const double* model_params,
                                        size_t num_chains, const wn_config* cfg, WalnutpyError** err);
WALNUTS_HIP_EXPORT void wn_engine_destroy(wn_engine* e);

/* InitConfig (config.hpp:74-185): positions [C*D], masses [C*D] (masses, not inverse
 * masses), step sizes [C]; host pointers. */
WALNUTS_HIP_EXPORT int wn_engine_set_positions(wn_engine* e, const double* positions, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_set_masses(wn_engine* e, const double* masses, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_set_step_sizes(wn_engine* e, const double* step_sizes, WalnutpyError** err);
/* device-side InitConfigBuilder steps (config.hpp:258-268,360-370,470-476 + util.hpp:242-303) on the
 * counter-based streams: positions ~ N(0, scale^2); mass = (1-s)*|grad| + s; step-size search */
WALNUTS_HIP_EXPORT int wn_engine_init_positions(wn_engine* e, uint64_t seed, uint32_t chain_offset, double scale,
                                                WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_init_masses_from_grad(wn_engine* e, double smoothing, WalnutpyError** err);
/* InitConfigBuilder::masses(logp_grad, s, average_masses = true) (config.hpp:371-380): after
 * wn_engine_init_masses_from_grad (or set_masses), replace every chain's masses by their geometric mean over the
 * chains of this engine. */
WALNUTS_HIP_EXPORT int wn_engine_average_masses(wn_engine* e, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_masses(wn_engine* e, double* out /*[C*D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_adapt_step(wn_engine* e, uint64_t seed, uint32_t chain_offset, WalnutpyError** err);
/* the same step-size search with the momentum normals [C*D] supplied by the host (lets a caller feed the
 * reference's own mt19937_64(seed_seq{seed,2}) stream, walnutpy.cpp:75-80) */
WALNUTS_HIP_EXPORT int wn_engine_adapt_step_with_normals(wn_engine* e, const double* normals, WalnutpyError** err);
/* counter-based generator: key = seed, chain ids chain_offset .. chain_offset+C-1
 * (the role of api.hpp:46-51's per-chain seed_seq{seed, m+1}) */
WALNUTS_HIP_EXPORT int wn_engine_seed(wn_engine* e, uint64_t seed, uint32_t chain_offset, WalnutpyError** err);
/* parity mode: reproduce the reference's own per-chain streams -- mt19937_64(seed_seq{seed, m+1}) with
 * libstdc++'s normal / uniform / bernoulli distributions (api.hpp:46-51, util.hpp:78-162) -- on the host and feed
 * them to every following transition.  One host round trip per transition: for small runs. */
WALNUTS_HIP_EXPORT int wn_engine_seed_reference_streams(wn_engine* e, uint64_t seed, WalnutpyError** err);
/* host-generated variates for the NEXT transition only (exact libstdc++ stream parity
 * runs): normals [C*D], canonical uniforms [C*u_per_chain] consumed in order. */
WALNUTS_HIP_EXPORT int wn_engine_set_variates(wn_engine* e, const double* normals, const double* uniforms,
                                              int u_per_chain, WalnutpyError** err);

/* AdaptiveWalnuts::operator() for all chains.  draws_dev: nullable DEVICE pointer; chain c's
 * position is written to draws_dev + c*draws_stride (doubles). */
WALNUTS_HIP_EXPORT int wn_engine_warmup_step(wn_engine* e, double* draws_dev, int64_t draws_stride,
                                             WalnutpyError** err);
/* AdaptiveWalnuts::sampler(): freeze step size, inverse mass and min micro steps. */
WALNUTS_HIP_EXPORT int wn_engine_freeze(wn_engine* e, WalnutpyError** err);
/* WalnutsSampler::operator() for all chains. */
WALNUTS_HIP_EXPORT int wn_engine_sample_step(wn_engine* e, double* draws_dev, int64_t draws_stride,
                                             WalnutpyError** err);
/* `transitions` consecutive AdaptiveWalnuts / WalnutsSampler transitions of every chain in ONE launch: the workgroup that
 * fetched a chain runs them back to back (the chains are independent, walnuts.hpp:682-692 called `transitions` times
 * per chain), so the per-launch tail -- the last chains finishing while the rest of the chip idles -- and the launch
 * itself are paid once per `transitions`.  Chain c's k-th position goes to draws_dev + c*draws_stride +
 * k*draws_transition_stride (doubles); the per-transition reports (wn_engine_get_depths, ...) show the LAST transition.
 * Same bits as `transitions` single steps.  Host-fed variates (reference streams, wn_engine_set_variates) cover one
 * transition: config error with transitions > 1. */
WALNUTS_HIP_EXPORT int wn_engine_warmup_steps(wn_engine* e, int transitions, double* draws_dev, int64_t draws_stride,
                                              int64_t draws_transition_stride, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_sample_steps(wn_engine* e, int transitions, double* draws_dev, int64_t draws_stride,
                                              int64_t draws_transition_stride, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_synchronize(wn_engine* e, WalnutpyError** err);
/* fails (generic error) if a transition of any chain since the previous check could not complete on the device (span
 * pool exhausted, host-fed variates exhausted); reading the flag clears it */
WALNUTS_HIP_EXPORT int wn_engine_check(wn_engine* e, WalnutpyError** err);

/* state -> host buffers */
WALNUTS_HIP_EXPORT int wn_engine_get_positions(wn_engine* e, double* out /*[C*D]*/, WalnutpyError** err);
/* after freeze: the sampler's inverse masses (walnuts.hpp:703-728); before: AdaptiveWalnuts::inv_mass(), the
 * estimate the next warmup transition integrates with (adaptive_walnuts.hpp:89-94) */
WALNUTS_HIP_EXPORT int wn_engine_get_inv_mass(wn_engine* e, double* out /*[C*D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_step_sizes(wn_engine* e, double* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_logp(wn_engine* e, double* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_min_micro(wn_engine* e, int32_t* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_depths(wn_engine* e, int32_t* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_grad_evals(wn_engine* e, int64_t* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_rng_draws(wn_engine* e, int32_t* out /*[C]*/, WalnutpyError** err);
/* The failure channel of a device model.  The reference traps a throwing model (NoExceptLogpGrad, util.hpp:336-346:
 * logp = -inf, grad = 0, handler.on_logp_exception); a device model cannot throw -- what a failing one produces is a
 * non-finite log density, and the leaf that meets it fails its energy test at every step size (walnuts.hpp:339-344),
 * the extension fails and the transition ends where it is (:543-545), exactly as logp = -inf does in the reference.
 * out[c] = 1 if an extension of chain c's LAST transition (of the last launch) failed that way -- which is also what a
 * finite energy error above max_hamiltonian_error at every step size, or a failed reversibility check, looks like: a
 * chain that reports it transition after transition is stuck at a point its model cannot leave.  Recording more (a
 * count of non-finite attempts) costs the headline kernel 2-8 %: profiles/r04/headline_attempts.md.  walnuts_hip.hpp
 * hands the flag to a handler's on_extension_failed, if it has one. */
WALNUTS_HIP_EXPORT int wn_engine_get_failed_extensions(wn_engine* e, int32_t* out /*[C]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_adam(wn_engine* e, double* out /*[C*6]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_get_estimator(wn_engine* e, double* draw_mean, double* draw_ssd,
                                               double* score_mean, double* score_ssd, double* weights /*[C*2]*/,
                                               WalnutpyError** err);
/* sum over chains of gradient evaluations so far (device-side reduction) */
WALNUTS_HIP_EXPORT int wn_engine_total_grad_evals(wn_engine* e, int64_t* out, WalnutpyError** err);

/* cross-chain monitors = the reference's controller loops, evaluated on the device.
 * wn_engine_rhat: sqrt(1 + var(chain lp means) / mean(chain lp sample variances)) over the sampling draws so far
 *   (sampler.hpp:132-145, WelfordAccumulator online_moments.hpp:22-86); _lp_sums/_lp_sq_dev are its two
 *   reduction stages for callers that all-reduce across GPUs in between.
 * wn_engine_warmup_spread: max over chains of the relative distance of the step size / of the mass vector from
 *   their geometric means over chains (adapt.hpp:193-221). */
WALNUTS_HIP_EXPORT int wn_engine_rhat(wn_engine* e, double* rhat, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_lp_sums(wn_engine* e, double* out3, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_lp_sq_dev(wn_engine* e, double mean_of_means, double* out1, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_warmup_spread(wn_engine* e, double* max_rel_diff_step, double* max_rel_diff_mass,
                                               WalnutpyError** err);
/* the same statistic in the two stages a multi-GPU driver all-reduces between: this engine's sums
 * over its chains of log step (1) and log mass ([D]) -> all-reduce SUM -> this engine's maxima given the sums over
 * all `total_chains` chains -> all-reduce MAX */
WALNUTS_HIP_EXPORT int wn_engine_warmup_sums(wn_engine* e, double* sum_log_step, double* colsum_log_mass /*[D]*/,
                                             WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_warmup_max_rel(wn_engine* e, double sum_log_step,
                                                const double* colsum_log_mass /*[D]*/, size_t total_chains,
                                                double* max_rel_diff_step, double* max_rel_diff_mass,
                                                WalnutpyError** err);

/* introspection */
WALNUTS_HIP_EXPORT int wn_engine_lanes(const wn_engine* e);        /* L = 64*NW: the reduction width   */
WALNUTS_HIP_EXPORT int wn_engine_dim_padded(const wn_engine* e);   /* Dp                                */
WALNUTS_HIP_EXPORT int wn_engine_is_streaming(const wn_engine* e); /* 1: vectors streamed from HBM      */
WALNUTS_HIP_EXPORT int wn_engine_held_tiles(const wn_engine* e);   /* streaming: pairs per lane of the moving end
                                                                      kept in registers (0: both ends streamed) */
WALNUTS_HIP_EXPORT int wn_engine_workgroups(const wn_engine* e);   /* persistent grid size              */
WALNUTS_HIP_EXPORT int wn_engine_chain_groups(const wn_engine* e); /* concurrent kernels per transition launch */
WALNUTS_HIP_EXPORT int wn_engine_lds_vectors(const wn_engine* e);  /* pool vectors resident in LDS      */
WALNUTS_HIP_EXPORT int64_t wn_engine_iteration(const wn_engine* e);
WALNUTS_HIP_EXPORT void* wn_engine_stream(const wn_engine* e);     /* the engine's main hipStream_t: everything but the
                                                                      transition launches of chain groups 1.. runs on
                                                                      it.  Work queued on it behind a transition is
                                                                      ordered after ALL groups only through
                                                                      wn_engine_release_stream / wn_engine_synchronize */
/* device pointer to the [C][Dp] position plane (for RCCL all-gather of draws) */
WALNUTS_HIP_EXPORT double* wn_engine_positions_device(const wn_engine* e);
/* HIP-event time of the last transition kernel launch, milliseconds */
WALNUTS_HIP_EXPORT int wn_engine_last_kernel_ms(wn_engine* e, float* ms, WalnutpyError** err);
/* per-launch HIP-event durations of the transition kernel since the last reset: wn_engine_timing_reset switches the
 * recording on (two events around every launch on the engine's stream; off by default, an event is a few
 * microseconds between two kernels); num_launches may exceed max_launches */
WALNUTS_HIP_EXPORT int wn_engine_timing_reset(wn_engine* e, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_kernel_times(wn_engine* e, float* ms_out, int max_launches, int* num_launches,
                                              WalnutpyError** err);
/* the same measurement with ONE pair of events for a whole region of launches (nothing between two kernels):
 * _region_begin records the start on the engine's stream, _region_ms records the end, waits for it and returns the
 * elapsed time and the number of transition launches in between (average launch duration = total / launches, the few
 * microseconds between two launches included) */
WALNUTS_HIP_EXPORT int wn_engine_region_begin(wn_engine* e, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_region_ms(wn_engine* e, float* total_ms, int* launches, WalnutpyError** err);
/* run on a caller-owned hipStream_t (e.g. the framework's current stream, so that RCCL collectives on the
 * draws are ordered after the kernels without host synchronisation) */
WALNUTS_HIP_EXPORT int wn_engine_set_stream(wn_engine* e, void* stream, WalnutpyError** err);
/* The same ordering without giving up the engine's own streams (and with them its chain groups): no host
 * synchronisation in either direction.
 *   _wait_stream:    the engine's next transition launches wait for everything `stream` holds at the time of the call
 *                    (e.g. the collective that last read the draw buffer the launch is about to overwrite);
 *   _release_stream: `stream` waits for every transition launch made so far (e.g. before the collective on the draws
 *                    is issued on it).
 * `stream` is a hipStream_t of the engine's device. */
WALNUTS_HIP_EXPORT int wn_engine_wait_stream(wn_engine* e, void* stream, WalnutpyError** err);
/* ... the same for ONE recorded hipEvent_t instead of everything a stream holds (a double-buffered consumer waits for
 * the copy of the block about to be overwritten, not for the copy it has just queued) */
WALNUTS_HIP_EXPORT int wn_engine_wait_event(wn_engine* e, void* event, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_engine_release_stream(wn_engine* e, void* stream, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_lanes_for_dim(int num_params, int waves_per_chain, int elems_per_lane);
/* the default launch geometry depends on the model (heavier gradients prefer one wavefront per chain) */
WALNUTS_HIP_EXPORT int wn_lanes_for_model_dim(int model, int num_params, int waves_per_chain, int elems_per_lane);
/* (internal, for the tests) the reference's host-side initial streams as walnutpie_sample_device produces them:
 * count_per_chain normals x scale per chain from mt19937_64(seed_seq{seed, stream}) through libstdc++'s
 * normal_distribution -- one distribution for all chains (initial positions, config.hpp:259-268) or a fresh one per
 * chain (step-size search, util.hpp:288) */
/* (internal, for the tests) the kernels' range-free square root (csrc/wn_devmath.h sqrt_normal) evaluated on the device
 * for n host arguments; checked != 0: the variant that patches 0 / inf back in.  -> 0, or -1 */
WALNUTS_HIP_EXPORT int wn_internal_sqrt_probe(const double* x, double* y, size_t n, int checked);
WALNUTS_HIP_EXPORT void wn_internal_reference_normals(unsigned int seed, unsigned int stream, size_t num_chains,
                                                      size_t count_per_chain, int fresh_per_chain, double scale,
                                                      double* out);
/* (internal) allocates the error object handed back through WalnutpyError** */
WALNUTS_HIP_EXPORT void* wn_internal_make_error(const char* msg, int type);

/* ---- posterior summaries over device-resident draws --------------------------------------------------------
 * include/walnutpie/summary.hpp:370-768 (mean, sample_variance, sample_standard_deviation, quantiles,
 * autocovariance, r_hat, effective_sample_size, monte_carlo_standard_error) for ragged collections of chains, as
 * MarkovChainsSplit / MarkovChainsUnified (summary.hpp:119-356) describe them.  The draws stay in HBM; results
 * ([dims] or [k][dims] row-major doubles) come back to host buffers the caller owns.  Same preconditions and
 * std::invalid_argument messages as the reference (-> config errors). */

/* Borrow draws that already live on the device: chain c's n-th draw is the `dims` doubles at
 * draws_dev + c * chain_stride + n * dims -- the layout walnutpie_sample_device fills and wn_engine_*_step writes
 * when called with (draws + n * dims, chain_stride).  lengths[c] <= max_len draws are valid in chain c (NULL: all
 * max_len).  `stream` (hipStream_t, may be NULL) orders the kernels after the producer. */
WALNUTS_HIP_EXPORT int wn_chains_view(wn_chains** out, const double* draws_dev, size_t num_chains, size_t max_len,
                                      size_t dims, int64_t chain_stride, const int64_t* lengths, int device,
                                      void* stream, WalnutpyError** err);
/* The same view, but the handle takes OWNERSHIP of draws_dev (a hipMalloc'd block of num_chains * chain_stride
 * doubles): wn_chains_destroy frees it.  What walnutpie_sample_device_resident hands back. */
WALNUTS_HIP_EXPORT int wn_chains_adopt(wn_chains** out, double* draws_dev, size_t num_chains, size_t max_len,
                                       size_t dims, int64_t chain_stride, const int64_t* lengths, int device,
                                       WalnutpyError** err);
/* Copy host draws in MarkovChainsUnified layout (chains stacked, [sum sizes][dims] row-major) to the device. */
WALNUTS_HIP_EXPORT int wn_chains_upload(wn_chains** out, const double* draws_host, size_t dims, const int64_t* sizes,
                                        size_t num_chains, int device, WalnutpyError** err);
WALNUTS_HIP_EXPORT void wn_chains_destroy(wn_chains* chains);
WALNUTS_HIP_EXPORT size_t wn_chains_num_chains(const wn_chains* chains);
WALNUTS_HIP_EXPORT size_t wn_chains_dims(const wn_chains* chains);
WALNUTS_HIP_EXPORT size_t wn_chains_num_draws(const wn_chains* chains);
WALNUTS_HIP_EXPORT size_t wn_chains_min_chain_size(const wn_chains* chains);
/* where the draws are: chain 0's first draw on device wn_chains_device() (chain c's n-th draw: see wn_chains_view) */
WALNUTS_HIP_EXPORT const double* wn_chains_device_draws(const wn_chains* chains);
WALNUTS_HIP_EXPORT int wn_chains_device(const wn_chains* chains);

WALNUTS_HIP_EXPORT int wn_summary_mean(wn_chains* chains, double* out /*[D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_summary_sample_variance(wn_chains* chains, double* out /*[D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_summary_sample_standard_deviation(wn_chains* chains, double* out /*[D]*/,
                                                            WalnutpyError** err);
/* out[k][d] = sorted_d[lo] + frac * (sorted_d[hi] - sorted_d[lo]) (summary.hpp:507-511), exact order statistics */
WALNUTS_HIP_EXPORT int wn_summary_quantiles(wn_chains* chains, const double* probs, size_t num_probs,
                                            double* out /*[num_probs*D]*/, WalnutpyError** err);
/* all lags of every chain, stacked like the draws: out[(first row of chain c) + lag][d] */
WALNUTS_HIP_EXPORT int wn_summary_autocovariance(wn_chains* chains, double* out /*[num_draws*D]*/,
                                                 WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_summary_r_hat(wn_chains* chains, double* out /*[D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_summary_effective_sample_size(wn_chains* chains, double* out /*[D]*/, WalnutpyError** err);
WALNUTS_HIP_EXPORT int wn_summary_monte_carlo_standard_error(wn_chains* chains, double* out /*[D]*/,
                                                             WalnutpyError** err);

/* ---- the reference's summary entry points, same symbols and arguments (walnutpy.cpp:333-369) -------------
 * draws: the stacked chains as Eigen::Map<const MatrixXd>(draws, num_draws, num_params) reads them
 * (walnutpy.cpp:89), i.e. COLUMN-major num_draws x num_params; lengths[num_chains] sum to num_draws;
 * out[num_params].  Computed on the device by wn_summary_* (upload, summarise, copy back). */
WALNUTS_HIP_EXPORT int walnutpie_ess(const double* draws, int num_draws, int num_params, const int* lengths,
                                     int num_chains, double* out, WalnutpyError** err);
WALNUTS_HIP_EXPORT int walnutpie_r_hat(const double* draws, int num_draws, int num_params, const int* lengths,
                                       int num_chains, double* out, WalnutpyError** err);
WALNUTS_HIP_EXPORT int walnutpie_mcse(const double* draws, int num_draws, int num_params, const int* lengths,
                                      int num_chains, double* out, WalnutpyError** err);

#ifdef __cplusplus
}
#endif
#endif /* WALNUTS_HIP_H */
