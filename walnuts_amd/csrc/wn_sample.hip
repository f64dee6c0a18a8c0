// wn_sample.hip -- walnutpie_sample_device(): the device-model sibling of the reference's ctypes
// entry point walnutpie_sample_cfunc (python/src/walnutpie/walnutpy.cpp:134-222 -> run_sampler
// :20-84 -> walnutpie::walnuts, api.hpp:35-69), built on the batched engine.
//
// What is kept: argument list after the model, validation and error types, output layout
// out[C][max_sampling + max_warmup*save_warmup][num_params] written chain-major, final_lengths
// [C warmup rows | C sampling rows], stepsize_out[C], inv_metric_out[C][num_params], progress lines.
// Initial positions and the step-size search consume the same libstdc++ streams as the reference
// (seed_seq{seed,1} and seed_seq{seed,2}; walnutpy.cpp:187-189,75-76): they are generated on the host,
// so those inputs are bit-identical to the reference's.
// What differs (documented in INTEGRATION.md): chains advance in lock step, so the controllers' stopping
// rules (adapt.hpp:172-229, sampler.hpp:117-158) are evaluated on whole iterations and every chain gets the same
// length (the reference's thread-per-chain workers stop wherever they happen to be), and the
// per-chain trajectory randomness comes from the counter-based generator keyed by seed+id+num_chains
// (walnutpy.cpp:82) instead of mt19937_64.
#include "wn_hip.h"

#include <cmath>
#include <iomanip>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/walnuts_hip.h"

extern "C" int wn_engine_adapt_step_with_normals(wn_engine* e, const double* normals, WalnutpyError** err);
extern "C" void* wn_internal_make_error(const char* msg, int type);

namespace {

struct EngineGuard {
  wn_engine* e = nullptr;
  ~EngineGuard() {
    if (e) wn_engine_destroy(e);
  }
};

[[noreturn]] void rethrow(WalnutpyError* err) {
  std::string msg = walnutpie_get_error_message(err);
  const WalnutpyErrorType t = walnutpie_get_error_type(err);
  walnutpie_destroy_error(err);
  if (t == config) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
#define WN_CALL(expr)                   \
  do {                                  \
    WalnutpyError* call_err_ = nullptr; \
    if ((expr) != 0) rethrow(call_err_); \
  } while (0)

void finite_positive(double v, const char* name) {  // validate.hpp: validate_finite_positive
  if (!(std::isfinite(v) && v > 0)) throw std::invalid_argument(std::string(name) + " must be finite and > 0");
}
void probability(double v, const char* name) {  // validate.hpp: validate_probability
  if (!(v > 0 && v < 1)) throw std::invalid_argument(std::string(name) + " must be in (0, 1)");
}

struct Printer {  // python/src/walnutpie/handlers.hpp:17-59
  PRINT_CALLBACK print;
  size_t refresh;
  size_t iter = 0;
  bool in_warmup = true;
  void progress(size_t num_chains) {
    ++iter;
    if (refresh == 0 || print == nullptr || iter % refresh != 0) return;
    for (size_t c = 0; c < num_chains; ++c) {
      std::stringstream ss;
      ss << "Chain [" << (c + 1) << "]: Iteration " << iter << "\t" << (in_warmup ? "(Warmup)" : "(Sampling)")
         << std::endl;
      const std::string s = ss.str();
      print(s.c_str(), s.length(), false);
    }
  }
};

}  // namespace

static int sample_device_impl(
    bool reference_streams,
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    WalnutpyError** err) {
  try {
    // walnutpy.cpp:151-160
    if (refresh < 0) {
      std::stringstream msg;
      msg << "refresh must be non-negative, was " << refresh;
      throw std::invalid_argument(msg.str());
    }
    if (num_params < 1) throw std::invalid_argument("num_params must be in {1, 2, ... }");
    if (max_sampling_iter < 0 || max_warmup_iter < 0) throw std::invalid_argument("iteration counts must be >= 0");
    const size_t rows = static_cast<size_t>(max_sampling_iter) + (save_warmup ? static_cast<size_t>(max_warmup_iter) : 0);
    const size_t draws_offset = static_cast<size_t>(num_params) * rows;
    if (out_size < num_chains * draws_offset) {
      std::stringstream ss;
      ss << "Output buffer too small. Expected at least " << num_chains << " chains of " << draws_offset
         << " doubles, got " << out_size;
      throw std::runtime_error(ss.str());
    }
    // WarmupConfigBuilder / SamplingConfigBuilder validation (config.hpp:656-850, 978-1059)
    if (static_cast<size_t>(min_warmup_iter) > static_cast<size_t>(max_warmup_iter))
      throw std::invalid_argument("min_iter cannot be greater than than max_iter");
    finite_positive(step_size_converge_tol, "step_size_converge_tol");
    finite_positive(mass_converge_tol, "mass_converge_tol");
    finite_positive(mass_init_count, "mass_init_count");
    finite_positive(mass_additive_smoothing, "mass_additive_smoothing");
    finite_positive(max_macro_steps_target, "max_macro_steps_target");
    probability(step_accept_rate_target, "step_accept_rate_target");
    finite_positive(step_learning_rate, "step_learning_rate");
    probability(step_gradient_decay, "step_gradient_decay");
    probability(step_sq_gradient_decay, "step_sq_gradient_decay");
    finite_positive(step_stabilization, "step_stabilization");
    probability(step_learn_rate_decay, "step_learn_rate_decay");
    if (static_cast<size_t>(min_sampling_iter) > static_cast<size_t>(max_sampling_iter))
      throw std::invalid_argument("min_iter must be <= max_iter");
    if (!(std::isfinite(rhat_converge_tol) && rhat_converge_tol > 1))
      throw std::invalid_argument("rhat_convergence_tol must be finite and > 1");
    finite_positive(max_hamiltonian_error, "max_hamiltonian_error");
    if (min_micro_steps < 1) throw std::invalid_argument("min_micro_steps must be in {1, 2, ... }");
    finite_positive(step_size_init, "step size");  // config.hpp:222

    wn_config cfg;
    wn_default_config(&cfg);
    cfg.max_trajectory_doublings = max_trajectory_doublings;
    cfg.max_step_halvings = max_step_halvings;
    cfg.min_micro_steps = min_micro_steps;
    cfg.max_hamiltonian_error = max_hamiltonian_error;
    cfg.mass_init_count = mass_init_count;
    cfg.max_macro_steps_target = max_macro_steps_target;
    cfg.step_accept_rate_target = step_accept_rate_target;
    cfg.step_learning_rate = step_learning_rate;
    cfg.step_gradient_decay = step_gradient_decay;
    cfg.step_sq_gradient_decay = step_sq_gradient_decay;
    cfg.step_stabilization = step_stabilization;
    cfg.step_learn_rate_decay = step_learn_rate_decay;

    EngineGuard guard;
    WN_CALL(wn_engine_create(&guard.e, model, num_params, model_params, num_chains, &cfg, &call_err_));
    wn_engine* e = guard.e;
    const size_t D = static_cast<size_t>(num_params);

    // initial positions (walnutpy.cpp:176-190)
    {
      std::vector<double> pos(num_chains * D);
      if (inits != nullptr) {
        for (size_t i = 0; i < pos.size(); ++i) {
          if (!std::isfinite(inits[i])) throw std::invalid_argument("positions must be finite");
          pos[i] = inits[i];
        }
      } else {
        finite_positive(init_radius, "init_scale");
        std::seed_seq ss{seed, 1u};
        std::mt19937_64 rng(ss);
        std::normal_distribution<double> normal(0.0, 1.0);  // one detail::Random for all chains, config.hpp:261-266
        for (size_t c = 0; c < num_chains; ++c) {
          for (size_t i = 0; i < D; ++i) pos[c * D + i] = normal(rng);
          for (size_t i = 0; i < D; ++i) pos[c * D + i] *= init_radius;
        }
      }
      WN_CALL(wn_engine_set_positions(e, pos.data(), &call_err_));
    }
    // masses (walnutpy.cpp:64-73).  NB the reference hands init_inv_metric to the builder's
    // masses(): reproduced as is.
    if (init_inv_metric != nullptr) {
      WN_CALL(wn_engine_set_masses(e, init_inv_metric, &call_err_));
    } else {
      WN_CALL(wn_engine_init_masses_from_grad(e, mass_additive_smoothing, &call_err_));
    }
    {
      std::vector<double> steps(num_chains, step_size_init);
      WN_CALL(wn_engine_set_step_sizes(e, steps.data(), &call_err_));
    }
    // adapt_step_build with mt19937_64(seed_seq{seed, 2}) (walnutpy.cpp:75-80): the engine is shared by
    // the chains in order and each chain starts a fresh normal distribution (util.hpp:288)
    {
      std::seed_seq ss{seed, 2u};
      std::mt19937_64 rng(ss);
      std::vector<double> z(num_chains * D);
      for (size_t c = 0; c < num_chains; ++c) {
        std::normal_distribution<double> normal(0.0, 1.0);
        for (size_t i = 0; i < D; ++i) z[c * D + i] = normal(rng);
      }
      WN_CALL(wn_engine_adapt_step_with_normals(e, z.data(), &call_err_));
    }
    // walnutpy.cpp:82: walnuts<mt19937_64>(seed + id + num_chains, ...)
    if (reference_streams) {
      WN_CALL(wn_engine_seed_reference_streams(e, static_cast<uint64_t>(seed) + id + num_chains, &call_err_));
    } else {
      WN_CALL(wn_engine_seed(e, static_cast<uint64_t>(seed) + id + num_chains, 0u, &call_err_));
    }

    // draws stay on the device in the caller's layout and come back in one copy
    double* d_out = nullptr;
    struct Free {
      double** p;
      ~Free() {
        if (*p) (void)hipFree(*p);
      }
    } free_out{&d_out};
    if (num_chains * draws_offset > 0) {
      if (hipMalloc(reinterpret_cast<void**>(&d_out), num_chains * draws_offset * sizeof(double)) != hipSuccess)
        throw std::runtime_error("cannot allocate the device draw buffer");
    }
    Printer printer{print, static_cast<size_t>(refresh)};
    size_t written = 0;
    for (int it = 1; it <= max_warmup_iter; ++it) {  // AdaptWorker loop, adapt.hpp:116-127
      double* dst = save_warmup ? d_out + written * D : nullptr;
      WN_CALL(wn_engine_warmup_step(e, dst, static_cast<int64_t>(draws_offset), &call_err_));
      if (save_warmup) ++written;
      printer.progress(num_chains);
      // controller_loop (adapt.hpp:172-229) on the snapshots published every publish_stride = 5 iterations
      if (it >= min_warmup_iter && it < max_warmup_iter && it % 5 == 0) {
        double rel_step = 0, rel_mass = 0;
        WN_CALL(wn_engine_warmup_spread(e, &rel_step, &rel_mass, &call_err_));
        if (rel_mass <= mass_converge_tol && rel_step <= step_size_converge_tol) break;
      }
    }
    const size_t written_warmup = written;
    WN_CALL(wn_engine_freeze(e, &call_err_));  // on_warmup_complete, handlers.hpp:91-101
    printer.in_warmup = false;
    if (stepsize_out != nullptr) WN_CALL(wn_engine_get_step_sizes(e, stepsize_out, &call_err_));
    if (inv_metric_out != nullptr) WN_CALL(wn_engine_get_inv_mass(e, inv_metric_out, &call_err_));
    for (int it = 1; it <= max_sampling_iter; ++it) {  // ChainWorker loop, sampler.hpp:82-93
      WN_CALL(wn_engine_sample_step(e, d_out + written * D, static_cast<int64_t>(draws_offset), &call_err_));
      ++written;
      printer.progress(num_chains);
      // controller_loop (sampler.hpp:117-158): R-hat of the log density once every chain has min_iter draws
      if (it >= min_sampling_iter && it >= 2 && it < max_sampling_iter && num_chains > 1) {
        double rhat = 0;
        WN_CALL(wn_engine_rhat(e, &rhat, &call_err_));
        if (print != nullptr && refresh != 0) {
          std::stringstream ss;
          ss << "Controller: R-hat at " << std::setprecision(10) << rhat << std::endl;  // handlers.hpp:160-176
          const std::string msg = ss.str();
          print(msg.c_str(), msg.length(), false);
        }
        if (rhat <= rhat_converge_tol) break;
      }
    }
    WN_CALL(wn_engine_check(e, &call_err_));
    if (num_chains * draws_offset > 0) {
      if (hipMemcpyAsync(out, d_out, num_chains * draws_offset * sizeof(double), hipMemcpyDeviceToHost,
                         reinterpret_cast<hipStream_t>(wn_engine_stream(e))) != hipSuccess ||
          hipStreamSynchronize(reinterpret_cast<hipStream_t>(wn_engine_stream(e))) != hipSuccess)
        throw std::runtime_error("copying draws to the host failed");
    }
    for (size_t c = 0; c < num_chains; ++c) {  // walnutpy.cpp:215-218
      final_lengths[c] = static_cast<int>(written_warmup);
      final_lengths[c + num_chains] = static_cast<int>(written - written_warmup);
    }
    return 0;
  } catch (const std::invalid_argument& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), config));
  } catch (const std::exception& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), generic));
  } catch (...) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("Unknown error", generic));
  }
  return -1;
}

#define WN_SAMPLE_ARGS                                                                                            \
  model, model_params, num_params, inits, num_chains, seed, id, init_radius, init_inv_metric, min_warmup_iter,   \
      max_warmup_iter, min_sampling_iter, max_sampling_iter, max_trajectory_doublings, max_step_halvings,        \
      min_micro_steps, max_hamiltonian_error, step_size_converge_tol, mass_converge_tol, rhat_converge_tol,      \
      mass_init_count, mass_additive_smoothing, max_macro_steps_target, step_size_init, step_accept_rate_target, \
      step_learning_rate, step_gradient_decay, step_sq_gradient_decay, step_stabilization, step_learn_rate_decay, \
      save_warmup, out, out_size, final_lengths, stepsize_out, inv_metric_out, refresh, print, err
#define WN_SAMPLE_PARAMS                                                                                          \
  int model, const double *model_params, int num_params, const double *inits, size_t num_chains,                 \
      unsigned int seed, unsigned int id, double init_radius, const double *init_inv_metric, int min_warmup_iter, \
      int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,           \
      int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,   \
      double mass_converge_tol, double rhat_converge_tol, double mass_init_count,                                \
      double mass_additive_smoothing, double max_macro_steps_target, double step_size_init,                      \
      double step_accept_rate_target, double step_learning_rate, double step_gradient_decay,                     \
      double step_sq_gradient_decay, double step_stabilization, double step_learn_rate_decay, bool save_warmup,  \
      double *out, size_t out_size, int *final_lengths, double *stepsize_out, double *inv_metric_out,            \
      int refresh, PRINT_CALLBACK print, WalnutpyError **err

extern "C" int walnutpie_sample_device(WN_SAMPLE_PARAMS) { return sample_device_impl(false, WN_SAMPLE_ARGS); }
extern "C" int walnutpie_sample_device_reference_streams(WN_SAMPLE_PARAMS) {
  return sample_device_impl(true, WN_SAMPLE_ARGS);
}
