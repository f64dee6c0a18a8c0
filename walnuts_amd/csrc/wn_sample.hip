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
#include <csignal>
#include <cstdlib>
#include <cstring>
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

// python/src/walnutpie/interrupts.hpp:34-102: while a sampling call runs, SIGINT sets a flag instead of killing
// the process (SA_RESETHAND: a second Ctrl-C gets the previous disposition back); the previous handler is restored
// on the way out.  The reference's controllers poll the flag (adapt.hpp:227, sampler.hpp:154); here it is polled
// between the launches of two iterations, and a raised flag ends the call with error type `interrupt`
// (errors.hpp:42-47: an empty message).
struct InterruptException {};
volatile std::sig_atomic_t wn_interrupted = 0;
class InterruptGuard {
 public:
  InterruptGuard() {
    wn_interrupted = 0;
    std::memset(&custom_, 0, sizeof(custom_));
    sigemptyset(&custom_.sa_mask);
    sigaddset(&custom_.sa_mask, SIGINT);
    custom_.sa_flags = SA_RESETHAND;
    custom_.sa_handler = &InterruptGuard::on_signal;
    sigaction(SIGINT, &custom_, &before_);
  }
  ~InterruptGuard() { sigaction(SIGINT, &before_, nullptr); }
  InterruptGuard(const InterruptGuard&) = delete;
  InterruptGuard& operator=(const InterruptGuard&) = delete;
  void throw_if_interrupted() const {
    if (wn_interrupted) throw InterruptException{};
  }

 private:
  static void on_signal(int) { wn_interrupted = 1; }
  struct sigaction before_, custom_;
};

// The draw sink (handlers.hpp:63-116 writes every draw straight into the caller's buffer, whatever its size).
// The device writes the draws of up to `span` consecutive iterations of all chains into one of two staging blocks
// [C][span][D]; a full block goes to the caller's out[C][rows][D] as ONE strided copy on a second stream while the
// next iterations fill the other block, so the device never holds more than two blocks of draws and
// [C][T][D] may exceed HBM (65 536 chains x 1 000 draws x 1 024 = 537 GB).
class DrawSink {
 public:
  DrawSink(size_t chains, size_t rows, size_t dim, double* out, hipStream_t compute)
      : C_(chains), rows_(rows), D_(dim), out_(out), compute_(compute) {
    if (C_ * rows_ * D_ == 0) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = size_t{1} << 30;
    size_t budget = free_b / 4;  // both blocks together: a quarter of what the engine left
    if (const char* env = std::getenv("WALNUTS_AMD_DRAW_STAGING_BYTES")) budget = std::strtoull(env, nullptr, 10);
    const size_t per_iter = C_ * D_ * sizeof(double);
    span_ = std::max<size_t>(1, std::min(rows_, budget / 2 / per_iter));
    for (int b = 0; b < 2; ++b) {
      if (hipMalloc(reinterpret_cast<void**>(&block_[b]), C_ * span_ * D_ * sizeof(double)) != hipSuccess)
        throw std::runtime_error("cannot allocate the device draw staging buffer");
      if (hipEventCreateWithFlags(&filled_[b], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&drained_[b], hipEventDisableTiming) != hipSuccess)
        throw std::runtime_error("cannot create the draw-sink events");
    }
    if (hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking) != hipSuccess)
      throw std::runtime_error("cannot create the draw-sink stream");
  }
  ~DrawSink() {
    for (int b = 0; b < 2; ++b) {
      if (block_[b]) (void)hipFree(block_[b]);
      if (filled_[b]) (void)hipEventDestroy(filled_[b]);
      if (drained_[b]) (void)hipEventDestroy(drained_[b]);
    }
    if (copy_) (void)hipStreamDestroy(copy_);
  }
  size_t stride() const { return span_ * D_; }  // doubles between two chains' rows in a staging block
  // where the next iteration's draws go (device pointer of chain 0's row)
  double* next_row() {
    if (fill_ == 0 && busy_[cur_]) {  // the block still feeds a copy: the kernels must not overwrite it yet
      if (hipStreamWaitEvent(compute_, drained_[cur_], 0) != hipSuccess) throw std::runtime_error("draw sink: wait failed");
      busy_[cur_] = false;
    }
    return block_[cur_] + fill_ * D_;
  }
  // the iteration launched into next_row() is queued on the compute stream
  void row_done() {
    ++fill_;
    ++written_;
    if (fill_ == span_) flush();
  }
  size_t written() const { return written_; }
  // everything written so far is in the caller's buffer when this returns
  void finish() {
    if (fill_ > 0) flush();
    if (copy_ && hipStreamSynchronize(copy_) != hipSuccess) throw std::runtime_error("copying draws to the host failed");
  }

 private:
  void flush() {
    const size_t first = written_ - fill_;
    if (hipEventRecord(filled_[cur_], compute_) != hipSuccess ||
        hipStreamWaitEvent(copy_, filled_[cur_], 0) != hipSuccess ||
        hipMemcpy2DAsync(out_ + first * D_, rows_ * D_ * sizeof(double), block_[cur_], span_ * D_ * sizeof(double),
                         fill_ * D_ * sizeof(double), C_, hipMemcpyDeviceToHost, copy_) != hipSuccess ||
        hipEventRecord(drained_[cur_], copy_) != hipSuccess)
      throw std::runtime_error("copying draws to the host failed");
    busy_[cur_] = true;
    cur_ ^= 1;
    fill_ = 0;
  }
  size_t C_, rows_, D_;
  double* out_;
  hipStream_t compute_, copy_ = nullptr;
  double* block_[2] = {nullptr, nullptr};
  hipEvent_t filled_[2] = {nullptr, nullptr}, drained_[2] = {nullptr, nullptr};
  bool busy_[2] = {false, false};
  size_t span_ = 1, fill_ = 0, written_ = 0;
  int cur_ = 0;
};

struct Printer {  // python/src/walnutpie/handlers.hpp:17-59
  PRINT_CALLBACK print;
  size_t refresh;
  size_t iter = 0;
  bool in_warmup = true;
  void progress(size_t num_chains) {
    ++iter;
    if (refresh == 0 || print == nullptr || iter % refresh != 0) return;
    if (num_chains > 16) {  // lock-step chains: one line for all of them instead of 65 536 callbacks per refresh
      std::stringstream ss;
      ss << "Chains [1-" << num_chains << "]: Iteration " << iter << "\t" << (in_warmup ? "(Warmup)" : "(Sampling)")
         << std::endl;
      const std::string s = ss.str();
      print(s.c_str(), s.length(), false);
      return;
    }
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

    InterruptGuard interrupt;  // walnutpy.cpp: interrupt::walnutpy_interrupt_handler on the stack of the call
    DrawSink sink(num_chains, rows, D, out, reinterpret_cast<hipStream_t>(wn_engine_stream(e)));
    Printer printer{print, static_cast<size_t>(refresh)};
    for (int it = 1; it <= max_warmup_iter; ++it) {  // AdaptWorker loop, adapt.hpp:116-127
      interrupt.throw_if_interrupted();
      double* dst = save_warmup ? sink.next_row() : nullptr;
      WN_CALL(wn_engine_warmup_step(e, dst, static_cast<int64_t>(sink.stride()), &call_err_));
      if (save_warmup) sink.row_done();
      printer.progress(num_chains);
      // controller_loop (adapt.hpp:172-229) on the snapshots published every publish_stride = 5 iterations
      if (it >= min_warmup_iter && it < max_warmup_iter && it % 5 == 0) {
        double rel_step = 0, rel_mass = 0;
        WN_CALL(wn_engine_warmup_spread(e, &rel_step, &rel_mass, &call_err_));
        if (rel_mass <= mass_converge_tol && rel_step <= step_size_converge_tol) break;
      }
    }
    const size_t written_warmup = sink.written();
    WN_CALL(wn_engine_freeze(e, &call_err_));  // on_warmup_complete, handlers.hpp:91-101
    printer.in_warmup = false;
    if (stepsize_out != nullptr) WN_CALL(wn_engine_get_step_sizes(e, stepsize_out, &call_err_));
    if (inv_metric_out != nullptr) WN_CALL(wn_engine_get_inv_mass(e, inv_metric_out, &call_err_));
    for (int it = 1; it <= max_sampling_iter; ++it) {  // ChainWorker loop, sampler.hpp:82-93
      interrupt.throw_if_interrupted();
      WN_CALL(wn_engine_sample_step(e, sink.next_row(), static_cast<int64_t>(sink.stride()), &call_err_));
      sink.row_done();
      printer.progress(num_chains);
      // controller_loop (sampler.hpp:117-158): R-hat of the log density once every chain has min_iter draws.  The
      // reference's controller looks on a 1 ms timer, not after every draw: here every `rhat_stride` iterations
      // (each look is a handful of small launches and a blocking read-back that would otherwise serialise every
      // transition with the host).
      constexpr int rhat_stride = 5;
      if (it >= min_sampling_iter && it >= 2 && it < max_sampling_iter && num_chains > 1 &&
          (it - min_sampling_iter) % rhat_stride == 0) {
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
    const size_t written = sink.written();
    WN_CALL(wn_engine_check(e, &call_err_));
    sink.finish();
    interrupt.throw_if_interrupted();
    for (size_t c = 0; c < num_chains; ++c) {  // walnutpy.cpp:215-218
      final_lengths[c] = static_cast<int>(written_warmup);
      final_lengths[c + num_chains] = static_cast<int>(written - written_warmup);
    }
    return 0;
  } catch (const InterruptException&) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("", interrupt));
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

// ---- walnutpie_ess / walnutpie_r_hat / walnutpie_mcse (walnutpy.cpp:333-369) ----------------------------------
// The reference's ctypes layer binds these three with (draws, num_draws, num_params, lengths, num_chains, out, err);
// `draws` is what Eigen::Map<const MatrixXd>(draws, num_draws, num_params) reads (walnutpy.cpp:89): a COLUMN-major
// num_draws x num_params matrix of the stacked chains.  Same symbols and arguments here: the draws are uploaded in
// the device layout, summarised by wn_summary_* (wn_summary.hip) and the num_params results copied back.
namespace {
template <class F>
int summary_shim(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains, double* out,
                 WalnutpyError** err, F summarise) {
  try {
    if (draws == nullptr || lengths == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    if (num_draws < 0 || num_params < 1 || num_chains < 1) throw std::invalid_argument("sizes must be positive");
    std::vector<int64_t> sizes(static_cast<size_t>(num_chains));
    int64_t total = 0;
    for (int m = 0; m < num_chains; ++m) {
      sizes[m] = lengths[m];
      total += lengths[m];
    }
    if (total != num_draws)  // MarkovChainsUnified, summary.hpp:266-280
      throw std::invalid_argument("The number of rows in draws and sum of chain_sizes must be equal.");
    const size_t N = static_cast<size_t>(num_draws), D = static_cast<size_t>(num_params);
    std::vector<double> rows(N * D);  // [draw][param]
    for (size_t d = 0; d < D; ++d)
      for (size_t n = 0; n < N; ++n) rows[n * D + d] = draws[d * N + n];
    wn_chains* ch = nullptr;
    WN_CALL(wn_chains_upload(&ch, rows.data(), D, sizes.data(), static_cast<size_t>(num_chains), 0, &call_err_));
    struct Guard {
      wn_chains* c;
      ~Guard() { wn_chains_destroy(c); }
    } guard{ch};
    WN_CALL(summarise(ch, out, &call_err_));
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
}  // namespace

extern "C" int walnutpie_ess(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                             double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err, wn_summary_effective_sample_size);
}
extern "C" int walnutpie_r_hat(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                               double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err, wn_summary_r_hat);
}
extern "C" int walnutpie_mcse(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                              double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err,
                      wn_summary_monte_carlo_standard_error);
}
