/* walnuts_hip.hpp -- C++20 surface of the GPU-resident many-chain Walnuts engine, header only, over the C ABI in
 * walnuts_hip.h.  It mirrors the reference's C++ interface for this path so that a program written against
 * `walnutpie::walnuts<RNG>(seed, chain_handlers, global_handler, interrupt, logp_grad, config)`
 * (include/walnutpie/api.hpp:35-69) switches by changing the namespace and naming a device model instead of a host
 * log-density callable:
 *
 *   reference (include/walnutpie/...)                      here (namespace walnuts_hip)
 *   ----------------------------------------------------   -------------------------------------------------------
 *   Eigen::VectorXd handed to handlers                     VectorView (converts to Eigen::VectorXd / std::vector)
 *   LogpGrad callable           concepts.hpp:258-262       DeviceModel (built-in device model id + parameters)
 *   SamplingConfig(+Builder)    config.hpp:885-1066        SamplingConfig(+Builder), same defaults and checks
 *   WarmupConfig(+Builder)      config.hpp:513-850         WarmupConfig(+Builder), same defaults and checks
 *   InitConfigBuilder           config.hpp:195-480         InitConfigBuilder: same verbs, executed ON THE DEVICE for
 *                                                          all chains at once; RNG objects become seeds
 *   WalnutsConfig               config.hpp:1089-1140       WalnutsConfig
 *   AdaptiveWalnuts<F,RNG,H>    adaptive_walnuts.hpp:182   BatchedAdaptiveWalnuts<H>: operator()() advances EVERY
 *                                                          chain one warmup transition; per-chain accessors
 *   WalnutsSampler<F,RNG,H>     walnuts.hpp:605-766        BatchedWalnutsSampler<H>
 *   Sampler concept             concepts.hpp:95-99         ChainView<H>: per-chain view, `double operator()()`,
 *                                                          `dim()`, usable in a reference-style chain loop
 *   ChainHandler/GlobalHandler/InterruptCallback           same member names and argument order
 *                               concepts.hpp:173-245
 *   mean / quantiles / r_hat / effective_sample_size ...   same names over MarkovChains (device-resident draws)
 *                               summary.hpp:119-768
 *   detail::adapt / detail::sample controllers             run_warmup / run_sampling: same stopping rules on whole
 *                               adapt.hpp:172-259,          iterations (all chains advance in lock step, so every
 *                               sampler.hpp:117-200         chain ends with the same length)
 *
 * Errors: the C ABI's `config` errors surface as std::invalid_argument, `interrupt` as walnuts_hip::Interrupted,
 * everything else as std::runtime_error -- the exception types the reference throws (errors.hpp:42-72). */
#ifndef WALNUTS_HIP_HPP
#define WALNUTS_HIP_HPP

#include <algorithm>
#include <cmath>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "walnuts_hip.h"

namespace walnuts_hip {

/** Raised when the C ABI reports an `interrupt` error (errors.hpp:10-14). */
class Interrupted : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

namespace detail {

[[noreturn]] inline void raise(WalnutpyError* err) {
  std::string msg = err ? walnutpie_get_error_message(err) : "unknown error";
  const WalnutpyErrorType type = err ? walnutpie_get_error_type(err) : generic;
  if (err) walnutpie_destroy_error(err);
  if (type == config) throw std::invalid_argument(msg);
  if (type == interrupt) throw Interrupted(msg);
  throw std::runtime_error(msg);
}

/** Call a C-ABI entry point whose last parameter is `WalnutpyError**`. */
template <class F, class... A>
inline void call(F f, A... a) {
  WalnutpyError* err = nullptr;
  if (f(a..., &err) != 0) raise(err);
}

inline void finite_positive(double v, const std::string& name) {  // validate.hpp:102-108
  if (!(std::isfinite(v) && v > 0)) throw std::invalid_argument(name + " must be finite and > 0");
}
inline void finite_gt1(double v, const std::string& name) {  // validate.hpp:86-92
  if (!(std::isfinite(v) && v > 1)) throw std::invalid_argument(name + " must be finite and > 1");
}
inline void probability(double v, const std::string& name) {
  if (!(v > 0 && v < 1)) throw std::invalid_argument(name + " must be in (0, 1)");
}

}  // namespace detail

/**
 * A borrowed, read-only run of doubles: what handlers receive where the reference hands them a
 * `const Eigen::VectorXd&`.  Converts implicitly to any owning vector type that can be built from a size and
 * exposes `double* data()` (Eigen::VectorXd, std::vector<double>), so handlers written for the reference keep
 * their signatures.
 */
class VectorView {
 public:
  VectorView() = default;
  VectorView(const double* p, std::size_t n) : p_(p), n_(n) {}
  std::size_t size() const noexcept { return n_; }
  const double* data() const noexcept { return p_; }
  double operator[](std::size_t i) const noexcept { return p_[i]; }
  double operator()(std::size_t i) const noexcept { return p_[i]; }
  const double* begin() const noexcept { return p_; }
  const double* end() const noexcept { return p_ + n_; }

  template <class T>
    requires(!std::same_as<T, VectorView> && std::constructible_from<T, std::size_t> &&
             requires(T& t) {
               { t.data() } -> std::convertible_to<double*>;
             })
  operator T() const {  // NOLINT: implicit on purpose
    T out(n_);
    std::copy(p_, p_ + n_, out.data());
    return out;
  }

 private:
  const double* p_ = nullptr;
  std::size_t n_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// handler concepts: member names and argument order of include/walnutpie/concepts.hpp:173-245
// ---------------------------------------------------------------------------------------------------------------
template <class H>
concept GlobalHandler = requires(H& h, double r_hat) {
  { h.on_r_hat(r_hat) } -> std::same_as<void>;
};
template <class H>
concept InterruptCallback = requires(const H& h) {
  { h.throw_if_interrupted() } -> std::same_as<void>;
};
template <class H>
concept SampleHandler = requires(H& h, VectorView position, double lp) {
  { h.on_sample(position, lp) } -> std::same_as<void>;
};
template <class H>
concept ChainHandler = SampleHandler<H> && requires(H& h, VectorView position, VectorView diag_inv_mass, double lp,
                                                    double step_size) {
  { h.on_warmup(position, lp, step_size, diag_inv_mass) } -> std::same_as<void>;
  { h.on_warmup_complete(step_size, diag_inv_mass) } -> std::same_as<void>;
};
/** Optional member of a chain handler: `on_extension_failed(position)`.  The reference's ErrorCallback
 *  (concepts.hpp:196-201: `on_logp_exception(position, exn)`) reports a THROWING model; a device model cannot throw --
 *  what a failing one produces is a non-finite log density, the leaf that meets it fails its energy test at every step
 *  size and the extension fails (walnuts.hpp:339-344,:543-545), as with logp = -inf in the reference.  The device flags
 *  the transitions in which an extension failed (wn_engine_get_failed_extensions; also set by a finite energy error
 *  above the bound at every step size) and the batched samplers pass the flag on -- with the chain's position after
 *  that transition -- to handlers that have this member. */
template <class H>
concept FailureCallback = requires(H& h, VectorView position) {
  { h.on_extension_failed(position) } -> std::same_as<void>;
};
/** One chain's sampler: concepts.hpp:95-99. */
template <class S>
concept Sampler = requires(S& s, const S& cs) {
  { s() } -> std::convertible_to<double>;
  { cs.dim() } -> std::convertible_to<std::size_t>;
};

/** A handler that ignores every event. */
struct NoOpChainHandler {
  void on_sample(VectorView, double) {}
  void on_warmup(VectorView, double, double, VectorView) {}
  void on_warmup_complete(double, VectorView) {}
};
struct NoOpGlobalHandler {
  void on_r_hat(double) {}
};
struct NeverInterrupted {
  void throw_if_interrupted() const {}
};

template <class H = NoOpChainHandler>
class BatchedAdaptiveWalnuts;
template <class H = NoOpChainHandler>
class BatchedWalnutsSampler;
template <class H = NoOpChainHandler>
class ChainView;

// ---------------------------------------------------------------------------------------------------------------
// the model: stands where the reference takes a LogpGrad callable.  Host callables cannot run inside the
// GPU-resident trajectory loop; the device evaluates one of the built-in models (wn_model).
// ---------------------------------------------------------------------------------------------------------------
struct DeviceModel {
  int id = WN_MODEL_STD_NORMAL;
  std::size_t dims = 0;
  std::vector<double> params;  // WN_MODEL_DIAG_NORMAL: sigma_sq[dims]

  static DeviceModel std_normal(std::size_t dims) { return {WN_MODEL_STD_NORMAL, dims, {}}; }
  static DeviceModel funnel(std::size_t dims) { return {WN_MODEL_FUNNEL, dims, {}}; }
  static DeviceModel diag_normal(std::vector<double> sigma_sq) {
    const std::size_t d = sigma_sq.size();
    return {WN_MODEL_DIAG_NORMAL, d, std::move(sigma_sq)};
  }
};

// ---------------------------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------------------------
class SamplingConfig {  // config.hpp:885-953
 public:
  std::size_t min_iter() const noexcept { return min_iter_; }
  std::size_t max_iter() const noexcept { return max_iter_; }
  std::size_t max_trajectory_doublings() const noexcept { return max_trajectory_doublings_; }
  std::size_t max_step_halvings() const noexcept { return max_step_halvings_; }
  double max_hamiltonian_error() const noexcept { return max_hamiltonian_error_; }
  std::size_t min_micro_steps() const noexcept { return min_micro_steps_; }
  double rhat_converge_tol() const noexcept { return rhat_converge_tol_; }

 private:
  friend class SamplingConfigBuilder;
  std::size_t min_iter_ = 50, max_iter_ = 1000;
  std::size_t max_trajectory_doublings_ = 5, max_step_halvings_ = 5;
  double max_hamiltonian_error_ = 0.5;
  std::size_t min_micro_steps_ = 1;
  double rhat_converge_tol_ = 1.01;
};

class SamplingConfigBuilder {  // config.hpp:967-1066
 public:
  SamplingConfigBuilder& min_max_iter(std::size_t lo, std::size_t hi) {
    if (lo > hi) throw std::invalid_argument("min_iter must be <= max_iter");
    c_.min_iter_ = lo;
    c_.max_iter_ = hi;
    return *this;
  }
  SamplingConfigBuilder& max_trajectory_doublings(std::size_t v) noexcept { return c_.max_trajectory_doublings_ = v, *this; }
  SamplingConfigBuilder& max_step_halvings(std::size_t v) noexcept { return c_.max_step_halvings_ = v, *this; }
  SamplingConfigBuilder& max_hamiltonian_error(double v) {
    detail::finite_positive(v, "max_hamiltonian_error");
    return c_.max_hamiltonian_error_ = v, *this;
  }
  SamplingConfigBuilder& min_micro_steps(std::size_t v) {
    if (v < 1) throw std::invalid_argument("min_micro_steps must be in {1, 2, ... }");
    return c_.min_micro_steps_ = v, *this;
  }
  SamplingConfigBuilder& rhat_converge_tol(double v) {
    detail::finite_gt1(v, "rhat_convergence_tol");
    return c_.rhat_converge_tol_ = v, *this;
  }
  SamplingConfig build() { return c_; }

 private:
  SamplingConfig c_;
};

class WarmupConfig {  // config.hpp:513-640
 public:
  std::size_t min_iter() const { return min_iter_; }
  std::size_t max_iter() const { return max_iter_; }
  double step_size_converge_tol() const { return step_size_converge_tol_; }
  double mass_converge_tol() const { return mass_converge_tol_; }
  double mass_init_count() const { return mass_init_count_; }
  double mass_additive_smoothing() const { return mass_additive_smoothing_; }
  double max_macro_steps_target() const { return max_macro_steps_target_; }
  double step_accept_rate_target() const { return step_accept_rate_target_; }
  double step_learning_rate() const { return step_learning_rate_; }
  double step_gradient_decay() const { return step_gradient_decay_; }
  double step_sq_gradient_decay() const { return step_sq_gradient_decay_; }
  double step_stabilization() const { return step_stabilization_; }
  double step_learn_rate_decay() const { return step_learn_rate_decay_; }
  /** The controller looks at the chains every `publish_stride` iterations (adapt.hpp:123-125). */
  std::size_t publish_stride() const { return publish_stride_; }

 private:
  friend class WarmupConfigBuilder;
  std::size_t min_iter_ = 50, max_iter_ = 1000;
  double step_size_converge_tol_ = 0.1, mass_converge_tol_ = 1.0;
  double mass_init_count_ = 4.0, mass_additive_smoothing_ = 1e-5, max_macro_steps_target_ = 15.0;
  double step_accept_rate_target_ = 0.8, step_learning_rate_ = 0.05, step_gradient_decay_ = 0.8;
  double step_sq_gradient_decay_ = 0.9, step_stabilization_ = 1e-4, step_learn_rate_decay_ = 0.5;
  std::size_t publish_stride_ = 5;
};

class WarmupConfigBuilder {  // config.hpp:646-850
 public:
  WarmupConfigBuilder& min_max_iter(std::size_t lo, std::size_t hi) {
    if (lo > hi) throw std::invalid_argument("min_iter cannot be greater than than max_iter");
    c_.min_iter_ = lo;
    c_.max_iter_ = hi;
    return *this;
  }
#define WN_HPP_SETTER(name, check)             \
  WarmupConfigBuilder& name(double v) {        \
    detail::check(v, #name);                   \
    c_.name##_ = v;                            \
    return *this;                              \
  }
  WN_HPP_SETTER(step_size_converge_tol, finite_positive)
  WN_HPP_SETTER(mass_converge_tol, finite_positive)
  WN_HPP_SETTER(mass_init_count, finite_positive)
  WN_HPP_SETTER(mass_additive_smoothing, finite_positive)
  WN_HPP_SETTER(max_macro_steps_target, finite_positive)
  WN_HPP_SETTER(step_accept_rate_target, probability)
  WN_HPP_SETTER(step_learning_rate, finite_positive)
  WN_HPP_SETTER(step_gradient_decay, probability)
  WN_HPP_SETTER(step_sq_gradient_decay, probability)
  WN_HPP_SETTER(step_stabilization, finite_positive)
  WN_HPP_SETTER(step_learn_rate_decay, probability)
#undef WN_HPP_SETTER
  WarmupConfigBuilder& publish_stride(std::size_t v) {
    if (v < 1) throw std::invalid_argument("publish_stride must be in {1, 2, ... }");
    return c_.publish_stride_ = v, *this;
  }
  WarmupConfig build() { return c_; }

 private:
  WarmupConfig c_;
};

/**
 * How the chains start (InitConfig, config.hpp:74-190).  Unlike the reference, which materialises one Eigen vector
 * per chain on the host, this records WHAT to do and the engine does it on the device for every chain at once
 * (wn_engine_init_positions / _init_masses_from_grad / _adapt_step).
 */
class InitConfig {
 public:
  std::size_t num_chains() const noexcept { return num_chains_; }
  std::size_t dims() const noexcept { return dims_; }

 private:
  friend class InitConfigBuilder;
  template <class H>
  friend class BatchedAdaptiveWalnuts;
  enum class Positions { zero, given, random };
  enum class Masses { one, given, from_gradient };
  std::size_t num_chains_ = 0, dims_ = 0;
  Positions positions_kind_ = Positions::zero;
  std::vector<double> positions_;  // [C*D] when given
  std::uint64_t positions_seed_ = 0;
  double positions_scale_ = 1.0;
  Masses masses_kind_ = Masses::one;
  std::vector<double> masses_;  // [C*D] when given
  double mass_smoothing_ = 1e-5;
  bool average_masses_ = false;
  std::vector<double> step_sizes_;
  bool adapt_step_ = false;
  std::uint64_t adapt_step_seed_ = 0;
};

class InitConfigBuilder {  // config.hpp:195-480
 public:
  InitConfigBuilder(std::size_t num_chains, std::size_t dims) {
    c_.num_chains_ = num_chains;
    c_.dims_ = dims;
    c_.step_sizes_.assign(num_chains, 0.1);  // config.hpp:205
  }
  InitConfigBuilder& step_sizes(double v) {
    detail::finite_positive(v, "step size");
    c_.step_sizes_.assign(c_.num_chains_, v);
    return *this;
  }
  InitConfigBuilder& step_sizes(const std::vector<double>& v) {
    if (v.size() != c_.num_chains_) throw std::invalid_argument("step_sizes size must match num_chains");
    for (double x : v) detail::finite_positive(x, "step_size");
    c_.step_sizes_ = v;
    return *this;
  }
  /** N(0, init_scale^2) in every coordinate; the reference's `RNG&` becomes a seed of the device stream. */
  InitConfigBuilder& positions(std::uint64_t seed, double init_scale) {
    detail::finite_positive(init_scale, "init_scale");
    c_.positions_kind_ = InitConfig::Positions::random;
    c_.positions_seed_ = seed;
    c_.positions_scale_ = init_scale;
    return *this;
  }
  /** One position for all chains (`dims` values) or one per chain (`num_chains * dims`, chain-major). */
  InitConfigBuilder& positions(const std::vector<double>& v) {
    c_.positions_ = spread(v, "positions");
    for (double x : c_.positions_)
      if (!std::isfinite(x)) throw std::invalid_argument("positions must be finite");
    c_.positions_kind_ = InitConfig::Positions::given;
    return *this;
  }
  /** mass = (1 - s)|grad| + s at the initial position, optionally the geometric mean over chains. */
  InitConfigBuilder& masses(double mass_smoothing, bool average_masses = false) {
    detail::probability(mass_smoothing, "mass_smoothing");
    c_.masses_kind_ = InitConfig::Masses::from_gradient;
    c_.mass_smoothing_ = mass_smoothing;
    c_.average_masses_ = average_masses;
    return *this;
  }
  InitConfigBuilder& masses(const std::vector<double>& v) {
    c_.masses_ = spread(v, "masses");
    for (double x : c_.masses_) detail::finite_positive(x, "masses");
    c_.masses_kind_ = InitConfig::Masses::given;
    return *this;
  }
  InitConfig build() { return c_; }
  /** The step-size search of util.hpp:285-303 for every chain, run on the device when the engine is built. */
  InitConfig adapt_step_build(std::uint64_t seed) {
    c_.adapt_step_ = true;
    c_.adapt_step_seed_ = seed;
    return c_;
  }

 private:
  std::vector<double> spread(const std::vector<double>& v, const char* what) const {
    const std::size_t C = c_.num_chains_, D = c_.dims_;
    if (v.size() == C * D) return v;
    if (v.size() != D) throw std::invalid_argument(std::string(what) + " size must match dims");
    std::vector<double> out(C * D);
    for (std::size_t c = 0; c < C; ++c) std::copy(v.begin(), v.end(), out.begin() + static_cast<std::ptrdiff_t>(c * D));
    return out;
  }
  InitConfig c_;
};

class WalnutsConfig {  // config.hpp:1089-1140
 public:
  WalnutsConfig(InitConfig init, WarmupConfig warmup, SamplingConfig sampling)
      : init_(std::move(init)), warmup_(std::move(warmup)), sampling_(std::move(sampling)) {}
  const InitConfig& init() const noexcept { return init_; }
  const WarmupConfig& warmup() const noexcept { return warmup_; }
  const SamplingConfig& sampling() const noexcept { return sampling_; }

 private:
  InitConfig init_;
  WarmupConfig warmup_;
  SamplingConfig sampling_;
};

/** Where and how the batch runs: no counterpart in the reference (one process per GPU). */
struct Placement {
  int device = 0;
  std::uint32_t chain_offset = 0;  // global id of this engine's first chain (rank * chains_per_rank)
  int waves_per_chain = 0, elems_per_lane = 0, workgroups_per_cu = 0, lds_vectors = -1, reserved_cus = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// engines
// ---------------------------------------------------------------------------------------------------------------
namespace detail {

struct EngineDeleter {
  void operator()(wn_engine* e) const noexcept {
    if (e) wn_engine_destroy(e);
  }
};

/** State shared by the adaptive engine, the sampler made from it and their chain views. */
template <class H>
struct Batch {
  std::unique_ptr<wn_engine, EngineDeleter> engine;
  std::size_t C = 0, D = 0;
  std::vector<H>* handlers = nullptr;
  std::vector<double> positions, logp, step_sizes, inv_mass;  // host copies of the latest iteration
  std::size_t produced = 0;                                    // sampling transitions done so far
  bool inv_mass_fresh = false, step_fresh = false;

  wn_engine* e() const { return engine.get(); }
  void fetch_draw() {
    positions.resize(C * D);
    call(wn_engine_get_positions, e(), positions.data());
  }
  void fetch_logp() {
    logp.resize(C);
    call(wn_engine_get_logp, e(), logp.data());
  }
  void fetch_steps() {
    step_sizes.resize(C);
    call(wn_engine_get_step_sizes, e(), step_sizes.data());
    step_fresh = true;
  }
  void fetch_inv_mass() {
    inv_mass.resize(C * D);
    call(wn_engine_get_inv_mass, e(), inv_mass.data());
    inv_mass_fresh = true;
  }
  VectorView row(const std::vector<double>& plane, std::size_t c) const { return {plane.data() + c * D, D}; }
  /** `on_extension_failed` for the chains whose last transition had a failed extension (positions fetched). */
  void report_model_failures() {
    if constexpr (FailureCallback<H>) {
      if (handlers == nullptr) return;
      failures.resize(C);
      call(wn_engine_get_failed_extensions, e(), failures.data());
      for (std::size_t c = 0; c < C; ++c)
        if (failures[c] != 0) (*handlers)[c].on_extension_failed(row(positions, c));
    }
  }
  std::vector<std::int32_t> failures;
  /** One sampling transition of every chain (walnuts.hpp:682-692), then `on_sample(position, lp)` per chain. */
  void sample_step() {
    call(wn_engine_sample_step, e(), static_cast<double*>(nullptr), std::int64_t{0});
    ++produced;
    fetch_logp();
    fetch_draw();
    report_model_failures();
    if (handlers != nullptr)
      for (std::size_t c = 0; c < C; ++c) (*handlers)[c].on_sample(row(positions, c), logp[c]);
  }
  void check_chain(std::size_t c) const {
    if (c >= C) throw std::out_of_range("chain index out of range");
  }
};

}  // namespace detail

/**
 * One chain of a BatchedWalnutsSampler as a `Sampler` (concepts.hpp:95-99): `double operator()()` returns the log
 * density of the chain's next draw.  The chains advance in lock step, so the first view asked for iteration t makes
 * the whole batch take transition t and the others read their value of it; a view may be at most one iteration
 * behind the batch.  A reference-style loop `for (t...) for (auto& s : samplers) lp = s();` therefore works as is.
 */
template <class H>
class ChainView {
 public:
  double operator()();
  std::size_t dim() const noexcept { return batch_->D; }
  std::size_t chain() const noexcept { return chain_; }
  /** The chain's latest draw. */
  VectorView position() const { return batch_->row(batch_->positions, chain_); }

 private:
  friend class BatchedWalnutsSampler<H>;
  ChainView(std::shared_ptr<detail::Batch<H>> b, std::size_t c)
      : batch_(std::move(b)), chain_(c), consumed_(batch_->produced) {}
  std::shared_ptr<detail::Batch<H>> batch_;
  std::size_t chain_, consumed_;
};

/** Fixed-parameter sampling of every chain: the batched WalnutsSampler (walnuts.hpp:605-766). */
template <class H>
class BatchedWalnutsSampler {
 public:
  /** One transition of every chain; `on_sample(position, lp)` per chain; returns the log densities. */
  const std::vector<double>& operator()() {
    batch_->sample_step();
    return batch_->logp;
  }
  std::size_t dim() const noexcept { return batch_->D; }
  std::size_t num_chains() const noexcept { return batch_->C; }
  std::size_t iter() const noexcept { return batch_->produced; }
  double step_size(std::size_t c) const {  // walnuts.hpp:703-728
    batch_->check_chain(c);
    if (!batch_->step_fresh) batch_->fetch_steps();
    return batch_->step_sizes[c];
  }
  VectorView inv_mass(std::size_t c) const {
    batch_->check_chain(c);
    if (!batch_->inv_mass_fresh) batch_->fetch_inv_mass();
    return batch_->row(batch_->inv_mass, c);
  }
  VectorView position(std::size_t c) const {
    batch_->check_chain(c);
    return batch_->row(batch_->positions, c);
  }
  /** R-hat of the log density over the chains (sampler.hpp:139-145). */
  double r_hat() const {
    double r = 0;
    detail::call(wn_engine_rhat, batch_->e(), &r);
    return r;
  }
  ChainView<H> chain(std::size_t c) {
    batch_->check_chain(c);
    return ChainView<H>(batch_, c);
  }
  std::vector<ChainView<H>> chains() {
    std::vector<ChainView<H>> v;
    v.reserve(batch_->C);
    for (std::size_t c = 0; c < batch_->C; ++c) v.push_back(chain(c));
    return v;
  }
  /** Fails if a chain's last transition could not complete on the device. */
  void check() const { detail::call(wn_engine_check, batch_->e()); }
  wn_engine* handle() const noexcept { return batch_->e(); }

 private:
  template <class>
  friend class BatchedAdaptiveWalnuts;
  explicit BatchedWalnutsSampler(std::shared_ptr<detail::Batch<H>> b) : batch_(std::move(b)) {}
  std::shared_ptr<detail::Batch<H>> batch_;
};

template <class H>
inline double ChainView<H>::operator()() {
  auto& b = *batch_;
  if (consumed_ == b.produced) {
    b.sample_step();
  } else if (consumed_ + 1 != b.produced) {
    throw std::logic_error("chain view is more than one iteration behind the batch");
  }
  ++consumed_;
  return b.logp[chain_];
}

/** Adaptive warmup of every chain: the batched AdaptiveWalnuts (adaptive_walnuts.hpp:182-363). */
template <class H>
class BatchedAdaptiveWalnuts {
 public:
  /**
   * @param seed chain m draws from the stream keyed (seed, place.chain_offset + m): the counterpart of
   *             `RNG(seed_seq{seed, m + 1})`, api.hpp:46-51
   * @param handlers one handler per chain, called back from operator()() and sampler(); may be null
   */
  BatchedAdaptiveWalnuts(const DeviceModel& model, const InitConfig& init, const WarmupConfig& warmup,
                         const SamplingConfig& sampling, std::uint64_t seed, std::vector<H>* handlers = nullptr,
                         const Placement& place = {})
      : batch_(std::make_shared<detail::Batch<H>>()) {
    if (init.dims() != model.dims) throw std::invalid_argument("init.dims() must be equal to the model's dims");
    if (handlers != nullptr && handlers->size() != init.num_chains())
      throw std::invalid_argument("chain_handlers.size() must be equal to config.init().num_chains()");
    wn_config cfg;
    wn_default_config(&cfg);
    cfg.max_trajectory_doublings = static_cast<std::int32_t>(sampling.max_trajectory_doublings());
    cfg.max_step_halvings = static_cast<std::int32_t>(sampling.max_step_halvings());
    cfg.min_micro_steps = static_cast<std::int32_t>(sampling.min_micro_steps());
    cfg.max_hamiltonian_error = sampling.max_hamiltonian_error();
    cfg.mass_init_count = warmup.mass_init_count();
    cfg.max_macro_steps_target = warmup.max_macro_steps_target();
    cfg.step_accept_rate_target = warmup.step_accept_rate_target();
    cfg.step_learning_rate = warmup.step_learning_rate();
    cfg.step_gradient_decay = warmup.step_gradient_decay();
    cfg.step_sq_gradient_decay = warmup.step_sq_gradient_decay();
    cfg.step_stabilization = warmup.step_stabilization();
    cfg.step_learn_rate_decay = warmup.step_learn_rate_decay();
    cfg.device = place.device;
    cfg.waves_per_chain = place.waves_per_chain;
    cfg.elems_per_lane = place.elems_per_lane;
    cfg.workgroups_per_cu = place.workgroups_per_cu;
    cfg.lds_vectors = place.lds_vectors;
    cfg.reserved_cus = place.reserved_cus;
    wn_engine* raw = nullptr;
    detail::call(wn_engine_create, &raw, model.id, static_cast<int>(model.dims),
                 model.params.empty() ? static_cast<const double*>(nullptr) : model.params.data(),
                 init.num_chains(), static_cast<const wn_config*>(&cfg));
    auto& b = *batch_;
    b.engine.reset(raw);
    b.C = init.num_chains();
    b.D = init.dims();
    b.handlers = handlers;
    // InitConfigBuilder's verbs, in the order the reference applies them (examples/walnutpie_api.cpp:57-63)
    switch (init.positions_kind_) {
      case InitConfig::Positions::zero: {
        const std::vector<double> z(b.C * b.D, 0.0);
        detail::call(wn_engine_set_positions, raw, z.data());
      } break;
      case InitConfig::Positions::given:
        detail::call(wn_engine_set_positions, raw, init.positions_.data());
        break;
      case InitConfig::Positions::random:
        detail::call(wn_engine_init_positions, raw, init.positions_seed_, place.chain_offset, init.positions_scale_);
        break;
    }
    switch (init.masses_kind_) {
      case InitConfig::Masses::one: {
        const std::vector<double> o(b.C * b.D, 1.0);
        detail::call(wn_engine_set_masses, raw, o.data());
      } break;
      case InitConfig::Masses::given:
        detail::call(wn_engine_set_masses, raw, init.masses_.data());
        break;
      case InitConfig::Masses::from_gradient:
        detail::call(wn_engine_init_masses_from_grad, raw, init.mass_smoothing_);
        if (init.average_masses_) detail::call(wn_engine_average_masses, raw);
        break;
    }
    detail::call(wn_engine_set_step_sizes, raw, init.step_sizes_.data());
    if (init.adapt_step_) detail::call(wn_engine_adapt_step, raw, init.adapt_step_seed_, place.chain_offset);
    detail::call(wn_engine_seed, raw, seed, place.chain_offset);
  }

  /**
   * One warmup transition of every chain (adaptive_walnuts.hpp:234-251), then per chain
   * `on_warmup(position, lp, step_size, diag_inv_mass)` with the step size AFTER this iteration's update and the
   * inverse masses the transition integrated with (adaptive_walnuts.hpp:249).
   */
  void operator()() {
    auto& b = *batch_;
    if (b.handlers != nullptr) b.fetch_inv_mass();  // the estimate this transition is about to use
    detail::call(wn_engine_warmup_step, b.e(), static_cast<double*>(nullptr), std::int64_t{0});
    ++iter_;
    b.step_fresh = false;
    if (b.handlers != nullptr) {
      b.fetch_draw();
      b.fetch_logp();
      b.fetch_steps();
      b.report_model_failures();
      for (std::size_t c = 0; c < b.C; ++c)
        (*b.handlers)[c].on_warmup(b.row(b.positions, c), b.logp[c], b.step_sizes[c], b.row(b.inv_mass, c));
    }
    b.inv_mass_fresh = false;
  }

  /** Freeze the adapted parameters (adaptive_walnuts.hpp:263-271): `on_warmup_complete(step_size, inv_mass)`
   *  per chain, and the fixed-parameter sampler over the same chains.  The adaptive engine must not be advanced
   *  afterwards. */
  BatchedWalnutsSampler<H> sampler() {
    auto& b = *batch_;
    detail::call(wn_engine_freeze, b.e());
    b.fetch_steps();
    b.fetch_inv_mass();
    b.fetch_draw();
    if (b.handlers != nullptr)
      for (std::size_t c = 0; c < b.C; ++c) (*b.handlers)[c].on_warmup_complete(b.step_sizes[c], b.row(b.inv_mass, c));
    return BatchedWalnutsSampler<H>(batch_);
  }

  std::size_t dim() const noexcept { return batch_->D; }
  std::size_t num_chains() const noexcept { return batch_->C; }
  std::size_t iter() const noexcept { return iter_; }
  double step_size(std::size_t c) const {
    batch_->check_chain(c);
    if (!batch_->step_fresh) batch_->fetch_steps();
    return batch_->step_sizes[c];
  }
  double log_step_size(std::size_t c) const { return std::log(step_size(c)); }
  /** The current inverse-mass estimate of chain c (adaptive_walnuts.hpp:297-299). */
  std::vector<double> inv_mass(std::size_t c) const {
    batch_->check_chain(c);
    if (!batch_->inv_mass_fresh) batch_->fetch_inv_mass();
    const VectorView v = batch_->row(batch_->inv_mass, c);
    return std::vector<double>(v.begin(), v.end());
  }
  std::vector<double> log_mass(std::size_t c) const {  // adaptive_walnuts.hpp:319-323
    std::vector<double> v = inv_mass(c);
    for (double& x : v) x = -std::log(x);
    return v;
  }
  /** The warmup controller's statistics (adapt.hpp:193-221): max over chains of the relative distance of the step
   *  size, and of the L2 relative distance of the masses, from their geometric means over chains. */
  void spread(double& max_rel_diff_step, double& max_rel_diff_mass) const {
    detail::call(wn_engine_warmup_spread, batch_->e(), &max_rel_diff_step, &max_rel_diff_mass);
  }
  wn_engine* handle() const noexcept { return batch_->e(); }

 private:
  std::shared_ptr<detail::Batch<H>> batch_;
  std::size_t iter_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// posterior summaries over device-resident draws: include/walnutpie/summary.hpp:119-768 with the reference's
// function names.  MarkovChains stands where MarkovChainsSplit / MarkovChainsUnified do; results are host vectors
// ([dims], or row-major [k][dims] for quantiles / [num_draws][dims] for autocovariance).
// ---------------------------------------------------------------------------------------------------------------
class MarkovChains {
 public:
  /** Chains stacked in one row-major [sum sizes][dims] block on the host (MarkovChainsUnified, summary.hpp:251-356);
   *  copied to the device. */
  MarkovChains(const std::vector<double>& draws, const std::vector<std::size_t>& chain_sizes, std::size_t dims,
               int device = 0) {
    std::size_t total = 0;
    std::vector<std::int64_t> sizes;
    for (std::size_t n : chain_sizes) {
      total += n;
      sizes.push_back(static_cast<std::int64_t>(n));
    }
    if (dims == 0 || total * dims != draws.size())
      throw std::invalid_argument("sum of chain sizes must equal number of rows in draws");  // summary.hpp:277-281
    wn_chains* raw = nullptr;
    detail::call(wn_chains_upload, &raw, draws.data(), dims, static_cast<const std::int64_t*>(sizes.data()),
                 sizes.size(), device);
    h_.reset(raw);
  }
  /** One matrix per chain (MarkovChainsSplit, summary.hpp:119-240): chain m is a row-major [rows][dims] block. */
  static MarkovChains split(const std::vector<std::vector<double>>& chains, std::size_t dims, int device = 0) {
    if (chains.empty()) throw std::invalid_argument("require at least one chain");
    std::vector<double> all;
    std::vector<std::size_t> sizes;
    for (const auto& c : chains) {
      if (dims == 0 || c.size() % dims != 0) throw std::invalid_argument("all chains must have the same number of columns");
      sizes.push_back(c.size() / dims);
      all.insert(all.end(), c.begin(), c.end());
    }
    return MarkovChains(all, sizes, dims, device);
  }
  /** Draws already on the device: chain c's n-th draw at draws_dev + c * chain_stride + n * dims (the sampler's draw
   *  buffer); `lengths` may be null (all max_len); `stream` orders the kernels after the producer. */
  static MarkovChains view(const double* draws_dev, std::size_t num_chains, std::size_t max_len, std::size_t dims,
                           std::int64_t chain_stride, const std::int64_t* lengths = nullptr, int device = 0,
                           void* stream = nullptr) {
    wn_chains* raw = nullptr;
    detail::call(wn_chains_view, &raw, draws_dev, num_chains, max_len, dims, chain_stride, lengths, device, stream);
    MarkovChains m;
    m.h_.reset(raw);
    return m;
  }
  std::size_t num_chains() const { return wn_chains_num_chains(h_.get()); }
  std::size_t dims() const { return wn_chains_dims(h_.get()); }
  std::size_t num_draws() const { return wn_chains_num_draws(h_.get()); }
  std::size_t min_chain_size() const { return wn_chains_min_chain_size(h_.get()); }
  wn_chains* handle() const noexcept { return h_.get(); }

 private:
  MarkovChains() = default;
  struct Deleter {
    void operator()(wn_chains* c) const noexcept {
      if (c) wn_chains_destroy(c);
    }
  };
  std::unique_ptr<wn_chains, Deleter> h_;
};

namespace detail {
template <class F>
inline std::vector<double> summary(F f, const MarkovChains& chains, std::size_t n) {
  std::vector<double> out(n);
  call(f, chains.handle(), out.data());
  return out;
}
}  // namespace detail

inline std::vector<double> mean(const MarkovChains& c) { return detail::summary(wn_summary_mean, c, c.dims()); }
inline std::vector<double> sample_variance(const MarkovChains& c) {
  return detail::summary(wn_summary_sample_variance, c, c.dims());
}
inline std::vector<double> sample_standard_deviation(const MarkovChains& c) {
  return detail::summary(wn_summary_sample_standard_deviation, c, c.dims());
}
inline std::vector<double> r_hat(const MarkovChains& c) { return detail::summary(wn_summary_r_hat, c, c.dims()); }
inline std::vector<double> effective_sample_size(const MarkovChains& c) {
  return detail::summary(wn_summary_effective_sample_size, c, c.dims());
}
inline std::vector<double> monte_carlo_standard_error(const MarkovChains& c) {
  return detail::summary(wn_summary_monte_carlo_standard_error, c, c.dims());
}
/** Row-major [num_draws][dims]: all lags of every chain, stacked like the draws (summary.hpp:529-545). */
inline std::vector<double> autocovariance(const MarkovChains& c) {
  return detail::summary(wn_summary_autocovariance, c, c.num_draws() * c.dims());
}
/** Row-major [probs.size()][dims] (summary.hpp:483-514). */
inline std::vector<double> quantiles(const MarkovChains& c, const std::vector<double>& probs) {
  std::vector<double> out(probs.size() * c.dims());
  detail::call(wn_summary_quantiles, c.handle(), probs.data(), probs.size(), out.data());
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// drivers
// ---------------------------------------------------------------------------------------------------------------

/** detail::adapt (adapt.hpp:172-259) in lock step: at least min_iter transitions, at most max_iter; in between,
 *  every publish_stride iterations, stop once step sizes and masses agree across chains within the tolerances. */
template <class H, InterruptCallback IC>
inline std::size_t run_warmup(BatchedAdaptiveWalnuts<H>& adapter, const WarmupConfig& cfg, const IC& interrupt) {
  std::size_t it = 0;
  while (it < cfg.max_iter()) {
    adapter();
    ++it;
    interrupt.throw_if_interrupted();
    if (it >= cfg.min_iter() && it < cfg.max_iter() && it % cfg.publish_stride() == 0) {
      double rel_step = 0, rel_mass = 0;
      adapter.spread(rel_step, rel_mass);
      if (rel_mass <= cfg.mass_converge_tol() && rel_step <= cfg.step_size_converge_tol()) break;
    }
  }
  return it;
}

/** detail::sample (sampler.hpp:117-200) in lock step: once every chain has min_iter draws, R-hat of the log density
 *  goes to `on_r_hat` after each iteration and sampling stops at R-hat <= rhat_converge_tol or at max_iter. */
template <class H, GlobalHandler GH, InterruptCallback IC>
inline std::size_t run_sampling(BatchedWalnutsSampler<H>& sampler, const SamplingConfig& cfg, GH& global_handler,
                                const IC& interrupt) {
  std::size_t it = 0;
  while (it < cfg.max_iter()) {
    sampler();
    ++it;
    interrupt.throw_if_interrupted();
    if (it >= cfg.min_iter() && it >= 2 && sampler.num_chains() > 1) {
      const double r_hat = sampler.r_hat();
      global_handler.on_r_hat(r_hat);
      if (r_hat <= cfg.rhat_converge_tol()) break;
    }
  }
  sampler.check();
  return it;
}

/**
 * Run Walnuts for every chain: warmup, freeze, sampling -- `walnutpie::walnuts` (api.hpp:35-69) for a device model.
 *
 * @throws std::invalid_argument if the number of handlers differs from `config.init().num_chains()` (api.hpp:41-44)
 */
template <ChainHandler H, GlobalHandler GH, InterruptCallback IC>
inline void walnuts(std::size_t seed, std::vector<H>& chain_handlers, GH& global_handler, const IC& interrupt_callback,
                    const DeviceModel& model, const WalnutsConfig& config, const Placement& place = {}) {
  if (chain_handlers.size() != config.init().num_chains())
    throw std::invalid_argument("chain_handlers.size() must be equal to config.init().num_chains()");
  BatchedAdaptiveWalnuts<H> adapter(model, config.init(), config.warmup(), config.sampling(), seed, &chain_handlers,
                                    place);
  run_warmup(adapter, config.warmup(), interrupt_callback);
  BatchedWalnutsSampler<H> sampler = adapter.sampler();
  run_sampling(sampler, config.sampling(), global_handler, interrupt_callback);
}

}  // namespace walnuts_hip

#endif  // WALNUTS_HIP_HPP
