// Test program for include/walnuts_hip.hpp (the C++ mirror of the reference's surface for the many-chain path).
// Built with g++ as a shared library linked against the library under test -- the CPU workgroup emulation (CPU
// tier, tests/test_cpp_surface.py builds it) or libwalnuts_hip.so (GPU tier, __graft_entry__.build() builds it so
// that nothing has to be spawned on the GPU box) -- and called in-process through
//   int cpp_surface_run(model: std_normal|diag_normal|funnel, chains, dims, warmup, sampling, seed, dump-file)
// (with -DCPP_SURFACE_MAIN also a command-line program with the same arguments).  Runs walnuts_hip::walnuts() with recording handlers (the reference's examples/handlers.hpp ChainStore, restated),
// checks the surface's contracts, and writes everything the handlers saw to <dump-file> as raw doubles so that the
// Python side can compare it bit for bit with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "walnuts_hip.hpp"

namespace wh = walnuts_hip;

// a handler written the way the reference's are: owning-vector parameters (std::vector<double> stands in for
// Eigen::VectorXd, which this image lacks) -- VectorView converts implicitly
struct ChainStore {
  std::vector<double> warmup_draws, warmup_lp, warmup_step, warmup_inv_mass, draws, lp, final_inv_mass;
  double final_step = 0;
  int completes = 0;
  void on_warmup(const std::vector<double>& position, double logp, double step_size,
                 const std::vector<double>& diag_inv_mass) {
    warmup_draws.insert(warmup_draws.end(), position.begin(), position.end());
    warmup_lp.push_back(logp);
    warmup_step.push_back(step_size);
    warmup_inv_mass.insert(warmup_inv_mass.end(), diag_inv_mass.begin(), diag_inv_mass.end());
  }
  void on_warmup_complete(double step_size, const std::vector<double>& diag_inv_mass) {
    final_step = step_size;
    final_inv_mass = diag_inv_mass;
    ++completes;
  }
  void on_sample(const std::vector<double>& position, double logp) {
    draws.insert(draws.end(), position.begin(), position.end());
    lp.push_back(logp);
  }
};
struct GlobalStore {
  std::vector<double> r_hats;
  void on_r_hat(double r) { r_hats.push_back(r); }
};
struct InterruptAfter {
  mutable long calls = 0;
  long limit;
  void throw_if_interrupted() const {
    if (++calls > limit) throw std::runtime_error("interrupted by the test");
  }
};

static_assert(wh::ChainHandler<ChainStore>);
static_assert(wh::ChainHandler<wh::NoOpChainHandler>);
static_assert(wh::GlobalHandler<GlobalStore>);
static_assert(wh::InterruptCallback<wh::NeverInterrupted>);
static_assert(wh::InterruptCallback<InterruptAfter>);
static_assert(wh::Sampler<wh::ChainView<ChainStore>>);
static_assert(wh::Sampler<wh::ChainView<>>);

static int failures = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) {                                                            \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      ++failures;                                                             \
    }                                                                         \
  } while (0)
template <class E, class F>
static bool throws(F f, const char* needle) {
  try {
    f();
  } catch (const E& e) {
    if (std::strstr(e.what(), needle) != nullptr) return true;
    std::fprintf(stderr, "exception message was: %s\n", e.what());
    return false;
  } catch (...) {
    return false;
  }
  return false;
}

static wh::DeviceModel make_model(const std::string& name, std::size_t D) {
  if (name == "std_normal") return wh::DeviceModel::std_normal(D);
  if (name == "funnel") return wh::DeviceModel::funnel(D);
  std::vector<double> s2(D);
  for (std::size_t d = 0; d < D; ++d) s2[d] = (1.0 + static_cast<double>(d % 16)) * (1.0 + static_cast<double>(d % 16));
  return wh::DeviceModel::diag_normal(s2);
}

static void put(std::ofstream& out, const std::vector<double>& v) {
  out.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(double)));
}

extern "C" __attribute__((visibility("default"))) int cpp_surface_run(const char* model_name_c, std::size_t C,
                                                                       std::size_t D, std::size_t W, std::size_t S,
                                                                       std::size_t seed, const char* dump_path) try {
  failures = 0;
  const std::string model_name = model_name_c;
  const wh::DeviceModel model = make_model(model_name, D);

  // ---- configuration classes: defaults and checks of config.hpp ------------------------------------------
  {
    const wh::SamplingConfig s = wh::SamplingConfigBuilder().build();
    EXPECT(s.min_iter() == 50 && s.max_iter() == 1000 && s.max_trajectory_doublings() == 5 && s.max_step_halvings() == 5);
    EXPECT(s.max_hamiltonian_error() == 0.5 && s.min_micro_steps() == 1 && s.rhat_converge_tol() == 1.01);
    const wh::WarmupConfig w = wh::WarmupConfigBuilder().build();
    EXPECT(w.min_iter() == 50 && w.max_iter() == 1000 && w.step_size_converge_tol() == 0.1 && w.mass_converge_tol() == 1.0);
    EXPECT(w.mass_init_count() == 4.0 && w.mass_additive_smoothing() == 1e-5 && w.max_macro_steps_target() == 15.0);
    EXPECT(w.step_accept_rate_target() == 0.8 && w.step_learning_rate() == 0.05 && w.step_gradient_decay() == 0.8);
    EXPECT(w.step_sq_gradient_decay() == 0.9 && w.step_stabilization() == 1e-4 && w.step_learn_rate_decay() == 0.5);
    EXPECT(w.publish_stride() == 5);
    EXPECT(throws<std::invalid_argument>([] { wh::SamplingConfigBuilder().min_max_iter(3, 2); }, "min_iter must be"));
    EXPECT(throws<std::invalid_argument>([] { wh::WarmupConfigBuilder().min_max_iter(3, 2); }, "min_iter cannot be greater"));
    EXPECT(throws<std::invalid_argument>([] { wh::SamplingConfigBuilder().rhat_converge_tol(1.0); }, "finite and > 1"));
    EXPECT(throws<std::invalid_argument>([] { wh::SamplingConfigBuilder().max_hamiltonian_error(0.0); }, "max_hamiltonian_error"));
    EXPECT(throws<std::invalid_argument>([] { wh::SamplingConfigBuilder().min_micro_steps(0); }, "min_micro_steps"));
    EXPECT(throws<std::invalid_argument>([] { wh::WarmupConfigBuilder().step_gradient_decay(1.0); }, "step_gradient_decay must be in (0, 1)"));
    EXPECT(throws<std::invalid_argument>([] { wh::WarmupConfigBuilder().mass_init_count(-1.0); }, "mass_init_count must be finite and > 0"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).step_sizes(0.0); }, "step size must be finite and > 0"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).step_sizes(std::vector<double>{0.1}); }, "step_sizes size must match num_chains"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).positions(1, -2.0); }, "init_scale"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).positions(std::vector<double>(4, 0.0)); }, "positions size must match"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).masses(std::vector<double>(3, 0.0)); }, "masses must be finite and > 0"));
    EXPECT(throws<std::invalid_argument>([] { wh::InitConfigBuilder(2, 3).masses(1.0); }, "mass_smoothing"));
  }

  // ---- the top-level call (api.hpp:35-69) ------------------------------------------------------------------
  const auto make_config = [&](std::size_t warm, std::size_t samp) {
    return wh::WalnutsConfig(wh::InitConfigBuilder(C, D).positions(seed + 5, 2.0).masses(1e-5).step_sizes(1.0).adapt_step_build(seed + 6),
                             wh::WarmupConfigBuilder().min_max_iter(warm, warm).build(),
                             wh::SamplingConfigBuilder().min_max_iter(samp, samp).build());
  };
  const wh::WalnutsConfig config = make_config(W, S);
  std::vector<ChainStore> stores(C);
  GlobalStore global;
  wh::NeverInterrupted never;
  wh::walnuts(seed, stores, global, never, model, config);
  for (std::size_t c = 0; c < C; ++c) {
    EXPECT(stores[c].warmup_lp.size() == W && stores[c].warmup_draws.size() == W * D);
    EXPECT(stores[c].warmup_inv_mass.size() == W * D && stores[c].warmup_step.size() == W);
    EXPECT(stores[c].lp.size() == S && stores[c].draws.size() == S * D);
    EXPECT(stores[c].completes == 1 && stores[c].final_inv_mass.size() == D && stores[c].final_step > 0);
  }
  // R-hat goes to the global handler once every chain has min_iter (= max_iter here) draws: exactly once
  EXPECT(global.r_hats.size() == (C > 1 && S >= 2 ? 1u : 0u));

  // wrong number of handlers (api.hpp:41-44)
  {
    std::vector<ChainStore> few(C + 1);
    EXPECT(throws<std::invalid_argument>([&] { wh::walnuts(seed, few, global, never, model, config); },
                                         "chain_handlers.size() must be equal to config.init().num_chains()"));
  }
  // the interrupt callback is polled and its exception propagates (sampler.hpp:154, adapt.hpp:226)
  {
    std::vector<ChainStore> st(C);
    GlobalStore g2;
    InterruptAfter stop{0, static_cast<long>(W) + 1};
    EXPECT(throws<std::runtime_error>([&] { wh::walnuts(seed, st, g2, stop, model, config); }, "interrupted by the test"));
    EXPECT(st[0].warmup_lp.size() == W && st[0].lp.size() == 2);
  }
  // a C-ABI config error surfaces as std::invalid_argument
  EXPECT(throws<std::invalid_argument>(
      [&] {
        wh::BatchedAdaptiveWalnuts<> bad(wh::DeviceModel::funnel(1), wh::InitConfigBuilder(C, 1).build(),
                                         config.warmup(), config.sampling(), seed);
      },
      "funnel"));
  EXPECT(throws<std::invalid_argument>(
      [&] {
        wh::BatchedAdaptiveWalnuts<> bad(model, wh::InitConfigBuilder(C, D + 1).build(), config.warmup(),
                                         config.sampling(), seed);
      },
      "dims"));

  // ---- batched engine driven by hand + per-chain Sampler views in a reference-style loop -------------------
  {
    wh::BatchedAdaptiveWalnuts<> adapter(model, config.init(), config.warmup(), config.sampling(), seed);
    EXPECT(adapter.dim() == D && adapter.num_chains() == C && adapter.iter() == 0);
    for (std::size_t it = 0; it < W; ++it) {
      // a.inv_mass() before the transition is what on_warmup reports for it; a.step_size() after it likewise
      const std::vector<double> im0 = adapter.inv_mass(0);
      adapter();
      EXPECT(std::memcmp(im0.data(), stores[0].warmup_inv_mass.data() + it * D, D * sizeof(double)) == 0);
      EXPECT(adapter.step_size(C - 1) == stores[C - 1].warmup_step[it]);
      EXPECT(adapter.log_step_size(0) == std::log(adapter.step_size(0)));
    }
    EXPECT(adapter.iter() == W);
    const std::vector<double> lm = adapter.log_mass(0), im = adapter.inv_mass(0);
    EXPECT(lm.size() == D && lm[0] == -std::log(im[0]));
    auto sampler = adapter.sampler();
    EXPECT(sampler.step_size(0) == stores[0].final_step);
    EXPECT(std::memcmp(sampler.inv_mass(C - 1).data(), stores[C - 1].final_inv_mass.data(), D * sizeof(double)) == 0);
    auto views = sampler.chains();  // std::vector<Sampler>
    for (std::size_t it = 0; it < S; ++it) {
      for (std::size_t c = 0; c < C; ++c) {
        const double lp = views[c]();  // the first view advances the batch, the others read the same iteration
        EXPECT(lp == stores[c].lp[it]);
        EXPECT(std::memcmp(views[c].position().data(), stores[c].draws.data() + it * D, D * sizeof(double)) == 0);
      }
    }
    EXPECT(sampler.iter() == S && views[0].dim() == D);
    if (C > 1 && S >= 2) EXPECT(sampler.r_hat() == global.r_hats.back());
    if (C > 1) {
      (void)views[0]();
      (void)views[0]();  // now two ahead of view 1
      EXPECT(throws<std::logic_error>([&] { (void)views[1](); }, "behind the batch"));
    }
    EXPECT(throws<std::out_of_range>([&] { (void)sampler.step_size(C); }, "chain index"));
    EXPECT(throws<std::runtime_error>([&] { adapter(); }, "after freeze"));
  }

  // ---- posterior summaries with the reference's names (summary.hpp; hand values of tests/summary_test.cpp) ------
  {
    // chain 0: [[1,2],[3,4]]  chain 1: [[5,6],[7,8],[9,10]]  chain 2: [[11,12],[13,14],[15,16]]
    const wh::MarkovChains ex = wh::MarkovChains::split({{1, 2, 3, 4}, {5, 6, 7, 8, 9, 10}, {11, 12, 13, 14, 15, 16}}, 2);
    EXPECT(ex.num_chains() == 3 && ex.dims() == 2 && ex.num_draws() == 8 && ex.min_chain_size() == 2);
    const auto near = [](double a, double b) { return std::fabs(a - b) <= 1e-10; };
    const std::vector<double> mu = wh::mean(ex), var = wh::sample_variance(ex), sd = wh::sample_standard_deviation(ex);
    EXPECT(near(mu[0], 8.0) && near(mu[1], 9.0) && near(var[0], 24.0) && near(var[1], 24.0));
    EXPECT(near(sd[0], std::sqrt(24.0)));
    const std::vector<double> q = wh::quantiles(ex, {0.0, 0.25, 0.5, 0.75, 1.0});   // numpy values, :534-543
    const double want[10] = {1.0, 2.0, 4.5, 5.5, 8.0, 9.0, 11.5, 12.5, 15.0, 16.0};
    for (int i = 0; i < 10; ++i) EXPECT(near(q[static_cast<std::size_t>(i)], want[i]));
    EXPECT(wh::quantiles(wh::MarkovChains({9, 11, 5, 3}, {4}, 1), {0.6})[0] == 8.2);   // doc example, :558-568
    EXPECT(wh::autocovariance(ex).size() == 16);
    EXPECT(throws<std::invalid_argument>([&] { (void)wh::quantiles(ex, {1.5}); }, "probs must be in [0, 1]"));
    EXPECT(throws<std::invalid_argument>([&] { (void)wh::r_hat(ex); }, "at least 3 draws"));
    const wh::MarkovChains three = wh::MarkovChains::split({{1, 10, 2, 8, 3, 9}, {4, 5, 6, 7, 5, 6}, {7, 2, 9, 4, 8, 3}}, 2);
    const std::vector<double> rh = wh::r_hat(three);                                 // :846-862
    EXPECT(std::fabs(rh[0] - std::sqrt(10.0)) <= 1e-14 && std::fabs(rh[1] - std::sqrt(10.0)) <= 1e-14);
    const std::vector<double> ess = wh::effective_sample_size(three), mcse = wh::monte_carlo_standard_error(three);
    const std::vector<double> sd3 = wh::sample_standard_deviation(three);
    EXPECT(ess[0] > 0 && near(mcse[0], sd3[0] / std::sqrt(ess[0])));
    EXPECT(throws<std::invalid_argument>([&] { (void)wh::r_hat(wh::MarkovChains({1, 2, 3}, {3}, 1)); }, "at least two chains"));
    EXPECT(throws<std::invalid_argument>([] { wh::MarkovChains({1, 2, 3}, {2}, 1); }, "sum of chain sizes"));
  }

  // ---- early stopping bounds (adapt.hpp:172-229, sampler.hpp:117-158) ----------------------------------------
  if (C > 1) {
    const wh::WalnutsConfig loose(
        config.init(),
        wh::WarmupConfigBuilder().min_max_iter(1, 60).publish_stride(2).step_size_converge_tol(1e6).mass_converge_tol(1e6).build(),
        wh::SamplingConfigBuilder().min_max_iter(3, 60).rhat_converge_tol(1e6).build());
    std::vector<ChainStore> st(C);
    GlobalStore g3;
    wh::walnuts(seed, st, g3, never, model, loose);
    EXPECT(st[0].warmup_lp.size() == 2);  // first controller look at a multiple of publish_stride >= min_iter
    EXPECT(st[0].lp.size() == 3 && g3.r_hats.size() == 1);
    for (std::size_t c = 1; c < C; ++c) EXPECT(st[c].lp.size() == st[0].lp.size());  // lock step: equal lengths
  }

  // ---- the failure channel of a device model (the counterpart of concepts.hpp:196-201, util.hpp:336-346) ------------
  // chain 0 starts where the standard normal's log density overflows to -inf: its first leaf meets a non-finite energy
  // at every step size, the extension fails (walnuts.hpp:543-545), the chain stays where it is -- and its handler hears
  // about it after every transition; the other chains' handlers do not
  if (model_name == "std_normal" && C > 1) {
    struct Listening : ChainStore {
      int failed = 0;
      double where = 0;
      void on_extension_failed(const std::vector<double>& position) {
        ++failed;
        where = position[0];
      }
    };
    static_assert(wh::FailureCallback<Listening>);
    static_assert(!wh::FailureCallback<ChainStore>);
    std::vector<double> pos(C * D, 0.25);
    for (std::size_t d = 0; d < D; ++d) pos[d] = 1e200;
    const wh::WalnutsConfig cfg(wh::InitConfigBuilder(C, D).positions(pos).masses(std::vector<double>(C * D, 1.0)).step_sizes(1e-3).build(),   // (so small a step that nothing else fails)
                                wh::WarmupConfigBuilder().min_max_iter(2, 2).build(),
                                wh::SamplingConfigBuilder().min_max_iter(2, 2).build());
    std::vector<Listening> ears(C);
    GlobalStore g4;
    wh::walnuts(seed, ears, g4, never, model, cfg);
    EXPECT(ears[0].failed == 4 && ears[0].where == 1e200);   // two warmup and two sampling transitions
    EXPECT(ears[0].draws.size() == 2 * D && ears[0].draws[0] == 1e200);
    for (std::size_t c = 1; c < C; ++c) EXPECT(ears[c].failed == 0);
  }

  // ---- dump for the oracle comparison -----------------------------------------------------------------------
  std::ofstream out(dump_path, std::ios::binary);
  for (const ChainStore& s : stores) {
    put(out, s.warmup_draws);
    put(out, s.warmup_lp);
    put(out, s.warmup_step);
    put(out, s.warmup_inv_mass);
    put(out, {s.final_step});
    put(out, s.final_inv_mass);
    put(out, s.draws);
    put(out, s.lp);
  }
  put(out, global.r_hats);
  out.close();
  if (failures != 0) {
    std::fprintf(stderr, "%d expectation(s) failed\n", failures);
    return 1;
  }
  return 0;
} catch (const std::exception& e) {
  std::fprintf(stderr, "cpp_surface_run: unexpected exception: %s\n", e.what());
  return 3;
}

#ifdef CPP_SURFACE_MAIN
int main(int argc, char** argv) {
  if (argc != 8) {
    std::fprintf(stderr, "usage: %s model chains dims warmup sampling seed dump\n", argv[0]);
    return 2;
  }
  const auto n = [&](int i) { return static_cast<std::size_t>(std::strtoul(argv[i], nullptr, 10)); };
  const int rc = cpp_surface_run(argv[1], n(2), n(3), n(4), n(5), n(6), argv[7]);
  if (rc == 0) std::puts("cpp surface ok");
  return rc;
}
#endif
