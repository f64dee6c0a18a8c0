// walnuts_hip_api.cpp -- the device counterpart of the reference's examples/walnutpie_api.cpp: many chains of a
// standard normal through the C++ surface in include/walnuts_hip.hpp, posterior summaries computed on the device.
//
//   g++ -std=c++20 -O2 -I include examples/walnuts_hip_api.cpp walnuts_amd/lib/libwalnuts_hip.so
//       -Wl,-rpath,$PWD/walnuts_amd/lib -o walnuts_hip_api            (one command line)
//   ./walnuts_hip_api [num_chains=4096] [dims=100] [max_warmup=2000] [max_sampling=1000]
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "walnuts_hip.hpp"

namespace wh = walnuts_hip;

// what the reference's examples/handlers.hpp ChainStore keeps, for a chain of this run
struct ChainStore {
  std::vector<double> draws, inv_mass;
  std::size_t num_warmup = 0, num_draws = 0;
  double step_size = 0;
  void on_warmup(wh::VectorView, double, double, wh::VectorView) { ++num_warmup; }
  void on_warmup_complete(double step, wh::VectorView diag_inv_mass) {
    step_size = step;
    inv_mass.assign(diag_inv_mass.begin(), diag_inv_mass.end());
  }
  void on_sample(wh::VectorView position, double) {
    draws.insert(draws.end(), position.begin(), position.end());
    ++num_draws;
  }
};
struct GlobalStore {
  std::vector<double> r_hats;
  void on_r_hat(double r) { r_hats.push_back(r); }
};

int main(int argc, char** argv) {
  const std::size_t num_chains = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 4096;
  const std::size_t dims = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 100;
  const std::size_t max_warmup = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 2000;
  const std::size_t max_sampling = argc > 4 ? std::strtoul(argv[4], nullptr, 10) : 1000;
  const std::size_t seed = 48;

  // 1) configure: zero initial positions, unit masses, an absurd initial step that the step-size search repairs
  wh::WalnutsConfig config(wh::InitConfigBuilder(num_chains, dims).step_sizes(100.2).adapt_step_build(seed),
                           wh::WarmupConfigBuilder().min_max_iter(std::min<std::size_t>(50, max_warmup), max_warmup).build(),
                           wh::SamplingConfigBuilder().min_max_iter(std::min<std::size_t>(50, max_sampling), max_sampling).build());
  std::vector<ChainStore> chain_handlers(num_chains);
  GlobalStore global_handler;
  wh::NeverInterrupted interrupt_callback;

  // 2) sample: every chain on the device, handlers called back from the host driver
  wh::walnuts(seed, chain_handlers, global_handler, interrupt_callback, wh::DeviceModel::std_normal(dims), config);

  // 3) summarise
  double sum_log_step = 0;
  for (const auto& h : chain_handlers) sum_log_step += std::log(h.step_size);
  std::cout << "ADAPTATION RESULT:\n  geom_mean(step_size) = " << std::exp(sum_log_step / static_cast<double>(num_chains))
            << "\n\nPER-CHAIN STATISTICS (first 4 chains):\n";
  for (std::size_t m = 0; m < std::min<std::size_t>(4, num_chains); ++m) {
    double norm = 0;
    for (double im : chain_handlers[m].inv_mass) norm += 1.0 / (im * im);
    std::cout << "  Chain " << m << "; step size = " << chain_handlers[m].step_size << "; ||mass|| = " << std::sqrt(norm)
              << "; # warmup_draws = " << chain_handlers[m].num_warmup << "; # draws = " << chain_handlers[m].num_draws << "\n";
  }
  if (!global_handler.r_hats.empty())
    std::cout << "\nNUMBER OF R-HAT EVALS: " << global_handler.r_hats.size() << ";  FINAL R-HAT: " << global_handler.r_hats.back()
              << "\n";

  // posterior summaries of all chains on the device (summary.hpp's functions)
  std::vector<std::vector<double>> chains;
  for (auto& h : chain_handlers) chains.push_back(std::move(h.draws));
  const wh::MarkovChains mc = wh::MarkovChains::split(chains, dims);
  const std::vector<double> mean = wh::mean(mc), sd = wh::sample_standard_deviation(mc), ess = wh::effective_sample_size(mc);
  const std::vector<double> q = wh::quantiles(mc, {0.05, 0.5, 0.95});
  std::cout << "\nPOSTERIOR (dimension 0 of " << dims << ", " << mc.num_draws() << " draws): mean = " << mean[0] << "; sd = " << sd[0]
            << "; 5% / 50% / 95% = " << q[0] << " / " << q[dims] << " / " << q[2 * dims] << "; ESS = " << ess[0] << "\n";
  if (num_chains > 1 && mc.min_chain_size() >= 3) std::cout << "R-HAT (dimension 0) = " << wh::r_hat(mc)[0] << "\n";
  std::cout << "\nFINISHED NORMALLY.\n";
  return 0;
}
