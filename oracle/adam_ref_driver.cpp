// adam_ref_driver.cpp -- exposes the REAL reference Adam optimiser
// (/root/reference/include/walnutpie/adam.hpp, the only hot-path header that
// does not need Eigen) through a C entry point, so tests can pin the oracle's
// Adam restatement against the reference's own object code.
// Built by oracle/Makefile into oracle/_ref/ (git-ignored, travels with gpurun).
#include <cstddef>

#include <walnutpie/adam.hpp>

extern "C" void adam_ref_run(double step_init, double target, double lr, double b1, double b2, double eps,
                             double decay, const double* alphas, std::size_t n, double* steps_out) {
  walnutpie::detail::Adam adam(step_init, target, lr, b1, b2, eps, decay);
  for (std::size_t i = 0; i < n; ++i) {
    adam(alphas[i]);
    steps_out[i] = adam.step_size();
  }
}
