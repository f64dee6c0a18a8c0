// TEST INFRASTRUCTURE: the host streams of walnutpie_sample_cfunc written the plain way -- std::normal_distribution
// over std::mt19937_64, as python/src/walnutpie/walnutpy.cpp:187-189,75-80 + config.hpp:259-268 + util.hpp:288 use
// them -- for tests/test_reference_streams_host.py to compare the library's two-pass generator against.
#include <cstddef>
#include <random>

extern "C" void plain_reference_normals(unsigned seed, unsigned stream, size_t num_chains, size_t count_per_chain,
                                        int fresh_per_chain, double scale, double* out) {
  std::seed_seq ss{seed, stream};
  std::mt19937_64 rng(ss);
  std::normal_distribution<double> shared(0.0, 1.0);
  for (size_t c = 0; c < num_chains; ++c) {
    std::normal_distribution<double> fresh(0.0, 1.0);
    std::normal_distribution<double>& normal = fresh_per_chain ? fresh : shared;
    for (size_t i = 0; i < count_per_chain; ++i) out[c * count_per_chain + i] = normal(rng);
    for (size_t i = 0; i < count_per_chain; ++i) out[c * count_per_chain + i] *= scale;
  }
}
