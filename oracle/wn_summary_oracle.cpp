/* wn_summary_oracle.cpp -- CPU oracle of the reference's posterior summaries.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain restatement of include/walnutpie/summary.hpp (flatironinstitute/walnuts) for ragged collections of Markov
 * chains: mean :370-378, sample_variance :396-405, sample_standard_deviation :423-426, quantiles :483-514,
 * autocovariance :529-545 (+ detail::autocovariance_col :55-73), r_hat :593-619, effective_sample_size :663-749,
 * monte_carlo_standard_error :764-768.  Only tests/ may load it; the product (walnuts_amd/, include/) never does.
 *
 * Chains come in the layout of MarkovChainsUnified (summary.hpp:251-356): one row-major [num_draws][dims] block of
 * draws with the chains stacked, plus the chain sizes.
 *
 * Where this differs from the reference in ARITHMETIC ORDER (results agree to rounding, ~1e-16 relative):
 *   - sums run left to right (over chains: in runs of 256 chains, see chain_sum); Eigen's colwise()/sum()
 *     reductions are packetised;
 *   - the autocovariance is the direct sum  acov[t] = (1/N) sum_n (y[n]-ybar)(y[n+t]-ybar)  that the reference's
 *     zero-padded FFT evaluates (summary.hpp:55-73; its own test checks the FFT against exactly this direct form,
 *     tests/summary_test.cpp:610-627,681-693).  Eigen::FFT (kissfft) is not in /root/reference.
 * Pinned by the reference's known answers: tests/summary_test.cpp (means/variances :248-345, quantiles :534-568,
 * autocovariance :644-679, R-hat :825-880, ESS/MCSE :1073-1085,1117-1132,1182-1192) -> tests/test_summary_oracle.py
 * through tests/golden/summary_reference.json.
 */
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Chains {
  const double* x;  // [N][D] row-major
  size_t D;
  std::vector<size_t> len, start;
  size_t N = 0;
  Chains(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains) : x(draws), D(dims) {
    if (num_chains == 0) throw std::invalid_argument("require at least one chain");
    for (size_t m = 0; m < num_chains; ++m) {
      if (sizes[m] < 1) throw std::invalid_argument("each chain must have at least one draw");  // summary.hpp:139-150
      start.push_back(N);
      len.push_back(static_cast<size_t>(sizes[m]));
      N += static_cast<size_t>(sizes[m]);
    }
  }
  size_t K() const { return len.size(); }
  double at(size_t m, size_t n, size_t d) const { return x[(start[m] + n) * D + d]; }
  size_t min_len() const { return *std::min_element(len.begin(), len.end()); }
};

// Sums over chains: runs of kChainBlock consecutive chains summed left to right, then the run totals left to right.
// Up to kChainBlock chains this is exactly the reference's left-to-right loop over chains; beyond, the grouping is
// the device's (walnuts_amd/csrc/wn_summary.hip, block_sum_kernel), which keeps the comparison bit-exact.
constexpr size_t kChainBlock = 256;
template <class F>
double chain_sum(size_t K, F term) {
  double total = 0;
  for (size_t b0 = 0; b0 < K; b0 += kChainBlock) {
    double s = 0;
    for (size_t k = b0; k < std::min(K, b0 + kChainBlock); ++k) s += term(k);
    total += s;
  }
  return total;
}

// detail::col_means (:19-22) and detail::sample_variance (:93-99) of one chain
void chain_moments(const Chains& c, size_t m, double* mean, double* var) {
  const size_t n_m = c.len[m];
  for (size_t d = 0; d < c.D; ++d) {
    double s = 0;
    for (size_t n = 0; n < n_m; ++n) s += c.at(m, n, d);
    mean[d] = s / static_cast<double>(n_m);
    double q = 0;
    for (size_t n = 0; n < n_m; ++n) {
      const double r = c.at(m, n, d) - mean[d];
      q += r * r;
    }
    var[d] = q / static_cast<double>(static_cast<int64_t>(n_m) - 1);  // 0/0 = NaN for a single draw (:341-353)
  }
}

// sample variance over the rows of a [K][D] matrix (detail::sample_variance(draws), :101-105)
void rows_sample_variance(const std::vector<double>& a, size_t K, size_t D, double* out) {
  for (size_t d = 0; d < D; ++d) {
    const double mu = chain_sum(K, [&](size_t k) { return a[k * D + d]; }) / static_cast<double>(K);
    const double q = chain_sum(K, [&](size_t k) { return (a[k * D + d] - mu) * (a[k * D + d] - mu); });
    out[d] = q / static_cast<double>(static_cast<int64_t>(K) - 1);
  }
}
void rows_mean(const std::vector<double>& a, size_t K, size_t D, double* out) {
  for (size_t d = 0; d < D; ++d) out[d] = chain_sum(K, [&](size_t k) { return a[k * D + d]; }) / static_cast<double>(K);
}

void mean(const Chains& c, double* out) {  // :370-378: per-chain column sums added chain by chain, / num_draws
  for (size_t d = 0; d < c.D; ++d) {
    const double total = chain_sum(c.K(), [&](size_t m) {
      double s = 0;
      for (size_t n = 0; n < c.len[m]; ++n) s += c.at(m, n, d);
      return s;
    });
    out[d] = total / static_cast<double>(c.N);
  }
}

void sample_variance(const Chains& c, double* out) {  // :396-405
  std::vector<double> mu(c.D);
  mean(c, mu.data());
  for (size_t d = 0; d < c.D; ++d) {
    const double sum_sq = chain_sum(c.K(), [&](size_t m) {
      double q = 0;
      for (size_t n = 0; n < c.len[m]; ++n) {
        const double r = c.at(m, n, d) - mu[d];
        q += r * r;
      }
      return q;
    });
    out[d] = sum_sq / static_cast<double>(static_cast<int64_t>(c.N) - 1);
  }
}

// all lags of one chain and one column, detail::autocovariance_col (:55-73) as a direct sum
void autocovariance_col(const Chains& c, size_t m, size_t d, double* ac /*[len]*/) {
  const size_t N = c.len[m];
  double s = 0;
  for (size_t n = 0; n < N; ++n) s += c.at(m, n, d);
  const double ybar = s / static_cast<double>(N);
  for (size_t t = 0; t < N; ++t) {
    double a = 0;
    for (size_t n = 0; n + t < N; ++n) a += (c.at(m, n, d) - ybar) * (c.at(m, n + t, d) - ybar);
    ac[t] = a / static_cast<double>(N);  // biased estimate, :70-71
  }
}

void autocovariance(const Chains& c, double* out /*[N][D]*/) {  // :529-545
  std::vector<double> ac;
  for (size_t m = 0; m < c.K(); ++m) {
    ac.resize(c.len[m]);
    for (size_t d = 0; d < c.D; ++d) {
      autocovariance_col(c, m, d, ac.data());
      for (size_t t = 0; t < c.len[m]; ++t) out[(c.start[m] + t) * c.D + d] = ac[t];
    }
  }
}

void r_hat(const Chains& c, double* out) {  // :593-619
  if (c.K() < 2) throw std::invalid_argument("require at least two chains to compute R-hat");
  for (size_t m = 0; m < c.K(); ++m)
    if (c.len[m] < 3) throw std::invalid_argument("each chain must have at least 3 draws");
  const size_t K = c.K(), D = c.D;
  std::vector<double> mu(K * D), sig(K * D), var_mu(D), mean_sig(D);
  for (size_t m = 0; m < K; ++m) chain_moments(c, m, &mu[m * D], &sig[m * D]);
  rows_sample_variance(mu, K, D, var_mu.data());
  rows_mean(sig, K, D, mean_sig.data());
  for (size_t d = 0; d < D; ++d) out[d] = std::sqrt(1.0 + var_mu[d] / mean_sig[d]);
}

void effective_sample_size(const Chains& c, double* out) {  // :663-749
  if (c.N < 3) throw std::invalid_argument("chains must have at least 3 draws");
  const size_t K = c.K(), D = c.D, min_len = c.min_len();
  // the reference indexes rho_hat_t(1) and rho_hat_t(max_t + 1) unconditionally (:705,:738): below three draws in
  // the shortest chain that is out of bounds there; stated as an error here
  if (min_len < 3) throw std::invalid_argument("each chain must have at least 3 draws");
  std::vector<double> means(K * D), vars(K * D), W(D), var_plus(D), between(D);
  for (size_t k = 0; k < K; ++k) chain_moments(c, k, &means[k * D], &vars[k * D]);
  rows_mean(vars, K, D, W.data());
  var_plus = W;
  if (K > 1) {
    rows_sample_variance(means, K, D, between.data());
    for (size_t d = 0; d < D; ++d) var_plus[d] += between[d];
  }
  std::vector<double> acov(c.N * D);
  autocovariance(c, acov.data());
  for (size_t d = 0; d < D; ++d) {
    const double w_d = W[d], vp_d = var_plus[d];
    auto mean_acov_at_lag = [&](size_t t) {
      return chain_sum(K, [&](size_t k) { return acov[(c.start[k] + t) * D + d]; }) / static_cast<double>(K);
    };
    std::vector<double> rho(min_len, 0.0);
    double even = 1.0;
    rho[0] = even;
    double odd = 1.0 - (w_d - mean_acov_at_lag(1)) / vp_d;
    rho[1] = odd;
    // Geyer's initial positive + monotone sequence on paired lags (:712-729)
    std::ptrdiff_t t = 1;
    const std::ptrdiff_t bound = static_cast<std::ptrdiff_t>(min_len) - 4;
    while (t < bound && (even + odd) > 0.0) {
      even = 1.0 - (w_d - mean_acov_at_lag(static_cast<size_t>(t + 1))) / vp_d;
      odd = 1.0 - (w_d - mean_acov_at_lag(static_cast<size_t>(t + 2))) / vp_d;
      if ((even + odd) >= 0.0) {
        rho[t + 1] = even;
        rho[t + 2] = odd;
      }
      if (rho[t + 1] + rho[t + 2] > rho[t - 1] + rho[t]) {
        rho[t + 1] = (rho[t - 1] + rho[t]) / 2.0;
        rho[t + 2] = rho[t + 1];
      }
      t += 2;
    }
    const std::ptrdiff_t max_t = t;
    if (even > 0.0) rho[max_t + 1] = even;  // antithetic-tail correction (:733-735)
    double head = 0;
    for (std::ptrdiff_t i = 0; i < max_t; ++i) head += rho[i];
    double tau = -1.0 + 2.0 * head + rho[max_t + 1];
    tau = std::max(tau, 1.0 / std::log10(static_cast<double>(c.N)));  // :741-742
    out[d] = static_cast<double>(c.N) / tau;
  }
}

void quantiles(const Chains& c, const double* probs, size_t K, double* out /*[K][D]*/) {  // :483-514
  for (size_t k = 0; k < K; ++k)
    if (!(probs[k] >= 0) || !(probs[k] <= 1)) throw std::invalid_argument("probs must be in [0, 1]");
  std::vector<double> col(c.N);
  const double n_minus_1 = static_cast<double>(static_cast<int64_t>(c.N) - 1);
  for (size_t d = 0; d < c.D; ++d) {
    for (size_t i = 0; i < c.N; ++i) col[i] = c.x[i * c.D + d];
    std::sort(col.begin(), col.end());
    for (size_t k = 0; k < K; ++k) {
      const double h = probs[k] * n_minus_1;
      const int64_t lo = static_cast<int64_t>(std::floor(h));
      const int64_t hi = std::min<int64_t>(lo + 1, static_cast<int64_t>(c.N) - 1);
      const double frac = h - static_cast<double>(lo);
      out[k * c.D + d] = col[lo] + frac * (col[hi] - col[lo]);
    }
  }
}

thread_local std::string last_error;

template <class F>
int guarded(F f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument& e) {
    last_error = e.what();
    return 1;  // config error: the reference throws std::invalid_argument
  } catch (const std::exception& e) {
    last_error = e.what();
    return 2;
  }
}

}  // namespace

extern "C" {

const char* wnso_last_error() { return last_error.c_str(); }

int wnso_mean(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains, double* out) {
  return guarded([&] { mean(Chains(draws, dims, sizes, num_chains), out); });
}
int wnso_sample_variance(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains, double* out) {
  return guarded([&] { sample_variance(Chains(draws, dims, sizes, num_chains), out); });
}
int wnso_sample_standard_deviation(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains,
                                   double* out) {
  return guarded([&] {  // :423-426
    sample_variance(Chains(draws, dims, sizes, num_chains), out);
    for (size_t d = 0; d < dims; ++d) out[d] = std::sqrt(out[d]);
  });
}
int wnso_quantiles(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains, const double* probs,
                   size_t num_probs, double* out) {
  return guarded([&] { quantiles(Chains(draws, dims, sizes, num_chains), probs, num_probs, out); });
}
int wnso_autocovariance(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains, double* out) {
  return guarded([&] { autocovariance(Chains(draws, dims, sizes, num_chains), out); });
}
int wnso_r_hat(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains, double* out) {
  return guarded([&] { r_hat(Chains(draws, dims, sizes, num_chains), out); });
}
int wnso_effective_sample_size(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains,
                               double* out) {
  return guarded([&] { effective_sample_size(Chains(draws, dims, sizes, num_chains), out); });
}
int wnso_monte_carlo_standard_error(const double* draws, size_t dims, const int64_t* sizes, size_t num_chains,
                                    double* out) {
  return guarded([&] {  // :764-768
    const Chains c(draws, dims, sizes, num_chains);
    std::vector<double> ess(dims), var(dims);
    effective_sample_size(c, ess.data());
    sample_variance(c, var.data());
    for (size_t d = 0; d < dims; ++d) out[d] = std::sqrt(var[d]) / std::sqrt(ess[d]);
  });
}
/* detail::fft_next_good_size (:39-52): the reference's FFT padding; kept for its known answers */
int64_t wnso_fft_next_good_size(int64_t n) {
  if (n <= 2) return 2;
  for (;; ++n) {
    int64_t m = n;
    for (int64_t f : {2, 3, 5})
      while (m % f == 0) m /= f;
    if (m <= 1) return n;
  }
}

}  // extern "C"
