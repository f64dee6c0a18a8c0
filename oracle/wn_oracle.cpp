// wn_oracle.cpp -- CPU oracle.  TEST INFRASTRUCTURE ONLY (see wn_oracle.h).
//
// A from-scratch restatement, over flat double arrays, of the reference's
// per-chain Walnuts path.  Every routine names the reference lines it follows
// (paths relative to the reference's include/walnutpie/).  Element-wise
// arithmetic keeps the reference's association order so that a build with
// -ffp-contract=off gives the reference's element-wise bits; reductions are
// left-to-right (reduce_lanes == 0) or in the device engine's order
// (reduce_lanes == L), see Reducer below.
//
// Build: oracle/Makefile (g++ -O3 -ffp-contract=off, no -march).

#include "wn_oracle.h"
#include "wn_oracle_math.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <thread>
#include <vector>

namespace {

using Vec = std::vector<double>;
constexpr double kInf = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------------------
// scalar maths: what the reference executes (libm) or what the device executes
// ---------------------------------------------------------------------------
struct MathOps {
  int mode = WNO_MATH_LIBM;
  double exp(double x) const { return mode == WNO_MATH_LIBM ? std::exp(x) : wno_exp(x); }
  double log(double x) const { return mode == WNO_MATH_LIBM ? std::log(x) : wno_log(x); }
  double pow(double x, double y) const {
    return mode == WNO_MATH_LIBM ? std::pow(x, y) : wno_pow_pos(x, y);
  }
};

// ---------------------------------------------------------------------------
// reductions.  L == 0: s = 0; s += f(0); s += f(1); ...
// L == WNO_REDUCE_EIGEN_SSE2 (-2): the order of Eigen 3.4's linear-vectorised redux with 2-lane packets -- what
//   `.sum()` / `.dot()` execute in the reference as its CMake files build it on x86-64 (plain -O3: SSE2, Packet2d;
//   util.hpp:222, walnuts.hpp:199-200, examples/walnutpie_api.cpp:41).  Eigen is NOT under /root/reference (cloned at
//   configure time, EIGEN_TAG 3.4.0, CMakeLists.txt:28-41), so this is a restatement of its published algorithm
//   (Eigen/src/Core/Redux.h, redux_impl<Func, Evaluator, LinearVectorizedTraversal, NoUnrolling>::run): with
//   P = 2, n2 = n rounded down to a multiple of 2P, n1 = n rounded down to a multiple of P,
//       fewer than P elements: res = x0; res += x1; ...
//       p0 = (x0, x1); if n1 > P: p1 = (x2, x3); for i = 2P, 4P, ... < n2: p0 += (x_i, x_i+1), p1 += (x_i+2, x_i+3);
//       p0 += p1; if n1 > n2: p0 += (x_n2, x_n2+1);
//       res = p0[0] + p0[1]  (predux<Packet2d>);  res += x_i for i = n1 .. n-1
//   (the element expression is evaluated per packet with pmul / padd: element-wise IEEE operations, no FMA in SSE2.)
// L > 0: the device order.  Lane l (0 <= l < L) owns elements
// (k*L + l)*2 + {0,1}, k = 0,1,...; it adds its elements in increasing index
// order into a partial that starts at +0.0; each group of 64 lanes (one
// wavefront) then runs an xor butterfly with offsets 32,1,2,4,8,16; wavefront
// totals are added left to right.
// ---------------------------------------------------------------------------
struct Reducer {
  int L = 0;
  // sum of a(i) * b(i) in the same order; `fused`: every partial is fma(a, b, partial), as the kernels' mad() does
  template <class A, class B>
  double sum_prod(size_t n, A a, B b, bool fused) const {
    if (!fused || L == WNO_REDUCE_EIGEN_SSE2) return sum(n, [&](size_t i) { return a(i) * b(i); });
    if (L <= 0) {
      double s = 0.0;
      for (size_t i = 0; i < n; ++i) s = std::fma(a(i), b(i), s);
      return s;
    }
    const size_t lanes = static_cast<size_t>(L);
    std::vector<double> part(lanes, 0.0);
    const size_t slots = (n + 2 * lanes - 1) / (2 * lanes);
    for (size_t l = 0; l < lanes; ++l) {
      double p = 0.0;
      for (size_t k = 0; k < slots; ++k) {
        size_t i0 = (k * lanes + l) * 2;
        if (i0 < n) p = std::fma(a(i0), b(i0), p);
        if (i0 + 1 < n) p = std::fma(a(i0 + 1), b(i0 + 1), p);
      }
      part[l] = p;
    }
    return butterfly(part);
  }
  template <class F>
  double sum(size_t n, F f) const {
    if (L == WNO_REDUCE_EIGEN_SSE2) {
      constexpr size_t P = 2;
      if (n == 0) return 0.0;
      const size_t n2 = (n / (2 * P)) * (2 * P), n1 = (n / P) * P;
      double res;
      if (n1 == 0) {
        res = f(0);
        for (size_t i = 1; i < n; ++i) res = res + f(i);
        return res;
      }
      double p0[P], p1[P];
      for (size_t k = 0; k < P; ++k) p0[k] = f(k);
      if (n1 > P) {
        for (size_t k = 0; k < P; ++k) p1[k] = f(P + k);
        for (size_t i = 2 * P; i < n2; i += 2 * P) {
          for (size_t k = 0; k < P; ++k) p0[k] = p0[k] + f(i + k);
          for (size_t k = 0; k < P; ++k) p1[k] = p1[k] + f(i + P + k);
        }
        for (size_t k = 0; k < P; ++k) p0[k] = p0[k] + p1[k];
        if (n1 > n2)
          for (size_t k = 0; k < P; ++k) p0[k] = p0[k] + f(n2 + k);
      }
      res = p0[0] + p0[1];
      for (size_t i = n1; i < n; ++i) res = res + f(i);
      return res;
    }
    if (L <= 0) {
      double s = 0.0;
      for (size_t i = 0; i < n; ++i) s += f(i);
      return s;
    }
    const size_t lanes = static_cast<size_t>(L);
    std::vector<double> part(lanes, 0.0);
    const size_t slots = (n + 2 * lanes - 1) / (2 * lanes);
    for (size_t l = 0; l < lanes; ++l) {
      double p = 0.0;
      for (size_t k = 0; k < slots; ++k) {
        size_t i0 = (k * lanes + l) * 2;
        if (i0 < n) p += f(i0);
        if (i0 + 1 < n) p += f(i0 + 1);
      }
      part[l] = p;
    }
    return butterfly(part);
  }
  // lane partials -> total: xor butterfly inside each group of 64 lanes, groups added left to right
  double butterfly(std::vector<double>& part) const {
    const size_t lanes = part.size();
    double total = 0.0;
    bool first = true;
    std::vector<double> tmp(64);
    for (size_t w = 0; w < lanes; w += 64) {
      size_t width = std::min<size_t>(64, lanes - w);
      double* v = part.data() + w;
      // The engine's widths are multiples of 64 (one or more wavefronts per chain) or 16 (the row kernels, four
      // chains per wavefront, walnuts_amd/csrc/wn_row.h): a row's butterfly has the offsets 1, 2, 4, 8 only -- an offset
      // that would leave the group does not exist there and is skipped.
      for (size_t off : {size_t{32}, size_t{1}, size_t{2}, size_t{4}, size_t{8}, size_t{16}}) {
        if (off >= width) continue;
        for (size_t l = 0; l < width; ++l) {
          size_t p = l ^ off;
          tmp[l] = v[l] + (p < width ? v[p] : 0.0);
        }
        for (size_t l = 0; l < width; ++l) v[l] = tmp[l];
      }
      if (first) {
        total = v[0];
        first = false;
      } else {
        total = total + v[0];
      }
    }
    return total;
  }
};

// ---------------------------------------------------------------------------
// target densities (the LogpGrad contract, concepts.hpp:258-262)
// ---------------------------------------------------------------------------
struct Model {
  int kind = WNO_MODEL_STD_NORMAL;
  size_t D = 0;
  Vec params;
  MathOps m;
  Reducer r;

  // `fused`: the log-density sums accumulate with fma (the device's Cx::mad inside a transition kernel running with
  // fused multiply-adds); gradients are element-wise and never fused
  void operator()(const double* x, double& logp, double* g, bool fused = false) const {
    switch (kind) {
      case WNO_MODEL_STD_NORMAL: {
        // examples/walnutpie_api.cpp:37-41, tests/test_util.hpp:16-22
        logp = -0.5 * r.sum_prod(D, [&](size_t i) { return x[i]; }, [&](size_t i) { return x[i]; }, fused);
        for (size_t i = 0; i < D; ++i) g[i] = -x[i];
        break;
      }
      case WNO_MODEL_DIAG_NORMAL: {
        // examples/examples.cpp:20-31 with sigma_sq supplied per coordinate
        const double* s2 = params.data();
        if (m.mode == WNO_MATH_PORTABLE) {
          // device arithmetic: the engine multiplies by 1/sigma_sq, rounded once (wn_engine_create)
          logp = r.sum_prod(D, [&](size_t i) { return -0.5 * x[i] * x[i]; }, [&](size_t i) { return 1.0 / s2[i]; }, fused);
          for (size_t i = 0; i < D; ++i) g[i] = -x[i] * (1.0 / s2[i]);
        } else {
          // the example's own scalar loop (examples.cpp:26-30): left to right whatever Eigen's redux does elsewhere
          Reducer loop;
          loop.L = r.L == WNO_REDUCE_EIGEN_SSE2 ? 0 : r.L;
          logp = loop.sum(D, [&](size_t i) { return -0.5 * x[i] * x[i] / s2[i]; });
          for (size_t i = 0; i < D; ++i) g[i] = -x[i] / s2[i];
        }
        break;
      }
      case WNO_MODEL_FUNNEL: {
        // Neal's funnel (not in the reference): v = x0 ~
        // N(0, 3^2), x_i | v ~ N(0, e^v), i >= 1.
        const double v = x[0];
        auto x_or_0 = [&](size_t i) { return i == 0 ? 0.0 : x[i]; };
        const double S = r.sum_prod(D, x_or_0, x_or_0, fused);
        const double ev = m.exp(-v);
        const double hd = 0.5 * static_cast<double>(D - 1);
        const double hev = 0.5 * ev;
        if (m.mode == WNO_MATH_PORTABLE) {
          // device arithmetic: the engine multiplies by the once-rounded reciprocals (wn_models.h)
          logp = ((-(v * v) * (1.0 / 18.0)) - hev * S) - hd * v;
          g[0] = ((-v * (1.0 / 9.0)) + hev * S) - hd;
        } else {
          logp = ((-(v * v) / 18.0) - hev * S) - hd * v;
          g[0] = ((-v / 9.0) + hev * S) - hd;
        }
        for (size_t i = 1; i < D; ++i) g[i] = -(x[i] * ev);
        break;
      }
      case WNO_MODEL_RW1: {
        // examples/examples.cpp:34-49: normal(0, Sigma), Sigma[i,j] = rho^|i-j|, rho = 0.99
        const double rho = 0.99;
        const double sigma_sq = 1.0 - rho * rho;
        const double inv_sigma_sq = 1.0 / sigma_sq;
        if (m.mode == WNO_MATH_PORTABLE) {
          // device arithmetic: the log density is -0.5 * (sum of the per-coordinate terms, device summation order)
          auto rr = [&](size_t n) { return n == 0 ? x[0] : x[n] - rho * x[n - 1]; };
          logp = -0.5 * r.sum_prod(D, rr, [&](size_t n) { return n == 0 ? x[0] : rr(n) * inv_sigma_sq; }, fused);
        } else {
          logp = -0.5 * x[0] * x[0];
          for (size_t n = 1; n < D; ++n) {
            const double rr = x[n] - rho * x[n - 1];
            const double w = rr * inv_sigma_sq;
            logp -= 0.5 * rr * w;
          }
        }
        // grad[n] = (0 - w[n]) + rho * w[n + 1] in both modes (element-wise: no summation order involved)
        for (size_t n = 0; n < D; ++n) {
          double gn = n == 0 ? -x[0] : -((x[n] - rho * x[n - 1]) * inv_sigma_sq);
          if (n + 1 < D) gn = gn + rho * ((x[n + 1] - rho * x[n]) * inv_sigma_sq);
          g[n] = gn;
        }
        break;
      }
      default:
        logp = -kInf;
        for (size_t i = 0; i < D; ++i) g[i] = 0.0;
    }
  }
};

// util.hpp:220-223
double logp_momentum(const Reducer& r, size_t n, const double* rho, const double* im, bool fused = false) {
  return -0.5 * r.sum_prod(n, [&](size_t i) { return im[i]; }, [&](size_t i) { return rho[i] * rho[i]; }, fused);
}

// util.hpp:174-183
double log_sum_exp(const MathOps& m, double x1, double x2) {
  double mx = std::fmax(x1, x2);
  if (std::isnan(x1) || std::isnan(x2)) return std::numeric_limits<double>::quiet_NaN();
  if (std::isinf(mx) || std::isnan(x1 + x2)) return std::fmax(x1, x2);
  return mx + m.log(m.exp(x1 - mx) + m.exp(x2 - mx));
}

// ---------------------------------------------------------------------------
// randomness (util.hpp:78-162)
// ---------------------------------------------------------------------------
struct RandomSource {
  int64_t scalar_draws = 0;
  virtual ~RandomSource() = default;
  virtual double uniform01() = 0;
  virtual bool bernoulli() = 0;
  virtual void normals(size_t n, double* out) = 0;
  virtual void begin_transition(uint32_t /*t*/) { scalar_draws = 0; }
  virtual void reset_distributions() {}
};

// libstdc++ distributions over a borrowed engine, exactly the members of
// detail::Random (util.hpp:91-92,102,112,124-127).  The normal distribution's
// cached second variate lives here, as it does in the reference object.
template <class Eng>
struct StdRandom final : RandomSource {
  Eng* eng;
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  std::bernoulli_distribution binary{0.5};
  std::normal_distribution<double> normal{0.0, 1.0};
  explicit StdRandom(Eng* e) : eng(e) {}
  double uniform01() override { ++scalar_draws; return unif(*eng); }
  bool bernoulli() override { ++scalar_draws; return binary(*eng); }
  void normals(size_t n, double* out) override {
    for (size_t i = 0; i < n; ++i) out[i] = normal(*eng);
  }
  // a new detail::Random is made over the same engine (walnuts.hpp:642)
  void reset_distributions() override {
    unif = std::uniform_real_distribution<double>(0.0, 1.0);
    binary = std::bernoulli_distribution(0.5);
    normal = std::normal_distribution<double>(0.0, 1.0);
  }
};

// the device engine's counter-based stream
struct PhiloxRandom final : RandomSource {
  uint64_t seed;
  uint32_t chain;
  uint32_t transition = 0;
  uint32_t index = 0;
  uint32_t normal_stream = WNO_STREAM_MOMENTUM;
  PhiloxRandom(uint64_t s, uint32_t c) : seed(s), chain(c) {}
  void begin_transition(uint32_t t) override { transition = t; index = 0; scalar_draws = 0; }
  double uniform01() override {
    ++scalar_draws;
    return wno_philox_uniform(seed, chain, transition, WNO_STREAM_TREE, index++);
  }
  bool bernoulli() override { return uniform01() < 0.5; }
  void normals(size_t n, double* out) override {
    for (size_t p = 0; 2 * p < n; ++p) {
      double z0, z1;
      wno_philox_normal_pair(seed, chain, transition, normal_stream, static_cast<uint32_t>(p), &z0, &z1);
      out[2 * p] = z0;
      if (2 * p + 1 < n) out[2 * p + 1] = z1;
    }
  }
};

// host-supplied variates for one transition (the device engine's kRngBuffer mode)
struct BufferRandom final : RandomSource {
  const double* z;
  const double* u;
  size_t n_u;
  BufferRandom(const double* zz, const double* uu, size_t nu) : z(zz), u(uu), n_u(nu) {}
  double uniform01() override {
    const size_t j = static_cast<size_t>(scalar_draws++);
    return j < n_u ? u[j] : 0.5;
  }
  bool bernoulli() override { return uniform01() < 0.5; }
  void normals(size_t n, double* out) override { std::copy(z, z + n, out); }
};

// ---------------------------------------------------------------------------
// Adam on log step size (adam.hpp:35-109)
// ---------------------------------------------------------------------------
struct Adam {
  double theta = 0, m = 0, v = 0, t = 0, b1pow = 1, b2pow = 1;
  double target = 0.8, lr = 0.05, b1 = 0.8, b2 = 0.9, eps = 1e-4, decay = 0.5;
  MathOps mo;
  void init(double step_init, const wno_config& c, const MathOps& mops) {
    mo = mops;
    theta = mo.log(step_init);  // adam.hpp:51
    m = v = t = 0;
    b1pow = b2pow = 1;
    target = c.step_accept_rate_target;
    lr = c.step_learning_rate;
    b1 = c.step_gradient_decay;
    b2 = c.step_sq_gradient_decay;
    eps = c.step_stabilization;
    decay = c.step_learn_rate_decay;
  }
  void observe(double alpha) {  // adam.hpp:70-86
    t += 1;
    b1pow *= b1;
    b2pow *= b2;
    double grad = target - alpha;
    m = b1 * m + (1 - b1) * grad;
    v = b2 * v + (1 - b2) * grad * grad;
    double m_hat = m / (1 - b1pow);
    double v_hat = v / (1 - b2pow);
    double lr_t = lr / mo.pow(t, decay);
    double denom = std::sqrt(v_hat) + eps;
    theta -= lr_t * m_hat / denom;
  }
  double step_size() const { return mo.exp(theta); }  // adam.hpp:93
};

// ---------------------------------------------------------------------------
// discounted Welford (online_moments.hpp:125-247)
// ---------------------------------------------------------------------------
// Device arithmetic (WNO_MATH_PORTABLE) divides a plane's elements by the estimator's weight the way the kernels do
// (walnuts_amd/csrc/wn_devmath.h, SharedDivisor): r = 1 / w once, then q0 = a r, q = fma(fma(-q0, w, a), r, q0) -- the
// correctly rounded quotient for every finite numerator in the normal range (the all-ones significand aside), restated
// here operation for operation so that the comparison is exact in the corner cases too.  Reference arithmetic: `/`.
static inline double div_shared(double a, double w, double r) {
  const double q0 = a * r;
  return std::fma(std::fma(-q0, w, a), r, q0);
}

struct OnlineMoments {
  double weight = 0;
  bool shared_div = false;
  Vec mean, ssd;
  void init(double w, const double* mean0, const double* var0, size_t n) {  // :151-159
    weight = w;
    mean.assign(mean0, mean0 + n);
    ssd.resize(n);
    for (size_t i = 0; i < n; ++i) ssd[i] = w * var0[i];
  }
  // :184-191.  `delta` there is a lazy Eigen expression over `mean_`, so the
  // second use of it sees the UPDATED mean: the increment is (y - mean_new)^2.
  void observe(double discount, const double* y) {
    weight = discount * weight + 1;
    const size_t n = mean.size();
    if (shared_div) {
      const double r = 1.0 / weight;
      for (size_t i = 0; i < n; ++i) mean[i] += div_shared(y[i] - mean[i], weight, r);
    } else {
      for (size_t i = 0; i < n; ++i) mean[i] += (y[i] - mean[i]) / weight;
    }
    for (size_t i = 0; i < n; ++i) ssd[i] = discount * ssd[i] + (y[i] - mean[i]) * (y[i] - mean[i]);
  }
  double variance(size_t i) const {  // :225-230
    if (!(weight > 0)) return 1.0;
    return shared_div ? div_shared(ssd[i], weight, 1.0 / weight) : ssd[i] / weight;
  }
};

// adaptive_walnuts.hpp:25-105
struct MassEstimator {
  double init_count = 4;
  OnlineMoments draw_var, score_var;
  void init(double count, const double* mass, size_t n, bool shared_div = false) {  // :54-62
    init_count = count;
    draw_var.shared_div = score_var.shared_div = shared_div;
    Vec zero(n, 0.0), inv(n);
    for (size_t i = 0; i < n; ++i) inv[i] = 1.0 / mass[i];
    score_var.init(count, zero.data(), mass, n);
    draw_var.init(count, zero.data(), inv.data(), n);
  }
  void observe(const double* theta, const double* grad, size_t iteration) {  // :74-80
    double discount = 1.0 - 1.0 / (init_count + static_cast<double>(iteration));
    draw_var.observe(discount, theta);
    score_var.observe(discount, grad);
  }
  void inv_mass(double* out) const {  // :89-94
    const size_t n = draw_var.mean.size();
    for (size_t i = 0; i < n; ++i) out[i] = std::sqrt(draw_var.variance(i) / score_var.variance(i));
  }
};

// adaptive_walnuts.hpp:119-164
struct MinMicro {
  double target = 15, total = 2.0, count = 1.0;
  size_t floor_ = 1;
  void observe(size_t macro_steps) { total += static_cast<double>(macro_steps); count += 1; }
  size_t value() const {
    double mean_micro = total / count;
    double mm = mean_micro / target;
    return std::max(floor_, static_cast<size_t>(std::lround(mm)));
  }
};

// ---------------------------------------------------------------------------
// the trajectory (walnuts.hpp, namespace detail)
// ---------------------------------------------------------------------------
struct Span {  // walnuts.hpp:34-131
  Vec th_bk, rho_bk, g_bk;
  double lj_bk = 0;
  Vec th_fw, rho_fw, g_fw;
  double lj_fw = 0;
  Vec th_sel, g_sel;
  double lp_sel = 0;
  double logsum = 0;
  // device arithmetic (WNO_MATH_PORTABLE): the span's weight in the linear domain, exp(energy - Ctx::w_ref), valid
  // for the reference energy of `epoch` (Ctx::weight_now brings it up to date)
  double w = 0;
  size_t epoch = 0;
};

struct TraceRec {
  double f[WNO_TRACE_FIELDS];
};

struct Ctx {
  const Model* model;
  const double* im;  // inverse mass diagonal
  size_t D;
  double step;
  size_t max_halvings;
  size_t min_micro;
  double max_error;
  MathOps mo;
  Reducer red;
  bool fma = false;  // the transition kernels' fused arithmetic (wno_config::fma)
  RandomSource* rng;
  Adam* adam;  // null: NoOpStepSizeAdapter (walnuts.hpp:572-587)
  int64_t grad_evals = 0;
  std::vector<TraceRec>* trace = nullptr;
  double last_alpha = 0;
  // Near-tie audit (SURVEY.md section 8d "parity gate"): how many of this transition's decisions sat within
  // tie_tol (relative to the magnitude of the compared quantities) of their threshold -- the only places where
  // a different summation order or a last-ulp exp/log can change the tree.
  // [0] |H0 - H1| <= max_error (walnuts.hpp:339, :234)  [1] U-turn signs (:199-200)  [2] log u < delta (:379)
  double tie_tol = 0.0;
  int64_t ties[3] = {0, 0, 0};
  int64_t decisions[3] = {0, 0, 0};
  void audit(int kind, double margin, double scale) {
    ++decisions[kind];
    if (tie_tol > 0 && std::fabs(margin) <= tie_tol * std::max(1.0, scale)) ++ties[kind];
  }
  // ---- the device's span weights (walnuts_amd/csrc/wn_traj.h, "span weights"; WNO_MATH_PORTABLE only) ----
  // The device carries combine()'s weights in the linear domain relative to a reference energy w_ref (the initial
  // point's, whose weight is exactly 1): a leaf weighs exp(logp_joint - w_ref), a merged span the sum of its halves,
  // and the acceptance tests compare u * total < w_new (Barker) / u * w_old < w_new (Metropolis).  A leaf whose
  // energy is more than 256 above w_ref moves the reference there: the device scales every live weight by
  // exp(old - new) on the spot; here the factors are kept in a list and a span's weight is brought up to date when
  // it is next read -- the same multiplications in the same order.
  double w_ref = 0;
  std::vector<double> rebase_factors;
  double leaf_weight(double lj) {
    const double x = lj - w_ref;
    if (x > 256.0) {
      rebase_factors.push_back(wno_exp_weight(-x));
      w_ref = lj;
      return 1.0;
    }
    return wno_exp_weight(x);
  }
  double weight_now(Span& s) {
    while (s.epoch < rebase_factors.size()) s.w = s.w * rebase_factors[s.epoch++];
    return s.w;
  }
};

inline void leap(Ctx& c, double step, double half, double* th, double* rho, double* g, double& logp_pos) {
  // walnuts.hpp:228-231 / :329-332
  const size_t D = c.D;
  if (c.fma) {  // the device's mad(): one rounding per multiply-add
    for (size_t i = 0; i < D; ++i) rho[i] = std::fma(half, g[i], rho[i]);
    for (size_t i = 0; i < D; ++i) th[i] = std::fma(step * c.im[i], rho[i], th[i]);
    (*c.model)(th, logp_pos, g, true);
    ++c.grad_evals;
    for (size_t i = 0; i < D; ++i) rho[i] = std::fma(half, g[i], rho[i]);
    return;
  }
  for (size_t i = 0; i < D; ++i) rho[i] += half * g[i];
  for (size_t i = 0; i < D; ++i) th[i] += step * c.im[i] * rho[i];
  (*c.model)(th, logp_pos, g);
  ++c.grad_evals;
  for (size_t i = 0; i < D; ++i) rho[i] += half * g[i];
}

// walnuts.hpp:218-235
bool within_tolerance(Ctx& c, double step, size_t num_steps, double logp_next, double* th, double* rho,
                      double* g) {
  double half = 0.5 * step;
  double logp = logp_next;
  for (size_t n = 0; n < num_steps; ++n) leap(c, step, half, th, rho, g, logp_next);
  logp_next += logp_momentum(c.red, c.D, rho, c.im, c.fma);
  c.audit(0, std::abs(logp_next - logp) - c.max_error, std::max(std::fabs(logp), std::fabs(logp_next)));
  return std::abs(logp_next - logp) <= c.max_error;
}

// walnuts.hpp:254-279
bool reversible(Ctx& c, double step, size_t num_steps, double logp_next, const Vec& th, const Vec& rho,
                const Vec& g) {
  if (num_steps == 1) return true;
  Vec th2(c.D), rho2(c.D), g2(c.D);
  while (num_steps >= 2 * c.min_micro) {
    th2 = th;
    for (size_t i = 0; i < c.D; ++i) rho2[i] = -rho[i];
    g2 = g;
    num_steps /= 2;
    step *= 2;
    if (within_tolerance(c, step, num_steps, logp_next, th2.data(), rho2.data(), g2.data())) return false;
  }
  return true;
}

// walnuts.hpp:307-345
bool macro_step(Ctx& c, bool forward, const Vec& th0, const Vec& rho0, const Vec& g0, double logp, Vec& th,
                Vec& rho, Vec& g, double& logp_pos_next, double& logp_next) {
  double step = forward ? c.step : -c.step;
  size_t num_steps = c.min_micro;
  for (size_t halvings = 0; halvings < c.max_halvings; ++halvings, num_steps *= 2, step *= 0.5) {
    th = th0;
    rho = rho0;
    g = g0;
    double half = 0.5 * step;
    for (size_t n = 0; n < num_steps; ++n) leap(c, step, half, th.data(), rho.data(), g.data(), logp_pos_next);
    logp_next = logp_pos_next + logp_momentum(c.red, c.D, rho.data(), c.im, c.fma);
    if (num_steps == c.min_micro) {
      double min_accept = c.mo.exp(-std::fabs(logp - logp_next));
      c.last_alpha = min_accept;
      if (c.adam) c.adam->observe(min_accept);
    }
    bool ok = std::fabs(logp - logp_next) <= c.max_error;
    c.audit(0, std::fabs(logp - logp_next) - c.max_error, std::max(std::fabs(logp), std::fabs(logp_next)));
    bool rev = false;
    if (ok) rev = reversible(c, step, num_steps, logp_next, th, rho, g);
    if (c.trace) {
      TraceRec r{{forward ? 1.0 : 0.0, static_cast<double>(halvings), static_cast<double>(num_steps), step, logp,
                  logp_next, ok ? 1.0 : 0.0, rev ? 1.0 : 0.0, logp_pos_next}};
      c.trace->push_back(r);
    }
    if (ok) return rev;
  }
  return false;
}

// walnuts.hpp:192-201 with order_forward_backward :153-160
bool uturn(Ctx& c, bool forward, const Span& s1, const Span& s2) {
  const Span& bk = forward ? s1 : s2;
  const Span& fw = forward ? s2 : s1;
  const double* im = c.im;
  auto sd = [&](size_t i) { return im[i] * (fw.th_fw[i] - bk.th_bk[i]); };
  double d_fw = c.red.sum_prod(c.D, [&](size_t i) { return fw.rho_fw[i]; }, sd, c.fma);
  double d_bk = c.red.sum_prod(c.D, [&](size_t i) { return bk.rho_bk[i]; }, sd, c.fma);
  if (c.tie_tol > 0) {  // scale of each product: the sum of the magnitudes of its terms
    double a_fw = 0, a_bk = 0;
    for (size_t i = 0; i < c.D; ++i) {
      a_fw += std::fabs(fw.rho_fw[i] * sd(i));
      a_bk += std::fabs(bk.rho_bk[i] * sd(i));
    }
    c.audit(1, d_fw, a_fw);
    c.audit(1, d_bk, a_bk);
  } else {
    c.decisions[1] += 2;
  }
  return d_fw < 0 || d_bk < 0;
}

// walnuts.hpp:368-387
Span combine(Ctx& c, bool metropolis, bool forward, Span&& s_old, Span&& s_new) {
  double total;
  bool update;
  double total_w = 0;
  if (c.mo.mode == WNO_MATH_PORTABLE) {
    // device arithmetic: the weights themselves (see Ctx::leaf_weight); log_sum_exp is never evaluated
    const double w_old = c.weight_now(s_old), w_new = c.weight_now(s_new);
    total_w = w_old + w_new;
    const double denom = metropolis ? w_old : total_w;
    const double u = c.rng->uniform01();
    update = u * denom < w_new;
    c.audit(2, std::log(u * denom) - std::log(w_new), 1.0);
    total = 0;  // (unused in this mode)
  } else {
    total = log_sum_exp(c.mo, s_old.logsum, s_new.logsum);
    double denom = metropolis ? s_old.logsum : total;
    double update_logprob = s_new.logsum - denom;
    const double log_u = c.mo.log(c.rng->uniform01());
    update = log_u < update_logprob;
    c.audit(2, log_u - update_logprob, std::max(std::fabs(s_new.logsum), std::fabs(denom)));
  }
  Span& sel = update ? s_new : s_old;
  Span out;
  out.th_sel = std::move(sel.th_sel);
  out.g_sel = std::move(sel.g_sel);
  out.lp_sel = sel.lp_sel;
  Span& bk = forward ? s_old : s_new;
  Span& fw = forward ? s_new : s_old;
  out.th_bk = std::move(bk.th_bk);
  out.rho_bk = std::move(bk.rho_bk);
  out.g_bk = std::move(bk.g_bk);
  out.lj_bk = bk.lj_bk;
  out.th_fw = std::move(fw.th_fw);
  out.rho_fw = std::move(fw.rho_fw);
  out.g_fw = std::move(fw.g_fw);
  out.lj_fw = fw.lj_fw;
  out.logsum = total;
  out.w = total_w;
  out.epoch = c.rebase_factors.size();
  return out;
}

Span single_state(const Vec& th, const Vec& rho, const Vec& g, double lp_pos, double lj) {
  // walnuts.hpp:47-63
  Span s;
  s.th_bk = th; s.rho_bk = rho; s.g_bk = g; s.lj_bk = lj;
  s.th_fw = th; s.rho_fw = rho; s.g_fw = g; s.lj_fw = lj;
  s.th_sel = th; s.g_sel = g; s.lp_sel = lp_pos; s.logsum = lj;
  return s;
}

// walnuts.hpp:420-442
std::optional<Span> build_leaf(Ctx& c, bool forward, const Span& span) {
  Vec th(c.D), rho(c.D), g(c.D);
  double lp_pos = -kInf, lj = -kInf;
  const Vec& th0 = forward ? span.th_fw : span.th_bk;
  const Vec& rho0 = forward ? span.rho_fw : span.rho_bk;
  const Vec& g0 = forward ? span.g_fw : span.g_bk;
  double logp = forward ? span.lj_fw : span.lj_bk;
  if (!macro_step(c, forward, th0, rho0, g0, logp, th, rho, g, lp_pos, lj)) return std::nullopt;
  Span leaf = single_state(th, rho, g, lp_pos, lj);
  if (c.mo.mode == WNO_MATH_PORTABLE) {
    leaf.w = c.leaf_weight(lj);
    leaf.epoch = c.rebase_factors.size();
  }
  return leaf;
}

// walnuts.hpp:464-495
std::optional<Span> build_span(Ctx& c, bool forward, size_t depth, const Span& last) {
  if (depth == 0) return build_leaf(c, forward, last);
  auto s1 = build_span(c, forward, depth - 1, last);
  if (!s1) return std::nullopt;
  auto s2 = build_span(c, forward, depth - 1, *s1);
  if (!s2) return std::nullopt;
  if (uturn(c, forward, *s1, *s2)) return std::nullopt;
  return combine(c, /*metropolis=*/false, forward, std::move(*s1), std::move(*s2));
}

// walnuts.hpp:520-563
void transition(Ctx& c, const double* chol, size_t max_depth, Vec& theta, size_t& depth, Vec& grad_sel,
                double& logp_sel) {
  const size_t D = c.D;
  Vec z(D), rho(D), g(D);
  c.rng->normals(D, z.data());
  for (size_t i = 0; i < D; ++i) rho[i] = chol[i] * z[i];
  double lp_pos;
  (*c.model)(theta.data(), lp_pos, g.data(), c.fma);
  ++c.grad_evals;
  double lj = lp_pos + logp_momentum(c.red, D, rho.data(), c.im, c.fma);
  Span acc = single_state(theta, rho, g, lp_pos, lj);
  c.w_ref = lj;  // device arithmetic: the initial point weighs exactly 1
  c.rebase_factors.clear();
  acc.w = 1.0;
  acc.epoch = 0;
  for (depth = 1; depth <= max_depth; ++depth) {
    bool forward = c.rng->bernoulli();
    auto next = build_span(c, forward, depth - 1, acc);
    if (!next) break;
    bool u = uturn(c, forward, acc, *next);
    acc = combine(c, /*metropolis=*/true, forward, std::move(acc), std::move(*next));
    if (u) break;
  }
  grad_sel = acc.g_sel;
  logp_sel = acc.lp_sel;
  theta = std::move(acc.th_sel);
}

// util.hpp:242-259
double leapfrog_error(const Model& model, const Reducer& red, size_t D, const double* theta, const double* rho,
                      const double* inv_m, double step, int64_t* evals) {
  Vec g(D), rs(D), ts(D);
  double logp;
  model(theta, logp, g.data());
  logp += logp_momentum(red, D, rho, inv_m);
  for (size_t i = 0; i < D; ++i) rs[i] = rho[i] + 0.5 * step * g[i];
  for (size_t i = 0; i < D; ++i) ts[i] = theta[i] + step * (inv_m[i] * rs[i]);
  double logp_star;
  model(ts.data(), logp_star, g.data());
  for (size_t i = 0; i < D; ++i) rs[i] = rs[i] + 0.5 * step * g[i];
  logp_star += logp_momentum(red, D, rs.data(), inv_m);
  if (evals) *evals += 2;
  return logp_star - logp;
}

// util.hpp:285-303
double adapt_step(RandomSource& rand, const Model& model, const Reducer& red, const MathOps& mo, size_t D,
                  const double* theta, const double* mass, double step, int64_t* evals) {
  Vec inv_m(D), rho(D), z(D);
  for (size_t i = 0; i < D; ++i) inv_m[i] = 1.0 / mass[i];
  rand.normals(D, z.data());
  for (size_t i = 0; i < D; ++i) rho[i] = z[i] * std::sqrt(mass[i]);
  const double log09 = mo.log(0.9), log06 = mo.log(0.6), rt = std::sqrt(0.5);
  while (leapfrog_error(model, red, D, theta, rho.data(), inv_m.data(), step, evals) > log09) step *= 2;
  while (leapfrog_error(model, red, D, theta, rho.data(), inv_m.data(), step, evals) < log06) step *= rt;
  return step;
}

// ---------------------------------------------------------------------------
// one chain = AdaptiveWalnuts (adaptive_walnuts.hpp:182-363) that can be frozen
// into a WalnutsSampler (walnuts.hpp:605-766)
// ---------------------------------------------------------------------------
struct Chain {
  Vec theta, mass, grad_sel;
  double step_init = 0.1;  // config.hpp:201
  Adam adam;
  MassEstimator est;
  MinMicro mm;
  size_t iteration = 0;
  bool frozen = false;
  Vec inv_mass, chol;  // frozen sampler parameters
  double step = 0;
  size_t min_micro = 1;
  double logp = 0;
  size_t depth = 0;
  int64_t grad_evals = 0;
  int64_t last_scalar_draws = 0;
  uint32_t transitions = 0;
  std::unique_ptr<RandomSource> rng;
  std::mt19937_64 eng64;
  std::mt19937 eng32;
  std::vector<TraceRec> trace;
  int64_t ties[3] = {0, 0, 0}, decisions[3] = {0, 0, 0};
  int64_t weight_rebases = 0;  // device arithmetic: how often a transition moved its weights' reference energy
  // WelfordAccumulator of the sampling log densities (online_moments.hpp:22-86, sampler.hpp:87-88)
  double lp_n = 0, lp_mean = 0, lp_m2 = 0;
  void observe_lp(double x) {
    lp_n += 1;
    const double delta = x - lp_mean;
    lp_mean += delta / lp_n;
    lp_m2 += delta * (x - lp_mean);
  }
  double lp_sample_variance() const {
    return lp_n > 1 ? lp_m2 / (lp_n - 1) : std::numeric_limits<double>::quiet_NaN();
  }
};

}  // namespace

struct wno_engine {
  Model model;
  wno_config cfg;
  MathOps mo;
  Reducer red;
  size_t C = 0, D = 0;
  std::vector<Chain> chains;
  bool adapt_ready = false;
  bool trace_on = false;
  double tie_tol = 0.0;
  int64_t iteration = 0;
  Vec var_z, var_u;  // pending host-supplied variates (one transition)
  size_t var_nu = 0;

  // the RandomSource a chain uses for the next transition
  struct RngLease {
    RandomSource* r;
    std::unique_ptr<RandomSource> owned;
  };
  RngLease lease_rng(Chain& ch) {
    if (var_nu == 0) return {ch.rng.get(), nullptr};
    const size_t c = static_cast<size_t>(&ch - chains.data());
    auto b = std::make_unique<BufferRandom>(var_z.data() + c * D, var_u.data() + c * var_nu, var_nu);
    RandomSource* raw = b.get();
    return {raw, std::move(b)};
  }

  void ensure_adapters() {
    if (adapt_ready) return;
    for (auto& ch : chains) {
      // adaptive_walnuts.hpp:205-223
      ch.adam.init(ch.step_init, cfg, mo);
      ch.est.init(cfg.mass_init_count, ch.mass.data(), D, cfg.math_mode == WNO_MATH_PORTABLE);
      ch.mm.target = cfg.max_macro_steps_target;
      ch.mm.floor_ = static_cast<size_t>(cfg.min_micro_steps);
      ch.mm.total = 2.0;
      ch.mm.count = 1.0;
      ch.iteration = 0;
    }
    adapt_ready = true;
  }

  Ctx make_ctx(Chain& ch, RandomSource* rng, const double* im, double step, size_t min_micro, Adam* adam) {
    Ctx c;
    c.model = &model;
    c.im = im;
    c.D = D;
    c.step = step;
    c.max_halvings = static_cast<size_t>(cfg.max_step_halvings);
    c.min_micro = min_micro;
    c.max_error = cfg.max_hamiltonian_error;
    c.mo = mo;
    c.red = red;
    c.fma = cfg.fma != 0;
    c.rng = rng;
    c.adam = adam;
    c.trace = trace_on ? &ch.trace : nullptr;
    if (trace_on) ch.trace.clear();
    c.tie_tol = tie_tol;
    return c;
  }
  static void collect_audit(Chain& ch, const Ctx& c) {
    for (int k = 0; k < 3; ++k) {
      ch.ties[k] += c.ties[k];
      ch.decisions[k] += c.decisions[k];
    }
    ch.weight_rebases += static_cast<int64_t>(c.rebase_factors.size());
  }

  void warmup_chain(Chain& ch) {  // adaptive_walnuts.hpp:234-251
    Vec im(D), chol(D);
    ch.est.inv_mass(im.data());
    for (size_t i = 0; i < D; ++i) chol[i] = std::sqrt(1.0 / im[i]);
    RngLease lease = lease_rng(ch);
    lease.r->begin_transition(ch.transitions);
    Ctx c = make_ctx(ch, lease.r, im.data(), ch.adam.step_size(), ch.mm.value(), &ch.adam);
    transition(c, chol.data(), static_cast<size_t>(cfg.max_trajectory_doublings), ch.theta, ch.depth, ch.grad_sel,
               ch.logp);
    ch.grad_evals += c.grad_evals;
    collect_audit(ch, c);
    ch.last_scalar_draws = lease.r->scalar_draws;
    ch.est.observe(ch.theta.data(), ch.grad_sel.data(), ch.iteration);
    ch.mm.observe(static_cast<size_t>(1) << ch.depth);
    ++ch.iteration;
    ++ch.transitions;
  }

  void freeze_chain(Chain& ch) {  // adaptive_walnuts.hpp:263-271, walnuts.hpp:637-660
    ch.inv_mass.resize(D);
    ch.chol.resize(D);
    ch.est.inv_mass(ch.inv_mass.data());
    for (size_t i = 0; i < D; ++i) ch.chol[i] = 1.0 / std::sqrt(ch.inv_mass[i]);
    ch.step = ch.adam.step_size();
    ch.min_micro = ch.mm.value();
    ch.rng->reset_distributions();
    ch.frozen = true;
  }

  void sample_chain(Chain& ch) {  // walnuts.hpp:682-692
    RngLease lease = lease_rng(ch);
    lease.r->begin_transition(ch.transitions);
    Ctx c = make_ctx(ch, lease.r, ch.inv_mass.data(), ch.step, ch.min_micro, nullptr);
    transition(c, ch.chol.data(), static_cast<size_t>(cfg.max_trajectory_doublings), ch.theta, ch.depth,
               ch.grad_sel, ch.logp);
    ch.grad_evals += c.grad_evals;
    collect_audit(ch, c);
    ch.last_scalar_draws = lease.r->scalar_draws;
    ch.observe_lp(ch.logp);
    ++ch.transitions;
  }

  // Worker threads live as long as the engine (one spawn per run, not per transition): a step hands them one
  // job and waits; chains are dealt out in blocks of a few through an atomic cursor, so a slow chain does not
  // hold up a whole pre-assigned range.
  struct Pool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(size_t)> job;
    std::atomic<size_t> cursor{0};
    size_t total = 0, block = 1, generation = 0, running = 0;
    bool stop = false;
    void work() {
      for (;;) {
        const size_t lo = cursor.fetch_add(block);
        if (lo >= total) break;
        const size_t hi = std::min(total, lo + block);
        for (size_t i = lo; i < hi; ++i) job(i);
      }
    }
    void loop() {
      size_t seen = 0;
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_go.wait(lk, [&] { return stop || generation != seen; });
          if (stop) return;
          seen = generation;
        }
        work();
        {
          std::lock_guard<std::mutex> lk(mu);
          if (--running == 0) cv_done.notify_all();
        }
      }
    }
    void ensure(size_t n) {
      if (workers.size() == n) return;
      shutdown();
      stop = false;
      for (size_t t = 0; t < n; ++t) workers.emplace_back([this] { loop(); });
    }
    void run(size_t n_items, size_t blk, std::function<void(size_t)> f) {
      {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f);
        total = n_items;
        block = blk;
        cursor = 0;
        running = workers.size();
        ++generation;
      }
      cv_go.notify_all();
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return running == 0; });
    }
    void shutdown() {
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv_go.notify_all();
      for (auto& w : workers) w.join();
      workers.clear();
    }
    ~Pool() { shutdown(); }
  };
  std::unique_ptr<Pool> pool;

  template <class F>
  void for_chains(int num_threads, F f) {
    if (num_threads <= 1 || C < 2) {
      for (auto& ch : chains) f(ch);
      return;
    }
    const size_t nt = std::min<size_t>(static_cast<size_t>(num_threads), C);
    if (!pool) pool = std::make_unique<Pool>();
    pool->ensure(nt);
    const size_t blk = std::max<size_t>(1, C / (nt * 4));
    pool->run(C, blk, [&](size_t i) { f(chains[i]); });
  }
};

namespace {
template <class Eng>
std::unique_ptr<RandomSource> make_std(Eng* e) {
  return std::make_unique<StdRandom<Eng>>(e);
}
}  // namespace

// Sums over chains in the two controllers below.  Reference order: left to right (adapt.hpp:200-206,
// sampler.hpp:139-142).  Device order (reduce_lanes > 0): runs of 256 consecutive chains left to right, then the run
// totals left to right -- what the engine's monitor kernels do (wn_elementwise.h: run_partial_sums), the same as left
// to right up to 256 chains.
template <class F>
static double chain_sum(const wno_engine* e, size_t M, F f) {
  if (e->red.L <= 0) {
    double s = 0.0;
    for (size_t m = 0; m < M; ++m) s += f(m);
    return s;
  }
  double total = 0.0;
  for (size_t lo = 0; lo < M; lo += 256) {
    double run = 0.0;
    for (size_t m = lo; m < std::min(M, lo + 256); ++m) run += f(m);
    total += run;
  }
  return total;
}

// util.hpp:380-383 l2_rel_diff: norm((a - b) / b).  block_order: the device's monitor block (256 strided partials, then
// a pairwise tree) instead of a left-to-right sum.
static double l2_rel_diff(const double* a, const double* b, size_t n, bool block_order) {
  auto rel_sq = [&](size_t d) {
    const double r = (a[d] - b[d]) / b[d];
    return r * r;
  };
  double ss = 0;
  if (!block_order) {
    for (size_t d = 0; d < n; ++d) ss += rel_sq(d);
  } else {
    double sh[256];
    for (size_t t = 0; t < 256; ++t) {
      double acc = 0.0;
      for (size_t d = t; d < n; d += 256) acc += rel_sq(d);
      sh[t] = acc;
    }
    for (size_t s = 128; s > 0; s >>= 1)
      for (size_t t = 0; t < s; ++t) sh[t] += sh[t + s];
    ss = sh[0];
  }
  return std::sqrt(ss);
}

// util.hpp:401-404 variance: sum((xs - mean(xs))^2) / (n - 1); `sum` is the order the caller's mode prescribes
template <class Sum, class At>
static double variance(size_t n, Sum sum, At at) {
  const double mean = sum([&](size_t m) { return at(m); }) / static_cast<double>(n);
  const double ss = sum([&](size_t m) { return (at(m) - mean) * (at(m) - mean); });
  return ss / static_cast<double>(n - 1);
}

extern "C" {

// the device's shared-divisor quotient by itself (tests/test_portable_math.py compares it with `/`)
double wno_div_shared(double a, double w) { return div_shared(a, w, 1.0 / w); }

// the two controller helpers by themselves, in the reference's order (tests/util_test.cpp:316-381 hold their vectors)
double wno_l2_rel_diff(size_t n, const double* a, const double* b) { return l2_rel_diff(a, b, n, false); }
double wno_variance(size_t n, const double* xs) {
  auto seq = [&](auto f) {
    double s = 0.0;
    for (size_t m = 0; m < n; ++m) s += f(m);
    return s;
  };
  return variance(n, seq, [&](size_t m) { return xs[m]; });
}

void wno_default_config(wno_config* c) {
  c->max_trajectory_doublings = 5;
  c->max_step_halvings = 5;
  c->min_micro_steps = 1;
  c->max_hamiltonian_error = 0.5;
  c->mass_init_count = 4.0;
  c->max_macro_steps_target = 15.0;
  c->step_accept_rate_target = 0.8;
  c->step_learning_rate = 0.05;
  c->step_gradient_decay = 0.8;
  c->step_sq_gradient_decay = 0.9;
  c->step_stabilization = 1e-4;
  c->step_learn_rate_decay = 0.5;
  c->math_mode = WNO_MATH_LIBM;
  c->reduce_lanes = 0;
  c->rng_mode = WNO_RNG_STD_MT64;
  c->fma = 0;
}

wno_engine* wno_create(int model, int dim, const double* params, size_t num_chains, const wno_config* cfg) {
  auto* e = new wno_engine();
  e->cfg = *cfg;
  e->mo.mode = cfg->math_mode;
  e->red.L = cfg->reduce_lanes;
  e->C = num_chains;
  e->D = static_cast<size_t>(dim);
  e->model.kind = model;
  e->model.D = e->D;
  e->model.m = e->mo;
  e->model.r = e->red;
  if (model == WNO_MODEL_DIAG_NORMAL && params) e->model.params.assign(params, params + dim);
  e->chains.resize(num_chains);
  for (auto& ch : e->chains) {
    ch.theta.assign(e->D, 0.0);  // config.hpp:202-204
    ch.mass.assign(e->D, 1.0);   // config.hpp:205-207
    ch.grad_sel.assign(e->D, 0.0);
  }
  return e;
}

void wno_destroy(wno_engine* e) { delete e; }
void wno_set_rng_mode(wno_engine* e, int mode) { e->cfg.rng_mode = mode; }

void wno_set_positions(wno_engine* e, const double* pos) {
  for (size_t c = 0; c < e->C; ++c) e->chains[c].theta.assign(pos + c * e->D, pos + (c + 1) * e->D);
}
void wno_set_masses(wno_engine* e, const double* mass) {
  for (size_t c = 0; c < e->C; ++c) e->chains[c].mass.assign(mass + c * e->D, mass + (c + 1) * e->D);
  e->adapt_ready = false;
}
void wno_set_step_sizes(wno_engine* e, const double* steps) {
  for (size_t c = 0; c < e->C; ++c) e->chains[c].step_init = steps[c];
  e->adapt_ready = false;
}

void wno_init_positions(wno_engine* e, uint64_t s0, uint64_t s1, double scale) {
  const size_t D = e->D;
  if (e->cfg.rng_mode == WNO_RNG_PHILOX) {
    for (size_t c = 0; c < e->C; ++c) {
      PhiloxRandom r(s0, static_cast<uint32_t>(s1 + c));
      r.normal_stream = WNO_STREAM_INIT_POS;
      r.normals(D, e->chains[c].theta.data());
      for (auto& x : e->chains[c].theta) x *= scale;
    }
    return;
  }
  // config.hpp:259-268: one Random over one engine for all chains
  std::seed_seq ss{s0, s1};
  auto fill = [&](RandomSource& r) {
    for (size_t c = 0; c < e->C; ++c) {
      r.normals(D, e->chains[c].theta.data());
      for (auto& x : e->chains[c].theta) x *= scale;
    }
  };
  if (e->cfg.rng_mode == WNO_RNG_STD_MT32) {
    std::mt19937 g(ss);
    StdRandom<std::mt19937> r(&g);
    fill(r);
  } else {
    std::mt19937_64 g(ss);
    StdRandom<std::mt19937_64> r(&g);
    fill(r);
  }
}

void wno_init_masses_from_grad(wno_engine* e, double s, int average) {
  const size_t D = e->D;
  Vec g(D);
  for (auto& ch : e->chains) {  // config.hpp:366-370
    double lp;
    e->model(ch.theta.data(), lp, g.data());
    ++ch.grad_evals;
    for (size_t i = 0; i < D; ++i) ch.mass[i] = (1 - s) * std::fabs(g[i]) + s;
  }
  if (average) {  // config.hpp:371-380
    Vec sum(D, 0.0);
    for (auto& ch : e->chains)
      for (size_t i = 0; i < D; ++i) sum[i] += e->mo.log(ch.mass[i]);
    for (size_t i = 0; i < D; ++i) sum[i] = e->mo.exp(sum[i] / static_cast<double>(e->C));
    for (auto& ch : e->chains) ch.mass = sum;
  }
  e->adapt_ready = false;
}

void wno_adapt_step(wno_engine* e, uint64_t s0, uint64_t s1) {
  const size_t D = e->D;
  if (e->cfg.rng_mode == WNO_RNG_PHILOX) {
    for (size_t c = 0; c < e->C; ++c) {
      auto& ch = e->chains[c];
      PhiloxRandom r(s0, static_cast<uint32_t>(s1 + c));
      r.normal_stream = WNO_STREAM_INIT_STEP;
      ch.step_init = adapt_step(r, e->model, e->red, e->mo, D, ch.theta.data(), ch.mass.data(), ch.step_init,
                                &ch.grad_evals);
    }
    e->adapt_ready = false;
    return;
  }
  // config.hpp:470-476: one engine, a fresh detail::Random per chain (util.hpp:288)
  std::seed_seq ss{s0, s1};
  if (e->cfg.rng_mode == WNO_RNG_STD_MT32) {
    std::mt19937 g(ss);
    for (auto& ch : e->chains) {
      StdRandom<std::mt19937> r(&g);
      ch.step_init = adapt_step(r, e->model, e->red, e->mo, D, ch.theta.data(), ch.mass.data(), ch.step_init,
                                &ch.grad_evals);
    }
  } else {
    std::mt19937_64 g(ss);
    for (auto& ch : e->chains) {
      StdRandom<std::mt19937_64> r(&g);
      ch.step_init = adapt_step(r, e->model, e->red, e->mo, D, ch.theta.data(), ch.mass.data(), ch.step_init,
                                &ch.grad_evals);
    }
  }
  e->adapt_ready = false;
}

void wno_seed_chains(wno_engine* e, uint64_t seed, uint32_t chain_offset) {
  for (size_t m = 0; m < e->C; ++m) {
    auto& ch = e->chains[m];
    if (e->cfg.rng_mode == WNO_RNG_PHILOX) {
      ch.rng = std::make_unique<PhiloxRandom>(seed, chain_offset + static_cast<uint32_t>(m));
    } else {
      std::seed_seq ss{static_cast<size_t>(seed), m + 1u};  // api.hpp:48-49
      if (e->cfg.rng_mode == WNO_RNG_STD_MT32) {
        ch.eng32 = std::mt19937(ss);
        ch.rng = make_std(&ch.eng32);
      } else {
        ch.eng64 = std::mt19937_64(ss);
        ch.rng = make_std(&ch.eng64);
      }
    }
    ch.transitions = 0;
  }
}

void wno_set_variates(wno_engine* e, const double* normals, const double* uniforms, size_t u_per_chain) {
  e->var_z.assign(normals, normals + e->C * e->D);
  e->var_u.assign(uniforms, uniforms + e->C * u_per_chain);
  e->var_nu = u_per_chain;
}

void wno_warmup_step(wno_engine* e, int num_threads) {
  e->ensure_adapters();
  e->for_chains(num_threads, [&](Chain& ch) { e->warmup_chain(ch); });
  e->var_nu = 0;
  ++e->iteration;
}

void wno_freeze(wno_engine* e) {
  e->ensure_adapters();
  for (auto& ch : e->chains) e->freeze_chain(ch);
}

void wno_sample_step(wno_engine* e, int num_threads) {
  e->for_chains(num_threads, [&](Chain& ch) { e->sample_chain(ch); });
  e->var_nu = 0;
  ++e->iteration;
}

/* frozen sampler parameters handed in as they are (a WalnutsSampler constructed from them, walnuts.hpp:637-660):
 * inverse mass [C*D], step size [C], min micro steps [C] */
void wno_set_sampler_state(wno_engine* e, const double* inv_mass, const double* step, const int64_t* min_micro) {
  for (size_t c = 0; c < e->C; ++c) {
    Chain& ch = e->chains[c];
    ch.inv_mass.assign(inv_mass + c * e->D, inv_mass + (c + 1) * e->D);
    ch.chol.resize(e->D);
    for (size_t i = 0; i < e->D; ++i) ch.chol[i] = 1.0 / std::sqrt(ch.inv_mass[i]);
    ch.step = step[c];
    ch.min_micro = static_cast<size_t>(min_micro[c]);
    if (ch.rng) ch.rng->reset_distributions();
    ch.frozen = true;
  }
}
/* adaptation state handed in as it is (an AdaptiveWalnuts mid-warmup, adaptive_walnuts.hpp:205-251): Adam's six
 * numbers per chain, the mass estimator's four planes and two weights, the min-micro-steps value in force, and the
 * number of warmup transitions done (the estimator's discount depends on it, :74-80).  For gates that replay ONE warmup
 * transition from another engine's state. */
void wno_set_adapt_state(wno_engine* e, const double* adam, const double* dm, const double* ds, const double* sm,
                         const double* ss, const double* w, const int64_t* min_micro, uint64_t iteration) {
  e->ensure_adapters();
  const size_t D = e->D;
  for (size_t c = 0; c < e->C; ++c) {
    Chain& ch = e->chains[c];
    const double* a = adam + 6 * c;
    ch.adam.theta = a[0]; ch.adam.m = a[1]; ch.adam.v = a[2]; ch.adam.t = a[3]; ch.adam.b1pow = a[4]; ch.adam.b2pow = a[5];
    ch.est.draw_var.mean.assign(dm + c * D, dm + (c + 1) * D);
    ch.est.draw_var.ssd.assign(ds + c * D, ds + (c + 1) * D);
    ch.est.score_var.mean.assign(sm + c * D, sm + (c + 1) * D);
    ch.est.score_var.ssd.assign(ss + c * D, ss + (c + 1) * D);
    ch.est.draw_var.weight = w[2 * c];
    ch.est.score_var.weight = w[2 * c + 1];
    // MinMicroStepsAdaptHandler: total / count chosen so that value() is exactly the handed-in one
    ch.mm.count = 1.0;
    ch.mm.total = static_cast<double>(min_micro[c]) * ch.mm.target;
    ch.iteration = static_cast<size_t>(iteration);
    ch.frozen = false;
  }
}
/* the counter-based streams are keyed by the transition index: make the next transition number `t` */
void wno_set_transition_index(wno_engine* e, uint32_t t) {
  for (auto& ch : e->chains) ch.transitions = t;
}

/* near-tie audit: tolerance (0 = off), and totals over all chains since the last reset:
 * out[0..2] = decisions within the tolerance of their threshold (energy error, U-turn sign, acceptance draw),
 * out[3..5] = decisions taken */
void wno_set_tie_tolerance(wno_engine* e, double tol) { e->tie_tol = tol; }
void wno_get_near_ties(wno_engine* e, int64_t* out, int reset) {
  for (int k = 0; k < 6; ++k) out[k] = 0;
  for (auto& ch : e->chains) {
    for (int k = 0; k < 3; ++k) {
      out[k] += ch.ties[k];
      out[3 + k] += ch.decisions[k];
      if (reset) ch.ties[k] = ch.decisions[k] = 0;
    }
  }
}

/* device arithmetic: moves of the span weights' reference energy (Ctx::leaf_weight) over all chains so far */
int64_t wno_get_weight_rebases(const wno_engine* e) {
  int64_t n = 0;
  for (auto& ch : e->chains) n += ch.weight_rebases;
  return n;
}

void wno_get_positions(const wno_engine* e, double* out) {
  for (size_t c = 0; c < e->C; ++c) std::copy(e->chains[c].theta.begin(), e->chains[c].theta.end(), out + c * e->D);
}
void wno_get_grad_select(const wno_engine* e, double* out) {
  for (size_t c = 0; c < e->C; ++c)
    std::copy(e->chains[c].grad_sel.begin(), e->chains[c].grad_sel.end(), out + c * e->D);
}
void wno_get_logp(const wno_engine* e, double* out) {
  for (size_t c = 0; c < e->C; ++c) out[c] = e->chains[c].logp;
}
void wno_get_step_sizes(const wno_engine* e, double* out) {
  for (size_t c = 0; c < e->C; ++c) {
    const auto& ch = e->chains[c];
    out[c] = ch.frozen ? ch.step : (e->adapt_ready ? ch.adam.step_size() : ch.step_init);
  }
}
void wno_get_masses(const wno_engine* e, double* out) {  // InitConfig::mass(m), config.hpp:74-120
  for (size_t c = 0; c < e->C; ++c) std::copy(e->chains[c].mass.begin(), e->chains[c].mass.end(), out + c * e->D);
}
// WalnutsSampler::inverse masses after freeze; before, AdaptiveWalnuts::inv_mass() = the estimator's current
// estimate (adaptive_walnuts.hpp:89-94,297-299), which for a fresh adapter is sqrt((1/m)/m), not 1/m bit for bit
void wno_get_inv_mass(const wno_engine* ce, double* out) {
  wno_engine* e = const_cast<wno_engine*>(ce);
  if (!e->chains.empty() && !e->chains[0].frozen) e->ensure_adapters();
  for (size_t c = 0; c < e->C; ++c) {
    const auto& ch = e->chains[c];
    if (ch.frozen) {
      std::copy(ch.inv_mass.begin(), ch.inv_mass.end(), out + c * e->D);
    } else {
      ch.est.inv_mass(out + c * e->D);
    }
  }
}
void wno_get_min_micro(const wno_engine* e, int64_t* out) {
  for (size_t c = 0; c < e->C; ++c) {
    const auto& ch = e->chains[c];
    out[c] = static_cast<int64_t>(ch.frozen ? ch.min_micro : ch.mm.value());
  }
}
void wno_get_depths(const wno_engine* e, int32_t* out) {
  for (size_t c = 0; c < e->C; ++c) out[c] = static_cast<int32_t>(e->chains[c].depth);
}
void wno_get_grad_evals(const wno_engine* e, int64_t* out) {
  for (size_t c = 0; c < e->C; ++c) out[c] = e->chains[c].grad_evals;
}
void wno_get_rng_draws(const wno_engine* e, int64_t* out) {
  for (size_t c = 0; c < e->C; ++c) out[c] = e->chains[c].last_scalar_draws;
}
void wno_get_estimator(const wno_engine* e, double* dm, double* ds, double* sm, double* ss, double* w) {
  const size_t D = e->D;
  for (size_t c = 0; c < e->C; ++c) {
    const auto& est = e->chains[c].est;
    std::copy(est.draw_var.mean.begin(), est.draw_var.mean.end(), dm + c * D);
    std::copy(est.draw_var.ssd.begin(), est.draw_var.ssd.end(), ds + c * D);
    std::copy(est.score_var.mean.begin(), est.score_var.mean.end(), sm + c * D);
    std::copy(est.score_var.ssd.begin(), est.score_var.ssd.end(), ss + c * D);
    w[2 * c] = est.draw_var.weight;
    w[2 * c + 1] = est.score_var.weight;
  }
}
void wno_get_adam(const wno_engine* e, double* out) {
  for (size_t c = 0; c < e->C; ++c) {
    const auto& a = e->chains[c].adam;
    double* o = out + 6 * c;
    o[0] = a.theta; o[1] = a.m; o[2] = a.v; o[3] = a.t; o[4] = a.b1pow; o[5] = a.b2pow;
  }
}
int64_t wno_iteration(const wno_engine* e) { return e->iteration; }

// sampler.hpp:132-145 with util.hpp:401-404 (variance) on the per-chain Welford statistics
double wno_rhat(const wno_engine* e) {
  const size_t M = e->C;
  double mean_of_vars = chain_sum(e, M, [&](size_t m) { return e->chains[m].lp_sample_variance(); });
  mean_of_vars /= static_cast<double>(M);
  const double variance_of_means = variance(M, [&](auto f) { return chain_sum(e, M, f); },
                                            [&](size_t m) { return e->chains[m].lp_mean; });
  return std::sqrt(1 + variance_of_means / mean_of_vars);
}

// adapt.hpp:193-221: max over chains of l2_rel_diff(mass_m, geom_mean_mass) and of the step's relative distance.
// exp / log are the mode's (libm, or the device's portable pair); in device order the squared relative differences of
// one chain are summed as the engine's block does it: 256 strided partials, then a pairwise tree.
void wno_warmup_spread(wno_engine* e, double* max_rel_step, double* max_rel_mass) {
  e->ensure_adapters();
  const MathOps& mo = e->mo;
  const size_t M = e->C, D = e->D;
  Vec mean_log_mass(D, 0.0), im(D);
  std::vector<Vec> mass(M, Vec(D));
  std::vector<double> log_step(M);
  for (size_t m = 0; m < M; ++m) {
    auto& ch = e->chains[m];
    log_step[m] = mo.log(ch.adam.step_size());  // log_step_size(), adaptive_walnuts.hpp:311
    ch.est.inv_mass(im.data());
    for (size_t d = 0; d < D; ++d) {
      const double lm = -mo.log(im[d]);  // log_mass(), adaptive_walnuts.hpp:319-323
      mean_log_mass[d] += lm;
      mass[m][d] = mo.exp(lm);           // adapt.hpp:141
    }
  }
  const double mean_log_step = chain_sum(e, M, [&](size_t m) { return log_step[m]; }) / static_cast<double>(M);
  for (size_t d = 0; d < D; ++d) mean_log_mass[d] = mo.exp(mean_log_mass[d] / static_cast<double>(M));
  const double gms = mo.exp(mean_log_step);
  double rm = 0, rs = 0;
  for (size_t m = 0; m < M; ++m) {
    rm = std::fmax(rm, l2_rel_diff(mass[m].data(), mean_log_mass.data(), D, e->red.L > 0));
    rs = std::fmax(rs, (mo.exp(log_step[m]) - gms) / gms);
  }
  *max_rel_step = rs;
  *max_rel_mass = rm;
}

void wno_enable_trace(wno_engine* e, int on) { e->trace_on = on != 0; }
size_t wno_get_trace(const wno_engine* e, size_t chain, double* out, size_t max_rec) {
  const auto& tr = e->chains[chain].trace;
  size_t n = std::min(max_rec, tr.size());
  for (size_t i = 0; i < n; ++i) std::memcpy(out + i * WNO_TRACE_FIELDS, tr[i].f, sizeof(tr[i].f));
  return tr.size();
}

// ---- function-level entry points -------------------------------------------
double wno_logp_momentum(size_t n, const double* rho, const double* inv_mass, int reduce_lanes) {
  Reducer r;
  r.L = reduce_lanes;
  return logp_momentum(r, n, rho, inv_mass);
}
double wno_reduce_sum(size_t n, const double* x, int reduce_lanes) {
  Reducer r;
  r.L = reduce_lanes;
  return r.sum(n, [&](size_t i) { return x[i]; });
}
double wno_log_sum_exp(double a, double b, int math_mode) {
  MathOps m;
  m.mode = math_mode;
  return log_sum_exp(m, a, b);
}
static Model make_model(int model, int dim, const double* params, int math_mode, int reduce_lanes) {
  Model md;
  md.kind = model;
  md.D = static_cast<size_t>(dim);
  md.m.mode = math_mode;
  md.r.L = reduce_lanes;
  if (model == WNO_MODEL_DIAG_NORMAL && params) md.params.assign(params, params + dim);
  return md;
}
int wno_model_logp_grad(int model, int dim, const double* params, const double* x, double* logp, double* grad,
                        int math_mode, int reduce_lanes) {
  Model md = make_model(model, dim, params, math_mode, reduce_lanes);
  md(x, *logp, grad);
  return 0;
}
double wno_leapfrog_error(int model, int dim, const double* params, const double* theta, const double* rho,
                          const double* inv_m, double step, int math_mode, int reduce_lanes) {
  Model md = make_model(model, dim, params, math_mode, reduce_lanes);
  return leapfrog_error(md, md.r, md.D, theta, rho, inv_m, step, nullptr);
}
int wno_uturn(size_t n, int forward, const double* th_in, const double* rho_in, const double* th_out,
              const double* rho_out, const double* inv_mass, int reduce_lanes) {
  // s1 = the earlier-built span (only its inner end matters), s2 = the later
  // one (only its outer end matters)
  Span s1, s2;
  Vec a(th_in, th_in + n), b(rho_in, rho_in + n), c(th_out, th_out + n), d(rho_out, rho_out + n);
  s1.th_bk = s1.th_fw = a;
  s1.rho_bk = s1.rho_fw = b;
  s2.th_bk = s2.th_fw = c;
  s2.rho_bk = s2.rho_fw = d;
  Ctx cx{};
  cx.im = inv_mass;
  cx.D = n;
  cx.red.L = reduce_lanes;
  return uturn(cx, forward != 0, s1, s2) ? 1 : 0;
}
void wno_adam_run(double step_init, double target, double lr, double b1, double b2, double eps, double decay,
                  const double* alphas, size_t n, double* steps_out, int math_mode) {
  wno_config c;
  wno_default_config(&c);
  c.step_accept_rate_target = target;
  c.step_learning_rate = lr;
  c.step_gradient_decay = b1;
  c.step_sq_gradient_decay = b2;
  c.step_stabilization = eps;
  c.step_learn_rate_decay = decay;
  MathOps m;
  m.mode = math_mode;
  Adam a;
  a.init(step_init, c, m);
  for (size_t i = 0; i < n; ++i) {
    a.observe(alphas[i]);
    steps_out[i] = a.step_size();
  }
}
void wno_online_moments_observe(size_t n, double discount, double* weight, double* mean, double* ssd,
                                const double* y) {
  OnlineMoments om;
  om.weight = *weight;
  om.mean.assign(mean, mean + n);
  om.ssd.assign(ssd, ssd + n);
  om.observe(discount, y);
  *weight = om.weight;
  std::copy(om.mean.begin(), om.mean.end(), mean);
  std::copy(om.ssd.begin(), om.ssd.end(), ssd);
}
int wno_macro_step(int model, int dim, const double* params, const wno_config* cfg, int forward, double step,
                   int min_micro, const double* inv_mass, const double* theta, const double* rho,
                   const double* grad, double logp_joint, double* theta_out, double* rho_out, double* grad_out,
                   double* logp_pos_out, double* logp_joint_out, double* alpha_out, int64_t* grad_evals_out) {
  Model md = make_model(model, dim, params, cfg->math_mode, cfg->reduce_lanes);
  const size_t D = md.D;
  Ctx c{};
  c.model = &md;
  c.im = inv_mass;
  c.D = D;
  c.step = step;
  c.max_halvings = static_cast<size_t>(cfg->max_step_halvings);
  c.min_micro = static_cast<size_t>(min_micro);
  c.max_error = cfg->max_hamiltonian_error;
  c.mo.mode = cfg->math_mode;
  c.red.L = cfg->reduce_lanes;
  Vec th0(theta, theta + D), rho0(rho, rho + D), g0(grad, grad + D), th(D), rh(D), g(D);
  double lp = -kInf, lj = -kInf;
  bool ok = macro_step(c, forward != 0, th0, rho0, g0, logp_joint, th, rh, g, lp, lj);
  std::copy(th.begin(), th.end(), theta_out);
  std::copy(rh.begin(), rh.end(), rho_out);
  std::copy(g.begin(), g.end(), grad_out);
  *logp_pos_out = lp;
  *logp_joint_out = lj;
  if (alpha_out) *alpha_out = c.last_alpha;
  if (grad_evals_out) *grad_evals_out = c.grad_evals;
  return ok ? 1 : 0;
}
double wno_stream_uniform(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t index) {
  return wno_philox_uniform(seed, chain, transition, stream, index);
}
void wno_stream_normals(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, size_t n,
                        double* out) {
  PhiloxRandom r(seed, chain);
  r.transition = transition;
  r.normal_stream = stream;
  r.normals(n, out);
}
double wno_math_exp(double x) { return wno_exp(x); }
double wno_math_log(double x) { return wno_log(x); }
double wno_math_exp_weight(double x) { return wno_exp_weight(x); }

}  // extern "C"
