// wn_devmath.h -- scalar maths and the counter-based random stream of the device engine.
//
// The trajectory kernels are GPU-resident, so the libm calls of the reference
// (std::exp / std::log / std::pow in util.hpp:174-183, walnuts.hpp:336,378,
// adam.hpp:83-93, and libstdc++'s <random> distributions behind
// util.hpp:78-162) have to exist as device code.  These are written with
// IEEE-754 binary64 +,-,*,/,sqrt and integer operations only and are compiled
// with -ffp-contract=off, which makes every result reproducible bit for bit on
// the host: that is what lets tests/ compare whole trajectories exactly.
//
// Polynomial schemes and coefficients: Sun fdlibm 5.3 (exp, log, sin/cos
// kernels).  Generator: Philox4x32-10 (Salmon et al., SC'11).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__) && !defined(WN_CPU_SIM)
#include <hip/hip_runtime.h>
#define WND_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define WND_HD inline
#endif

namespace wnd {

WND_HD uint64_t as_u64(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return static_cast<uint64_t>(__double_as_longlong(d));
#else
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
#endif
}
WND_HD double as_f64(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double(static_cast<long long>(u));
#else
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
#endif
}

WND_HD double two_to(int k) { return as_f64(static_cast<uint64_t>(k + 1023) << 52); }

WND_HD double dexp(double x) {
  constexpr double kLn2Hi = 6.93147180369123816490e-01;
  constexpr double kLn2Lo = 1.90821492927058770002e-10;
  constexpr double kInvLn2 = 1.44269504088896338700e+00;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return __builtin_inf();
  if (x < -7.45133219101941108420e+02) return 0.0;
  const double kf = __builtin_floor(x * kInvLn2 + 0.5);
  const int k = static_cast<int>(kf);
  const double hi = x - kf * kLn2Hi;
  const double lo = kf * kLn2Lo;
  const double r = hi - lo;
  const double t = r * r;
  double p = 4.13813679705723846039e-08;
  p = -1.65339022054652515390e-06 + t * p;
  p = 6.61375632143793436117e-05 + t * p;
  p = -2.77777777770155933842e-03 + t * p;
  p = 1.66666666666666019037e-01 + t * p;
  const double c = r - t * p;
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k == 0) return y;
  const int ka = k / 2;
  const int kb = k - ka;
  return (y * two_to(ka)) * two_to(kb);
}

WND_HD double dlog(double x) {
  constexpr double kLn2Hi = 6.93147180369123816490e-01;
  constexpr double kLn2Lo = 1.90821492927058770002e-10;
  if (x != x) return x;
  if (x < 0.0) return __builtin_nan("");
  if (x == 0.0) return -__builtin_inf();
  if (x == __builtin_inf()) return x;
  int k = 0;
  uint64_t bits = as_u64(x);
  if ((bits >> 52) == 0) {
    x = x * 18014398509481984.0;  // 2^54
    bits = as_u64(x);
    k = -54;
  }
  k += static_cast<int>(bits >> 52) - 1023;
  const uint64_t frac = bits & 0x000fffffffffffffULL;
  if (frac >= 0x6a09e667f3bcdULL) {
    k += 1;
    x = as_f64(frac | (static_cast<uint64_t>(1022) << 52));
  } else {
    x = as_f64(frac | (static_cast<uint64_t>(1023) << 52));
  }
  const double f = x - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 =
      z * (6.666666666666735130e-01 +
           w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = static_cast<double>(k);
  return dk * kLn2Hi - ((hfsq - (s * (hfsq + R) + dk * kLn2Lo)) - f);
}

// x^y for x > 0; the path's only use is Adam's t^decay (adam.hpp:83)
WND_HD double dpow_pos(double x, double y) {
  if (y == 0.0) return 1.0;
  if (y == 1.0) return x;
  if (y == 0.5) return __builtin_sqrt(x);
  return dexp(y * dlog(x));
}

// sin(pi a), cos(pi a), a in [0, 2)
WND_HD void dsincospi(double a, double& sn, double& cs) {
  constexpr double kPi = 3.14159265358979311600e+00;
  const double qf = __builtin_floor(a * 2.0 + 0.5);
  const int q = static_cast<int>(qf);
  const double r = a - qf * 0.5;
  const double x = kPi * r;
  const double z = x * x;
  double ps = 1.58969099521155010221e-10;
  ps = -2.50507602534068634195e-08 + z * ps;
  ps = 2.75573137070700676789e-06 + z * ps;
  ps = -1.98412698298579493134e-04 + z * ps;
  ps = 8.33333333332248946124e-03 + z * ps;
  ps = -1.66666666666666324348e-01 + z * ps;
  const double s = x + x * (z * ps);
  double pc = -1.13596475577881948265e-11;
  pc = 2.08757232129817482790e-09 + z * pc;
  pc = -2.75573143513906633035e-07 + z * pc;
  pc = 2.48015872894767294178e-05 + z * pc;
  pc = -1.38888888888741095749e-03 + z * pc;
  pc = 4.16666666666666019037e-02 + z * pc;
  const double c = (1.0 - 0.5 * z) + (z * z) * pc;
  const int m = q & 3;
  sn = (m == 0) ? s : (m == 1) ? c : (m == 2) ? -s : -c;
  cs = (m == 0) ? c : (m == 1) ? -s : (m == 2) ? -c : s;
}

// ---------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

WND_HD U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t pa = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t pb = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(pb >> 32) ^ c1 ^ k0;
    const uint32_t n2 = static_cast<uint32_t>(pa >> 32) ^ c3 ^ k1;
    c1 = static_cast<uint32_t>(pb);
    c3 = static_cast<uint32_t>(pa);
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

enum : uint32_t { kStreamMomentum = 0, kStreamTree = 1, kStreamInitPos = 2, kStreamInitStep = 3 };

// 64 bits -> open-interval uniform, exact in binary64
WND_HD double open01(uint32_t lo, uint32_t hi) {
  const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
  return (static_cast<double>(v >> 12) + 0.5) * 2.220446049250313080847e-16;
}

// counter = (index, transition, chain, stream); key = seed
WND_HD double stream_uniform(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t index) {
  const U4 o = philox(index, transition, chain, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  return open01(o.x, o.y);
}

// standard normals for vector elements (2*pair, 2*pair+1): Box-Muller
WND_HD void stream_normal_pair(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t pair,
                               double& z0, double& z1) {
  const U4 o = philox(pair, transition, chain, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  const double u1 = open01(o.x, o.y);
  const double u2 = open01(o.z, o.w);
  const double rad = __builtin_sqrt(-2.0 * dlog(u1));
  double sn, cs;
  dsincospi(2.0 * u2, sn, cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

}  // namespace wnd
