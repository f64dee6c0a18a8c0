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
// exp / log: table-driven schemes written for short dependency chains (below).  Their polynomial steps -- and only
// theirs: these are this engine's own functions, not the reference's element-wise arithmetic -- are explicit fused
// multiply-adds (wnd::fmad = one v_fma_f64, or libm's correctly rounded fma() on the host): half the instructions and
// half the depth of the mul-then-add form, identical bits on both sides.
// sin/cos kernels: Sun fdlibm 5.3 polynomial schemes and coefficients.
// Generator: Philox4x32-7 (Salmon et al., SC'11).
#pragma once

#include <stdint.h>

#include "wn_math_tables.h"

#if defined(__HIPCC__)
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

// a * b + c with one rounding (IEEE fusedMultiplyAdd on the device and on the host alike)
WND_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a / b for MANY numerators and ONE divisor (the mass estimator's weights: online_moments.hpp:184-191 divides every
// element of a plane by the same weight, adaptive_walnuts.hpp:89-94 every sum of squared deviations): with r = RN(1/b)
// from ONE true division, q0 = RN(a r), the exact remainder a - b q0 in one fused multiply-add and q = RN(q0 + rem r)
// is the correctly rounded quotient (Markstein, "Computation of elementary functions on the IBM RISC System/6000",
// 1990; Cornea, Harrison, Tang 2002, thm. 1) -- three instructions where v_div_scale / v_rcp / v_div_fmas / v_div_fixup
// take eleven.  It equals IEEE division for every finite numerator with a quotient in the normal range, unless b's
// significand is all ones (RN(1/b) then sits on a rounding boundary; a weight `discount * w + 1` never has been);
// a non-finite numerator gives NaN where division gives +-inf.  The oracle's device-order mode restates exactly these
// three operations, so parity with it is bit for bit in every case; tests/test_portable_math.py compares with `/`.
struct SharedDivisor {
  double b, r;
  WND_HD explicit SharedDivisor(double d) : b(d), r(1.0 / d) {}
};
WND_HD double operator/(double a, const SharedDivisor& d) {
#if defined(WN_PLAIN_ESTIMATOR_DIVISION)  // (A/B probe builds only: tests/gpu_probes/build_variant.sh)
  return a / d.b;
#endif
  const double q0 = a * d.r;
  return __builtin_fma(__builtin_fma(-q0, d.b, a), d.r, q0);
}

// sqrt(x) for x known to be a NORMAL positive number >= 2^-767 (the Box-Muller radicand -2 log u of an open-interval
// uniform lies in [2^-52, 73]): the refinement the compiler emits for fp64 sqrt on this target -- v_rsq_f64, one
// Goldschmidt step, two residual corrections -- without what it wraps around it for the rest of the domain (a compare, a
// select and a v_ldexp_f64 to scale tiny arguments up, another v_ldexp_f64 to scale back, a class test and two selects
// for 0 / inf / NaN): 10 instructions for 18, the same correctly rounded result (an exact power-of-two scaling does not
// change a rounding).  The host evaluates sqrt(); tests/test_gpu_parity.py compares the two on 2^24 arguments per binade
// sample through wn_internal_sqrt_probe, and every bit-exact trajectory test depends on it.
// `checked`: the same with the special operands patched back in (0, inf, NaN and negative arguments give what sqrt
// gives) for arguments that are normal when they are finite and positive -- the mass estimator's variance ratios.
template <bool checked = false>
WND_HD double sqrt_normal(double x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WN_PLAIN_SQRT)
  const double y = __builtin_amdgcn_rsq(x);
  const double g0 = x * y, h0 = 0.5 * y;
  const double r0 = __builtin_fma(-h0, g0, 0.5);
  const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
  const double g2 = __builtin_fma(__builtin_fma(-g1, g1, x), h1, g1);
  const double g3 = __builtin_fma(__builtin_fma(-g2, g2, x), h1, g2);
  if constexpr (checked) {
    // +-0 -> itself, +inf -> itself (class mask 0x260 = -0 | +0 | +inf); NaN and negative arguments: the sequence
    // already ends in NaN (rsq of a negative number / of NaN is NaN)
    return __builtin_amdgcn_class(x, 0x260) ? x : g3;
  } else {
    return g3;
  }
#else
  return __builtin_sqrt(x);
#endif
}

WND_HD double two_to(int k) { return as_f64(static_cast<uint64_t>(k + 1023) << 52); }

// ---------------------------------------------------------------------------
// exp and log: table-driven, division-free, short dependency chains.
//
// On the device these run on wave-uniform values in the middle of the tree loop (log_sum_exp at every merge,
// util.hpp:174-183; the Barker / Metropolis comparisons; Adam), where one wavefront's dependent fp64 chain is
// exposed latency: measured on MI355X, log_sum_exp built on the fdlibm schemes (two divisions, Horner chains,
// ~75 dependent operations) was 17 % of the headline step time.  Here: 64-entry 2^(j/64) table + degree-5
// polynomial for exp, 49-entry (1/c, log c) table + degree-8 polynomial for log, Estrin-style grouping --
// about 35 dependent operations for log_sum_exp.  Every step is a plain binary64 +,-,* or an integer
// operation, so the host reproduces the bits (the test suite keeps its own copy of both functions).
// Accuracy against libm: within 2 ulp (tests/test_portable_math.py).
//
// `Tab` supplies the table entries: exp2(j), rcp(i), logc(i) (wn_math_tables.h on the host, VGPR lanes on the
// device -- see wn_traj.h LaneTables).
// ---------------------------------------------------------------------------
// Both functions evaluate the main path for every lane (arguments outside the domain are clamped first) and patch
// the special values in at the end: the table look-up of a per-lane call is a cross-lane gather, which every lane
// of the wavefront has to take part in.
template <class Tab>
WND_HD double dexp(double x, const Tab& tab) {
  constexpr double kInvStep = 9.23324826168936567e+01;   // 64 / ln 2
  constexpr double kStepHi = 1.08304246095940471e-02;    // ln 2 / 64, upper 28 bits (k * kStepHi is exact)
  constexpr double kStepLo = 8.66550983900947049e-11;
  constexpr double kOver = 7.09782712893383973096e+02, kUnder = -7.45133219101941108420e+02;
  const double xc = (x != x) ? 0.0 : (x > kOver ? kOver : (x < kUnder ? kUnder : x));
  const double kf = __builtin_floor(fmad(xc, kInvStep, 0.5));
  const int k = static_cast<int>(kf);
  const double r = fmad(-kf, kStepLo, fmad(-kf, kStepHi, xc));   // |r| <= ln2/128 (kf * kStepHi is exact)
  const double t = tab.exp2(k & 63);
  const int e = k >> 6;
  const double r2 = r * r;
  // exp(r) - 1 = r + r^2/2 + r^3/6 + r^4/24 + r^5/120   (r^6/720 < 4e-17)
  const double lo = fmad(r, 1.66666666666666657e-01, 0.5);
  const double hi = fmad(r, 8.33333333333333322e-03, 4.16666666666666644e-02);
  const double p = fmad(r2, fmad(r2, hi, lo), r);
  double y = fmad(t, p, t);
  const int ea = e / 2;
  const int eb = e - ea;
  if (e != 0) y = (y * two_to(ea)) * two_to(eb);
  if (x > kOver) y = __builtin_inf();
  if (x < kUnder) y = 0.0;
  if (x != x) y = x;
  return y;
}

template <class Tab>
WND_HD double dlog(double x, const Tab& tab) {
  constexpr double kLn2Hi = 6.93147180369123816490e-01;
  constexpr double kLn2Lo = 1.90821492927058770002e-10;
  uint64_t bits = as_u64(x);
  const bool tiny = ((bits >> 52) & 0x7ff) == 0;    // subnormal (or zero): scale by 2^54
  if (tiny) bits = as_u64(x * 18014398509481984.0);
  int k = (tiny ? -54 : 0) + (static_cast<int>((bits >> 52) & 0x7ff) - 1023);
  const uint64_t frac = bits & 0x000fffffffffffffULL;
  // s in [0.75, 1.5): the mantissa, halved when it is 1.5 or more
  const bool upper = frac >= 0x0008000000000000ULL;
  k += upper ? 1 : 0;
  const double s = as_f64(frac | (static_cast<uint64_t>(upper ? 1022 : 1023) << 52));
  // nearest multiple of 1/64: c = i/64, 48 <= i <= 96; s - c is exact, |s - c| <= 1/128
  const int i = static_cast<int>(fmad(s, 64.0, 0.5));
  const double c = static_cast<double>(i) * 0.015625;
  const double u = (s - c) * tab.rcp(i - 48);
  const double lc = tab.logc(i - 48);
  // log(1 + u) = u - u^2 (1/2 - u/3 + u^2/4 - u^3/5 + u^4/6 - u^5/7 + u^6/8),  |u| < 0.0105
  const double q = u * u;
  const double a0 = fmad(-u, 3.33333333333333315e-01, 0.5);
  const double a1 = fmad(-u, 2.00000000000000011e-01, 0.25);
  const double a2 = fmad(-u, 1.42857142857142849e-01, 1.66666666666666657e-01);
  const double q2 = q * q;
  const double pl = fmad(q2, fmad(q, 0.125, a2), fmad(q, a1, a0));
  const double l1 = fmad(-q, pl, u);
  const double dk = static_cast<double>(k);
  double y = fmad(dk, kLn2Hi, lc) + fmad(dk, kLn2Lo, l1);
  if (x == __builtin_inf()) y = x;
  if (x == 0.0) y = -__builtin_inf();
  if (x < 0.0) y = __builtin_nan("");
  if (x != x) y = x;
  return y;
}

// dlog() for an argument known to be a positive NORMAL number (the generator's open-interval uniforms): the same main
// path, hence the same bits, without the special-value patches and the subnormal rescaling.
template <class Tab>
WND_HD double dlog_normal(double x, const Tab& tab) {
  constexpr double kLn2Hi = 6.93147180369123816490e-01;
  constexpr double kLn2Lo = 1.90821492927058770002e-10;
  const uint64_t bits = as_u64(x);
  int k = static_cast<int>(bits >> 52) - 1023;
  const uint64_t frac = bits & 0x000fffffffffffffULL;
  const bool upper = frac >= 0x0008000000000000ULL;
  k += upper ? 1 : 0;
  const double s = as_f64(frac | (static_cast<uint64_t>(upper ? 1022 : 1023) << 52));
  const int i = static_cast<int>(fmad(s, 64.0, 0.5));
  const double c = static_cast<double>(i) * 0.015625;
  const double u = (s - c) * tab.rcp(i - 48);
  const double lc = tab.logc(i - 48);
  const double q = u * u;
  const double a0 = fmad(-u, 3.33333333333333315e-01, 0.5);
  const double a1 = fmad(-u, 2.00000000000000011e-01, 0.25);
  const double a2 = fmad(-u, 1.42857142857142849e-01, 1.66666666666666657e-01);
  const double q2 = q * q;
  const double pl = fmad(q2, fmad(q, 0.125, a2), fmad(q, a1, a0));
  const double l1 = fmad(-q, pl, u);
  const double dk = static_cast<double>(k);
  return fmad(dk, kLn2Hi, lc) + fmad(dk, kLn2Lo, l1);
}

// exp(x) for the weight of a trajectory span relative to the transition's reference energy (wn_traj.h, "span weights"):
// the caller keeps x <= kWeightRebase (it moves the reference when a state's energy runs ahead of it), and arguments
// below -700 -- or NaN -- stand at -700: such a span weighs e^-700 instead of less, beside spans of weight ~1.  The
// result is therefore a NORMAL number and the final scaling by 2^e is exact: one v_ldexp_f64 on the device, ldexp() on
// the host, the same bits.  The main path of dexp() without its range patches, ~18 operations around two lane reads.
constexpr double kWeightFloor = -700.0;
template <class Tab>
WND_HD double dexp_weight(double x, const Tab& tab) {
  constexpr double kInvStep = 9.23324826168936567e+01;   // 64 / ln 2
  constexpr double kStepHi = 1.08304246095940471e-02;    // ln 2 / 64, upper 28 bits (k * kStepHi is exact)
  constexpr double kStepLo = 8.66550983900947049e-11;
  const double xc = (x > kWeightFloor) ? x : kWeightFloor;  // (NaN compares false)
  const double kf = __builtin_floor(fmad(xc, kInvStep, 0.5));
  const int k = static_cast<int>(kf);
  const double r = fmad(-kf, kStepLo, fmad(-kf, kStepHi, xc));
  const double t = tab.exp2(k & 63);
  const double r2 = r * r;
  const double lo = fmad(r, 1.66666666666666657e-01, 0.5);
  const double hi = fmad(r, 8.33333333333333322e-03, 4.16666666666666644e-02);
  const double p = fmad(r2, fmad(r2, hi, lo), r);
  return __builtin_ldexp(fmad(t, p, t), k >> 6);
}

// the tables as plain arrays (host: tests, engine set-up; device: constant memory for the rarely used call sites)
struct ArrayTables {
  const unsigned long long* e2;
  const unsigned long long* rc;
  const unsigned long long* lc;
  WND_HD double exp2(int j) const { return as_f64(e2[j]); }
  WND_HD double rcp(int i) const { return as_f64(rc[i]); }
  WND_HD double logc(int i) const { return as_f64(lc[i]); }
};

WND_HD ArrayTables array_tables() { return ArrayTables{wn_tab_exp2_bits, wn_tab_rcp_bits, wn_tab_logc_bits}; }
// forms that read the tables from memory: host code and the element-wise set-up kernels
WND_HD double dexp(double x) { return dexp(x, array_tables()); }
WND_HD double dexp_weight(double x) { return dexp_weight(x, array_tables()); }
WND_HD double dlog(double x) { return dlog(x, array_tables()); }

// x^y for x > 0; the path's only use is Adam's t^decay (adam.hpp:83)
template <class Tab>
WND_HD double dpow_pos(double x, double y, const Tab& tab) {
  if (y == 0.0) return 1.0;
  if (y == 1.0) return x;
  if (y == 0.5) return __builtin_sqrt(x);
  return dexp(y * dlog(x, tab), tab);
}

// sin(pi a), cos(pi a), a in [0, 2)
WND_HD void dsincospi(double a, double& sn, double& cs) {
  constexpr double kPi = 3.14159265358979311600e+00;
  const double qf = __builtin_floor(fmad(a, 2.0, 0.5));
  const int q = static_cast<int>(qf);
  const double r = fmad(-qf, 0.5, a);  // exact
  const double x = kPi * r;
  const double z = x * x;
  double ps = 1.58969099521155010221e-10;
  ps = fmad(z, ps, -2.50507602534068634195e-08);
  ps = fmad(z, ps, 2.75573137070700676789e-06);
  ps = fmad(z, ps, -1.98412698298579493134e-04);
  ps = fmad(z, ps, 8.33333333332248946124e-03);
  ps = fmad(z, ps, -1.66666666666666324348e-01);
  const double s = fmad(x, z * ps, x);
  double pc = -1.13596475577881948265e-11;
  pc = fmad(z, pc, 2.08757232129817482790e-09);
  pc = fmad(z, pc, -2.75573143513906633035e-07);
  pc = fmad(z, pc, 2.48015872894767294178e-05);
  pc = fmad(z, pc, -1.38888888888741095749e-03);
  pc = fmad(z, pc, 4.16666666666666019037e-02);
  const double c = fmad(z * z, pc, fmad(-0.5, z, 1.0));
  // quadrant m = q mod 4: (sin, cos) = (s, c), (c, -s), (-s, -c), (-c, s): one swap, then sign flips on the high words
  const uint64_t m = static_cast<uint64_t>(q & 3);
  const bool swap = (m & 1) != 0;
  const double a0 = swap ? c : s, b0 = swap ? s : c;
  sn = as_f64(as_u64(a0) ^ ((m >> 1) << 63));
  cs = as_f64(as_u64(b0) ^ (((m ^ (m >> 1)) & 1) << 63));
}

// ---------------------------------------------------------------------------
// Philox4x32
// ---------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

// a ^ b ^ c: one instruction on gfx950 (v_bitop3_b32 with the parity truth table), which the compiler does not form
// from two xors when one operand is a round key in a scalar register
WND_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// The engine's streams use SEVEN rounds (stream version 2, round 5): the fewest with which Philox4x32 passes the full
// BigCrush battery (Salmon et al., SC'11, table 2; Random123 ships known answers for 7 and for 10 rounds, and
// tests/test_portable_math.py checks both).  The momentum refresh is ~30 % of a headline transition's vector
// instructions and the generator a third of that; v_mad_u64_u32 runs at half rate.  Version 1 (rounds 1-4) used ten.
constexpr int kPhiloxRounds = 7;
constexpr int kStreamVersion = 2;
template <int ROUNDS = kPhiloxRounds>
WND_HD U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const uint64_t pa = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t pb = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = xor3(static_cast<uint32_t>(pb >> 32), c1, k0);
    const uint32_t n2 = xor3(static_cast<uint32_t>(pa >> 32), c3, k1);
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

WND_HD double dpow_pos(double x, double y) { return dpow_pos(x, y, array_tables()); }

// 64 bits -> open-interval uniform, exact in binary64
// (k + 0.5) * 2^-52 for the top 52 bits k, built in the mantissa: 1 + k 2^-52 is exact, so are the subtraction and the
// addition of 2^-53 (the result (2k + 1) 2^-53 is representable) -- the same value as converting k to double, without
// the 64-bit integer conversion.
WND_HD double open01(uint32_t lo, uint32_t hi) {
  const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
  const double one_plus = as_f64(0x3ff0000000000000ULL | (v >> 12));
  return (one_plus - 1.0) + 1.1102230246251565404e-16;
}

// counter = (index, transition, chain, stream); key = seed
WND_HD double stream_uniform(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t index) {
  const U4 o = philox(index, transition, chain, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  return open01(o.x, o.y);
}

// standard normals for vector elements (2*pair, 2*pair+1): Box-Muller
template <class Tab>
WND_HD void stream_normal_pair(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t pair,
                               double& z0, double& z1, const Tab& tab) {
  const U4 o = philox(pair, transition, chain, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  const double u1 = open01(o.x, o.y);
  const double u2 = open01(o.z, o.w);
  const double rad = sqrt_normal(-2.0 * dlog_normal(u1, tab));
  double sn, cs;
  dsincospi(2.0 * u2, sn, cs);
  z0 = rad * cs;
  z1 = rad * sn;
}
WND_HD void stream_normal_pair(uint64_t seed, uint32_t chain, uint32_t transition, uint32_t stream, uint32_t pair,
                               double& z0, double& z1) {
  stream_normal_pair(seed, chain, transition, stream, pair, z0, z1, array_tables());
}

}  // namespace wnd
