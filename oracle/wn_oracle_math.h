/* wn_oracle_math.h -- TEST INFRASTRUCTURE (oracle side), not shipped in the product.
 *
 * The oracle's own restatement of the "portable" scalar maths and the
 * counter-based random stream that the device engine uses.  Everything here
 * is built only from IEEE-754 binary64 +, -, *, /, sqrt and integer bit
 * manipulation, so that a build with -ffp-contract=off gives the same bits on
 * the host as on gfx950.  The product has an independently written copy in
 * walnuts_amd/csrc/wn_devmath.h; tests/test_portable_math.py compiles that
 * header on the host and checks the two bit-for-bit against each other and
 * both against libm (few-ulp).
 *
 * Algorithms (published, restated here):
 *   exp / log / sin,cos kernels: Sun fdlibm 5.3 (e_exp.c, e_log.c, k_sin.c,
 *     k_cos.c) polynomial schemes and coefficients.
 *   Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as
 *     easy as 1, 2, 3", SC'11 (Random123 known-answer vectors are checked in
 *     tests/test_portable_math.py).
 *   Normal variates: Box-Muller on two open-interval uniforms.
 *
 * Where the reference uses libm (`std::exp`, `std::log`, `std::pow`:
 * util.hpp:174-183, walnuts.hpp:336,378, adam.hpp:83-93) the oracle can be
 * run in either maths mode: WNO_MATH_LIBM (what the reference executes on the
 * host) or WNO_MATH_PORTABLE (what the device executes); see wn_oracle.h.
 */
#ifndef WN_ORACLE_MATH_H
#define WN_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t wno_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double wno_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* 2^k for -1022 <= k <= 1023 */
static inline double wno_pow2i(int k) { return wno_u2d((uint64_t)(k + 1023) << 52); }

/* ---- exp ------------------------------------------------------------- */
static inline double wno_exp(double x) {
  const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return INFINITY;
  if (x < -7.45133219101941108420e+02) return 0.0;
  double fk = floor(x * invln2 + 0.5);
  int k = (int)fk;
  double hi = x - fk * ln2hi;
  double lo = fk * ln2lo;
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k == 0) return y;
  int k1 = k / 2, k2 = k - k1;
  return (y * wno_pow2i(k1)) * wno_pow2i(k2);
}

/* ---- log ------------------------------------------------------------- */
static inline double wno_log(double x) {
  const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  if (x != x) return x;
  if (x < 0.0) return NAN;
  if (x == 0.0) return -INFINITY;
  if (x == INFINITY) return x;
  int k = 0;
  uint64_t u = wno_d2u(x);
  if ((u >> 52) == 0) { /* subnormal: scale up by 2^54 */
    x = x * 18014398509481984.0;
    u = wno_d2u(x);
    k = -54;
  }
  k += (int)(u >> 52) - 1023;
  uint64_t m = u & 0x000fffffffffffffULL;
  /* 1+f in [sqrt(2)/2, sqrt(2)) */
  if (m >= 0x6a09e667f3bcdULL) { /* mantissa >= sqrt(2) */
    k += 1;
    x = wno_u2d(m | ((uint64_t)1022 << 52));
  } else {
    x = wno_u2d(m | ((uint64_t)1023 << 52));
  }
  double f = x - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)k;
  return dk * ln2hi - ((hfsq - (s * (hfsq + R) + dk * ln2lo)) - f);
}

/* x^y for x > 0 (the only use is Adam's t^decay, adam.hpp:83). */
static inline double wno_pow_pos(double x, double y) {
  if (y == 0.0) return 1.0;
  if (y == 1.0) return x;
  if (y == 0.5) return sqrt(x);
  return wno_exp(y * wno_log(x));
}

/* ---- sin(pi a), cos(pi a) for a in [0, 2) ----------------------------- */
static inline void wno_sincospi(double a, double* sn, double* cs) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pi = 3.14159265358979311600e+00;
  double fj = floor(a * 2.0 + 0.5);
  int j = (int)fj;
  double r = a - fj * 0.5;      /* exact; |r| <= 1/4 */
  double x = pi * r;
  double z = x * x;
  double s = x + x * (z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))))));
  double c = (1.0 - 0.5 * z) + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  switch (j & 3) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
  }
}

/* ---- Philox4x32-10 ----------------------------------------------------- */
static inline void wno_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream tags (counter word 3) */
#define WNO_STREAM_MOMENTUM 0u
#define WNO_STREAM_TREE 1u
#define WNO_STREAM_INIT_POS 2u
#define WNO_STREAM_INIT_STEP 3u

/* 64 random bits -> uniform on the open interval (0,1); exact in binary64 */
static inline double wno_u01(uint32_t lo, uint32_t hi) {
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return ((double)(x >> 12) + 0.5) * 2.220446049250313080847e-16; /* 2^-52 */
}

/* one scalar uniform: counter = (index, transition, chain, stream) */
static inline double wno_philox_uniform(uint64_t seed, uint32_t chain, uint32_t transition,
                                        uint32_t stream, uint32_t index) {
  uint32_t ctr[4] = {index, transition, chain, stream};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  wno_philox4x32(ctr, key, o);
  return wno_u01(o[0], o[1]);
}

/* a pair of standard normals (elements 2*pair, 2*pair+1 of the vector);
 * `lg` is the log implementation in force (always the portable one for the
 * counter-based stream: the stream is DEFINED with wno_log). */
static inline void wno_philox_normal_pair(uint64_t seed, uint32_t chain, uint32_t transition,
                                          uint32_t stream, uint32_t pair, double* z0, double* z1) {
  uint32_t ctr[4] = {pair, transition, chain, stream};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  wno_philox4x32(ctr, key, o);
  double u1 = wno_u01(o[0], o[1]);
  double u2 = wno_u01(o[2], o[3]);
  double r = sqrt(-2.0 * wno_log(u1));
  double sn, cs;
  wno_sincospi(2.0 * u2, &sn, &cs);
  *z0 = r * cs;
  *z1 = r * sn;
}

#endif /* WN_ORACLE_MATH_H */
