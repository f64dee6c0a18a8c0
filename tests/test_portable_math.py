"""CPU tier: the product's device maths header (walnuts_amd/csrc/wn_devmath.h), compiled for the host,
against the oracle's independently written copy (bit for bit), libm (few ulp) and the Random123 known-answer
vectors for Philox4x32 (7 and 10 rounds)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIM = r'''
#include "wn_devmath.h"
extern "C" {
double p_exp(double x) { return wnd::dexp(x); }
double p_log(double x) { return wnd::dlog(x); }
double p_exp_weight(double x) { return wnd::dexp_weight(x); }
double p_pow(double x, double y) { return wnd::dpow_pos(x, y); }
void p_sincospi(double a, double* s, double* c) { wnd::dsincospi(a, *s, *c); }
void p_philox(const unsigned* c, const unsigned* k, unsigned* o) {
  wnd::U4 r = wnd::philox(c[0], c[1], c[2], c[3], k[0], k[1]); o[0]=r.x; o[1]=r.y; o[2]=r.z; o[3]=r.w; }
void p_philox10(const unsigned* c, const unsigned* k, unsigned* o) {
  wnd::U4 r = wnd::philox<10>(c[0], c[1], c[2], c[3], k[0], k[1]); o[0]=r.x; o[1]=r.y; o[2]=r.z; o[3]=r.w; }
int p_rounds() { return wnd::kPhiloxRounds; }
void p_div_shared(const double* a, double w, double* out, long n) {
  const wnd::SharedDivisor d(w);
  for (long i = 0; i < n; ++i) out[i] = a[i] / d;
}
double p_uniform(unsigned long long seed, unsigned chain, unsigned t, unsigned stream, unsigned idx) {
  return wnd::stream_uniform(seed, chain, t, stream, idx); }
void p_normal_pair(unsigned long long seed, unsigned chain, unsigned t, unsigned stream, unsigned pair, double* z) {
  wnd::stream_normal_pair(seed, chain, t, stream, pair, z[0], z[1]); }
}
'''


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("devmath")
    src = d / "shim.cpp"
    src.write_text(SHIM)
    so = d / "libshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                           os.path.join(ROOT, "walnuts_amd", "csrc"), str(src), "-o", str(so)])
    L = C.CDLL(str(so))
    for f in ("p_exp", "p_log", "p_exp_weight"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.c_double]
    L.p_pow.restype = C.c_double
    L.p_pow.argtypes = [C.c_double, C.c_double]
    L.p_uniform.restype = C.c_double
    L.p_uniform.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    return L


def test_philox_known_answers(shim):
    # Random123 kat_vectors: philox4x32 with 10 rounds (the round function) and with 7, the engine's stream generator
    kats10 = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
              ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
              ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
               (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    kats7 = [((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
             ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
              (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a))]
    assert shim.p_rounds() == 7
    for fn, kats in ((shim.p_philox10, kats10), (shim.p_philox, kats7)):
        for ctr, key, want in kats:
            c, k, o = (C.c_uint * 4)(*ctr), (C.c_uint * 2)(*key), (C.c_uint * 4)()
            fn(c, k, o)
            assert tuple(o) == want


def test_exp_log_bitwise_equal_to_oracle_copy_and_close_to_libm(shim, oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-745, 709, 20000), rng.normal(0, 3, 20000), [0.0, -0.0, 1.0, -1.0, 709.7, -745.0]])
    L = oracle.lib()
    for x in xs:
        a, b = shim.p_exp(x), L.wno_math_exp(x)
        assert a == b or (np.isnan(a) and np.isnan(b))
        ref = np.exp(x)
        assert abs(a - ref) <= 2.0 * np.spacing(ref) or ref == 0.0   # table-driven exp: within 2 ulp of libm
    ys = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0.5, 2.0, 20000), [1.0, 5e-324, 1e308]])
    for y in ys:
        a, b = shim.p_log(y), L.wno_math_log(y)
        assert a == b
        ref = np.log(y)
        assert abs(a - ref) <= 2.0 * np.spacing(abs(ref)) + 1e-320   # table-driven log: within 2 ulp of libm
    assert shim.p_exp(0.0) == 1.0 and shim.p_log(1.0) == 0.0
    assert shim.p_log(0.0) == -np.inf and np.isnan(shim.p_log(-1.0)) and shim.p_exp(1000.0) == np.inf
    assert shim.p_pow(49.0, 0.5) == 7.0 and shim.p_pow(3.0, 0.0) == 1.0


def test_span_weight_exp_bitwise_equal_to_oracle_copy_and_close_to_libm(shim, oracle):
    """exp for the span weights (wn_devmath.h: dexp_weight): arguments up to the rebase threshold, a floor at -700."""
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-700, 256, 20000), rng.normal(0, 3, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                         [0.0, -0.0, 256.0, -700.0, -699.999, 16.0, -16.0]])
    L = oracle.lib()
    L.wno_math_exp_weight.restype = C.c_double
    L.wno_math_exp_weight.argtypes = [C.c_double]
    for x in xs:
        a, b = shim.p_exp_weight(x), L.wno_math_exp_weight(x)
        assert a == b
        ref = np.exp(x)
        assert abs(a - ref) <= 2.0 * np.spacing(ref)
        assert a == shim.p_exp(x)   # the general exp's main path: the same bits wherever both are defined
    floor = shim.p_exp_weight(-700.0)
    assert 0 < floor < 1e-303 and floor == L.wno_math_exp_weight(-700.0)
    for x in (-701.0, -1e6, -np.inf, np.nan):   # anything below the floor -- and NaN -- stands at the floor
        assert shim.p_exp_weight(x) == floor == L.wno_math_exp_weight(x)
    assert shim.p_exp_weight(0.0) == 1.0


def test_streams_bitwise_equal_to_oracle_copy(shim, oracle):
    for seed, chain, t in ((1, 0, 0), (2**40 + 17, 65535, 1234), (2**63 + 5, 2**31, 2**31 + 1)):
        for idx in range(50):
            assert shim.p_uniform(seed, chain, t, 1, idx) == oracle.stream_uniform(seed, chain, t, 1, idx)
        zs = oracle.stream_normals(seed, chain, t, 0, 101)
        for p in range(51):
            z = (C.c_double * 2)()
            shim.p_normal_pair(C.c_uint64(seed), chain, t, 0, p, z)
            assert z[0] == zs[2 * p]
            if 2 * p + 1 < 101:
                assert z[1] == zs[2 * p + 1]


def test_stream_moments(oracle):
    z = np.concatenate([oracle.stream_normals(9, c, 3, 0, 4096) for c in range(64)])
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01 and abs((z**4).mean() - 3) < 0.08
    u = np.array([oracle.stream_uniform(9, 1, 2, 1, i) for i in range(20000)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_shared_divisor_quotients_are_the_correctly_rounded_ones(shim, oracle):
    """wnd::SharedDivisor (the mass estimator's divisions by its weight: online_moments.hpp:184-191,
    adaptive_walnuts.hpp:89-94) against IEEE division: equal bit for bit on 6e6 numerators over the weights the
    estimator's recurrence w <- (1 - 1/(count + i)) w + 1 actually takes (adaptive_walnuts.hpp:74-80) and over random
    divisors, numerators from subnormal-free 1e-300 to 1e300 in both signs plus exact zeros -- and equal to the oracle's
    restatement (the comparison the parity tests rest on)."""
    dp = C.POINTER(C.c_double)
    shim.p_div_shared.argtypes = [dp, C.c_double, dp, C.c_long]
    rng = np.random.default_rng(17)
    weights = []
    for count in (4.0, 1.0, 10.0, 0.5):
        w = count
        for i in range(400):
            weights.append(w)
            w = (1.0 - 1.0 / (count + i)) * w + 1.0
    weights += list(np.exp(rng.uniform(-50, 50, size=1200)))
    n = 2500
    bad = 0
    for w in weights:
        a = np.concatenate([rng.normal(size=n // 2) * np.exp(rng.uniform(-600, 600, size=n // 2)),
                            rng.normal(size=n // 2 - 2), [0.0, -0.0]])
        out = np.empty_like(a)
        shim.p_div_shared(a.ctypes.data_as(dp), float(w), out.ctypes.data_as(dp), a.size)
        want = a / w
        normal = (np.abs(want) > 2.3e-308) | (want == 0)   # (a subnormal quotient may differ in its last bit)
        bad += int(np.sum((out != want) & normal))
        for k in (0, 1, n // 2, n - 1):
            assert oracle.div_shared(float(a[k]), float(w)) == out[k]
    assert bad == 0
