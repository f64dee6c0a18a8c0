"""ctypes binding of the posterior-summary oracle (oracle/wn_summary_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Chains are a list of [n_m, D] arrays (MarkovChainsSplit, summary.hpp:119-240) or one stacked [N, D] array plus
sizes (MarkovChainsUnified, :251-356); function names are the reference's."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwn_summary_oracle.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB_PATH)
        _lib.wnso_last_error.restype = C.c_char_p
        _lib.wnso_fft_next_good_size.restype = C.c_int64
        _lib.wnso_fft_next_good_size.argtypes = [C.c_int64]
        for name in ("mean", "sample_variance", "sample_standard_deviation", "autocovariance", "r_hat",
                     "effective_sample_size", "monte_carlo_standard_error"):
            f = getattr(_lib, "wnso_" + name)
            f.restype = C.c_int
            f.argtypes = [_dp, C.c_size_t, _ip, C.c_size_t, _dp]
        _lib.wnso_quantiles.restype = C.c_int
        _lib.wnso_quantiles.argtypes = [_dp, C.c_size_t, _ip, C.c_size_t, _dp, C.c_size_t, _dp]
    return _lib


def unify(chains, sizes=None):
    """-> (stacked [N, D] float64 C-contiguous, sizes int64[K])"""
    if sizes is None:
        chains = [np.atleast_2d(np.asarray(c, dtype=np.float64)) for c in chains]
        sizes = np.array([c.shape[0] for c in chains], dtype=np.int64)
        draws = np.ascontiguousarray(np.concatenate(chains, axis=0)) if chains else np.zeros((0, 0))
    else:
        draws = np.ascontiguousarray(np.asarray(chains, dtype=np.float64))
        sizes = np.asarray(sizes, dtype=np.int64)
        if sizes.sum() != draws.shape[0]:
            raise ValueError("sum of chain sizes must equal number of rows in draws")  # summary.hpp:277-281
    return draws, sizes


def _call(name, chains, sizes, out_shape, *extra):
    L = _load()
    draws, sz = unify(chains, sizes)
    D = draws.shape[1] if draws.ndim == 2 else 0
    shape = out_shape(draws.shape[0], D)
    out = np.zeros(shape, dtype=np.float64)
    args = [draws.ctypes.data_as(_dp), D, sz.ctypes.data_as(_ip), len(sz)] + list(extra) + [out.ctypes.data_as(_dp)]
    rc = getattr(L, "wnso_" + name)(*args)
    if rc == 1:
        raise ValueError(L.wnso_last_error().decode())
    if rc != 0:
        raise RuntimeError(L.wnso_last_error().decode())
    return out


def mean(chains, sizes=None):
    return _call("mean", chains, sizes, lambda n, d: (d,))


def sample_variance(chains, sizes=None):
    return _call("sample_variance", chains, sizes, lambda n, d: (d,))


def sample_standard_deviation(chains, sizes=None):
    return _call("sample_standard_deviation", chains, sizes, lambda n, d: (d,))


def quantiles(chains, probs, sizes=None):
    p = np.ascontiguousarray(np.asarray(probs, dtype=np.float64))
    return _call("quantiles", chains, sizes, lambda n, d: (len(p), d), p.ctypes.data_as(_dp), len(p))


def autocovariance(chains, sizes=None):
    return _call("autocovariance", chains, sizes, lambda n, d: (n, d))


def r_hat(chains, sizes=None):
    return _call("r_hat", chains, sizes, lambda n, d: (d,))


def effective_sample_size(chains, sizes=None):
    return _call("effective_sample_size", chains, sizes, lambda n, d: (d,))


def monte_carlo_standard_error(chains, sizes=None):
    return _call("monte_carlo_standard_error", chains, sizes, lambda n, d: (d,))


def fft_next_good_size(n: int) -> int:
    return int(_load().wnso_fft_next_good_size(n))
