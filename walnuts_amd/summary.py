"""Posterior summaries on the device: the reference's include/walnutpie/summary.hpp:370-768 (``mean``,
``sample_variance``, ``sample_standard_deviation``, ``quantiles``, ``autocovariance``, ``r_hat``,
``effective_sample_size``, ``monte_carlo_standard_error``) over draws that stay in HBM.

``MarkovChains`` plays the role of the reference's ``MarkovChainsSplit`` / ``MarkovChainsUnified`` containers
(summary.hpp:119-356): possibly ragged chains with a common number of dimensions."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class MarkovChains:
    """Device-resident chains.  Build with ``from_host`` (list of [n_m, D] arrays, or one stacked [N, D] array plus
    sizes) or ``from_device`` (a pointer to [C][max_len][D] draws already in HBM, e.g. the sampler's draw buffer)."""

    def __init__(self, handle, lib):
        self._h = handle
        self.lib = lib

    @classmethod
    def from_host(cls, chains, sizes: Optional[Sequence[int]] = None, device: int = 0, lib_path: Optional[str] = None):
        lib = _ffi.load_library(lib_path)
        if sizes is None:  # MarkovChainsSplit
            mats = [np.asarray(c, dtype=np.float64) for c in chains]
            if len(mats) == 0:
                raise ValueError("require at least one chain")
            if any(m.ndim != 2 for m in mats):
                raise ValueError("each chain must be a [draws, dims] matrix")
            if any(m.shape[1] != mats[0].shape[1] for m in mats):
                raise ValueError("all chains must have the same number of columns")  # summary.hpp:151-156
            sizes = [m.shape[0] for m in mats]
            draws = np.ascontiguousarray(np.concatenate(mats, axis=0))
        else:              # MarkovChainsUnified
            draws = np.ascontiguousarray(np.asarray(chains, dtype=np.float64))
            if draws.ndim != 2:
                raise ValueError("draws must be a [num_draws, dims] matrix")
            if int(np.sum(sizes)) != draws.shape[0]:
                raise ValueError("sum of chain sizes must equal number of rows in draws")  # summary.hpp:277-281
        sz = np.ascontiguousarray(np.asarray(sizes, dtype=np.int64))
        h, err = C.c_void_p(), C.c_void_p()
        rc = lib.wn_chains_upload(C.byref(h), draws.ctypes.data_as(_dp), draws.shape[1], sz.ctypes.data_as(_ip), len(sz),
                                  device, C.byref(err))
        _ffi.check(lib, rc, err)
        return cls(h, lib)

    @classmethod
    def from_device(cls, draws_ptr: int, num_chains: int, max_len: int, dims: int, *, chain_stride: Optional[int] = None,
                    lengths: Optional[Sequence[int]] = None, device: int = 0, stream: int = 0,
                    lib_path: Optional[str] = None):
        lib = _ffi.load_library(lib_path)
        stride = max_len * dims if chain_stride is None else chain_stride
        ln = None if lengths is None else np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        h, err = C.c_void_p(), C.c_void_p()
        rc = lib.wn_chains_view(C.byref(h), C.c_void_p(draws_ptr), num_chains, max_len, dims, stride,
                                None if ln is None else ln.ctypes.data_as(_ip), device, C.c_void_p(stream or None),
                                C.byref(err))
        _ffi.check(lib, rc, err)
        return cls(h, lib)

    # MarkovChainSequence accessors (concepts.hpp / summary.hpp:160-240)
    def num_chains(self) -> int:
        return int(self.lib.wn_chains_num_chains(self._h))

    def dims(self) -> int:
        return int(self.lib.wn_chains_dims(self._h))

    def num_draws(self) -> int:
        return int(self.lib.wn_chains_num_draws(self._h))

    def min_chain_size(self) -> int:
        return int(self.lib.wn_chains_min_chain_size(self._h))

    def close(self):
        if self._h:
            self.lib.wn_chains_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, fn, shape, *extra):
        out = np.zeros(shape, dtype=np.float64)
        err = C.c_void_p()
        rc = fn(self._h, *extra, out.ctypes.data_as(_dp), C.byref(err))
        _ffi.check(self.lib, rc, err)
        return out

    # the free functions below as methods (what a caller holding the chains of walnuts_device(keep_on_device=True) uses)
    def mean(self):
        return mean(self)

    def sample_variance(self):
        return sample_variance(self)

    def quantiles(self, probs):
        return quantiles(self, probs)

    def r_hat(self):
        return r_hat(self)

    def effective_sample_size(self):
        return effective_sample_size(self)

    def monte_carlo_standard_error(self):
        return monte_carlo_standard_error(self)


def mean(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_mean, (chains.dims(),))


def sample_variance(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_sample_variance, (chains.dims(),))


def sample_standard_deviation(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_sample_standard_deviation, (chains.dims(),))


def quantiles(chains: MarkovChains, probs) -> np.ndarray:
    p = np.ascontiguousarray(np.asarray(probs, dtype=np.float64).reshape(-1))
    return chains._vec(chains.lib.wn_summary_quantiles, (len(p), chains.dims()), p.ctypes.data_as(_dp), len(p))


def autocovariance(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_autocovariance, (chains.num_draws(), chains.dims()))


def r_hat(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_r_hat, (chains.dims(),))


def effective_sample_size(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_effective_sample_size, (chains.dims(),))


def monte_carlo_standard_error(chains: MarkovChains) -> np.ndarray:
    return chains._vec(chains.lib.wn_summary_monte_carlo_standard_error, (chains.dims(),))


class Summarizer:
    """The reference's ``walnutpie.Summarizer`` (python/src/walnutpie/summary.py:11-150) on the device library: the
    simple statistics in numpy, ``ess`` / ``r_hat`` / ``mcse`` through the same three C symbols the reference's
    ctypes layer binds (``walnutpie_ess``, ``walnutpie_r_hat``, ``walnutpie_mcse``, walnutpy.cpp:333-369).  Those
    read ``draws`` as Eigen::Map<const MatrixXd>(draws, num_draws, num_params) does, i.e. column-major, so the
    stacked draws are handed over in Fortran order."""

    def __init__(self, draws, lib_path: Optional[str] = None):
        mats = [np.asarray(c, dtype=np.float64) for c in draws]
        self._stacked = np.concatenate(mats)
        self._num_draws, self._num_params = self._stacked.shape
        self._lengths = np.array([c.shape[0] for c in mats], dtype=np.intc)
        self._num_chains = len(mats)
        self.lib = _ffi.load_library(lib_path)

    def mean(self):
        return np.mean(self._stacked, axis=0)

    def variance(self):
        return np.var(self._stacked, axis=0, ddof=1)

    def standard_deviation(self):
        return np.std(self._stacked, axis=0, ddof=1)

    def _call(self, fn):
        col_major = np.asfortranarray(self._stacked)
        out = np.zeros((self._num_params,))
        err = C.c_void_p()
        rc = fn(col_major.ctypes.data_as(_dp), self._num_draws, self._num_params,
                self._lengths.ctypes.data_as(C.POINTER(C.c_int)), self._num_chains, out.ctypes.data_as(_dp),
                C.byref(err))
        _ffi.check(self.lib, rc, err)
        return out

    def ess(self) -> np.ndarray:
        return self._call(self.lib.walnutpie_ess)

    def r_hat(self) -> np.ndarray:
        return self._call(self.lib.walnutpie_r_hat)

    def mcse(self) -> np.ndarray:
        return self._call(self.lib.walnutpie_mcse)
