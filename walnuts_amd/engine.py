"""Batched engine handle: the per-transition verbs of include/walnuts_hip.h.

``DeviceEngine`` advances ALL chains by one transition per call; its methods are named after the reference
objects they batch: ``warmup_step`` = ``AdaptiveWalnuts::operator()`` (adaptive_walnuts.hpp:234-251),
``freeze`` = ``AdaptiveWalnuts::sampler()`` (:263-271), ``sample_step`` = ``WalnutsSampler::operator()``
(walnuts.hpp:682-692); the init methods follow ``InitConfigBuilder`` (config.hpp:195-484).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi

MODEL_STD_NORMAL, MODEL_DIAG_NORMAL, MODEL_FUNNEL, MODEL_RW1 = 0, 1, 2, 3
_dp = _ffi._dp


def default_config(lib_path: Optional[str] = None, **overrides) -> _ffi.Config:
    cfg = _ffi.Config()
    _ffi.load_library(lib_path).wn_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def model_id(name: str, lib_path: Optional[str] = None) -> int:
    """Id of the device model registered under `name` (see walnuts_amd/csrc/wn_model_api.h); ValueError if none."""
    i = _ffi.load_library(lib_path).wn_model_id(name.encode())
    if i < 0:
        raise ValueError(f"no device model named {name!r} in this build of the library")
    return i


def stream_version(lib_path: Optional[str] = None) -> int:
    """Version of the library's counter-based random streams (wn_stream_version): same seed + same version = same run."""
    return int(_ffi.load_library(lib_path).wn_stream_version())


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class DeviceEngine:
    def __init__(self, model: int, dim: int, num_chains: int, cfg: Optional[_ffi.Config] = None,
                 params: Optional[np.ndarray] = None, lib_path: Optional[str] = None):
        self.lib = _ffi.load_library(lib_path)
        self.cfg = cfg if cfg is not None else default_config(lib_path)
        self.C, self.D = int(num_chains), int(dim)
        p = None if params is None else _f64(params)
        if p is not None and p.size != dim:
            raise ValueError("model params must have num_params entries")
        h, err = C.c_void_p(), C.c_void_p()
        rc = self.lib.wn_engine_create(C.byref(h), model, dim, None if p is None else p.ctypes.data_as(_dp),
                                       num_chains, C.byref(self.cfg), C.byref(err))
        _ffi.check(self.lib, rc, err)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.wn_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, *args):
        err = C.c_void_p()
        _ffi.check(self.lib, fn(self.h, *args, C.byref(err)), err)

    # ---- InitConfigBuilder
    def set_positions(self, pos):
        a = _f64(pos).reshape(self.C, self.D)
        self._call(self.lib.wn_engine_set_positions, a.ctypes.data_as(_dp))

    def set_masses(self, mass):
        a = _f64(mass).reshape(self.C, self.D)
        self._call(self.lib.wn_engine_set_masses, a.ctypes.data_as(_dp))

    def set_step_sizes(self, steps):
        a = _f64(np.broadcast_to(np.asarray(steps, dtype=np.float64), (self.C,)))
        self._call(self.lib.wn_engine_set_step_sizes, a.ctypes.data_as(_dp))

    def init_positions(self, seed: int, chain_offset: int, scale: float):
        self._call(self.lib.wn_engine_init_positions, seed, chain_offset, scale)

    def init_masses_from_grad(self, smoothing: float, average: bool = False):
        """InitConfigBuilder::masses(logp_grad, smoothing, average_masses) (config.hpp:360-382)."""
        self._call(self.lib.wn_engine_init_masses_from_grad, smoothing)
        if average:
            self._call(self.lib.wn_engine_average_masses)

    def adapt_step(self, seed: int, chain_offset: int = 0):
        self._call(self.lib.wn_engine_adapt_step, seed, chain_offset)

    def adapt_step_with_normals(self, normals):
        a = _f64(normals).reshape(self.C, self.D)
        self._call(self.lib.wn_engine_adapt_step_with_normals, a.ctypes.data_as(_dp))

    def seed_chains(self, seed: int, chain_offset: int = 0):
        self._call(self.lib.wn_engine_seed, seed, chain_offset)

    def seed_reference_streams(self, seed: int):
        """Parity mode: the reference's mt19937_64(seed_seq{seed, m+1}) + libstdc++ distributions, generated on
        the host for every following transition (api.hpp:46-51, util.hpp:78-162)."""
        self._call(self.lib.wn_engine_seed_reference_streams, seed)

    def set_variates(self, normals, uniforms):
        z = _f64(normals).reshape(self.C, self.D)
        u = _f64(uniforms).reshape(self.C, -1)
        self._call(self.lib.wn_engine_set_variates, z.ctypes.data_as(_dp), u.ctypes.data_as(_dp), u.shape[1])

    # ---- transitions.  draws_ptr: integer device address (e.g. torch tensor .data_ptr()) or None
    def warmup_step(self, draws_ptr: Optional[int] = None, stride: int = 0):
        self._call(self.lib.wn_engine_warmup_step, C.c_void_p(draws_ptr), stride)

    def freeze(self):
        self._call(self.lib.wn_engine_freeze)

    def sample_step(self, draws_ptr: Optional[int] = None, stride: int = 0):
        self._call(self.lib.wn_engine_sample_step, C.c_void_p(draws_ptr), stride)

    def warmup_steps(self, transitions: int, draws_ptr: Optional[int] = None, stride: int = 0, transition_stride: int = 0):
        """`transitions` warmup transitions of every chain in one launch (same bits as as many warmup_step calls);
        chain c's k-th position at draws_ptr + c*stride + k*transition_stride doubles."""
        self._call(self.lib.wn_engine_warmup_steps, int(transitions), C.c_void_p(draws_ptr), stride, transition_stride)

    def sample_steps(self, transitions: int, draws_ptr: Optional[int] = None, stride: int = 0, transition_stride: int = 0):
        """`transitions` sampling transitions of every chain in one launch (same bits as as many sample_step calls)."""
        self._call(self.lib.wn_engine_sample_steps, int(transitions), C.c_void_p(draws_ptr), stride, transition_stride)

    def synchronize(self):
        self._call(self.lib.wn_engine_synchronize)

    def check(self):
        """Raise if any chain's last transition could not complete on the device."""
        self._call(self.lib.wn_engine_check)

    # ---- state
    def _get(self, fn, shape, dtype=np.float64, ptr=_dp):
        out = np.empty(shape, dtype=dtype)
        self._call(fn, out.ctypes.data_as(ptr))
        return out

    def positions(self):
        return self._get(self.lib.wn_engine_get_positions, (self.C, self.D))

    def masses(self):
        return self._get(self.lib.wn_engine_get_masses, (self.C, self.D))

    def inv_mass(self):
        return self._get(self.lib.wn_engine_get_inv_mass, (self.C, self.D))

    def step_sizes(self):
        return self._get(self.lib.wn_engine_get_step_sizes, (self.C,))

    def logp(self):
        return self._get(self.lib.wn_engine_get_logp, (self.C,))

    def adam(self):
        return self._get(self.lib.wn_engine_get_adam, (self.C, 6))

    def min_micro(self):
        return self._get(self.lib.wn_engine_get_min_micro, (self.C,), np.int32, _ffi._i32p)

    def depths(self):
        return self._get(self.lib.wn_engine_get_depths, (self.C,), np.int32, _ffi._i32p)

    def grad_evals(self):
        return self._get(self.lib.wn_engine_get_grad_evals, (self.C,), np.int64, _ffi._i64p)

    def failed_extensions(self):
        """Per chain: 1 if an extension of the last transition failed (a leaf's energy error above the bound at every
        step size -- where a model returning non-finite values ends up; the device counterpart of the reference's
        on_logp_exception events, util.hpp:336-346 -- or a failed reversibility check)."""
        return self._get(self.lib.wn_engine_get_failed_extensions, (self.C,), np.int32, _ffi._i32p)

    def rng_draws(self):
        return self._get(self.lib.wn_engine_get_rng_draws, (self.C,), np.int32, _ffi._i32p)

    def estimator(self):
        dm, ds, sm, ss = (np.empty((self.C, self.D)) for _ in range(4))
        w = np.empty((self.C, 2))
        self._call(self.lib.wn_engine_get_estimator, *(x.ctypes.data_as(_dp) for x in (dm, ds, sm, ss, w)))
        return dict(draw_mean=dm, draw_ssd=ds, score_mean=sm, score_ssd=ss, weights=w)

    def total_grad_evals(self) -> int:
        v = C.c_int64()
        self._call(self.lib.wn_engine_total_grad_evals, C.byref(v))
        return v.value

    def last_kernel_ms(self) -> float:
        v = C.c_float()
        self._call(self.lib.wn_engine_last_kernel_ms, C.byref(v))
        return v.value

    def rhat(self) -> float:
        """R-hat of the log density over the sampling draws so far (sampler.hpp:132-145)."""
        v = C.c_double()
        self._call(self.lib.wn_engine_rhat, C.cast(C.byref(v), _dp))
        return v.value

    def lp_sums(self):
        """Stage 1 of R-hat for a multi-GPU driver: (sum of chain means, sum of chain sample variances, chains) of the
        log density -- all-reduce SUM."""
        out = np.zeros(3)
        self._call(self.lib.wn_engine_lp_sums, out.ctypes.data_as(_dp))
        return out

    def lp_sq_dev(self, mean_of_means: float) -> float:
        """Stage 2: sum over this engine's chains of (chain mean - mean of means)^2 -- all-reduce SUM."""
        v = C.c_double()
        self._call(self.lib.wn_engine_lp_sq_dev, mean_of_means, C.cast(C.byref(v), _dp))
        return v.value

    def warmup_spread(self):
        """(max rel. step-size distance, max rel. mass distance) from the chains' geometric means
        (adapt.hpp:193-221)."""
        a, b = C.c_double(), C.c_double()
        self._call(self.lib.wn_engine_warmup_spread, C.cast(C.byref(a), _dp), C.cast(C.byref(b), _dp))
        return a.value, b.value

    def warmup_sums(self):
        """Stage 1 of the warmup statistic for a multi-GPU driver: (sum over this engine's chains of log step,
        [D] sums of log mass) -- all-reduce SUM these D+1 doubles."""
        s = C.c_double()
        col = np.zeros(self.D)
        self._call(self.lib.wn_engine_warmup_sums, C.cast(C.byref(s), _dp), col.ctypes.data_as(_dp))
        return s.value, col

    def warmup_max_rel(self, sum_log_step: float, colsum_log_mass, total_chains: int):
        """Stage 2: this engine's (max rel. step distance, max rel. mass distance) from the geometric means over
        ALL `total_chains` chains -- all-reduce MAX."""
        col = _f64(colsum_log_mass).reshape(self.D)
        a, b = C.c_double(), C.c_double()
        self._call(self.lib.wn_engine_warmup_max_rel, sum_log_step, col.ctypes.data_as(_dp), total_chains,
                   C.cast(C.byref(a), _dp), C.cast(C.byref(b), _dp))
        return a.value, b.value

    def region_begin(self):
        """Start of a timed region of launches (one pair of HIP events around all of them)."""
        self._call(self.lib.wn_engine_region_begin)

    def region_ms(self):
        """-> (elapsed ms since region_begin on the engine's stream, transition launches in between)."""
        ms, n = C.c_float(), C.c_int()
        self._call(self.lib.wn_engine_region_ms, C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)

    def timing_reset(self):
        self._call(self.lib.wn_engine_timing_reset)

    def kernel_times_ms(self, max_launches: int = 1 << 16) -> np.ndarray:
        buf = np.zeros(max_launches, dtype=np.float32)
        n = C.c_int()
        self._call(self.lib.wn_engine_kernel_times, buf.ctypes.data_as(C.POINTER(C.c_float)), max_launches,
                   C.byref(n))
        return buf[: min(n.value, max_launches)].astype(np.float64)

    def set_stream(self, stream_handle: int):
        self._call(self.lib.wn_engine_set_stream, C.c_void_p(stream_handle))

    def wait_stream(self, stream_handle: int):
        """The next transition launches wait for what the caller's stream holds now (wn_engine_wait_stream)."""
        self._call(self.lib.wn_engine_wait_stream, C.c_void_p(stream_handle))

    def release_stream(self, stream_handle: int):
        """The caller's stream waits for every transition launch made so far (wn_engine_release_stream)."""
        self._call(self.lib.wn_engine_release_stream, C.c_void_p(stream_handle))

    @property
    def lanes(self) -> int:
        return self.lib.wn_engine_lanes(self.h)

    @property
    def dim_padded(self) -> int:
        return self.lib.wn_engine_dim_padded(self.h)

    @property
    def streaming(self) -> bool:
        return bool(self.lib.wn_engine_is_streaming(self.h))

    @property
    def held_tiles(self) -> int:
        """Streaming kernels: 16-byte pairs per lane of the trajectory's moving end kept in registers (0: both ends of
        every micro step stream through HBM)."""
        return self.lib.wn_engine_held_tiles(self.h)

    @property
    def workgroups(self) -> int:
        return self.lib.wn_engine_workgroups(self.h)

    @property
    def chain_groups(self) -> int:
        """Kernels one transition launch consists of (wn_config::chain_groups): contiguous chain blocks, one stream each."""
        return self.lib.wn_engine_chain_groups(self.h)

    @property
    def lds_vectors(self) -> int:
        return self.lib.wn_engine_lds_vectors(self.h)

    @property
    def iteration(self) -> int:
        return self.lib.wn_engine_iteration(self.h)

    @property
    def stream(self) -> int:
        return self.lib.wn_engine_stream(self.h) or 0

    @property
    def positions_device_ptr(self) -> int:
        return self.lib.wn_engine_positions_device(self.h) or 0
