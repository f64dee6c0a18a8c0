"""``walnuts_device``: the reference's ``walnuts_pyfunc`` (python/src/walnutpie/pyfunc.py:45-286) for built-in
device models.  Same keyword arguments, defaults, output arrays and error mapping; the host log-density
callable is replaced by ``model`` (+ ``model_params``)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from collections.abc import Sequence
from typing import Generic, Optional, TypeVar

import numpy as np

from . import _ffi

T = TypeVar("T")


@dataclass
class WarmupInfo(Generic[T]):  # python/src/walnutpie/util.py:47-66
    stepsize: float
    inv_metric: Optional[np.ndarray]
    warmup_draws: Optional[T]
    # not in the reference: which definition of the library's counter-based random streams produced this run
    # (wn_stream_version()): a stored result is reproducible from its seed only under the same version
    stream_version: Optional[int] = None


class WalnutsOutputArray(np.ndarray):  # python/src/walnutpie/pyfunc.py (ndarray with a .warmup attribute)
    warmup: WarmupInfo

    def __new__(cls, input_array, warmup: WarmupInfo):
        obj = np.asarray(input_array).view(cls)
        obj.warmup = warmup
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.warmup = getattr(obj, "warmup", None)


class ChainResults(Sequence):
    """The per-chain results of a call with MANY chains, built on access (``walnuts_device(..., lazy_results=True)``):
    ``results[c]`` is what the reference's list holds at index c (pyfunc.py:270-286: the chain's sampling draws as an
    array carrying ``.warmup``).  Creating 65 536 array views and WarmupInfo objects up front costs 0.2 s -- more than
    the 0.14 s the device needs for 20 + 32 iterations of 65 536 x 1 024 (profiles/r04/sample_device_e2e.txt).  A
    read-only sequence (len, indexing, slicing, iteration, ``list(results)``); an item is built once and kept, so
    ``results[c] is results[c]``.  The default of ``walnuts_device`` is the reference's plain list."""

    def __init__(self, make, n):
        self._make, self._n, self._built = make, n, {}

    def __len__(self):
        return self._n

    def _item(self, c):
        item = self._built.get(c)
        if item is None:
            item = self._built[c] = self._make(c)
        return item

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._item(c) for c in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError("chain index out of range")
        return self._item(i)


def _prepare_output_buffer(*, num_chains, num_params, max_sampling_iter, max_warmup_iter, save_warmup):
    # python/src/walnutpie/util.py:16-32
    if num_chains < 1:
        raise ValueError("num_chains must be at least 1")
    if max_warmup_iter < 0:
        raise ValueError("max_warmup_iter must be non-negative")
    if max_sampling_iter < 1:
        raise ValueError("max_sampling_iter must be at least 1")
    num_draws = max_sampling_iter + max_warmup_iter * save_warmup
    return np.zeros((num_chains, num_draws, num_params), dtype=np.float64)


def _prepare_inv_metric(init_inv_metric, metric_size, num_chains):
    # python/src/walnutpie/util.py:35-46
    if init_inv_metric is None:
        return None
    init_inv_metric = np.asarray(init_inv_metric, dtype=np.float64)
    if init_inv_metric.shape == metric_size:
        return np.ascontiguousarray(np.repeat(init_inv_metric[np.newaxis], num_chains, axis=0))
    if init_inv_metric.shape == (num_chains, *metric_size):
        return np.ascontiguousarray(init_inv_metric)
    raise ValueError(f"Invalid initial metric size. Expected a {metric_size} or {(num_chains, *metric_size)} matrix.")


def walnuts_device(
    model: int,
    *,
    model_params: Optional[np.ndarray] = None,
    num_params: Optional[int] = None,
    inits: Optional[np.ndarray] = None,
    num_chains: int = 4,
    seed: Optional[int] = None,
    id: int = 1,
    init_radius: float = 2.0,
    init_inv_metric: Optional[np.ndarray] = None,
    save_inv_metric: bool = False,
    min_warmup_iter: int = 50,
    max_warmup_iter: int = 1000,
    min_sampling_iter: int = 50,
    max_sampling_iter: int = 1000,
    max_trajectory_doublings: int = 5,
    max_step_halvings: int = 5,
    min_micro_steps: int = 1,
    max_hamiltonian_error: float = 0.5,
    step_size_converge_tol: float = 0.1,
    mass_converge_tol: float = 1.0,
    rhat_converge_tol: float = 1.01,
    mass_init_count: float = 4.0,
    mass_additive_smoothing: float = 1e-5,
    max_macro_steps_target: float = 15.0,
    step_size_init: float = 1.0,
    step_accept_rate_target: float = 0.8,
    step_learning_rate: float = 0.05,
    step_gradient_decay: float = 0.8,
    step_sq_gradient_decay: float = 0.9,
    step_stabilization: float = 1e-4,
    step_learn_rate_decay: float = 0.5,
    save_warmup: bool = False,
    refresh: int = 0,
    reference_streams: bool = False,
    keep_on_device: bool = False,
    thin: int = 0,
    devices=None,
    lazy_results: bool = False,
    all_gather: bool = False,
    lib_path: Optional[str] = None,
    print_callback=None,
):
    """The device-model sibling of the reference's ``walnuts_pyfunc`` (pyfunc.py:45-286): same keywords, same result
    (a list of per-chain draw arrays carrying ``.warmup``).

    ``keep_on_device=True`` (walnutpie_sample_device_resident): the sampling draws stay in HBM and the call returns
    ``(results, chains)`` -- ``chains`` a :class:`walnuts_amd.summary.MarkovChains` over ALL sampling draws (mean,
    variance, quantiles, R-hat, ESS, MCSE computed on the device), ``results[c]`` holding only every ``thin``-th draw
    (``thin=0``: none): 65 536 chains x 1 024 parameters are 512 MiB per iteration, six times what PCIe moves in the
    time the GPU needs to produce them.

    ``devices=[0, 1, ...]`` (walnutpie_sample_device_multi): the chains are sharded over these HIP devices of the node,
    one host thread + engine + stream each, every shard writing its own slice of the output; same result as the
    one-device call (random streams keyed by global chain id, controllers reduced over all shards).  An ordinal may
    repeat: ``devices=[0, 0]`` runs two half-size engines on one device, each filling the other's launch tail.  With
    ``keep_on_device=True`` (walnutpie_sample_device_multi_resident) every shard keeps its draws on its own device and
    the blocks are gathered on ``devices[0]`` by peer-to-peer copies at the end: one ``MarkovChains`` handle there;
    with ``all_gather=True`` as well (walnutpie_sample_device_multi_allgather) EVERY listed device ends with the whole
    block -- the call returns ``(results, [chains on devices[0], chains on devices[1], ...])``."""
    lib = _ffi.load_library(lib_path)
    if devices is not None and reference_streams:
        raise ValueError("devices is not available with reference_streams")
    if keep_on_device and reference_streams:
        raise ValueError("keep_on_device is not available with reference_streams")
    if thin < 0:
        raise ValueError("thin must be non-negative")
    if all_gather and not (keep_on_device and devices is not None):
        raise ValueError("all_gather needs devices=[...] and keep_on_device=True")
    if inits is not None:
        inits = np.asarray(inits, dtype=np.float64)
        if inits.ndim == 1:
            inits = np.repeat(inits[np.newaxis], num_chains, axis=0)
        if inits.shape[0] != num_chains:
            raise ValueError("inits must have one row per chain")
        inits = np.ascontiguousarray(inits)
        if num_params is None:
            num_params = inits.shape[1]
        elif num_params != inits.shape[1]:
            raise ValueError("num_params does not match inits")
    if num_params is None:
        raise ValueError("At least one of num_params or inits must be specified")
    if seed is None:
        seed = int(np.random.randint(0, 2**32 - 1, dtype=np.uint32))
    mp = None if model_params is None else np.ascontiguousarray(np.asarray(model_params, dtype=np.float64))
    if mp is not None and mp.size != num_params:
        raise ValueError("model_params must have num_params entries")

    out = _prepare_output_buffer(num_chains=num_chains, num_params=num_params, max_sampling_iter=max_sampling_iter,
                                 max_warmup_iter=max_warmup_iter, save_warmup=save_warmup) if not keep_on_device else None
    if keep_on_device:   # only every thin-th sampling draw comes to the host
        _prepare_output_buffer(num_chains=num_chains, num_params=1, max_sampling_iter=max_sampling_iter,
                               max_warmup_iter=0, save_warmup=False)   # (the same argument checks)
        rows_sampling = 0 if thin == 0 else -(-max_sampling_iter // thin)
        out = np.zeros((num_chains, rows_sampling + max_warmup_iter * save_warmup, num_params), dtype=np.float64)
    inv_metric_init = _prepare_inv_metric(init_inv_metric, (num_params,), num_chains)
    final_lengths = np.zeros(2 * num_chains, dtype=np.intc)
    stepsize_out = np.zeros(num_chains, dtype=np.float64)
    inv_metric_out = np.zeros((num_chains, num_params), dtype=np.float64) if save_inv_metric else None

    def _print(msg, length, bad):
        text = msg[:length].decode("utf-8", "replace")
        if print_callback is not None:
            print_callback(text)
        else:
            print(text, end="", flush=True)

    stream_version = int(lib.wn_stream_version())
    cb = _ffi.PRINT_CALLBACK(_print)
    dp = _ffi._dp
    err = C.c_void_p()
    entry = lib.walnutpie_sample_device_reference_streams if reference_streams else lib.walnutpie_sample_device
    chains_handle = C.c_void_p()
    tail = (refresh, cb, C.byref(err))
    if keep_on_device:
        entry = lib.walnutpie_sample_device_resident
        tail = (refresh, cb, thin, C.byref(chains_handle), C.byref(err))
    if devices is not None:
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        entry = lib.walnutpie_sample_device_multi
        tail = (refresh, cb, dev, len(devices), C.byref(err))
        if keep_on_device:   # every shard's draws stay on its device, gathered on devices[0] at the end
            entry = lib.walnutpie_sample_device_multi_resident
            tail = (refresh, cb, dev, len(devices), thin, C.byref(chains_handle), C.byref(err))
            if all_gather:   # ... and every device gets every shard's draws
                entry = lib.walnutpie_sample_device_multi_allgather
                handles = (C.c_void_p * len(devices))()
                tail = (refresh, cb, dev, len(devices), thin, handles, C.byref(err))
    import os
    import sys
    import time

    timing = os.environ.get("WALNUTS_AMD_TIMING") is not None   # the C side prints its phases under the same switch
    t_call = time.perf_counter()
    rc = entry(
        model, None if mp is None else mp.ctypes.data_as(dp), num_params,
        None if inits is None else inits.ctypes.data_as(dp), num_chains, seed, id, init_radius,
        None if inv_metric_init is None else inv_metric_init.ctypes.data_as(dp), min_warmup_iter, max_warmup_iter,
        min_sampling_iter, max_sampling_iter, max_trajectory_doublings, max_step_halvings, min_micro_steps,
        max_hamiltonian_error, step_size_converge_tol, mass_converge_tol, rhat_converge_tol, mass_init_count,
        mass_additive_smoothing, max_macro_steps_target, step_size_init, step_accept_rate_target, step_learning_rate,
        step_gradient_decay, step_sq_gradient_decay, step_stabilization, step_learn_rate_decay, save_warmup,
        out.ctypes.data_as(dp) if out.size else None, out.size, final_lengths.ctypes.data_as(C.POINTER(C.c_int)),
        stepsize_out.ctypes.data_as(dp), None if inv_metric_out is None else inv_metric_out.ctypes.data_as(dp),
        *tail)
    _ffi.check(lib, rc, err)
    t_done = time.perf_counter()

    def result_of(c):  # python/src/walnutpie/pyfunc.py:270-286
        n_warm, n_samp = int(final_lengths[c]), int(final_lengths[num_chains + c])
        warm = out[c, :n_warm] if save_warmup else None
        info = WarmupInfo(stepsize=float(stepsize_out[c]),
                          inv_metric=None if inv_metric_out is None else inv_metric_out[c], warmup_draws=warm,
                          stream_version=None if reference_streams else stream_version)
        if keep_on_device:   # rows 0, thin, 2 thin, ... of the n_samp draws the device holds
            n_samp = 0 if thin == 0 else -(-n_samp // thin)
        return WalnutsOutputArray(out[c, n_warm:n_warm + n_samp], info)

    # a list, as the reference returns it (pyfunc.py:270-286); on request the same thing built on access
    results = ChainResults(result_of, num_chains) if lazy_results else [result_of(c) for c in range(num_chains)]
    if timing:
        print(f"[walnuts_amd] {'C entry point (all phases above)':34s} {(t_done - t_call) * 1e3:9.3f} ms\n"
              f"[walnuts_amd] {'per-chain result objects (Python)':34s} {(time.perf_counter() - t_done) * 1e3:9.3f} ms",
              file=sys.stderr, flush=True)
    if keep_on_device:
        from .summary import MarkovChains

        if all_gather and devices is not None:
            return results, [MarkovChains(C.c_void_p(h), lib) for h in handles]
        return results, MarkovChains(chains_handle, lib)
    return results
