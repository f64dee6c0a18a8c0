"""ctypes binding of include/walnuts_hip.h (the role of python/src/walnutpie/_ffi.py in the reference)."""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libwalnuts_hip.so")

_dp = C.POINTER(C.c_double)
_errpp = C.POINTER(C.c_void_p)
PRINT_CALLBACK = C.CFUNCTYPE(None, C.c_char_p, C.c_size_t, C.c_bool)


class WalnutsHipError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("max_trajectory_doublings", C.c_int32),
        ("max_step_halvings", C.c_int32),
        ("min_micro_steps", C.c_int32),
        ("device", C.c_int32),
        ("max_hamiltonian_error", C.c_double),
        ("mass_init_count", C.c_double),
        ("max_macro_steps_target", C.c_double),
        ("step_accept_rate_target", C.c_double),
        ("step_learning_rate", C.c_double),
        ("step_gradient_decay", C.c_double),
        ("step_sq_gradient_decay", C.c_double),
        ("step_stabilization", C.c_double),
        ("step_learn_rate_decay", C.c_double),
        ("waves_per_chain", C.c_int32),
        ("elems_per_lane", C.c_int32),
        ("workgroups_per_cu", C.c_int32),
        ("lds_vectors", C.c_int32),
        ("reserved_cus", C.c_int32),
        ("fused_multiply_add", C.c_int32),
        ("chain_groups", C.c_int32),
    ]


# every symbol include/walnuts_hip.h declares: (name, restype, argtypes)
_vp, _sz, _i32, _i64, _u32, _u64, _dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
_i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
# the reference's trailing sampling arguments (python/src/walnutpie/_ffi.py: _common_sampling_argtypes)
_REFERENCE_SAMPLING_ARGS = [_sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl, _dbl,
                            _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz,
                            C.POINTER(C.c_int), _dp, _dp, _i32, PRINT_CALLBACK, _errpp]
LOGP_CFUNC = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                         C.c_void_p)
SYMBOLS = [
    ("walnutpie_get_error_message", C.c_char_p, [_vp]),
    ("walnutpie_get_error_type", _i32, [_vp]),
    ("walnutpie_destroy_error", None, [_vp]),
    # host-model entry points of the reference: exported so that the library loads where libwalnutpy is expected; a call
    # fails with a config error (no CPU path exists here)
    ("walnutpie_sample_cfunc", _i32, [LOGP_CFUNC, _vp, _i32, _dp] + _REFERENCE_SAMPLING_ARGS),
    ("walnutpie_sample_bridgestan", _i32, [C.c_char_p, C.c_char_p, PRINT_CALLBACK, C.c_uint, C.c_char_p]
     + _REFERENCE_SAMPLING_ARGS),
    ("walnutpie_separator_char", C.c_char, []),
    ("walnutpie_sample_device", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, _errpp]),
    ("walnutpie_sample_device_reference_streams", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, _errpp]),
    ("walnutpie_sample_device_resident", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, _i32, C.POINTER(_vp), _errpp]),
    ("walnutpie_sample_device_multi", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, C.POINTER(C.c_int), _i32, _errpp]),
    ("walnutpie_sample_device_multi_resident", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, C.POINTER(C.c_int), _i32, _i32, C.POINTER(_vp), _errpp]),
    ("walnutpie_sample_device_multi_allgather", _i32,
     [_i32, _dp, _i32, _dp, _sz, C.c_uint, C.c_uint, _dbl, _dp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl,
      _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_bool, _dp, _sz, C.POINTER(C.c_int),
      _dp, _dp, _i32, PRINT_CALLBACK, C.POINTER(C.c_int), _i32, _i32, C.POINTER(_vp), _errpp]),
    ("wn_internal_sqrt_probe", _i32, [_dp, _dp, _sz, _i32]),
    ("wn_internal_reference_normals", None, [C.c_uint, C.c_uint, _sz, _sz, _i32, _dbl, _dp]),
    ("walnutpie_ess", _i32, [_dp, _i32, _i32, C.POINTER(C.c_int), _i32, _dp, _errpp]),
    ("walnutpie_r_hat", _i32, [_dp, _i32, _i32, C.POINTER(C.c_int), _i32, _dp, _errpp]),
    ("walnutpie_mcse", _i32, [_dp, _i32, _i32, C.POINTER(C.c_int), _i32, _dp, _errpp]),
    ("wn_default_config", None, [C.POINTER(Config)]),
    ("wn_model_id", _i32, [C.c_char_p]),
    ("wn_stream_version", _i32, []),
    ("wn_build_flags", C.c_char_p, []),
    ("wn_build_compiler", C.c_char_p, []),
    ("wn_plugin_register_model", _i32, [_vp, _vp]),
    ("wn_model_error", C.c_char_p, []),
    ("wn_model_clear_error", None, []),
    ("wn_geometry_for", _i32, [_i32, _i32, _i32, _i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _errpp]),
    ("wn_geometry_for_model", _i32, [_i32, _i32, _i32, _i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _errpp]),
    ("wn_geometry_candidates", _i32, [_i32, _i32, _i32, _i32, C.POINTER(C.c_int), _i32, C.POINTER(C.c_int), _errpp]),
    ("wn_engine_create", _i32, [C.POINTER(_vp), _i32, _i32, _dp, _sz, C.POINTER(Config), _errpp]),
    ("wn_engine_destroy", None, [_vp]),
    ("wn_engine_set_positions", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_set_masses", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_set_step_sizes", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_init_positions", _i32, [_vp, _u64, _u32, _dbl, _errpp]),
    ("wn_engine_init_masses_from_grad", _i32, [_vp, _dbl, _errpp]),
    ("wn_engine_average_masses", _i32, [_vp, _errpp]),
    ("wn_engine_get_masses", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_adapt_step", _i32, [_vp, _u64, _u32, _errpp]),
    ("wn_engine_adapt_step_with_normals", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_seed", _i32, [_vp, _u64, _u32, _errpp]),
    ("wn_engine_seed_reference_streams", _i32, [_vp, _u64, _errpp]),
    ("wn_engine_set_variates", _i32, [_vp, _dp, _dp, _i32, _errpp]),
    ("wn_engine_warmup_step", _i32, [_vp, _vp, _i64, _errpp]),
    ("wn_engine_freeze", _i32, [_vp, _errpp]),
    ("wn_engine_sample_step", _i32, [_vp, _vp, _i64, _errpp]),
    ("wn_engine_warmup_steps", _i32, [_vp, _i32, _vp, _i64, _i64, _errpp]),
    ("wn_engine_sample_steps", _i32, [_vp, _i32, _vp, _i64, _i64, _errpp]),
    ("wn_engine_synchronize", _i32, [_vp, _errpp]),
    ("wn_engine_check", _i32, [_vp, _errpp]),
    ("wn_engine_get_positions", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_get_inv_mass", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_get_step_sizes", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_get_logp", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_get_min_micro", _i32, [_vp, _i32p, _errpp]),
    ("wn_engine_get_depths", _i32, [_vp, _i32p, _errpp]),
    ("wn_engine_get_grad_evals", _i32, [_vp, _i64p, _errpp]),
    ("wn_engine_get_rng_draws", _i32, [_vp, _i32p, _errpp]),
    ("wn_engine_get_failed_extensions", _i32, [_vp, _i32p, _errpp]),
    ("wn_engine_get_adam", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_get_estimator", _i32, [_vp, _dp, _dp, _dp, _dp, _dp, _errpp]),
    ("wn_engine_total_grad_evals", _i32, [_vp, _i64p, _errpp]),
    ("wn_engine_rhat", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_lp_sums", _i32, [_vp, _dp, _errpp]),
    ("wn_engine_lp_sq_dev", _i32, [_vp, _dbl, _dp, _errpp]),
    ("wn_engine_warmup_spread", _i32, [_vp, _dp, _dp, _errpp]),
    ("wn_engine_warmup_sums", _i32, [_vp, _dp, _dp, _errpp]),
    ("wn_engine_warmup_max_rel", _i32, [_vp, _dbl, _dp, _sz, _dp, _dp, _errpp]),
    ("wn_engine_lanes", _i32, [_vp]),
    ("wn_engine_dim_padded", _i32, [_vp]),
    ("wn_engine_is_streaming", _i32, [_vp]),
    ("wn_engine_workgroups", _i32, [_vp]),
    ("wn_engine_chain_groups", _i32, [_vp]),
    ("wn_engine_held_tiles", _i32, [_vp]),
    ("wn_engine_lds_vectors", _i32, [_vp]),
    ("wn_engine_iteration", _i64, [_vp]),
    ("wn_engine_stream", _vp, [_vp]),
    ("wn_engine_positions_device", _vp, [_vp]),
    ("wn_engine_last_kernel_ms", _i32, [_vp, C.POINTER(C.c_float), _errpp]),
    ("wn_engine_timing_reset", _i32, [_vp, _errpp]),
    ("wn_engine_region_begin", _i32, [_vp, _errpp]),
    ("wn_engine_region_ms", _i32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int), _errpp]),
    ("wn_engine_kernel_times", _i32, [_vp, C.POINTER(C.c_float), _i32, C.POINTER(C.c_int), _errpp]),
    ("wn_engine_set_stream", _i32, [_vp, _vp, _errpp]),
    ("wn_engine_wait_stream", _i32, [_vp, _vp, _errpp]),
    ("wn_engine_release_stream", _i32, [_vp, _vp, _errpp]),
    ("wn_engine_wait_event", _i32, [_vp, _vp, _errpp]),
    ("wn_lanes_for_dim", _i32, [_i32, _i32, _i32]),
    ("wn_lanes_for_model_dim", _i32, [_i32, _i32, _i32, _i32]),
    # posterior summaries (summary.hpp:370-768)
    ("wn_chains_view", _i32, [C.POINTER(_vp), _vp, _sz, _sz, _sz, C.c_int64, C.POINTER(C.c_int64), _i32, _vp, _errpp]),
    ("wn_chains_adopt", _i32, [C.POINTER(_vp), _vp, _sz, _sz, _sz, C.c_int64, C.POINTER(C.c_int64), _i32, _errpp]),
    ("wn_chains_upload", _i32, [C.POINTER(_vp), _dp, _sz, C.POINTER(C.c_int64), _sz, _i32, _errpp]),
    ("wn_chains_destroy", None, [_vp]),
    ("wn_chains_num_chains", _sz, [_vp]),
    ("wn_chains_dims", _sz, [_vp]),
    ("wn_chains_num_draws", _sz, [_vp]),
    ("wn_chains_min_chain_size", _sz, [_vp]),
    ("wn_chains_device_draws", _vp, [_vp]),
    ("wn_chains_device", _i32, [_vp]),
    ("wn_summary_mean", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_sample_variance", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_sample_standard_deviation", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_quantiles", _i32, [_vp, _dp, _sz, _dp, _errpp]),
    ("wn_summary_autocovariance", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_r_hat", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_effective_sample_size", _i32, [_vp, _dp, _errpp]),
    ("wn_summary_monte_carlo_standard_error", _i32, [_vp, _dp, _errpp]),
    ("wn_internal_make_error", _vp, [C.c_char_p, _i32]),
]

_cache = {}


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libwalnuts_hip.so and bind every declared symbol; raises if the library was not built."""
    path = path or os.environ.get("WALNUTS_AMD_LIB") or DEFAULT_LIB
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise WalnutsHipError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C walnuts_amd/csrc). There is no CPU fallback.")
    if "torch" not in sys.modules and not os.environ.get("WALNUTS_AMD_NO_TORCH"):
        # PyTorch-ROCm ships its own copy of the HIP runtime.  A process that loads the system copy first (through
        # this library) and torch's afterwards ends up with two runtimes, and the second finds no device: let
        # torch's load first so that both bind to one.  (Only the load order matters; nothing of torch is used here.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _cache[path] = lib
    return lib


def check(lib: C.CDLL, rc: int, err: C.c_void_p):
    """Map the C error object to Python exceptions as python/src/walnutpie/_ffi.py:170-215 does."""
    if rc == 0:
        return
    msg = lib.walnutpie_get_error_message(err).decode("utf-8", "replace")
    kind = lib.walnutpie_get_error_type(err)
    if err:
        lib.walnutpie_destroy_error(err)
    if kind == 1:
        raise ValueError(msg)
    if kind == 2:
        raise KeyboardInterrupt(msg)
    raise RuntimeError(msg)
