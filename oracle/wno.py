"""ctypes binding of the CPU oracle (oracle/libwn_oracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg and by
nothing under ``walnuts_amd/``.  Build the library with ``make -C oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwn_oracle.so")

MODEL_STD_NORMAL, MODEL_DIAG_NORMAL, MODEL_FUNNEL, MODEL_RW1 = 0, 1, 2, 3
MATH_LIBM, MATH_PORTABLE = 0, 1
RNG_STD_MT64, RNG_STD_MT32, RNG_PHILOX = 0, 1, 2
STREAM_MOMENTUM, STREAM_TREE, STREAM_INIT_POS, STREAM_INIT_STEP = 0, 1, 2, 3
TRACE_FIELDS = 9


class Config(C.Structure):
    _fields_ = [
        ("max_trajectory_doublings", C.c_int32),
        ("max_step_halvings", C.c_int32),
        ("min_micro_steps", C.c_int32),
        ("max_hamiltonian_error", C.c_double),
        ("mass_init_count", C.c_double),
        ("max_macro_steps_target", C.c_double),
        ("step_accept_rate_target", C.c_double),
        ("step_learning_rate", C.c_double),
        ("step_gradient_decay", C.c_double),
        ("step_sq_gradient_decay", C.c_double),
        ("step_stabilization", C.c_double),
        ("step_learn_rate_decay", C.c_double),
        ("math_mode", C.c_int32),
        ("reduce_lanes", C.c_int32),
        ("rng_mode", C.c_int32),
        ("fma", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, sz, i32, i64, u32, u64, dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    L.wno_default_config.argtypes = [C.POINTER(Config)]
    L.wno_create.restype = vp
    L.wno_create.argtypes = [i32, i32, _dp, sz, C.POINTER(Config)]
    L.wno_destroy.argtypes = [vp]
    L.wno_set_rng_mode.argtypes = [vp, i32]
    for name in ("wno_set_positions", "wno_set_masses", "wno_set_step_sizes", "wno_get_positions",
                 "wno_get_grad_select", "wno_get_logp", "wno_get_step_sizes", "wno_get_inv_mass", "wno_get_masses", "wno_get_adam"):
        getattr(L, name).argtypes = [vp, _dp]
    L.wno_init_positions.argtypes = [vp, u64, u64, dbl]
    L.wno_init_masses_from_grad.argtypes = [vp, dbl, i32]
    L.wno_adapt_step.argtypes = [vp, u64, u64]
    L.wno_seed_chains.argtypes = [vp, u64, u32]
    L.wno_set_variates.argtypes = [vp, _dp, _dp, sz]
    L.wno_warmup_step.argtypes = [vp, i32]
    L.wno_freeze.argtypes = [vp]
    L.wno_sample_step.argtypes = [vp, i32]
    L.wno_get_min_micro.argtypes = [vp, C.POINTER(C.c_int64)]
    L.wno_get_depths.argtypes = [vp, C.POINTER(C.c_int32)]
    L.wno_get_grad_evals.argtypes = [vp, C.POINTER(C.c_int64)]
    L.wno_get_rng_draws.argtypes = [vp, C.POINTER(C.c_int64)]
    L.wno_get_estimator.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
    L.wno_iteration.restype = i64
    L.wno_iteration.argtypes = [vp]
    L.wno_rhat.restype = dbl
    L.wno_rhat.argtypes = [vp]
    L.wno_warmup_spread.argtypes = [vp, _dp, _dp]
    L.wno_enable_trace.argtypes = [vp, i32]
    L.wno_get_trace.restype = sz
    L.wno_get_trace.argtypes = [vp, sz, _dp, sz]
    L.wno_logp_momentum.restype = dbl
    L.wno_logp_momentum.argtypes = [sz, _dp, _dp, i32]
    L.wno_reduce_sum.restype = dbl
    L.wno_reduce_sum.argtypes = [sz, _dp, i32]
    L.wno_log_sum_exp.restype = dbl
    L.wno_log_sum_exp.argtypes = [dbl, dbl, i32]
    L.wno_model_logp_grad.argtypes = [i32, i32, _dp, _dp, _dp, _dp, i32, i32]
    L.wno_leapfrog_error.restype = dbl
    L.wno_leapfrog_error.argtypes = [i32, i32, _dp, _dp, _dp, _dp, dbl, i32, i32]
    L.wno_uturn.restype = i32
    L.wno_uturn.argtypes = [sz, i32, _dp, _dp, _dp, _dp, _dp, i32]
    L.wno_adam_run.argtypes = [dbl, dbl, dbl, dbl, dbl, dbl, dbl, _dp, sz, _dp, i32]
    L.wno_online_moments_observe.argtypes = [sz, dbl, _dp, _dp, _dp, _dp]
    L.wno_macro_step.restype = i32
    L.wno_macro_step.argtypes = [i32, i32, _dp, C.POINTER(Config), i32, dbl, i32, _dp, _dp, _dp, _dp, dbl, _dp, _dp,
                                 _dp, _dp, _dp, _dp, C.POINTER(C.c_int64)]
    L.wno_stream_uniform.restype = dbl
    L.wno_stream_uniform.argtypes = [u64, u32, u32, u32, u32]
    L.wno_stream_normals.argtypes = [u64, u32, u32, u32, sz, _dp]
    L.wno_math_exp.restype = dbl
    L.wno_math_exp.argtypes = [dbl]
    L.wno_math_log.restype = dbl
    L.wno_math_log.argtypes = [dbl]
    L.wno_math_exp_weight.restype = dbl
    L.wno_math_exp_weight.argtypes = [dbl]
    L.wno_set_sampler_state.argtypes = [vp, _dp, _dp, C.POINTER(C.c_int64)]
    L.wno_set_transition_index.argtypes = [vp, u32]
    L.wno_set_adapt_state.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int64), u64]
    L.wno_l2_rel_diff.restype = dbl
    L.wno_l2_rel_diff.argtypes = [sz, _dp, _dp]
    L.wno_div_shared.restype = dbl
    L.wno_div_shared.argtypes = [dbl, dbl]
    L.wno_variance.restype = dbl
    L.wno_variance.argtypes = [sz, _dp]
    L.wno_set_tie_tolerance.argtypes = [vp, dbl]
    L.wno_get_near_ties.argtypes = [vp, C.POINTER(C.c_int64), i32]
    L.wno_get_weight_rebases.restype = C.c_int64
    L.wno_get_weight_rebases.argtypes = [vp]
    _lib = L
    return L


def default_config(**overrides) -> Config:
    cfg = Config()
    lib().wno_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Engine:
    """A batch of C independent chains stepped by the CPU restatement."""

    def __init__(self, model: int, dim: int, num_chains: int, cfg: Optional[Config] = None,
                 params: Optional[np.ndarray] = None):
        self.L = lib()
        self.cfg = cfg if cfg is not None else default_config()
        self.C, self.D = int(num_chains), int(dim)
        self._params = None if params is None else _f64(params)
        self.h = self.L.wno_create(model, dim, _p(self._params), num_chains, C.byref(self.cfg))

    def close(self):
        if self.h:
            self.L.wno_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- init
    def set_rng_mode(self, mode: int):
        self.L.wno_set_rng_mode(self.h, mode)

    def set_positions(self, pos):
        a = _f64(pos).reshape(self.C, self.D)
        self.L.wno_set_positions(self.h, _p(a))

    def set_masses(self, mass):
        a = _f64(mass).reshape(self.C, self.D)
        self.L.wno_set_masses(self.h, _p(a))

    def set_step_sizes(self, steps):
        a = _f64(np.broadcast_to(np.asarray(steps, dtype=np.float64), (self.C,)))
        self.L.wno_set_step_sizes(self.h, _p(a))

    def init_positions(self, s0: int, s1: int, scale: float):
        self.L.wno_init_positions(self.h, s0, s1, scale)

    def init_masses_from_grad(self, smoothing: float, average: bool = False):
        self.L.wno_init_masses_from_grad(self.h, smoothing, int(average))

    def adapt_step(self, s0: int, s1: int):
        self.L.wno_adapt_step(self.h, s0, s1)

    def seed_chains(self, seed: int, chain_offset: int = 0):
        self.L.wno_seed_chains(self.h, seed, chain_offset)

    def set_variates(self, normals, uniforms):
        z = _f64(normals).reshape(self.C, self.D)
        u = _f64(uniforms).reshape(self.C, -1)
        self.L.wno_set_variates(self.h, _p(z), _p(u), u.shape[1])

    # --- stepping
    def warmup_step(self, threads: int = 1):
        self.L.wno_warmup_step(self.h, threads)

    def freeze(self):
        self.L.wno_freeze(self.h)

    def sample_step(self, threads: int = 1):
        self.L.wno_sample_step(self.h, threads)

    def set_sampler_state(self, inv_mass, step, min_micro):
        """Frozen sampler parameters handed in as they are (inverse mass [C, D], step [C], min micro steps [C])."""
        im = np.ascontiguousarray(inv_mass, dtype=np.float64).reshape(self.C, self.D)
        st = np.ascontiguousarray(np.broadcast_to(step, (self.C,)), dtype=np.float64)
        mm = np.ascontiguousarray(np.broadcast_to(min_micro, (self.C,)), dtype=np.int64)
        self.L.wno_set_sampler_state(self.h, im.ctypes.data_as(_dp), st.ctypes.data_as(_dp),
                                     mm.ctypes.data_as(C.POINTER(C.c_int64)))

    def set_adapt_state(self, adam, estimator, min_micro, iteration: int):
        """Adam [C, 6], the estimator dict of Engine.estimator(), min micro steps [C], warmup transitions done."""
        a = _f64(adam).reshape(self.C, 6)
        planes = [_f64(estimator[k]).reshape(self.C, self.D) for k in ("draw_mean", "draw_ssd", "score_mean", "score_ssd")]
        w = _f64(estimator["weights"]).reshape(self.C, 2)
        mm = np.ascontiguousarray(np.asarray(min_micro, dtype=np.int64))
        self.L.wno_set_adapt_state(self.h, _p(a), *(_p(x) for x in planes), _p(w),
                                   mm.ctypes.data_as(C.POINTER(C.c_int64)), int(iteration))

    def set_transition_index(self, t: int):
        self.L.wno_set_transition_index(self.h, int(t))

    def set_tie_tolerance(self, tol: float):
        self.L.wno_set_tie_tolerance(self.h, float(tol))

    def near_ties(self, reset: bool = True):
        """-> dict kind -> (decisions within the tolerance of their threshold, decisions taken)."""
        out = (C.c_int64 * 6)()
        self.L.wno_get_near_ties(self.h, out, 1 if reset else 0)
        return {k: (int(out[i]), int(out[3 + i])) for i, k in enumerate(("energy_error", "uturn_sign", "acceptance"))}

    def weight_rebases(self) -> int:
        """Device arithmetic: how often a transition moved the reference energy of its span weights (all chains)."""
        return int(self.L.wno_get_weight_rebases(self.h))

    # --- state
    def _vec(self, fn, shape, dtype=np.float64, ptr=_dp):
        out = np.empty(shape, dtype=dtype)
        fn(self.h, out.ctypes.data_as(ptr))
        return out

    def positions(self):
        return self._vec(self.L.wno_get_positions, (self.C, self.D))

    def grad_select(self):
        return self._vec(self.L.wno_get_grad_select, (self.C, self.D))

    def logp(self):
        return self._vec(self.L.wno_get_logp, (self.C,))

    def step_sizes(self):
        return self._vec(self.L.wno_get_step_sizes, (self.C,))

    def inv_mass(self):
        return self._vec(self.L.wno_get_inv_mass, (self.C, self.D))

    def masses(self):
        return self._vec(self.L.wno_get_masses, (self.C, self.D))

    def adam(self):
        return self._vec(self.L.wno_get_adam, (self.C, 6))

    def min_micro(self):
        return self._vec(self.L.wno_get_min_micro, (self.C,), np.int64, C.POINTER(C.c_int64))

    def depths(self):
        return self._vec(self.L.wno_get_depths, (self.C,), np.int32, C.POINTER(C.c_int32))

    def grad_evals(self):
        return self._vec(self.L.wno_get_grad_evals, (self.C,), np.int64, C.POINTER(C.c_int64))

    def rng_draws(self):
        return self._vec(self.L.wno_get_rng_draws, (self.C,), np.int64, C.POINTER(C.c_int64))

    def estimator(self):
        dm, ds, sm, ss = (np.empty((self.C, self.D)) for _ in range(4))
        w = np.empty((self.C, 2))
        self.L.wno_get_estimator(self.h, _p(dm), _p(ds), _p(sm), _p(ss), _p(w))
        return dict(draw_mean=dm, draw_ssd=ds, score_mean=sm, score_ssd=ss, weights=w)

    def rhat(self) -> float:
        return self.L.wno_rhat(self.h)

    def warmup_spread(self):
        a, b = C.c_double(), C.c_double()
        self.L.wno_warmup_spread(self.h, C.cast(C.byref(a), _dp), C.cast(C.byref(b), _dp))
        return a.value, b.value

    def enable_trace(self, on: bool = True):
        self.L.wno_enable_trace(self.h, int(on))

    def trace(self, chain: int, max_rec: int = 4096) -> np.ndarray:
        buf = np.empty((max_rec, TRACE_FIELDS))
        n = self.L.wno_get_trace(self.h, chain, _p(buf), max_rec)
        return buf[: min(n, max_rec)].copy()


# ---- function-level helpers -------------------------------------------------
def logp_momentum(rho, inv_mass, reduce_lanes: int = 0) -> float:
    r, m = _f64(rho), _f64(inv_mass)
    return lib().wno_logp_momentum(r.size, _p(r), _p(m), reduce_lanes)


REDUCE_EIGEN_SSE2 = -2   # reduce_lanes: Eigen 3.4's vectorised redux order, 2-lane packets (wn_oracle.cpp: Reducer)


def reduce_sum(x, reduce_lanes: int = 0) -> float:
    v = _f64(x).reshape(-1)
    return lib().wno_reduce_sum(v.size, _p(v), reduce_lanes)


def l2_rel_diff(a, b) -> float:
    a, b = _f64(a), _f64(b)
    return lib().wno_l2_rel_diff(a.size, _p(a), _p(b))


def div_shared(a: float, w: float) -> float:
    return lib().wno_div_shared(a, w)


def variance(xs) -> float:
    xs = _f64(xs)
    return lib().wno_variance(xs.size, _p(xs))


def log_sum_exp(a: float, b: float, math_mode: int = MATH_LIBM) -> float:
    return lib().wno_log_sum_exp(a, b, math_mode)


def model_logp_grad(model, x, params=None, math_mode=MATH_LIBM, reduce_lanes=0):
    x = _f64(x)
    p = None if params is None else _f64(params)
    g = np.empty_like(x)
    lp = C.c_double()
    lib().wno_model_logp_grad(model, x.size, _p(p), _p(x), C.cast(C.byref(lp), _dp), _p(g), math_mode, reduce_lanes)
    return lp.value, g


def leapfrog_error(model, theta, rho, inv_m, step, params=None, math_mode=MATH_LIBM, reduce_lanes=0) -> float:
    t, r, m = _f64(theta), _f64(rho), _f64(inv_m)
    p = None if params is None else _f64(params)
    return lib().wno_leapfrog_error(model, t.size, _p(p), _p(t), _p(r), _p(m), step, math_mode, reduce_lanes)


def uturn(forward, th_in, rho_in, th_out, rho_out, inv_mass, reduce_lanes=0) -> bool:
    a, b, c, d, m = map(_f64, (th_in, rho_in, th_out, rho_out, inv_mass))
    return bool(lib().wno_uturn(a.size, int(forward), _p(a), _p(b), _p(c), _p(d), _p(m), reduce_lanes))


def adam_run(alphas, step_init=1.0, target=0.8, lr=0.05, b1=0.8, b2=0.9, eps=1e-4, decay=0.5, math_mode=MATH_LIBM):
    a = _f64(alphas)
    out = np.empty_like(a)
    lib().wno_adam_run(step_init, target, lr, b1, b2, eps, decay, _p(a), a.size, _p(out), math_mode)
    return out


def online_moments_observe(discount, weight, mean, ssd, y):
    mean, ssd, y = _f64(mean).copy(), _f64(ssd).copy(), _f64(y)
    w = C.c_double(weight)
    lib().wno_online_moments_observe(mean.size, discount, C.cast(C.byref(w), _dp), _p(mean), _p(ssd), _p(y))
    return w.value, mean, ssd


def macro_step(model, cfg: Config, forward, step, min_micro, inv_mass, theta, rho, grad, logp_joint, params=None):
    im, t, r, g = map(_f64, (inv_mass, theta, rho, grad))
    p = None if params is None else _f64(params)
    to, ro, go = np.empty_like(t), np.empty_like(t), np.empty_like(t)
    lp, lj, al = C.c_double(), C.c_double(), C.c_double()
    ne = C.c_int64()
    ok = lib().wno_macro_step(model, t.size, _p(p), C.byref(cfg), int(forward), step, min_micro, _p(im), _p(t), _p(r),
                              _p(g), logp_joint, _p(to), _p(ro), _p(go), C.cast(C.byref(lp), _dp),
                              C.cast(C.byref(lj), _dp), C.cast(C.byref(al), _dp), C.byref(ne))
    return dict(ok=bool(ok), theta=to, rho=ro, grad=go, logp_pos=lp.value, logp_joint=lj.value, alpha=al.value,
                grad_evals=ne.value)


def stream_uniform(seed, chain, transition, stream, index) -> float:
    return lib().wno_stream_uniform(seed, chain, transition, stream, index)


def stream_normals(seed, chain, transition, stream, n) -> np.ndarray:
    out = np.empty(n)
    lib().wno_stream_normals(seed, chain, transition, stream, n, _p(out))
    return out


def adam_ref_lib():
    """The REAL reference Adam (oracle/_ref/libadam_ref.so) or None when it was never built."""
    path = os.path.join(_HERE, "_ref", "libadam_ref.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.adam_ref_run.argtypes = [C.c_double] * 7 + [_dp, C.c_size_t, _dp]
    return L
