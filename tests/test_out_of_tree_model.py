"""CPU tier: the documented user path for a device model of one's own (INTEGRATION.md "Adding a device model",
wn_model_api.h) -- a header + a five-line wn_kernels_<name>.hip OUTSIDE the tree, built in with `make MODELS=...`:
the gfx950 object compiles with hipcc, and under the workgroup emulation the model registers, resolves by name and
samples the density it states.  Also: a model that claims an id already taken must not kill the process at load time."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "cpusim"))
import build as simbuild  # noqa: E402
import walnuts_amd as wa  # noqa: E402

MODEL_H = r'''// a user's model: independent normals with per-coordinate means, logp = -0.5 * sum (x_i - mu_i)^2
#pragma once
#include "wn_model_api.h"
namespace user {
struct ShiftedNormal {
  static constexpr bool kUsesParams = true;     // mp = mu
  static constexpr bool kElementwise = true;
  static constexpr bool kGradIsNegTheta = false;
  static constexpr bool kCheapGrad = true;
  __device__ __forceinline__ static double grad_elem(double th, double mu) { return mu - th; }
  struct Aux {};
  template <int EPL, class Cx>
  __device__ __forceinline__ static void eval(Cx& cx, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&mu)[EPL], Aux&, double& acc) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const double d = cx.valid(j) ? th[j] - mu[j] : 0.0;   // (parameter padding is 1.0)
      g[j] = -d;
      acc = Cx::mad(d, d, acc);
    }
  }
  template <int EPL, class Cx>
  __device__ __forceinline__ static void grad(Cx& cx, const double (&th)[EPL], double (&g)[EPL],
                                              const double (&mu)[EPL], Aux&) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) g[j] = cx.valid(j) ? mu[j] - th[j] : 0.0;
  }
  __device__ __forceinline__ static double finish(double sum, const Aux&, int) { return -0.5 * sum; }
};
}  // namespace user
'''
# the same density declared as "not element-wise" and WITHOUT a streaming form: register kernels only
PLAIN_H = MODEL_H.replace("struct ShiftedNormal", "struct PlainNormal").replace(
    "static constexpr bool kElementwise = true;", "static constexpr bool kElementwise = false;").replace(
    "static constexpr bool kCheapGrad = true;", "static constexpr bool kCheapGrad = false;")

MODEL_HIP = '''#include "shifted_normal.h"
#define WN_MODEL_ID %d
#define WN_MODEL_TAG %s
#define WN_MODEL_TYPE user::ShiftedNormal
#include "wn_kernels.inc"
'''


@pytest.fixture(scope="module")
def user_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("user_model")
    (d / "shifted_normal.h").write_text(MODEL_H)
    (d / "wn_kernels_shifted_normal.hip").write_text(MODEL_HIP % (7, "shifted_normal"))
    (d / "wn_kernels_clash.hip").write_text(MODEL_HIP % (0, "clash"))   # id 0 is std_normal's
    (d / "plain_normal.h").write_text(PLAIN_H)
    (d / "wn_kernels_plain_normal.hip").write_text(
        (MODEL_HIP % (9, "plain_normal")).replace("shifted_normal.h", "plain_normal.h").replace("ShiftedNormal",
                                                                                                "PlainNormal"))
    return d


@pytest.mark.timeout(900)
def test_out_of_tree_model_object_compiles_for_gfx950(user_dir):
    csrc = os.path.join(ROOT, "walnuts_amd", "csrc")
    src = str(user_dir / "wn_kernels_shifted_normal.hip")
    obj = str(user_dir / "obj" / "wn_kernels_shifted_normal.o")
    subprocess.check_call(["make", "-C", csrc, "-s", f"MODELS={src}", f"OBJDIR={user_dir / 'obj'}",
                           f"EXTRA=-DWN_FAST_BUILD -I{user_dir}", obj])
    assert os.path.getsize(obj) > 100000   # a fat object with the model's gfx950 kernels in it


@pytest.mark.timeout(900)
def test_out_of_tree_model_registers_and_samples_its_density(user_dir):
    lib_path = simbuild.build_with_models([str(user_dir / "wn_kernels_shifted_normal.hip")], str(user_dir))
    assert wa.model_id("shifted_normal", lib_path) == 7
    assert wa.model_id("rw1", lib_path) == 3               # the in-tree models are still there
    D, Cn = 10, 3
    mu = np.linspace(-2.0, 2.0, D)
    for fma in (0, 1):
        e = wa.DeviceEngine(7, D, Cn, wa.default_config(lib_path, fused_multiply_add=fma), params=mu, lib_path=lib_path)
        e.set_positions(np.random.default_rng(1).normal(size=(Cn, D)))
        e.set_step_sizes(0.4)
        e.seed_chains(3, 0)
        for _ in range(3):
            e.warmup_step()
        e.freeze()
        for _ in range(3):
            e.sample_step()
        e.synchronize()
        x = e.positions()
        assert np.allclose(e.logp(), -0.5 * np.sum((x - mu) ** 2, axis=1), rtol=1e-13, atol=0)
        assert np.all(e.depths() >= 1) and np.all(e.grad_evals() > 6)
    with pytest.raises(ValueError, match="parameter vector"):       # kUsesParams: the engine insists on mu
        wa.DeviceEngine(7, D, Cn, wa.default_config(lib_path), lib_path=lib_path)


@pytest.mark.timeout(900)
def test_model_without_a_streaming_form_is_limited_to_the_register_kernels(user_dir):
    lib_path = simbuild.build_with_models([str(user_dir / "wn_kernels_plain_normal.hip")], str(user_dir))
    mu = np.zeros(12)
    e = wa.DeviceEngine(9, 12, 2, wa.default_config(lib_path), params=mu, lib_path=lib_path)   # register kernels: fine
    e.seed_chains(1, 0)
    e.warmup_step()
    e.synchronize()
    assert np.all(e.depths() >= 1)
    with pytest.raises(ValueError, match="streaming"):     # vectors in HBM need kElementwise or the streaming form
        wa.DeviceEngine(9, 12, 2, wa.default_config(lib_path, elems_per_lane=-1), params=mu, lib_path=lib_path)


@pytest.mark.timeout(900)
def test_model_id_clash_is_a_config_error_not_a_crash_at_load_time(user_dir):
    lib_path = simbuild.build_with_models([str(user_dir / "wn_kernels_clash.hip")], str(user_dir))
    lib = wa.load_library(lib_path)                        # loading must survive the clash
    with pytest.raises(ValueError, match="already taken by 'std_normal'"):
        wa.DeviceEngine(wa.MODEL_STD_NORMAL, 4, 2, wa.default_config(lib_path), lib_path=lib_path)
