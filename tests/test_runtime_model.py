"""Device models compiled at RUN time (walnuts_amd/models.py): the model's translation unit is compiled against the
installed headers into a shared object of its own -- only the launch geometry its engine will use --, loaded, and
registered in the library through wn_plugin_register_model; libwalnuts_hip.so is not rebuilt.  The device counterpart
of the reference taking any host callable as a model (pyfunc.py:45-286, walnutpy.cpp:131-132).

CPU tier: the whole route under the workgroup emulation (g++ in place of hipcc).  GPU tier (-m gpu): the real thing --
hipcc on the GPU box, a fifth model beside the four built-in ones, bit for bit against the oracle's model of the same
density."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "cpusim"))
import parity  # noqa: E402
import walnuts_amd as wa  # noqa: E402
from walnuts_amd import models  # noqa: E402

HEADER = os.path.join(HERE, "helpers", "user_diag_model.h")


def check_against_oracle(model_id, lib_path, cases):
    import wno
    parity.MODELS["user_diag"] = (model_id, wno.MODEL_DIAG_NORMAL)
    for D, C, geometry, kw in cases:
        parity.run_case("user_diag", D, C, warmup=4, sampling=4, lib_path=lib_path, geometry=geometry, **kw)


@pytest.mark.timeout(900)
def test_runtime_model_under_the_emulation(oracle, tmp_path):
    import build as simbuild
    sim = simbuild.build()
    gxx = ["g++", "-x", "c++", "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-pthread",
           "-DWN_CPU_SIM", "-I", os.path.join(HERE, "cpusim")]
    with pytest.raises(ValueError, match="no device model named"):
        wa.model_id("user_diag_130", sim)
    assert models.geometry_for(130, elems_per_lane=4, lib_path=sim) == (1, 4, False)
    so = models.build_device_model(HEADER, "user::MyDiagNormal", "user_diag_130", 11, 130, out_dir=str(tmp_path),
                                   elems_per_lane=4, lib_path=sim, compiler=gxx)
    mid = models.load_device_model(so, "user_diag_130", lib_path=sim)
    assert mid == 11 == wa.model_id("user_diag_130", sim) and wa.model_id("rw1", sim) == 3
    check_against_oracle(mid, sim, [(130, 2, (1, 4), dict(fused_multiply_add=1)),
                                    (130, 2, (1, 4), dict(fused_multiply_add=0))])
    # the object holds ONE geometry: an engine that asks for another is told so
    with pytest.raises(ValueError, match="no kernel for this geometry"):
        e = wa.DeviceEngine(mid, 130, 2, wa.default_config(sim, waves_per_chain=2, elems_per_lane=2),
                            params=np.ones(130), lib_path=sim)
        e.warmup_step()
    # a streaming build (vectors in HBM) of the same model under another name and id
    so2 = models.build_device_model(HEADER, "user::MyDiagNormal", "user_diag_stream", 12, 300, out_dir=str(tmp_path),
                                    waves_per_chain=1, elems_per_lane=-1, lib_path=sim, compiler=gxx)
    mid2 = models.load_device_model(so2, "user_diag_stream", lib_path=sim)
    check_against_oracle(mid2, sim, [(300, 2, (1, -1), {})])
    # an id that is taken: a ValueError naming the holder, and the library keeps working
    so3 = models.build_device_model(HEADER, "user::MyDiagNormal", "user_clash", 11, 130, out_dir=str(tmp_path),
                                    elems_per_lane=4, lib_path=sim, compiler=gxx)
    with pytest.raises(ValueError, match="already taken by 'user_diag_130'"):
        models.load_device_model(so3, "user_clash", lib_path=sim)
    check_against_oracle(mid, sim, [(130, 2, (1, 4), {})])


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_fifth_model_without_rebuilding_the_library(gpu, oracle, tmp_path):
    """hipcc on the GPU box: the user's header -> its own shared object (headline geometry (1, 16) and the small (1, 2))
    -> registered in the loaded libwalnuts_hip.so -> chains bit-identical to the oracle's diagonal normal."""
    lib = wa.load_library()
    before = os.path.getmtime(lib._name)
    so = models.build_device_model(HEADER, "user::MyDiagNormal", "user_diag_1024", 20, 1024, out_dir=str(tmp_path))
    mid = models.load_device_model(so, "user_diag_1024")
    assert mid == 20 and wa.model_id("user_diag_1024") == 20 and wa.model_id("diag_normal") == 1
    check_against_oracle(mid, None, [(1024, 48, None, dict(fused_multiply_add=1)),
                                     (1000, 16, None, dict(fused_multiply_add=0))])
    so2 = models.build_device_model(HEADER, "user::MyDiagNormal", "user_diag_100", 21, 100, out_dir=str(tmp_path))
    mid2 = models.load_device_model(so2, "user_diag_100")
    check_against_oracle(mid2, None, [(100, 64, None, {})])
    # 6 000 parameters: the register kernels' (16, 8) AND the streaming kernels that hold the moving end in registers
    # are built (models.geometry_defines); the engine picks the latter for an element-wise gradient
    so3 = models.build_device_model(HEADER, "user::MyDiagNormal", "user_diag_6000", 22, 6000, out_dir=str(tmp_path))
    mid3 = models.load_device_model(so3, "user_diag_6000")
    e = wa.DeviceEngine(mid3, 6000, 4, wa.default_config(), params=np.linspace(0.5, 2.0, 6000))
    assert e.streaming and e.held_tiles == 16 and e.lanes == 512
    del e
    check_against_oracle(mid3, None, [(6000, 6, None, {}), (6000, 4, (16, 8), {})])
    assert os.path.getmtime(lib._name) == before   # the library itself was not touched
