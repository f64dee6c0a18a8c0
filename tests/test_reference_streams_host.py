"""CPU tier: the two-pass generator of the reference's host-side initial streams (walnuts_amd/csrc/wn_refstream.h:
sequential engine + rejection pass, parallel sqrt/log pass) gives exactly the numbers of the plain
std::normal_distribution loops it replaces (2.6 s -> ~0.3 s per stream at 65 536 chains x 1 024 parameters)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "cpusim"))
import build as simbuild  # noqa: E402
import walnuts_amd as wa  # noqa: E402


@pytest.fixture(scope="module")
def plain(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("refn") / "libplain_normals.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "cpp", "ref_normals.cpp")])
    lib = C.CDLL(so)
    lib.plain_reference_normals.argtypes = [C.c_uint, C.c_uint, C.c_size_t, C.c_size_t, C.c_int, C.c_double,
                                            C.POINTER(C.c_double)]
    lib.plain_reference_normals.restype = None
    return lib


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chains,count,fresh,scale", [
    (1, 1, 0, 1.0), (5, 1, 0, 2.0), (5, 1, 1, 1.0),          # one value per chain: the saved variate crosses / is dropped
    (7, 3, 0, 2.0), (7, 3, 1, 1.0), (4, 100, 0, 2.0), (4, 100, 1, 1.0),
    (3, 200001, 0, 0.5), (3, 200001, 1, 1.0),                # several worker chunks, odd length
    (300, 1025, 1, 1.0), (300, 1024, 0, 2.0),
])
def test_two_pass_host_streams_equal_the_plain_distribution_loops(plain, chains, count, fresh, scale):
    lib = wa.load_library(simbuild.build())
    a, b = np.full(chains * count, np.nan), np.full(chains * count, np.nan)
    dp = C.POINTER(C.c_double)
    for seed, stream in ((11, 1), (48, 2)):
        lib.wn_internal_reference_normals(seed, stream, chains, count, fresh, scale, a.ctypes.data_as(dp))
        plain.plain_reference_normals(seed, stream, chains, count, fresh, scale, b.ctypes.data_as(dp))
        assert np.array_equal(a, b), (chains, count, fresh, np.flatnonzero(a != b)[:5])
