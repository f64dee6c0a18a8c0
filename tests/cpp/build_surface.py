"""Builds tests/cpp/cpp_surface.cpp (the C++ test of include/walnuts_hip.hpp) into a shared library linked against
the library under test.  Used by tests/test_cpp_surface.py (CPU tier: against the emulation) and by
__graft_entry__.build() (GPU tier: against libwalnuts_hip.so, so that nothing is compiled on the GPU box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def surface_library(tag: str) -> str:
    return os.path.join(HERE, f"libcpp_surface_{tag}.so")


def build_surface_library(lib_path: str, tag: str) -> str:
    out = surface_library(tag)
    src = os.path.join(HERE, "cpp_surface.cpp")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("walnuts_hip.hpp", "walnuts_hip.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(f) for f in [src, lib_path] + hdrs):
        return out
    libdir = os.path.dirname(lib_path)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I",
                           os.path.join(ROOT, "include"), src, "-o", out, lib_path, f"-Wl,-rpath,{libdir}", "-pthread"])
    return out
