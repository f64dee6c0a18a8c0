"""Builds tests/cpusim/libwalnuts_sim.so: the product's host + kernel sources compiled by g++ against
the lock-step workgroup emulation in wn_cpusim.h.  TEST INFRASTRUCTURE (see wn_cpusim.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "walnuts_amd", "csrc")
OUT = os.path.join(HERE, "libwalnuts_sim.so")
SOURCES = ["wn_engine.hip", "wn_sample.hip", "wn_summary.hip"] + sorted(
    f for f in os.listdir(CSRC) if f.startswith("wn_kernels_") and f.endswith(".hip"))  # one per device model


def build(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [
        os.path.join(CSRC, "models", f) for f in os.listdir(os.path.join(CSRC, "models"))] + [
        os.path.join(HERE, "wn_cpusim.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(HERE, os.path.basename(s) + ".sim.o")
        objs.append(o)
        procs.append(subprocess.Popen(
            ["g++", "-x", "c++", "-std=c++20", "-O1", "-g", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-pthread", "-DWN_CPU_SIM",
             "-DWN_SIM_GEOMETRIES", "-I", HERE, "-I", CSRC, "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("cpusim compile failed")
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", OUT] + objs)
    return OUT


def build_with_models(model_sources, out_dir: str) -> str:
    """The emulation library plus OUT-OF-TREE device models (what `make MODELS="..."` does for the HIP build): the
    extra wn_kernels_<name>.hip files are compiled against the same headers and linked with the in-tree objects."""
    build()
    objs = [os.path.join(HERE, os.path.basename(os.path.join(CSRC, s)) + ".sim.o") for s in SOURCES]
    for src in model_sources:
        o = os.path.join(out_dir, os.path.basename(src) + ".sim.o")
        subprocess.check_call(
            ["g++", "-x", "c++", "-std=c++20", "-O1", "-g", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-pthread", "-DWN_CPU_SIM",
             "-DWN_SIM_GEOMETRIES", "-I", HERE, "-I", CSRC, "-I", os.path.dirname(src), "-c", src, "-o", o])
        objs.append(o)
    tag = "_".join(os.path.splitext(os.path.basename(m))[0].replace("wn_kernels_", "") for m in model_sources)
    out = os.path.join(out_dir, f"libwalnuts_sim_{tag}.so")   # (a name of its own: loaded libraries are cached by path)
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force=True))
