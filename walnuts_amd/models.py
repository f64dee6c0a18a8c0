"""Device models compiled at run time -- the device counterpart of handing the reference a host callable.

The reference takes any host function as a model: a ``LOGP_CFUNC`` pointer, a numba cfunc or a ctypes trampoline
(python/src/walnutpie/pyfunc.py:45-286, :216; walnutpy.cpp:131-132).  A host function cannot be called from a
GPU-resident trajectory, so here a model is a header of static device functions (``walnuts_amd/csrc/wn_model_api.h``)
that is compiled INTO the transition kernels.  ``build_device_model`` does that at call time, without touching
``libwalnuts_hip.so``: it writes the model's five-line translation unit, compiles it against the installed headers
(``walnuts_amd/csrc``) with one ``hipcc -shared`` -- instantiating only the launch geometries the engine may use
for ``num_params`` parameters (``wn_geometry_candidates``: one, or up to three where the choice depends on the model's
traits), which takes seconds where the whole table takes minutes -- and
``load_device_model`` loads the result; the shared object's static initialiser enters the model into the library's
registry (``wn_plugin_register_model``).  From then on ``model_id(name)`` resolves it and every entry point takes it.
"""
import ctypes as C
import os
import subprocess
import tempfile
from typing import Optional, Sequence

from . import _ffi

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def hipcc_flags(lib_path: Optional[str] = None) -> Sequence[str]:
    """The code-generation flags the library itself was built with (csrc/Makefile CODEGEN_FLAGS, kept in the library:
    wn_build_flags): same arithmetic (-ffp-contract=off), same code -- read from the library, not repeated here."""
    flags = _ffi.load_library(lib_path).wn_build_flags().decode().split()
    if not flags:
        raise _ffi.WalnutsHipError("this libwalnuts_hip.so does not state its build flags (built without csrc/Makefile?)")
    return flags


def check_compiler(lib_path: Optional[str] = None, hipcc: str = "hipcc") -> None:
    """A run-time model and the library must come from the same compiler (the code-generation flags include an
    LLVM-internal option, and the kernels share inlined headers): compare `hipcc --version`'s "HIP version:" with the
    one the library recorded at build time (wn_build_compiler)."""
    built = _ffi.load_library(lib_path).wn_build_compiler().decode()
    if not built:
        return
    out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    here = next((line.split(":", 1)[1].strip() for line in out.splitlines() if line.startswith("HIP version:")), "")
    if here != built:
        raise _ffi.WalnutsHipError(f"libwalnuts_hip.so was built with HIP {built}, this hipcc is HIP {here or '?'}: "
                                   "rebuild the library (make -C walnuts_amd/csrc) or use its compiler")


_loaded = {}   # path -> CDLL (kept alive: the registry holds pointers into the object)


def translation_unit(header: str, type_name: str, tag: str, model_id: int) -> str:
    """The five lines that enter a model into the registry (wn_model_api.h, "Registration")."""
    return (f'#include "{os.path.basename(header)}"\n#define WN_MODEL_ID {int(model_id)}\n#define WN_MODEL_TAG {tag}\n'
            f"#define WN_MODEL_TYPE {type_name}\n" '#include "wn_kernels.inc"\n')


def geometry_for(num_params: int, *, model: Optional[int] = None, waves_per_chain: int = 0, elems_per_lane: int = 0,
                 preferred_elems_per_lane: int = 0, lib_path: Optional[str] = None):
    """-> (waves per chain, elements per lane, streaming?) of the engine's kernel for ``num_params`` parameters.

    ``model``: the id of a REGISTERED model -- the answer is then the one geometry ``wn_engine_create`` picks for it
    (``wn_geometry_for_model``).  Without it: the choice for a model that has no held streaming kernels
    (``wn_geometry_for``); for 4 097-16 384 parameters a model with such kernels runs them instead --
    :func:`geometry_candidates` lists every possibility."""
    lib = _ffi.load_library(lib_path)
    nw, epl, mem, err = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
    if model is not None:
        rc = lib.wn_geometry_for_model(int(model), num_params, waves_per_chain, elems_per_lane, C.byref(nw), C.byref(epl),
                                       C.byref(mem), C.byref(err))
    else:
        rc = lib.wn_geometry_for(num_params, waves_per_chain, elems_per_lane, preferred_elems_per_lane,
                                 C.byref(nw), C.byref(epl), C.byref(mem), C.byref(err))
    _ffi.check(lib, rc, err)
    return nw.value, epl.value, bool(mem.value)


def geometry_candidates(num_params: int, *, waves_per_chain: int = 0, elems_per_lane: int = 0,
                        preferred_elems_per_lane: int = 0, lib_path: Optional[str] = None):
    """Every (waves per chain, elements per lane, streaming?) the engine may pick for these requests, over all traits a
    model can have (``wn_geometry_candidates``): what a model compiled at run time has to instantiate."""
    lib = _ffi.load_library(lib_path)
    out, n, err = (C.c_int * 9)(), C.c_int(), C.c_void_p()
    _ffi.check(lib, lib.wn_geometry_candidates(num_params, waves_per_chain, elems_per_lane, preferred_elems_per_lane,
                                               out, 3, C.byref(n), C.byref(err)), err)
    return [(out[3 * i], out[3 * i + 1], bool(out[3 * i + 2])) for i in range(n.value)]


def geometry_defines(candidates) -> Sequence[str]:
    """-DWN_ONLY_* switches (wn_launch.h) that instantiate exactly the candidate geometries: at most one register
    geometry and two streaming ones -- the library decides (geometry_candidates), nothing about the rule lives here."""
    chip = [(nw, epl) for nw, epl, mem in candidates if not mem]
    mem = [nw for nw, epl, m in candidates if m]
    if len(chip) > 1 or len(mem) > 2:
        raise _ffi.WalnutsHipError(f"more candidate geometries than a model object holds: {candidates}")
    flags = []
    if chip:
        flags += [f"-DWN_ONLY_NW={chip[0][0]}", f"-DWN_ONLY_EPL={chip[0][1]}"]
    if mem:
        flags.append(f"-DWN_ONLY_MEM_NW={mem[0]}")
    if len(mem) > 1:
        flags.append(f"-DWN_ONLY_MEM_NW_ALSO={mem[1]}")
    return flags


def build_device_model(header: str, type_name: str, tag: str, model_id: int, num_params: int, *,
                       out_dir: Optional[str] = None, waves_per_chain: int = 0, elems_per_lane: int = 0,
                       preferred_elems_per_lane: int = 0, lib_path: Optional[str] = None,
                       compiler: Optional[Sequence[str]] = None, extra_flags: Sequence[str] = ()) -> str:
    """Compile the model in ``header`` (a struct ``type_name`` implementing wn_model_api.h) for engines of
    ``num_params`` parameters -> path of its shared object (pass it to :func:`load_device_model`).

    ``tag`` is the name ``model_id(tag)`` will find, ``model_id`` its registry slot (4..63; 0-3 are the built-in
    models).  ``waves_per_chain`` / ``elems_per_lane``: the same requests a ``wn_config`` can make (0 = the engine's
    choice); ``preferred_elems_per_lane``: the model's ``kPreferredElemsPerLane`` if it states one.  An engine created
    with other requests than the ones given here finds no kernel and says so.  ``compiler``: the command in front of
    the flags (default ``["hipcc"] + hipcc_flags()``; the CPU test tier passes g++ with the emulation's flags)."""
    header = os.path.abspath(header)
    lib_file = os.path.abspath(lib_path or os.environ.get("WALNUTS_AMD_LIB") or _ffi.DEFAULT_LIB)
    candidates = geometry_candidates(num_params, waves_per_chain=waves_per_chain, elems_per_lane=elems_per_lane,
                                     preferred_elems_per_lane=preferred_elems_per_lane, lib_path=lib_path)
    out_dir = out_dir or tempfile.mkdtemp(prefix="wn_model_")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(out_dir, f"wn_kernels_{tag}.hip")
    with open(src, "w") as f:
        f.write(translation_unit(header, type_name, tag, model_id))
    geo = "_".join(f"mem{nw}" if mem else f"{nw}x{epl}" for nw, epl, mem in candidates)
    out = os.path.join(out_dir, f"libwn_model_{tag}_{geo}.so")
    if compiler is None:
        check_compiler(lib_path)
    cmd = list(compiler) if compiler is not None else ["hipcc"] + list(hipcc_flags(lib_path))
    cmd += ["-DWN_MODEL_PLUGIN",
            *geometry_defines(candidates),
            "-I", CSRC, "-I", os.path.dirname(header),
            *extra_flags, "-shared", src, "-o", out,
            # the registration call resolves against the library the engines come from
            "-L", os.path.dirname(lib_file), "-l:" + os.path.basename(lib_file),
            "-Wl,-rpath," + os.path.dirname(lib_file)]
    done = subprocess.run(cmd, capture_output=True, text=True)
    if done.returncode != 0:
        raise _ffi.WalnutsHipError("compiling the device model failed:\n" + " ".join(cmd) + "\n" + done.stderr[-4000:])
    return out


def load_device_model(path: str, tag: str, *, lib_path: Optional[str] = None) -> int:
    """Load a model built by :func:`build_device_model` -> its id (what ``model_id(tag)`` returns from now on)."""
    lib = _ffi.load_library(lib_path)
    path = os.path.abspath(path)
    if path not in _loaded:
        lib.wn_model_clear_error()
        _loaded[path] = C.CDLL(path)   # (its static initialiser calls wn_plugin_register_model)
        msg = lib.wn_model_error()
        if msg:
            text = msg.decode("utf-8", "replace")
            lib.wn_model_clear_error()
            raise ValueError(text)
    mid = lib.wn_model_id(tag.encode())
    if mid < 0:
        raise ValueError(f"{path} did not register a device model named {tag!r}")
    return mid
