"""walnuts_amd -- MI355X-native many-chain Walnuts/NUTS leapfrog engine.

Host-side mirror of the reference's Python package surface for ONE path: ``walnuts_device`` plays the role of
``walnutpie.walnuts_pyfunc`` (python/src/walnutpie/pyfunc.py:45-286) for built-in device models, and
``DeviceEngine`` exposes the batched per-transition verbs of the C ABI in ``include/walnuts_hip.h``.
Everything runs through ``walnuts_amd/lib/libwalnuts_hip.so`` (hand-written HIP for gfx950); there is no
CPU fallback: importing the binding without the built library raises.
"""
from ._ffi import WalnutsHipError, load_library  # noqa: F401
from .engine import (MODEL_DIAG_NORMAL, MODEL_FUNNEL, MODEL_RW1, MODEL_STD_NORMAL, DeviceEngine, model_id,  # noqa: F401
                     default_config, stream_version)
from .device import WalnutsOutputArray, WarmupInfo, walnuts_device  # noqa: F401
from . import models, summary  # noqa: F401,E402
from .summary import MarkovChains, Summarizer  # noqa: F401,E402
