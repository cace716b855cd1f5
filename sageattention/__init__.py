"""Import-name drop-in for the reference package: ``import sageattention`` / ``from sageattention import
sageattn_qk_int8_pv_fp16_cuda`` (what diffusers and xDiT do, reference sageattention/__init__.py:86-95) resolve to
the MI355X-native implementation in ``sageattention_amd`` with no change in the caller.

Layout mirrors the reference package: ``sageattention.core`` / ``.quant`` (Python API), ``sageattention._fused`` and
``sageattention._qattn_rocm`` (the names of its pybind modules, served by ctypes shims over libsageattn_hip.so), and a
``qattn`` alias as in sageattention/__init__.py:8-13.  ``sageattn`` / ``sageattn_varlen`` are served lazily through
``__getattr__`` like there (:25-29).  There is no stub fallback: if the HIP library is missing, the first call raises."""
from importlib import import_module

from sageattention_amd import (sageattn_qk_int8_pv_fp16_cuda, sageattn_qk_int8_pv_fp16_triton,  # noqa: F401
                               sageattn_qk_int8_pv_fp8_cuda, sageattn_qk_int8_pv_fp8_cuda_sm90, __version__)
from . import _qattn_rocm as qattn  # noqa: F401

__all__ = ["qattn", "sageattn_qk_int8_pv_fp16_cuda", "sageattn_qk_int8_pv_fp8_cuda",
           "sageattn_qk_int8_pv_fp8_cuda_sm90", "sageattn_qk_int8_pv_fp16_triton"]

_LAZY = {"sageattn": ".core", "sageattn_varlen": ".core", "ring_sageattn": ".core", "ulysses_sageattn": ".core"}


def __getattr__(name):
    if name in _LAZY:
        return getattr(import_module(_LAZY[name], __name__), name)
    if name in ("_fused", "core", "quant", "_qattn_rocm"):
        return import_module("." + name, __name__)
    raise AttributeError(f"module {__name__} has no attribute {name}")
