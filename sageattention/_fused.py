"""``sageattention._fused`` (reference pybind module, csrc/fused/pybind.cpp) -> ctypes shim over libsageattn_hip.so."""
from sageattention_amd._fused import *  # noqa: F401,F403
