"""Installable build of the MI355X-native SageAttention drop-in (replaces the reference's setup.py:187-260 ROCm branch).

``pip install .`` / ``python setup.py build_ext --inplace`` compile the HIP sources for gfx950 with hipcc
(sageattention_amd/_build.py: explicit ``hipcc --offload-arch=gfx950``, no torch cpp_extension, no rocWMMA, no
rocblas/hipblas link) into ``sageattention_amd/libsageattn_hip.so`` and install two packages: ``sageattention_amd``
(the implementation) and ``sageattention`` (the import-name shim, so existing ``from sageattention import ...`` callers
need no change)."""
import os
import sys

from setuptools import Command, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_hip():
    sys.path.insert(0, HERE)
    from sageattention_amd import _build
    return _build.build()


class BuildHip(Command):
    description = "compile the gfx950 HIP library (libsageattn_hip.so) with hipcc"
    user_options = [("inplace", "i", "kept for `build_ext --inplace` compatibility (the build is always in-tree)")]
    boolean_options = ["inplace"]

    def initialize_options(self):
        self.inplace = False

    def finalize_options(self):
        pass

    def run(self):
        print("built", _build_hip())


class BuildPyWithHip(build_py):
    def run(self):
        _build_hip()  # the .so must exist before package_data is collected
        super().run()


setup(
    name="sageattention-amd",
    version="0.2.0",
    description="SageAttention (INT8 QK^T, FP16/FP8 PV) for AMD MI355X: hand-written gfx950 HIP kernels behind the "
                "reference's sageattention API",
    packages=["sageattention_amd", "sageattention"],
    package_data={"sageattention_amd": ["libsageattn_hip.so", "csrc/*.hip", "csrc/*.h"]},
    data_files=[("include", ["include/sageattn_hip.h"])],
    python_requires=">=3.9",
    install_requires=[],  # torch (ROCm build) is expected to be present; not pinned, as in the reference
    cmdclass={"build_ext": BuildHip, "build_hip": BuildHip, "build_py": BuildPyWithHip},
)
