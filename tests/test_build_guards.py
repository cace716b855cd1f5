"""The build refuses a library whose attention kernels spill, leave the three-waves-per-SIMD register budget at head_dim 64, or
touch M0 outside the tile-copy helper (sageattention_amd/_build.py).  CPU-only: the checks parse compiler remarks / a
disassembly."""
import os

import pytest

from sageattention_amd import _build

NAME_D64 = "_ZN4sage14attn_i8_kernelILi64ELi4ELb1ELb1ELb0ELb1ELb0EEEvNS_10AttnParamsE"
NAME_D128 = "_ZN4sage14attn_i8_kernelILi128ELi8ELb0ELb1ELb0ELb0ELb0EEEvNS_10AttnParamsE"
NAME_MASK = "_ZN4sage14attn_i8_kernelILi64ELi4ELb0ELb1ELb0ELb0ELb1EEEvNS_10AttnParamsE"


def _remarks(name, vgprs, scratch=0):
    return (f"x.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
            f"x.hip:1:1: remark:     VGPRs: {vgprs} [-Rpass-analysis=kernel-resource-usage]\n"
            f"x.hip:1:1: remark:     ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]\n")


def test_scratch_is_refused():
    _build._check_no_scratch("x.hip", _remarks(NAME_D128, 230, 0))
    with pytest.raises(RuntimeError, match="scratch"):
        _build._check_no_scratch("x.hip", _remarks(NAME_D128, 256, 8))


def test_head_dim_64_occupancy_is_guarded():
    _build._check_occupancy("x.hip", _remarks(NAME_D64, 168))
    _build._check_occupancy("x.hip", _remarks(NAME_D128, 240))     # head_dim 128 runs two waves per SIMD by design
    _build._check_occupancy("x.hip", _remarks(NAME_MASK, 202))     # attn_mask variants are exempt
    with pytest.raises(RuntimeError, match="168"):
        _build._check_occupancy("x.hip", _remarks(NAME_D64, 169))


def test_m0_is_private_to_the_tile_copies_in_the_built_object():
    obj = os.path.join(_build.CSRC, "sage_attn.o")
    if not os.path.exists(obj):
        _build.build()
    _build._check_m0_private(obj)
