"""CPU: ``import sageattention`` -- the reference's package name -- resolves to the gfx950 implementation exactly as
the reference's callers use it (sageattention/__init__.py:25-29,86-95; example/parallel_sageattn_cogvideo.py:8-14,
44-52; example/cogvideox-2b.py:6,16-17).  No compute (no GPU here)."""
import functools
import inspect
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_names_diffusers_imports_by_name():
    from sageattention import (sageattn_qk_int8_pv_fp16_cuda, sageattn_qk_int8_pv_fp16_triton,  # noqa: F401
                               sageattn_qk_int8_pv_fp8_cuda, sageattn_qk_int8_pv_fp8_cuda_sm90)
    import sageattention_amd as impl
    assert sageattn_qk_int8_pv_fp16_cuda is impl.sageattn_qk_int8_pv_fp16_cuda
    assert sageattn_qk_int8_pv_fp8_cuda is impl.sageattn_qk_int8_pv_fp8_cuda
    assert sageattn_qk_int8_pv_fp8_cuda_sm90 is impl.sageattn_qk_int8_pv_fp8_cuda_sm90
    assert sageattn_qk_int8_pv_fp16_triton is impl.sageattn_qk_int8_pv_fp16_triton
    # the xDiT launcher wraps it in functools.partial (parallel_sageattn_cogvideo.py:44-51)
    f = functools.partial(sageattn_qk_int8_pv_fp8_cuda, pv_accum_dtype="fp32+fp32")
    assert "pv_accum_dtype" in inspect.signature(f.func).parameters


def test_lazy_names_and_submodules():
    import sageattention
    import sageattention_amd as impl
    from sageattention import sageattn, sageattn_varlen          # lazily served (__init__.py:25-29)
    assert sageattn is impl.sageattn and sageattn_varlen is impl.sageattn_varlen
    from sageattention.core import sageattn as s2
    assert s2 is impl.sageattn
    from sageattention.quant import per_block_int8, per_warp_int8, sub_mean, per_channel_fp8  # noqa: F401 (quant.py:23,106,183,225)
    assert per_warp_int8 is impl.quant.per_warp_int8
    # the reference's pybind module names
    assert sageattention.qattn.qk_int8_sv_f8_accum_f32_attn is impl._qattn.qk_int8_sv_f8_accum_f32_attn  # core.py:893
    from sageattention import _fused
    for name in ("quant_per_block_int8_cuda", "quant_per_block_int8_fuse_sub_mean_cuda", "quant_per_warp_int8_cuda",
                 "sub_mean_cuda"):  # csrc/fused/pybind.cpp:23-32
        assert callable(getattr(_fused, name))
    try:
        sageattention.does_not_exist
    except AttributeError:
        pass
    else:
        raise AssertionError("unknown attribute must raise AttributeError")


def test_reference_signatures_are_kept():
    """Parameter names and defaults of the public entry points equal the reference's (core.py:80-89,161-173,363-375,
    480-493,656-669,908-920), read from its source text (the reference package cannot be imported without its
    compiled extension; the file is parsed, not executed).  Skipped where /root/reference is absent (GPU box)."""
    import ast
    ref = "/root/reference/sageattention/core.py"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference tree not present")
    import sageattention
    tree = ast.parse(open(ref).read())
    want = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name.startswith("sageattn"):
            args = node.args
            names = [a.arg for a in args.args]
            defaults = [None] * (len(names) - len(args.defaults)) + [ast.literal_eval(d) for d in args.defaults]
            want[node.name] = (names, defaults, args.kwarg.arg if args.kwarg else None)
    assert len(want) >= 6
    for fname, (names, defaults, kwarg) in want.items():
        sig = inspect.signature(getattr(sageattention, fname))
        params = [p for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
        assert [p.name for p in params] == names, fname
        for p, d in zip(params, defaults):
            if p.default is not inspect.Parameter.empty or d is not None:
                assert p.default == d, (fname, p.name, p.default, d)
        if kwarg:
            assert any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()), fname


def test_packaging_metadata():
    setup_py = open(os.path.join(ROOT, "setup.py")).read()
    assert re.search(r'packages=\["sageattention_amd", "sageattention"\]', setup_py)
    assert "_build.build()" in setup_py and os.path.exists(os.path.join(ROOT, "pyproject.toml"))
