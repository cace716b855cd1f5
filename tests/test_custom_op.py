"""torch.library registration (sageattention_amd/ops.py): the op exists with the documented schema, its fake
implementation describes the outputs (no GPU needed), and -- on the GPU -- a function calling it compiles as ONE graph
(fullgraph=True; the reference's @torch.compiler.disable entry points force a graph break) with results identical to
the eager operator."""
import pytest
import torch

import sageattention_amd.ops as ops


def test_ops_registered_with_schema():
    op = torch.ops.sageattention_amd.attn.default
    s = str(op._schema)
    assert "Tensor q, Tensor k, Tensor v, str tensor_layout, bool is_causal, float sm_scale, str pv, str qk_quant_gran" in s
    assert str(torch.ops.sageattention_amd.attn_lse.default._schema).endswith("-> (Tensor, Tensor)")


@pytest.mark.parametrize("layout", ["HND", "NHD"])
def test_fake_implementation_shapes(layout):
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        shp = (2, 8, 100, 128) if layout == "HND" else (2, 100, 8, 128)
        kshp = (2, 4, 300, 128) if layout == "HND" else (2, 300, 4, 128)
        q = torch.empty(shp, dtype=torch.bfloat16, device="cuda")
        k = torch.empty(kshp, dtype=torch.bfloat16, device="cuda")
        o, lse = ops.sageattn_compilable(q, k, k, tensor_layout=layout, return_lse=True)
        assert o.shape == q.shape and o.dtype == q.dtype and o.device == q.device
        assert lse.shape == (2, 8, 100) and lse.dtype == torch.float32
        o2 = ops.sageattn_compilable(q, k, k, tensor_layout=layout, dropout_p=0.0)
        assert o2.shape == q.shape


def test_wrapper_validates_layout():
    q = torch.zeros(1, 1, 4, 64)
    with pytest.raises(ValueError):
        ops.sageattn_compilable(q, q, q, tensor_layout="BHSD")


@pytest.mark.gpu
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_compiles_as_one_graph(pv):
    import sageattention_amd as sa
    torch.manual_seed(0)
    q = torch.randn(1, 4, 256, 128, dtype=torch.float16, device="cuda")
    k = torch.randn(1, 4, 320, 128, dtype=torch.float16, device="cuda")
    v = torch.randn(1, 4, 320, 128, dtype=torch.float16, device="cuda")

    def block(q, k, v):
        o, lse = ops.sageattn_compilable(q * 1.0, k, v, pv=pv, return_lse=True)
        return o + 1.0, lse

    # aot_eager: traces, functionalises and runs through the dispatcher without generating Triton code
    compiled = torch.compile(block, backend="aot_eager", fullgraph=True)
    oc, lc = compiled(q, k, v)
    entry = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    oe, le = entry(q, k, v, return_lse=True)
    assert torch.equal(oc, oe + 1.0)
    assert torch.equal(lc, le)


@pytest.mark.gpu
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_operator_captures_into_a_hip_graph(pv):
    """Every launch goes to the caller's current stream and nothing synchronises or allocates outside the caching
    allocator: the whole operator (K mean, quantizers, attention, LSE fix) records into a HIP graph and replays with
    bit-identical results on new input values."""
    import sageattention_amd as sa
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    torch.manual_seed(1)
    q, k, v = (torch.randn(2, 4, 384, 128, dtype=torch.float16, device="cuda") for _ in range(3))
    for _ in range(2):
        fn(q, k, v, is_causal=True, return_lse=True)   # warm up: module load, function attributes
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o_g, l_g = fn(q, k, v, is_causal=True, return_lse=True)
    for seed in (2, 3):
        torch.manual_seed(seed)
        q.copy_(torch.randn_like(q)); k.copy_(torch.randn_like(k)); v.copy_(torch.randn_like(v))
        g.replay()
        torch.cuda.synchronize()
        o_e, l_e = fn(q, k, v, is_causal=True, return_lse=True)
        assert torch.equal(o_g, o_e) and torch.equal(l_g, l_e)


def test_fake_output_is_contiguous_for_strided_inputs():
    """The real op returns a freshly allocated contiguous tensor (also when the head dim was padded or q is a strided
    view); the fake must describe that layout, not q's."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q = torch.empty((2, 100, 8, 96), dtype=torch.float16, device="cuda").transpose(1, 2)  # HND view of NHD memory
        o = ops.sageattn_compilable(q, q, q, pv="fp16")
        assert o.shape == q.shape and o.is_contiguous()


@pytest.mark.gpu
def test_compiles_at_padded_head_dim_and_strided_q():
    """head_dim 96 (padded to 128 inside, output sliced back) and a transposed q: real and fake layouts agree, so the
    compiled graph (inductor included) indexes the result correctly."""
    import sageattention_amd as sa
    torch.manual_seed(0)
    q = torch.randn(1, 200, 4, 96, dtype=torch.float16, device="cuda").transpose(1, 2)
    k = torch.randn(1, 4, 260, 96, dtype=torch.float16, device="cuda")
    v = torch.randn(1, 4, 260, 96, dtype=torch.float16, device="cuda")

    def block(q, k, v):
        return ops.sageattn_compilable(q, k, v, pv="fp16") * 2.0

    want = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v) * 2.0
    for backend in ("aot_eager", "inductor"):
        try:
            got = torch.compile(block, backend=backend, fullgraph=True)(q, k, v)
        except torch._dynamo.exc.BackendCompilerFailed:
            if backend == "inductor":   # no usable code generator for the surrounding pointwise op on this box
                continue
            raise
        assert got.shape == want.shape and torch.equal(got, want), backend
