"""The one-call C-ABI operators sage_sageattn_pv_{f16,f8} (csrc/sage_op.hip): one crossing, one workspace.  They sequence
the same entry points as the multi-call path of sageattention_amd/core.py, so every result must be BIT-identical to it
(which the oracle / fixture tests of test_gpu_parity.py pin), and the per-call `nwaves` option must not leak into the
calling thread's tuning state."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, Hq, Hk, M, N, D, layout, dtype, causal, gran
    (2, 4, 4, 256, 256, 64, "HND", torch.float16, False, "per_thread"),
    (1, 8, 2, 300, 300, 128, "NHD", torch.bfloat16, True, "per_warp"),
    (1, 2, 2, 100, 1000, 128, "HND", torch.float16, False, "per_thread"),   # M != N
    (1, 2, 2, 4200, 4200, 64, "HND", torch.float16, True, "per_thread"),
    (1, 2, 1, 4160, 520, 128, "NHD", torch.bfloat16, False, "per_warp"),
]


@pytest.mark.parametrize("fused_q", [True, False], ids=["fusedq", "separate_q"])
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x).replace("torch.", "") for x in c))
def test_one_call_is_bit_identical_to_the_multi_call_path(case, pv, fused_q):
    import sageattention_amd as sa
    from sageattention_amd import core
    B, Hq, Hk, M, N, D, layout, dt, causal, gran = case
    if pv == "fp8" and M != N:
        pytest.skip("the fp8 one-call path prepares K and V together: equal shapes only")
    torch.manual_seed(M + N)
    shp = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
    q = torch.randn(shp(Hq, M), dtype=dt, device="cuda")
    k = (torch.randn(shp(Hk, N), device="cuda") + 1.0).to(dt)
    v = torch.randn(shp(Hk, N), dtype=dt, device="cuda")
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    keep, keep_f = core.ONE_CALL, core.FUSE_Q_QUANT
    core.FUSE_Q_QUANT = fused_q   # False: the stand-alone Q quantizer + finish-LSE legs of both paths
    try:
        res = {}
        for one in (True, False):
            core.ONE_CALL = one
            res[one] = fn(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, return_lse=True)
            res[(one, "nolse")] = fn(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran)
        torch.cuda.synchronize()
    finally:
        core.ONE_CALL, core.FUSE_Q_QUANT = keep, keep_f
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[(True, "nolse")], res[(False, "nolse")]) and torch.equal(res[True][0], res[(True, "nolse")])
    assert torch.isfinite(res[True][0].float()).all() and torch.isfinite(res[True][1]).all()


def test_one_call_nwaves_is_per_call():
    from sageattention_amd import _lib as L
    B, H, N, D = 1, 2, 512, 128
    torch.manual_seed(5)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    lib = L.lib()
    outs = []
    for nw in (0, 4, 8):
        opts = L.OpOpts(3, 32, 1, -1, nw)
        nbytes = lib.sage_sageattn_workspace_bytes(0, B, H, H, N, N, D, 0, opts)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        o = torch.empty_like(q)
        L.check(lib.sage_sageattn_pv_f16(L.desc(q, "HND"), L.desc(k, "HND"), L.desc(v, "HND"), 0, L.desc(o, "HND"), None, B, H, H, N,
                                         N, D, 0, D ** -0.5, opts, ws.data_ptr(), nbytes,
                                         torch.cuda.current_stream().cuda_stream), "one call")
        torch.cuda.synchronize()
        assert lib.sage_get_tuning(0) == 0      # the calling thread's setting is untouched
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])  # geometry never changes results


@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_custom_sm_scale_and_padded_head_dim_vs_oracle(pv):
    """A caller-chosen sm_scale and a head_dim that is padded (96 -> 128, sm_scale from the ORIGINAL head_dim, core.py:606)
    through the default (one-call) path against the oracle's end-to-end restatement."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(3)
    for D, sm in ((64, 0.2), (96, None), (128, 0.05)):
        q, k, v = (torch.randn(1, 4, 300, D).to(torch.float16) for _ in range(3))
        fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
        o, lse = fn(q.cuda(), k.cuda(), v.cuda(), sm_scale=sm, return_lse=True)
        torch.cuda.synchronize()
        oo, ol = O.sageattn_oracle(q, k, v, qk_quant_gran="per_thread", pv=pv, sm_scale=sm, return_lse=True)
        assert o.shape == q.shape
        assert (o.cpu().float() - oo.float()).abs().max() < (2e-3 if pv == "fp16" else 0.06), (D, sm)
        assert (lse.cpu() - ol).abs().max() < 2e-3, (D, sm)
