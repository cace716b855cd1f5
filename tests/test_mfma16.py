"""The 16x16-MFMA attention kernel (csrc/sage_attn16.hip; SURVEY 2.2 K5/K6 names both fragment families) against the
32x32 kernel on the same quantized operands and against the reference fixtures.  Same arithmetic (fp32 row sums of the
unrounded p, fp32 P.V accumulation); the two differ in the ORDER in which P.V is accumulated (k-steps of 32 keys
instead of 16) and, at head_dim 64, in the 32x32 kernel's row sums of the fp16-rounded P."""
import pytest
import torch

from conftest import Golden, calc_diff

pytestmark = pytest.mark.gpu


def _run(L, q8, k8, v, o, lse, qs, ks, dims, causal, gran, shape, nw=0):
    B, Hq, Hk, M, N, D = dims
    lib = L.lib()
    lib.sage_set_tuning(1, shape)
    lib.sage_set_tuning(0, nw)
    try:
        L.check(lib.sage_attn_qk_int8_pv_f16(
            L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), L.dtype_code(v.dtype), L.desc(o, "HND"),
            L.dtype_code(o.dtype), qs.data_ptr(), ks.data_ptr(), None, lse.data_ptr(), B, Hq, Hk, M, N, D, int(causal),
            gran, 128, 32, D ** -0.5, 0, torch.cuda.current_stream().cuda_stream), "attn")
        torch.cuda.synchronize()
    finally:
        lib.sage_set_tuning(1, 0)
        lib.sage_set_tuning(0, 0)


CASES = [  # B, Hq, Hk, M, N, D, causal, dtype, gran
    (2, 4, 4, 256, 256, 128, False, torch.float16, "per_thread"),
    (1, 4, 2, 300, 300, 128, True, torch.float16, "per_thread"),     # ragged, GQA, causal
    (1, 2, 2, 100, 333, 128, False, torch.float16, "per_warp"),      # M != N, ragged both
    (2, 4, 4, 384, 384, 64, False, torch.float16, "per_thread"),
    (1, 8, 2, 520, 520, 64, True, torch.float16, "per_warp"),
    (1, 2, 2, 200, 1000, 64, False, torch.bfloat16, "per_thread"),
    (1, 2, 2, 512, 512, 128, True, torch.bfloat16, "per_thread"),
    (1, 2, 2, 1, 1, 64, False, torch.float16, "per_thread"),         # one row, one key
    (1, 2, 2, 33, 65, 128, False, torch.float16, "per_thread"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x).replace("torch.", "") for x in c))
@pytest.mark.parametrize("nw", [4, 8])
def test_mfma16_matches_mfma32(case, nw):
    import sageattention_amd as sa
    from sageattention_amd import _lib as L, core
    B, Hq, Hk, M, N, D, causal, dt, gran = case
    torch.manual_seed(B * 1000 + M + N + D)
    q = torch.randn(B, Hq, M, D, dtype=dt, device="cuda")
    k = (torch.randn(B, Hk, N, D, device="cuda") + torch.randn(1, Hk, 1, D, device="cuda")).to(dt)
    v = torch.randn(B, Hk, N, D, dtype=dt, device="cuda")
    km = sa.quant.k_mean(k)
    q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", gran, D ** -0.5, 32, False, Hq, Hk)
    code = core._GRAN_CODE[gran]
    outs = {}
    for shape in (32, 16):
        o = torch.full_like(q, float("nan"))
        lse = torch.full((B, Hq, M), float("nan"), dtype=torch.float32, device="cuda")
        _run(L, q8, k8, v, o, lse, qs, ks, (B, Hq, Hk, M, N, D), causal, code, shape, nw)
        outs[shape] = (o.float().cpu(), lse.cpu())
    o32, l32 = outs[32]
    o16, l16 = outs[16]
    assert torch.isfinite(o16).all() and torch.isfinite(l16).all()
    # one output ulp at |o| <= 4 (bf16: 2^-6; fp16: 2^-9 -- 2e-3 also covers the head_dim-64 rounded-P row sums of the 32x32 kernel)
    tol = 3.2e-2 if dt == torch.bfloat16 else 2e-3
    assert (o16 - o32).abs().max() < tol, (o16 - o32).abs().max()
    assert calc_diff(o16, o32) < 1e-5
    assert (l16 - l32).abs().max() < (1.5e-3 if D == 64 else 2e-5)


@pytest.mark.parametrize("name", ["c1_hnd", "d128_ragged", "cross_100x200", "bf16_d128", "d128_causal_384", "gqa_causal_320"])
def test_mfma16_vs_reference_fixture(name):
    """The reference Triton kernel's output for the fixture's int8 operands (tolerances of test_gpu_parity)."""
    from sageattention_amd import _lib as L
    g = Golden(name)
    m = g.meta
    if m["layout"] != "HND":
        hnd = lambda x: x.transpose(1, 2).contiguous()
    else:
        hnd = lambda x: x
    dt = torch.float16 if m["dtype"] == "fp16" else torch.bfloat16
    q8, k8 = hnd(g.pb_q8).cuda(), hnd(g.pb_k8).cuda()
    v = hnd(g.v).to(torch.float16).cuda()   # core.py:289-290
    o = torch.empty(q8.shape, dtype=dt, device="cuda")
    lse = torch.empty(m["B"], m["Hq"], m["M"], dtype=torch.float32, device="cuda")
    qs, ks = g.pb_qs.cuda(), g.pb_ks.cuda()
    lib = L.lib()
    lib.sage_set_tuning(1, 16)
    try:
        L.check(lib.sage_attn_qk_int8_pv_f16(
            L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), 0, L.desc(o, "HND"), L.dtype_code(dt),
            qs.data_ptr(), ks.data_ptr(), None, lse.data_ptr(), m["B"], m["Hq"], m["Hk"], m["M"],
            m["N"], m["D"], m["causal"], 1, 128, 128, m["sm_scale"], 1, torch.cuda.current_stream().cuda_stream), "attn16")
        torch.cuda.synchronize()
    finally:
        lib.sage_set_tuning(1, 0)
    ref = hnd(g.pb_o).float()
    assert (o.cpu().float() - ref).abs().max() < (4e-3 if dt == torch.float16 else 1.6e-2)
    assert (lse.cpu() - g.pb_lse2).abs().max() < 5e-4
