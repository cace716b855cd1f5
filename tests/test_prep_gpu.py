"""Single-pass pre-pass kernels (csrc/sage_prep.hip): sage_k_prep (K mean + smoothing + INT8 quantization, K read once)
and sage_v_prep_fp8 (per-channel statistics + FP8 quantization + transpose, V read once) must be BIT-identical to the
multi-launch kernels they replace (which are pinned against the reference fixtures / the oracle in test_gpu_parity.py),
on ragged and large shapes, both layouts, both dtypes -- also when every workgroup is forced onto the self-help path
that guarantees forward progress (SAGE_TUNE_PREP_POLL = 1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1, 64), (2, 3, 63, 128), (1, 2, 257, 64), (2, 4, 1000, 128), (1, 5, 4096, 64), (4, 32, 2048, 64),
          (2, 16, 8192, 128)]


def _mk(B, H, N, D, dt, layout, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    shape = (B, H, N, D) if layout == "HND" else (B, N, H, D)
    x = torch.randn(shape, device="cuda", generator=g) * 2 + torch.randn((1, 1, 1, D), device="cuda", generator=g) * 3
    return x.to(dt)


@pytest.mark.parametrize("poll", [0, 1])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_k_prep_bit_identical_to_mean_plus_quantizer(dt, layout, poll):
    import sageattention_amd as sa
    from sageattention_amd import _lib as L
    from sageattention_amd.quant import _quant, k_mean, k_smooth_quant
    lib = L.lib()
    try:
        assert lib.sage_set_tuning(2, poll) == 0
        for i, (B, H, N, D) in enumerate(SHAPES):
            k = _mk(B, H, N, D, dt, layout, 100 + i)
            for gran, rnd in ((L.GRAN_PER_THREAD, L.ROUND_TRITON), (L.GRAN_PER_BLOCK, L.ROUND_CUDA), (L.GRAN_PER_BLOCK, L.ROUND_TRITON)):
                km = k_mean(k, layout)
                k8, ks, _ = _quant(k, layout, gran, True, 64, 64, 1.0, rnd, mean=km, dense_heads=True)
                k8b, ksb, kmb = k_smooth_quant(k, layout, gran, rnd)
                assert torch.equal(kmb, km), (B, H, N, D, gran, rnd)
                assert torch.equal(ksb, ks), (B, H, N, D, gran, rnd)
                assert torch.equal(k8b, k8), (B, H, N, D, gran, rnd)
    finally:
        lib.sage_set_tuning(2, 0)


@pytest.mark.parametrize("poll", [0, 1])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_v_prep_bit_identical_to_three_launch_quantizer(dt, layout, poll, monkeypatch):
    import sageattention_amd as sa
    from sageattention_amd import _lib as L, quant
    lib = L.lib()
    try:
        assert lib.sage_set_tuning(2, poll) == 0
        for i, (B, H, N, D) in enumerate(SHAPES):
            v = _mk(B, H, N, D, dt, layout, 200 + i)
            for smooth in (False, True):
                monkeypatch.setattr(quant, "SINGLE_PASS", False)
                v8, vs, vm = quant.per_channel_fp8(v, tensor_layout=layout, smooth_v=smooth)
                monkeypatch.setattr(quant, "SINGLE_PASS", True)
                v8b, vsb, vmb = quant.per_channel_fp8(v, tensor_layout=layout, smooth_v=smooth)
                assert torch.equal(vsb, vs), (B, H, N, D, smooth)
                assert (vm is None and vmb is None) or torch.equal(vmb, vm)
                assert torch.equal(v8b.view(torch.uint8), v8.view(torch.uint8)), (B, H, N, D, smooth)
    finally:
        lib.sage_set_tuning(2, 0)


@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_operator_is_bit_identical_with_and_without_single_pass(pv, monkeypatch):
    import sageattention_amd as sa
    from sageattention_amd import quant
    torch.manual_seed(4)
    q = torch.randn(2, 8, 1500, 128, dtype=torch.float16, device="cuda")
    k = (torch.randn(2, 4, 1500, 128, device="cuda") + 2 * torch.randn(1, 4, 1, 128, device="cuda")).half()
    v = torch.randn(2, 4, 1500, 128, dtype=torch.float16, device="cuda")
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    monkeypatch.setattr(quant, "SINGLE_PASS", True)
    o1, l1 = fn(q, k, v, is_causal=True, return_lse=True)
    monkeypatch.setattr(quant, "SINGLE_PASS", False)
    o0, l0 = fn(q, k, v, is_causal=True, return_lse=True)
    assert torch.equal(o1, o0) and torch.equal(l1, l0)
