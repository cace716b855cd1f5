"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/sageattn_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from sageattention_amd import _build
    return _build.build()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sageattn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sage_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 13, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sageattn_hip.h but not exported"


def test_binding_covers_header(built_lib):
    from sageattention_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    l = _lib.lib()
    assert l.sage_abi_version() == 3
    assert l.sage_target_arch() == b"gfx950"
    assert b"head_dim" in l.sage_status_string(-2)


def test_argument_validation_without_gpu(built_lib):
    """Host-side validation returns status codes before anything touches a device."""
    from sageattention_amd import _lib as L
    l = L.lib()
    t = L.SageTensor(None, 0, 0, 0)
    assert l.sage_k_mean(t, 0, 1, 1, 1, 64, None, None, None) == -1
    bad = L.SageTensor(16, 8, 8, 8)
    assert l.sage_quant_qk_int8(bad, 0, 1, 1, 8, 96, None, bad, 16, 1, 0, 128, 128, 1.0, 0, None, 1, None, None) == -2
    assert l.sage_quant_qk_int8(bad, 7, 1, 1, 8, 64, None, bad, 16, 1, 0, 128, 128, 1.0, 0, None, 1, None, None) == -1
    assert l.sage_set_tuning(0, 5) == -1 and l.sage_set_tuning(0, 0) == 0
    assert l.sage_set_tuning(1, 1) == -1
    assert l.sage_set_tuning(0, 8) == 0 and l.sage_get_tuning(0) == 8 and l.sage_set_tuning(0, 0) == 0
    assert l.sage_get_tuning(0) == 0 and l.sage_get_tuning(9) == -1
    assert l.sage_set_tuning(7, 0) == -1


def test_sequence_parallel_entry_points_validate_arguments(built_lib):
    """The round-2 building blocks (kv tile layouts, statistics, tile-major quantizers, merge_ex) reject bad arguments on
    the host, before any launch."""
    import ctypes as C
    from sageattention_amd import _lib as L
    l = L.lib()
    t = L.SageTensor(16, 64, 64, 64)
    lay = L.KvLayout(0, 0, 0, 0, 0)
    args = (1, 1, 1, 64, 64, 64, 0, 3, 128, 32, 0.125, None)
    assert l.sage_attn_qk_int8_pv_f16_kvtiles(t, t, t, 0, t, 0, 16, 16, None, None, *args) == -1          # no layout
    assert l.sage_attn_qk_int8_pv_f8_kvtiles(t, t, t, t, 0, 16, 16, 16, None, None, *args) == -1
    bad = L.KvLayout(-64, 0, 0, 0, 0)
    assert l.sage_attn_qk_int8_pv_f16_kvtiles(t, t, t, 0, t, 0, 16, 16, bad, None, *args) == -1           # negative stride
    odd = L.KvLayout(0, 0, 6, 2, 2)
    assert l.sage_attn_qk_int8_pv_f16_kvtiles(t, t, t, 0, t, 0, 16, 16, odd, None, *args) == -1           # scale strides < 4
    assert l.sage_seq_stats(t, 0, 1, 1, 64, 96, 16, 16, None) == -2                                       # head_dim
    assert l.sage_seq_stats(t, 0, 1, 1, 64, 64, None, 16, None) == -1
    assert l.sage_kv_stats_reduce(None, None, 1, 192, 1, 64, 64, 0, 448.0, None, None, None, None) == -1
    assert l.sage_kv_stats_reduce(16, None, 2, 10, 1, 64, 64, 0, 448.0, 16, None, None, None) == -1       # part stride too small
    st3 = (C.c_int64 * 3)(4, 4, 4)
    assert l.sage_quant_k_int8_kvtiles(t, 0, 1, 1, 64, 64, None, t, 0, 16, st3, 3, 0, None) == -1         # no tile stride
    assert l.sage_quant_k_int8_kvtiles(t, 0, 1, 1, 64, 64, None, t, 4096, 16, st3, 2, 0, None) == -1      # per_warp is a Q granularity
    assert l.sage_quant_v_fp8_apply(t, 0, 1, 1, 64, 64, t, 24, 16, None) == -1                            # unaligned tile stride
    assert l.sage_k_smooth_quant(t, 0, 1, 1, 64, 64, t, 16, None, 3, 0, 16, None) == -1                   # km missing
    assert l.sage_kv_prepare_fp8(t, t, 0, 1, 1, 64, 64, t, 16, None, 3, 0, t, 16, 448.0, 16, None) == -1  # km missing
    assert l.sage_kv_prepare_fp8(t, t, 0, 1, 1, 64, 96, t, 16, 16, 3, 0, t, 16, 448.0, 16, None) == -2    # head_dim
    assert l.sage_kv_prepare_fp8(t, t, 0, 1, 1, 64, 64, t, 16, 16, 2, 0, t, 16, 448.0, 16, None) == -1    # per_warp is a Q granularity
    assert l.sage_kv_prepare_fp8_workspace_bytes(2, 3, 1000, 64) >= 2 * 2 * 3 * 4 * 64 * 4
    op = (C.c_void_p * 1)(16)
    assert l.sage_merge_attn_states_multi_ex(op, op, 1, 0, 16, None, 4, 64, 0.0, None, 0.0, None) == -1   # lse multiplier must be > 0


def test_one_call_operator_validates_arguments(built_lib):
    """sage_sageattn_pv_{f16,f8}: workspace sizing and host-side validation, no launch."""
    from sageattention_amd import _lib as L
    l = L.lib()
    t = L.SageTensor(16, 64, 64, 64)
    ok = L.OpOpts(3, 32, 1, -1, 0)
    n16 = l.sage_sageattn_workspace_bytes(0, 2, 4, 2, 1000, 1000, 64, 1, ok)
    n8 = l.sage_sageattn_workspace_bytes(1, 2, 4, 2, 1000, 1000, 64, 1, ok)
    assert n16 >= 2 * 2 * 1000 * 64 and n8 >= n16 + 2 * 2 * 64 * 1024              # int8 K; + fp8 V^T
    big = l.sage_sageattn_workspace_bytes(0, 1, 4, 2, 8192, 8192, 128, 1, L.OpOpts(3, 32, 1, 0, 0))  # fuse_q = 0: stand-alone Q quantizer
    assert big >= 2 * 8192 * 128 + 4 * 8192 * 128 + 2 * 4 * 8192 * 4
    assert l.sage_sageattn_workspace_bytes(0, 1, 1, 1, 64, 64, 96, 0, ok) == 0       # head_dim
    assert l.sage_sageattn_workspace_bytes(0, 1, 1, 1, 64, 64, 64, 0, L.OpOpts(1, 32, 1, -1, 0)) == 0  # per_block: not here
    args = (1, 2, 2, 64, 64, 64, 0, 0.125)
    assert l.sage_sageattn_pv_f16(t, t, t, 0, t, None, *args, ok, 16, 8, None) == -1             # workspace too small
    assert l.sage_sageattn_pv_f16(t, t, t, 0, t, None, *args, ok, None, 1 << 20, None) == -1
    assert l.sage_sageattn_pv_f16(t, t, t, 0, t, None, *args, L.OpOpts(3, 32, 0, -1, 0), 16, 1 << 20, None) == -3  # smooth_k = 0
    assert l.sage_sageattn_pv_f16(t, t, t, 0, t, None, *args, L.OpOpts(3, 32, 1, -1, 5), 16, 1 << 20, None) == -1  # nwaves
    assert l.sage_sageattn_pv_f16(t, t, t, 0, t, None, 1, 3, 2, 64, 64, 64, 0, 0.125, ok, 16, 1 << 20, None) == -1  # Hq % Hk
    assert l.sage_sageattn_pv_f8(t, t, t, 0, t, None, *args, 0.0, ok, 16, 1 << 20, None) == -1   # scale_max


def test_product_path_has_no_oracle_import():
    """The shipped package must never import the oracle or fall back to CPU."""
    pkg = os.path.join(ROOT, "sageattention_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+[^#\n]*oracle", src, flags=re.M), fn
            assert "sage_oracle" not in src, fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sageattention_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
