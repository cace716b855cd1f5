// One-call operators: everything sageattn_qk_int8_pv_fp16_cuda / sageattn_qk_int8_pv_fp8_cuda do below their argument
// checks (core.py:604-651, 786-905) behind ONE C-ABI crossing and ONE caller-provided workspace -- K mean + INT8 K
// (sage_k_smooth_quant / sage_kv_prepare_fp8), FP8 V, Q quantizer (folded into the attention kernel's prologue unless
// fuse_q = 0), attention, LSE fix.  Host code only: it sequences the library's own entry points on the caller's stream, so the
// results are bit-identical to calling them one by one (what the Python mirror did until round 3: 3-4 crossings and 6-9
// allocations per call, 46 us of host time against a 64 us GPU step at (4,32,1024,64)).
#include "sage_common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Plan {
  bool fuse_q;
  size_t k8, ks, km, pre_ws, v8, v_scale, q8, qs, corr, lse2, total;  // byte offsets (valid where the piece exists)
  int gk, gq, npad;
};

bool make_plan(Plan& pl, int pv_fp8, int B, int Hq, int Hk, int M, int N, int D, int want_lse, const sage_op_opts* o) {
  if (B <= 0 || Hq <= 0 || Hk <= 0 || M <= 0 || N <= 0 || (D != 64 && D != 128) || !o) return false;
  const int gran = o->qk_gran, warpq = o->warpq ? o->warpq : 32;
  if (gran != SAGE_GRAN_PER_WARP && gran != SAGE_GRAN_PER_THREAD) return false;
  if (warpq != 16 && warpq != 32) return false;
  pl.fuse_q = o->fuse_q != 0;  // -1 (the library's choice) = fused: it is at least as fast at every length (core.py, round 3)
  pl.npad = (N + 63) / 64 * 64;
  pl.gk = (N + 63) / 64 * (gran == SAGE_GRAN_PER_THREAD ? 4 : 1);
  const int nblkq = (M + 127) / 128;
  pl.gq = gran == SAGE_GRAN_PER_WARP ? nblkq * (128 / warpq) : nblkq * (128 / warpq) * 8;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += up(bytes); return at; };
  pl.k8 = take((size_t)B * Hk * N * D);
  pl.ks = take((size_t)B * Hk * pl.gk * 4);
  pl.km = take((size_t)B * Hk * D * 2);
  const size_t pre = pv_fp8 ? sage_kv_prepare_fp8_workspace_bytes(B, Hk, N, D) : sage_k_mean_workspace_bytes(B, Hk, N, D);
  pl.pre_ws = take(pre > 4 ? pre : 4);
  pl.v8 = pl.v_scale = 0;
  if (pv_fp8) {
    pl.v8 = take((size_t)B * Hk * D * pl.npad);
    pl.v_scale = take((size_t)B * Hk * D * 4);
  }
  pl.q8 = pl.qs = pl.corr = pl.lse2 = 0;
  if (!pl.fuse_q) {
    pl.q8 = take((size_t)B * Hq * M * D);
    pl.qs = take((size_t)B * Hq * pl.gq * 4);
    if (want_lse) {
      pl.corr = take((size_t)B * Hq * M * 4);
      pl.lse2 = take((size_t)B * Hq * M * 4);
    }
  }
  pl.total = off;
  return true;
}

int run(int pv_fp8, const sage_tensor* q, const sage_tensor* k, const sage_tensor* v, int dtype, const sage_tensor* o, float* lse,
        int B, int Hq, int Hk, int M, int N, int D, int is_causal, float sm_scale, float scale_max, const sage_op_opts* opts,
        void* workspace, size_t workspace_bytes, sage_stream_t stream) {
  if (!q || !k || !v || !o || !opts || !workspace) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  if (Hk <= 0 || Hq % Hk != 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (!opts->smooth_k) return SAGE_ERR_UNSUPPORTED;  // the un-smoothed variant goes through the separate entry points
  if (opts->nwaves != 0 && opts->nwaves != 4 && opts->nwaves != 8) return SAGE_ERR_INVALID_ARGUMENT;
  Plan pl;
  if (!make_plan(pl, pv_fp8, B, Hq, Hk, M, N, D, lse != nullptr, opts)) return SAGE_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < pl.total || !sage::aligned16(workspace)) return SAGE_ERR_INVALID_ARGUMENT;
  char* const ws = static_cast<char*>(workspace);
  const int gran = opts->qk_gran, warpq = opts->warpq ? opts->warpq : 32;
  const int rounding = gran == SAGE_GRAN_PER_THREAD ? SAGE_ROUND_TRITON : SAGE_ROUND_CUDA;      // core.py:621-624
  const int k_gran = gran == SAGE_GRAN_PER_THREAD ? SAGE_GRAN_PER_THREAD : SAGE_GRAN_PER_BLOCK;  // per_warp: K per block
  // internal operands are head-major whatever the caller's layout: the attention kernel then streams the rows of one
  // head from consecutive lines
  const sage_tensor k8{ws + pl.k8, (int64_t)Hk * N * D, (int64_t)N * D, D};
  float* const ks = reinterpret_cast<float*>(ws + pl.ks);
  void* const km = ws + pl.km;
  int st;
  sage_tensor v8{nullptr, 0, 0, 0};
  float* v_scale = nullptr;
  if (pv_fp8) {
    v8 = sage_tensor{ws + pl.v8, (int64_t)Hk * D * pl.npad, (int64_t)D * pl.npad, pl.npad};
    v_scale = reinterpret_cast<float*>(ws + pl.v_scale);
    st = sage_kv_prepare_fp8(k, v, dtype, B, Hk, N, D, &k8, ks, km, k_gran, rounding, &v8, v_scale, scale_max, ws + pl.pre_ws, stream);
  } else {
    st = sage_k_smooth_quant(k, dtype, B, Hk, N, D, &k8, ks, km, k_gran, rounding, ws + pl.pre_ws, stream);
  }
  if (st != SAGE_OK) return st;
  // per-call workgroup geometry: the calling thread's tuning value is set for this call's launches and restored
  const int prev_nw = sage_get_tuning(SAGE_TUNE_NWAVES);
  if (opts->nwaves) sage_set_tuning(SAGE_TUNE_NWAVES, opts->nwaves);
  if (pl.fuse_q) {
    st = pv_fp8 ? sage_attn_fusedq_pv_f8(q, dtype, &k8, &v8, o, dtype, ks, km, v_scale, nullptr, lse, B, Hq, Hk, M, N, D, is_causal,
                                         gran, warpq, sm_scale, stream)
                : sage_attn_fusedq_pv_f16(q, dtype, &k8, v, dtype, o, dtype, ks, km, nullptr, lse, B, Hq, Hk, M, N, D, is_causal,
                                          gran, warpq, sm_scale, stream);
  } else {
    const sage_tensor q8{ws + pl.q8, (int64_t)Hq * M * D, (int64_t)M * D, D};
    float* const qs = reinterpret_cast<float*>(ws + pl.qs);
    float* const corr = lse ? reinterpret_cast<float*>(ws + pl.corr) : nullptr;
    float* const lse2 = lse ? reinterpret_cast<float*>(ws + pl.lse2) : nullptr;
    st = sage_quant_qk_int8(q, dtype, B, Hq, M, D, nullptr, &q8, qs, gran, 0, 128, warpq, 1.0f, rounding, lse ? km : nullptr,
                            Hq / Hk, corr, stream);
    if (st == SAGE_OK)
      st = pv_fp8 ? sage_attn_qk_int8_pv_f8(&q8, &k8, &v8, o, dtype, qs, ks, v_scale, nullptr, lse2, B, Hq, Hk, M, N, D, is_causal,
                                            gran, 128, warpq, sm_scale, 0, stream)
                  : sage_attn_qk_int8_pv_f16(&q8, &k8, v, dtype, o, dtype, qs, ks, nullptr, lse2, B, Hq, Hk, M, N, D, is_causal,
                                             gran, 128, warpq, sm_scale, 0, stream);
    if (st == SAGE_OK && lse) st = sage_finish_lse(lse2, corr, sm_scale, lse, (int64_t)B * Hq * M, stream);
  }
  if (opts->nwaves) sage_set_tuning(SAGE_TUNE_NWAVES, prev_nw);
  return st;
}

}  // namespace

extern "C" size_t sage_sageattn_workspace_bytes(int pv_fp8, int B, int Hq, int Hk, int M, int N, int D, int want_lse,
                                                const sage_op_opts* opts) {
  Plan pl;
  return make_plan(pl, pv_fp8, B, Hq, Hk, M, N, D, want_lse, opts) ? pl.total : 0;
}

extern "C" int sage_sageattn_pv_f16(const sage_tensor* q, const sage_tensor* k, const sage_tensor* v, int dtype,
                                    const sage_tensor* o, float* lse, int B, int Hq, int Hk, int M, int N, int D, int is_causal,
                                    float sm_scale, const sage_op_opts* opts, void* workspace, size_t workspace_bytes,
                                    sage_stream_t stream) {
  return run(0, q, k, v, dtype, o, lse, B, Hq, Hk, M, N, D, is_causal, sm_scale, 0.f, opts, workspace, workspace_bytes, stream);
}

extern "C" int sage_sageattn_pv_f8(const sage_tensor* q, const sage_tensor* k, const sage_tensor* v, int dtype,
                                   const sage_tensor* o, float* lse, int B, int Hq, int Hk, int M, int N, int D, int is_causal,
                                   float sm_scale, float scale_max, const sage_op_opts* opts, void* workspace,
                                   size_t workspace_bytes, sage_stream_t stream) {
  if (!(scale_max > 0.f)) return SAGE_ERR_INVALID_ARGUMENT;
  return run(1, q, k, v, dtype, o, lse, B, Hq, Hk, M, N, D, is_causal, sm_scale, scale_max, opts, workspace, workspace_bytes, stream);
}
