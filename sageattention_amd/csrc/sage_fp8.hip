// K3: FP8 (OCP e4m3fn) per-channel V quantizer for gfx950.
// Replaces TransposePadPermuteKernel + MeanScaleKernel (csrc/fused/fused.cu:262-427, quant.py:225-322) with
//   pass 1  per-channel max / min / sum over the sequence (deterministic two-level reduction, no atomics)
//   pass 2  (x - mean) * scale_max/amax -> e4m3 (RNE, saturating), transposed to [d][token] through LDS
// HBM-bound; algorithmic traffic 2 B + 2 B read, 1 B written per element (the reference: ~7 B/element through its
// fp16 transposed temporary).
//
// Token order inside each 64-token block ("MFMA order"): the PV product O^T += V^T . P^T runs on
// v_mfma_scale_f32_32x32x64_f8f6f4 whose B operand is P^T straight out of the S^T accumulators: lane half h, byte j
// (j = 16*mt + reg) carries key kv(h,j) = 32*(j>>4) + (j&3) + 8*((j&15)>>2) + 4*h.  Storing token kv(pos>>5, pos&31) at
// position pos makes every A fragment 32 contiguous bytes of a V^T row.  This plays the role of the reference's
// NVIDIA-fragment permutation (quant.py:234) for the gfx950 fragment; the layout is private to this library.
#include "sage_common.h"
#include "sage_fp8_kernels.h"

namespace sage {

template <int D, bool BF16>
__global__ __launch_bounds__(256) void v_stats_partial_kernel(const uint16_t* __restrict__ v, int64_t sb, int64_t sh,
                                                              int64_t sn, int N, float* __restrict__ part, int S) {
  constexpr int TPR = D / 8, RPP = 256 / TPR;
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const uint16_t* base = v + b * sb + h * sh + tc * 8;
  float mx[8], mn[8], sm[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mx[j] = -1000000.0f; mn[j] = 1000000.0f; sm[j] = 0.f; }  // fused.cu:345-347
  if ((s + 1) * VQ_ROWS <= N) {
    // whole chunk: unconditional loads, eight in flight (hipcc sinks a conditional load into its branch and waits for it
    // before the next one is issued: see k_mean_partial_body); same operations in the same order
#pragma unroll 8
    for (int i = 0; i < VQ_ROWS / RPP; ++i) {
      float f[8];
      unpack8<BF16>(*reinterpret_cast<const uint4*>(base + (int64_t)(s * VQ_ROWS + i * RPP + tr) * sn), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { mx[j] = fmaxf(mx[j], f[j]); mn[j] = fminf(mn[j], f[j]); sm[j] += f[j]; }
    }
  } else {
#pragma unroll 4
    for (int i = 0; i < VQ_ROWS / RPP; ++i) {
      const int row = s * VQ_ROWS + i * RPP + tr;
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tokens in [N, ceil16(N)) count as zeros (fused.cu:335)
      const int n16 = (N + 15) / 16 * 16;
      if (row < n16) {
        if (row < N) unpack8<BF16>(*reinterpret_cast<const uint4*>(base + (int64_t)row * sn), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = fmaxf(mx[j], f[j]); mn[j] = fminf(mn[j], f[j]); sm[j] += f[j]; }
      }
    }
  }
  __shared__ float red[3][RPP][D + 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][tr][tc * 8 + j] = mx[j]; red[1][tr][tc * 8 + j] = mn[j]; red[2][tr][tc * 8 + j] = sm[j]; }
  __syncthreads();
  if (threadIdx.x < D) {
    float a = red[0][0][threadIdx.x], c = red[1][0][threadIdx.x], e = red[2][0][threadIdx.x];
    for (int r = 1; r < RPP; ++r) {
      a = fmaxf(a, red[0][r][threadIdx.x]); c = fminf(c, red[1][r][threadIdx.x]); e += red[2][r][threadIdx.x];
    }
    float* o = part + ((((int64_t)b * gridDim.y + h) * S + s) * 3) * D + threadIdx.x;
    o[0] = a; o[D] = c; o[2 * D] = e;
  }
}

// per (b,h): finish the reduction; v_scale = amax/scale_max; keep mean and scale_max/amax for pass 2
__global__ void v_stats_final_kernel(const float* __restrict__ part, int S, int D, int N, float scale_max, int smooth,
                                     float* __restrict__ v_scale, float* __restrict__ v_mean, float* __restrict__ coef) {
  const int64_t bh = blockIdx.x;
  const int d = threadIdx.x;
  if (d >= D) return;
  float a = -1000000.0f, c = 1000000.0f, e = 0.f;
  // 8 chunks' partials in flight, combined in chunk order (a plain loop pays one memory round trip per chunk)
  for (int s0 = 0; s0 < S; s0 += 8) {
    float va[8], vc[8], ve[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = min(s0 + u, S - 1);
      const float* q = part + ((bh * S + s) * 3) * D + d;
      va[u] = q[0]; vc[u] = q[D]; ve[u] = q[2 * D];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < S) { a = fmaxf(a, va[u]); c = fminf(c, vc[u]); e += ve[u]; }
  }
  const int n16 = (N + 15) / 16 * 16;
  const float mean = smooth ? e / (float)n16 : 0.f;  // fused.cu:381: divides by the 16-padded token count
  const float amax = smooth ? fmaxf(fabsf(a - mean), fabsf(c - mean)) : fmaxf(fabsf(a), fabsf(c));
  v_scale[bh * D + d] = amax / scale_max;
  if (smooth) v_mean[bh * D + d] = mean;
  coef[(bh * 2) * D + d] = mean;
  coef[(bh * 2 + 1) * D + d] = scale_max / amax;
}

// Pass 2: quantize + transpose (body: sage_fp8_kernels.h)
template <int D, bool BF16>
__global__ __launch_bounds__(256) void v_quant_transpose_kernel(const uint16_t* __restrict__ v, int64_t sb, int64_t sh,
                                                                int64_t sn, int N, const float* __restrict__ coef,
                                                                uint8_t* __restrict__ out, int64_t ob, int64_t oh,
                                                                int64_t od, int64_t o_tile) {
  using G = VQuantGeom<D>;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int tc = threadIdx.x % G::TPR;
  __shared__ __attribute__((aligned(16))) uint32_t tile[G::BLKS * G::IMG];
  float mean[8], rcp[8];
  {
    const float* cf = coef + (((int64_t)b * H + h) * 2) * D + tc * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = cf[j]; rcp[j] = cf[D + j]; }
  }
  v_quant_transpose_body<D, BF16>(v, sb, sh, sn, N, mean, rcp, out, ob, oh, od, o_tile, blockIdx.x, h, b, tile);
}

// raw per-channel statistics (max, min, sum) of one tensor: second level of v_stats_partial_kernel
__global__ void seq_stats_final_kernel(const float* __restrict__ part, int S, int D, float* __restrict__ stats) {
  const int64_t bh = blockIdx.x;
  const int d = threadIdx.x;
  if (d >= D) return;
  float a = -1000000.0f, c = 1000000.0f, e = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {
    float va[8], vc[8], ve[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = min(s0 + u, S - 1);
      const float* q = part + ((bh * S + s) * 3) * D + d;
      va[u] = q[0]; vc[u] = q[D]; ve[u] = q[2 * D];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < S) { a = fmaxf(a, va[u]); c = fminf(c, vc[u]); e += ve[u]; }
  }
  float* o = stats + bh * 3 * D + d;
  o[0] = a; o[D] = c; o[2 * D] = e;
}

// statistics of `parts` shards -> km (whole-sequence mean, storage dtype), v_scale, v_coef; fixed order over the shards
template <bool BF16>
__global__ void kv_stats_reduce_kernel(const float* __restrict__ ks, const float* __restrict__ vs, int parts,
                                       int64_t part_stride, int BH, int D,
                                       float n_total, float scale_max, uint16_t* __restrict__ km, float* __restrict__ v_scale,
                                       float* __restrict__ v_coef) {
  const int64_t bh = blockIdx.x;
  const int d = threadIdx.x;
  if (d >= D) return;
  if (ks && km) {
    float sum = 0.f;
    for (int p = 0; p < parts; ++p) sum += ks[p * part_stride + (bh * 3 + 2) * D + d];
    km[bh * D + d] = f32_to_elem_bits<BF16>(sum / n_total);
  }
  if (vs) {
    float a = -1000000.0f, c = 1000000.0f;
    for (int p = 0; p < parts; ++p) {
      const float* q = vs + p * part_stride + (bh * 3) * D + d;
      a = fmaxf(a, q[0]); c = fminf(c, q[D]);
    }
    const float amax = fmaxf(fabsf(a), fabsf(c));
    v_scale[bh * D + d] = amax / scale_max;
    v_coef[(bh * 2) * D + d] = 0.f;
    v_coef[(bh * 2 + 1) * D + d] = scale_max / amax;
  }
}

}  // namespace sage

using namespace sage;

extern "C" size_t sage_quant_v_fp8_workspace_bytes(int B, int H, int N, int D) {
  const size_t S = (size_t)(N + VQ_ROWS - 1) / VQ_ROWS;
  return ((size_t)B * H * S * 3 * D + (size_t)B * H * 2 * D) * sizeof(float);
}

extern "C" int sage_quant_v_fp8(const sage_tensor* v, int dtype, int B, int H, int N, int D, const sage_tensor* v_fp8,
                                float* v_scale, float* v_mean, float scale_max, void* workspace, sage_stream_t stream) {
  if (!v || !v->data || !aligned16(v->data) || v->stride_b % 8 || v->stride_h % 8 || v->stride_n % 8) return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_fp8 || !v_fp8->data || !aligned16(v_fp8->data) || v_fp8->stride_b % 16 || v_fp8->stride_h % 16 || v_fp8->stride_n % 16)
    return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_scale || !workspace || B <= 0 || H <= 0 || N <= 0 || !(scale_max > 0.f)) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int S = (N + VQ_ROWS - 1) / VQ_ROWS;
  float* part = (float*)workspace;
  float* coef = part + (size_t)B * H * S * 3 * D;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const uint16_t* vp = (const uint16_t*)v->data;
  const int vq_blks = D == 128 ? 1 : 2;  // 64-token blocks per workgroup of v_quant_transpose_kernel
  const dim3 g1(S, H, B), g2(((N + 63) / 64 + vq_blks - 1) / vq_blks, H, B);
#define L1(DD, BF) hipLaunchKernelGGL((v_stats_partial_kernel<DD, BF>), g1, dim3(256), 0, st, vp, v->stride_b, v->stride_h, v->stride_n, N, part, S)
#define L2(DD, BF)                                                                                                    \
  hipLaunchKernelGGL((v_quant_transpose_kernel<DD, BF>), g2, dim3(256), 0, st, vp, v->stride_b, v->stride_h, v->stride_n, N, \
                     coef, (uint8_t*)v_fp8->data, v_fp8->stride_b, v_fp8->stride_h, v_fp8->stride_n, (int64_t)64)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) L1(64, true); else L1(64, false); } else { if (bf) L1(128, true); else L1(128, false); }
  hipLaunchKernelGGL(v_stats_final_kernel, dim3(B * H), dim3(128), 0, st, part, S, D, N, scale_max, v_mean ? 1 : 0, v_scale,
                     v_mean, coef);
  if (D == 64) { if (bf) L2(64, true); else L2(64, false); } else { if (bf) L2(128, true); else L2(128, false); }
#undef L1
#undef L2
  return launch_status();
}

extern "C" size_t sage_seq_stats_workspace_bytes(int B, int H, int N, int D) {
  const size_t S = (size_t)(N + VQ_ROWS - 1) / VQ_ROWS;
  return (size_t)B * H * S * 3 * D * sizeof(float);
}

extern "C" int sage_seq_stats(const sage_tensor* x, int dtype, int B, int H, int N, int D, float* stats, void* workspace,
                              sage_stream_t stream) {
  if (!x || !x->data || !aligned16(x->data) || x->stride_b % 8 || x->stride_h % 8 || x->stride_n % 8) return SAGE_ERR_INVALID_ARGUMENT;
  if (!stats || !workspace || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int S = (N + VQ_ROWS - 1) / VQ_ROWS;
  float* part = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const uint16_t* xp = (const uint16_t*)x->data;
  const dim3 g1(S, H, B);
  // rows in [N, ceil16(N)) would count as zeros in v_stats_partial_kernel (the reference's padded amax, fused.cu:335):
  // harmless for max|x| and for the sum
#define L1(DD, BF) hipLaunchKernelGGL((v_stats_partial_kernel<DD, BF>), g1, dim3(256), 0, st, xp, x->stride_b, x->stride_h, x->stride_n, N, part, S)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) L1(64, true); else L1(64, false); } else { if (bf) L1(128, true); else L1(128, false); }
#undef L1
  hipLaunchKernelGGL(seq_stats_final_kernel, dim3(B * H), dim3(128), 0, st, part, S, D, stats);
  return launch_status();
}

extern "C" int sage_kv_stats_reduce(const float* k_stats, const float* v_stats, int parts, int64_t part_stride, int BH, int D, int64_t n_total,
                                    int dtype, float scale_max, void* km, float* v_scale, float* v_coef, sage_stream_t stream) {
  if ((!k_stats && !v_stats) || parts <= 0 || BH <= 0 || n_total <= 0 || part_stride < (int64_t)BH * 3 * D) return SAGE_ERR_INVALID_ARGUMENT;
  if ((k_stats != nullptr) != (km != nullptr)) return SAGE_ERR_INVALID_ARGUMENT;
  if (v_stats && (!v_scale || !v_coef || !(scale_max > 0.f))) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  launch_begin();
  if (dtype == SAGE_BF16)
    hipLaunchKernelGGL((kv_stats_reduce_kernel<true>), dim3(BH), dim3(128), 0, (hipStream_t)stream, k_stats, v_stats, parts, part_stride, BH,
                       D, (float)n_total, scale_max, (uint16_t*)km, v_scale, v_coef);
  else
    hipLaunchKernelGGL((kv_stats_reduce_kernel<false>), dim3(BH), dim3(128), 0, (hipStream_t)stream, k_stats, v_stats, parts, part_stride, BH,
                       D, (float)n_total, scale_max, (uint16_t*)km, v_scale, v_coef);
  return launch_status();
}

extern "C" int sage_quant_v_fp8_apply(const sage_tensor* v, int dtype, int B, int H, int N, int D, const sage_tensor* v_fp8,
                                      int64_t out_tile_stride, const float* v_coef, sage_stream_t stream) {
  if (!v || !v->data || !aligned16(v->data) || v->stride_b % 8 || v->stride_h % 8 || v->stride_n % 8) return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_fp8 || !v_fp8->data || !aligned16(v_fp8->data) || v_fp8->stride_b % 16 || v_fp8->stride_h % 16 || v_fp8->stride_n % 16)
    return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_coef || B <= 0 || H <= 0 || N <= 0 || out_tile_stride < 0 || (out_tile_stride & 15)) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int64_t o_tile = out_tile_stride ? out_tile_stride : 64;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const uint16_t* vp = (const uint16_t*)v->data;
  const int vq_blks = D == 128 ? 1 : 2;
  const dim3 g2(((N + 63) / 64 + vq_blks - 1) / vq_blks, H, B);
#define L2(DD, BF)                                                                                                    \
  hipLaunchKernelGGL((v_quant_transpose_kernel<DD, BF>), g2, dim3(256), 0, st, vp, v->stride_b, v->stride_h, v->stride_n, N, \
                     v_coef, (uint8_t*)v_fp8->data, v_fp8->stride_b, v_fp8->stride_h, v_fp8->stride_n, o_tile)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) L2(64, true); else L2(64, false); } else { if (bf) L2(128, true); else L2(128, false); }
#undef L2
  return launch_status();
}
