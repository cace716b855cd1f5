// Small helpers of the hot path: library info, LSE finishing (core.py:651) and the ring-attention
// state merge (new component; rule in SURVEY.md section 5).  Elementwise, HBM-bound.
#include "sage_common.h"

namespace sage {

template <bool BF16>
__global__ __launch_bounds__(256) void merge_states_kernel(float* __restrict__ o_acc, float* __restrict__ lse_acc,
                                                           const uint16_t* __restrict__ o_blk,
                                                           const float* __restrict__ lse_blk, int64_t rows, int D) {
  // one thread per 8 output elements; D/8 threads per row
  const int tpr = D / 8;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = gid / tpr;
  const int c = (int)(gid % tpr);
  if (row >= rows) return;
  const float la = lse_acc[row], lb = lse_blk[row];
  const float mx = fmaxf(la, lb);
  // logaddexp; la = -inf (empty accumulator) gives wa = 0, wb = 1
  const float ea = (la == -INFINITY) ? 0.f : __expf(la - mx), eb = (lb == -INFINITY) ? 0.f : __expf(lb - mx);
  const float sum = ea + eb;
  const float lse = (sum > 0.f) ? mx + __logf(sum) : -INFINITY;
  const float wa = (sum > 0.f) ? ea / sum : 0.f, wb = (sum > 0.f) ? eb / sum : 0.f;
  float* oa = o_acc + row * D + c * 8;
  float fb[8];
  unpack8<BF16>(*reinterpret_cast<const uint4*>(o_blk + row * D + c * 8), fb);
  float4 a0 = *reinterpret_cast<float4*>(oa), a1 = *reinterpret_cast<float4*>(oa + 4);
  a0.x = a0.x * wa + fb[0] * wb; a0.y = a0.y * wa + fb[1] * wb; a0.z = a0.z * wa + fb[2] * wb; a0.w = a0.w * wa + fb[3] * wb;
  a1.x = a1.x * wa + fb[4] * wb; a1.y = a1.y * wa + fb[5] * wb; a1.z = a1.z * wa + fb[6] * wb; a1.w = a1.w * wa + fb[7] * wb;
  *reinterpret_cast<float4*>(oa) = a0;
  *reinterpret_cast<float4*>(oa + 4) = a1;
  __syncthreads();  // all threads of a row have read lse_acc[row] (rows never straddle a block: 256 % tpr == 0)
  if (c == 0) lse_acc[row] = lse;
}

// Multi-way merge: (o, lse) = merge of up to SAGE_MERGE_MAX block results in ONE pass -- a ring step over P shards
// otherwise reads and writes the fp32 accumulator P times (sage_merge_attn_states per block).
struct MergeMany {
  const uint16_t* o[SAGE_MERGE_MAX];
  const float* lse[SAGE_MERGE_MAX];
  int count;
};
// MAXC: compile-time bound of `count` (2, 4, 8 or 16): the slots are unrolled
template <bool BF16, int MAXC>
__global__ __launch_bounds__(256) void merge_many_kernel(const MergeMany m, uint16_t* __restrict__ o_out,
                                                         float* __restrict__ lse_out, int64_t rows, int D,
                                                         const float in_mult, const float* __restrict__ corr,
                                                         const float corr_mult) {
  const int tpr = D / 8;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = gid / tpr;
  const int c = (int)(gid % tpr);
  if (row >= rows) return;
  // every lse and every o row of this thread is requested up front (slots unrolled; a slot past `count` repeats the last
  // one and is not used): as a run-time loop with the o load under `if (l != -inf)` each block cost two dependent round
  // trips -- hipcc sinks a load into the branch that uses it.  The uses below are selects, not branches, for that reason.
  // The combination runs in slot order as before: deterministic, bit-identical.
  float l[MAXC];
  uint4 raw[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int s = i < m.count ? i : m.count - 1;
    l[i] = m.lse[s][row] * in_mult;
    raw[i] = *reinterpret_cast<const uint4*>(m.o[s] + row * D + c * 8);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) mx = i < m.count ? fmaxf(mx, l[i]) : mx;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {  // fixed order: deterministic
    const bool use = i < m.count && l[i] != -INFINITY;  // empty block (e.g. fully masked): weight 0, its o may be anything
    const float w = __expf(l[i] - mx);
    sum = use ? sum + w : sum;
    float f[8];
    unpack8<BF16>(raw[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = use ? acc[j] + f[j] * w : acc[j];
  }
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    w[j] = (uint32_t)f32_to_elem_bits<BF16>(acc[2 * j] * inv) | ((uint32_t)f32_to_elem_bits<BF16>(acc[2 * j + 1] * inv) << 16);
  *reinterpret_cast<uint4*>(o_out + row * D + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  if (c == 0 && lse_out) lse_out[row] = (sum > 0.f ? mx + __logf(sum) : -INFINITY) + (corr ? corr[row] * corr_mult : 0.f);
}

__global__ __launch_bounds__(256) void finish_lse_kernel(const float* __restrict__ lse2, const float* __restrict__ corr,
                                                         float sm_scale, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = lse2[i] / 1.44269504f;  // core.py:651
  if (corr) v += corr[i] * sm_scale;
  out[i] = v;
}

}  // namespace sage

using namespace sage;

extern "C" int sage_abi_version(void) { return SAGEATTN_HIP_ABI_VERSION; }
extern "C" const char* sage_target_arch(void) { return "gfx950"; }
extern "C" const char* sage_status_string(int status) {
  switch (status) {
    case SAGE_OK: return "ok";
    case SAGE_ERR_INVALID_ARGUMENT: return "invalid argument (null/unaligned pointer, bad enum or inconsistent sizes)";
    case SAGE_ERR_UNSUPPORTED_HEAD_DIM: return "unsupported head_dim (must be 64 or 128 after padding)";
    case SAGE_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case SAGE_ERR_TOO_LARGE: return "tensor slice too large";
    case SAGE_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown status";
  }
}

extern "C" int sage_merge_attn_states(float* o_acc, float* lse_acc, const void* o_blk, int o_dtype, const float* lse_blk,
                                      int64_t rows, int D, sage_stream_t stream) {
  if (!o_acc || !lse_acc || !o_blk || !lse_blk || rows <= 0 || !aligned16(o_acc) || !aligned16(o_blk)) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (o_dtype != SAGE_F16 && o_dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int64_t threads = rows * (D / 8);
  const dim3 grid((unsigned)((threads + 255) / 256));
  launch_begin();
  if (o_dtype == SAGE_BF16)
    hipLaunchKernelGGL((merge_states_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, o_acc, lse_acc, (const uint16_t*)o_blk, lse_blk, rows, D);
  else
    hipLaunchKernelGGL((merge_states_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, o_acc, lse_acc, (const uint16_t*)o_blk, lse_blk, rows, D);
  return launch_status();
}

extern "C" int sage_merge_attn_states_multi_ex(const void* const* o_blks, const float* const* lse_blks, int count, int o_dtype,
                                               void* o_out, float* lse_out, int64_t rows, int D, float lse_in_mult,
                                               const float* corr, float corr_mult, sage_stream_t stream) {
  if (!(lse_in_mult > 0.f)) return SAGE_ERR_INVALID_ARGUMENT;
  if (!o_blks || !lse_blks || !o_out || count <= 0 || count > SAGE_MERGE_MAX || rows <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (o_dtype != SAGE_F16 && o_dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  MergeMany m;
  m.count = count;
  for (int i = 0; i < count; ++i) {
    if (!o_blks[i] || !lse_blks[i] || !aligned16(o_blks[i])) return SAGE_ERR_INVALID_ARGUMENT;
    m.o[i] = (const uint16_t*)o_blks[i];
    m.lse[i] = lse_blks[i];
  }
  for (int i = count; i < SAGE_MERGE_MAX; ++i) { m.o[i] = nullptr; m.lse[i] = nullptr; }
  if (!aligned16(o_out)) return SAGE_ERR_INVALID_ARGUMENT;
  const int64_t threads = rows * (D / 8);
  const dim3 grid((unsigned)((threads + 255) / 256));
  launch_begin();
#define LAUNCH(BF, MC)                                                                                                  \
  hipLaunchKernelGGL((merge_many_kernel<BF, MC>), grid, dim3(256), 0, (hipStream_t)stream, m, (uint16_t*)o_out, lse_out, rows, D, \
                     lse_in_mult, corr, corr_mult)
#define BY_COUNT(BF)                                                                                                    \
  do {                                                                                                                  \
    if (count <= 2) LAUNCH(BF, 2); else if (count <= 4) LAUNCH(BF, 4); else if (count <= 8) LAUNCH(BF, 8); else LAUNCH(BF, 16); \
  } while (0)
  if (o_dtype == SAGE_BF16) BY_COUNT(true); else BY_COUNT(false);
#undef BY_COUNT
#undef LAUNCH
  return launch_status();
}

extern "C" int sage_merge_attn_states_multi(const void* const* o_blks, const float* const* lse_blks, int count, int o_dtype,
                                           void* o_out, float* lse_out, int64_t rows, int D, sage_stream_t stream) {
  return sage_merge_attn_states_multi_ex(o_blks, lse_blks, count, o_dtype, o_out, lse_out, rows, D, 1.0f, nullptr, 0.f, stream);
}

extern "C" int sage_finish_lse(const float* lse2, const float* corr, float sm_scale, float* lse_out, int64_t n,
                               sage_stream_t stream) {
  if (!lse2 || !lse_out || n <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  launch_begin();
  hipLaunchKernelGGL(finish_lse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lse2, corr, sm_scale, lse_out, n);
  return launch_status();
}
