// Single-pass pre-pass kernels for gfx950: K smoothing + INT8 quantization, and the per-channel FP8 V
// quantizer, each as ONE launch that reads its input ONCE.
//
//   sage_k_prep      replaces  k.mean(seq) (core.py:612)  +  quant_per_block_int8_fuse_sub_mean_cuda / the K half of
//                    per_thread_int8 (fused.cu:594-682, triton/quant_per_thread.py:48-102): 2 B read + 1 B written per
//                    element instead of 4 B + 1 B over three launches.
//   sage_v_prep_fp8  replaces  transpose_pad_permute_cuda + [mean_]scale_fuse_quant_cuda (fused.cu:850-1083): 2 B + 1 B
//                    instead of 4 B + 1 B over three launches (the reference: ~7 B through its fp16 transposed copy).
//
// Both need a statistic over the WHOLE sequence (column mean / max, min, sum) before the first element can be quantized.
// A workgroup owns 256 consecutive rows of one (b, h) and keeps them in registers; the workgroups of a (b, h) hand their
// partial statistics to each other through global memory:
//     partials (plain stores) -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: agent-scope release, wait, counter += 1
//     lane 0 polls the counter (relaxed, sc1) until all C chunks of the (b, h) have arrived -> agent-scope acquire -> barrier
//     every workgroup reduces the C partials in the SAME fixed order -> identical bits everywhere, and identical to the
//     two-level reductions of sage_k_mean / sage_quant_v_fp8 (same chunking, same order)
// (MI355X_MICROARCH.md "inter-workgroup visibility": per-XCD L2s are not coherent, hence release / acquire at agent scope.)
// Forward progress does NOT depend on dispatch order or co-residency: the poll is bounded, and a workgroup whose poll runs
// out recomputes the missing partials itself from the input (same routine, same bits).  With the in-order dispatch the
// hardware shows, the chunks of a (b, h) are co-resident (consecutive block ids) and the fallback never runs.
// Roofline: HBM.  Compiled with -ffp-contract=off (bit-exact vs the oracle and vs the multi-launch path).
#include <type_traits>
#include "sage_common.h"

namespace sage {

constexpr int PREP_ROWS = 256;           // rows per workgroup = KMEAN_ROWS = VQ_ROWS of the multi-launch kernels
constexpr int PREP_POLL_LIMIT = 1 << 16;  // polls (each >= ~0.3 us) before a workgroup helps itself

__device__ __forceinline__ void wait_vmem() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// lane 0 of the workgroup: publish this workgroup's stores, then wait for the other chunks of the group.
// Returns (to every thread, through `flag`) whether all `need` arrivals were seen.
__device__ __forceinline__ bool cluster_handoff(unsigned* counter, const unsigned need, unsigned* flag, const int poll_limit) {
  wait_vmem();       // every wave: its partial stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wait_vmem();     // keep: the compiler may drop the fence's own wait (guide, compiler hazard)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned ok = 0;
    for (int it = 0; it < poll_limit; ++it) {
      if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    wait_vmem();
    *flag = ok;
  }
  __syncthreads();
  return *flag != 0;
}

// ------------------------------------------------------------------------------------------------
// K: mean + subtract + INT8 quantization (per_block or per_thread K granularity, 64-row blocks)
// ------------------------------------------------------------------------------------------------
struct KPrepParams {
  const uint16_t* x; int64_t xsb, xsh, xsn;
  int8_t* out; int64_t osb, osh, osn, o_blk;
  float* scale; int64_t ss_b, ss_h, ss_blk;
  uint16_t* km;      // [B,H,D] out (storage dtype)
  float* part;       // [B*H][C][D]
  unsigned* count;   // [B*H], zeroed before the launch
  int H, N, C, per_thread, rounding, poll_limit;
};

template <int D, bool BF16>
__global__ __launch_bounds__(256) void k_prep_kernel(const KPrepParams p) {
  constexpr int TPR = D / 8, RPP = 256 / TPR, NP = PREP_ROWS / RPP, NPB = 64 / RPP;  // passes per chunk / per 64-row block
  const int bh = blockIdx.x / p.C, c = blockIdx.x % p.C;
  const int b = bh / p.H, h = bh % p.H;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const uint16_t* xbase = p.x + b * p.xsb + h * p.xsh + tc * 8;
  __shared__ float red[RPP][D + 1];
  __shared__ unsigned gmax[16];
  __shared__ unsigned flag;

  // column sums of one 256-row chunk, in the order of k_mean_partial_kernel; result in red[0][0..D) (all threads call)
  // (the chunk's own rows come from the registers: `raw` must never have its address taken, or it lives in scratch)
  uint4 raw[NP];
  auto chunk_colsum = [&](const int chunk, auto own_tag) __attribute__((always_inline)) {
    constexpr bool OWN = decltype(own_tag)::value;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = chunk * PREP_ROWS + i * RPP + tr;
      if (row < p.N) {
        uint4 u;
        if constexpr (OWN) u = raw[i]; else u = *reinterpret_cast<const uint4*>(xbase + (int64_t)row * p.xsn);
        float f[8];
        unpack8<BF16>(u, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    }
    __syncthreads();  // red free
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tr][tc * 8 + j] = acc[j];
    __syncthreads();
    float sum = 0.f;
    if (threadIdx.x < D)
      for (int r = 0; r < RPP; ++r) sum += red[r][threadIdx.x];  // fixed order
    __syncthreads();
    if (threadIdx.x < D) red[0][threadIdx.x] = sum;
    __syncthreads();
  };

  // ---- this chunk: one read of its rows, kept in registers
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = c * PREP_ROWS + i * RPP + tr;
    raw[i] = make_uint4(0, 0, 0, 0);
    if (row < p.N) raw[i] = *reinterpret_cast<const uint4*>(xbase + (int64_t)row * p.xsn);
  }
  if (threadIdx.x < 16) gmax[threadIdx.x] = 0u;
  chunk_colsum(c, std::true_type{});
  float* part = p.part + (int64_t)bh * p.C * D;
  if (threadIdx.x < D) part[c * D + threadIdx.x] = red[0][threadIdx.x];
  const bool all_in = cluster_handoff(p.count + bh, (unsigned)p.C, &flag, p.poll_limit);

  // ---- whole-sequence mean: the C partials in chunk order (k_mean_final_kernel)
  float total = 0.f;
  if (all_in) {
    if (threadIdx.x < D)
      for (int s = 0; s < p.C; ++s) total += part[s * D + threadIdx.x];
  } else {  // never with in-order dispatch: some chunk has not arrived in time -- recompute them all here, same bits
    for (int s = 0; s < p.C; ++s) {
      if (s == c) chunk_colsum(s, std::true_type{}); else chunk_colsum(s, std::false_type{});
      if (threadIdx.x < D) total += red[0][threadIdx.x];
    }
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const uint16_t kmb = f32_to_elem_bits<BF16>(total / (float)p.N);
    red[1][threadIdx.x] = elem_to_f32<BF16>(kmb);
    if (c == 0) p.km[(int64_t)bh * D + threadIdx.x] = kmb;
  }
  __syncthreads();
  float mean_f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mean_f[j] = red[1][tc * 8 + j];

  // ---- quantize the chunk's (up to) four 64-row blocks: arithmetic of quant_qk_int8_kernel, is_key = 1, BLK = 64
  const bool triton = p.rounding == SAGE_ROUND_TRITON;
  float xf[NP][8];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int lr = (i % NPB) * RPP + tr;  // row inside its 64-row block
    const int row = c * PREP_ROWS + i * RPP + tr;
    const bool valid = row < p.N;
    unpack8<BF16>(raw[i], xf[i]);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = xf[i][j] - mean_f[j];
      if (triton) v = round_to_elem<BF16>(v);  // torch `k - km` in the input dtype
      v = v * (valid ? 1.0f : 0.f);
      xf[i][j] = v;
      amax = fmaxf(amax, fabsf(v));
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const int grp = p.per_thread ? (lr & 7) >> 1 : 0;
    if (tc == 0) atomicMax(&gmax[(i / NPB) * 4 + grp], __float_as_uint(amax));
  }
  __syncthreads();
  const int gpb = p.per_thread ? 4 : 1;
  const float eps = p.per_thread ? 0.0000001f : 0.f;
  if (threadIdx.x < 4 * gpb) {
    const int blk_in = threadIdx.x / gpb, g = threadIdx.x % gpb;
    const int blk = c * 4 + blk_in;
    if (blk * 64 < p.N) {
      const float a = __uint_as_float(gmax[blk_in * 4 + g]);
      p.scale[b * p.ss_b + h * p.ss_h + blk * p.ss_blk + g] = triton ? a / 127.f + eps : fmaxf(a, 0.0000001f) / 127.f;
    }
  }
  int8_t* obase = p.out + b * p.osb + h * p.osh + tc * 8;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int lr = (i % NPB) * RPP + tr;
    const int blk = c * 4 + i / NPB;
    const int row = blk * 64 + lr;
    const int grp = p.per_thread ? (lr & 7) >> 1 : 0;
    const float a = __uint_as_float(gmax[(i / NPB) * 4 + grp]);
    int q[8];
    if (triton) {
      const float sc = a / 127.f + eps;
      const float r = 1.0f / sc;
      bool near = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ya = xf[i][j] * r;
        const float f = __builtin_amdgcn_fractf(fabsf(ya) + 0.5f);
        near |= fabsf(f - 0.5f) > 0.5f - 6.1035156e-5f;
        q[j] = (int)(ya + __builtin_copysignf(0.5f, ya));
      }
      if (__builtin_amdgcn_ballot_w64(near || !(fabsf(r) < 3.0e38f)) != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = xf[i][j] / sc;  // IEEE division (quant_per_thread.py:41)
          y = y + (y >= 0.f ? 0.5f : -0.5f);
          q[j] = (int)y;
        }
      }
    } else {
      const float inv = 127.f / fmaxf(a, 0.0000001f);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (int)rintf(xf[i][j] * inv);  // cvt.rni (fused.cu:176-181)
    }
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w0 |= (uint32_t)(min(max(q[j], -128), 127) & 0xff) << (8 * j);
      w1 |= (uint32_t)(min(max(q[4 + j], -128), 127) & 0xff) << (8 * j);
    }
    if (row < p.N) *reinterpret_cast<uint2*>(obase + blk * p.o_blk + (int64_t)lr * p.osn) = make_uint2(w0, w1);
  }
}

// ------------------------------------------------------------------------------------------------
// V: per-channel statistics + FP8 quantization + transpose to the MFMA-order V^T layout (sage_fp8.hip)
// ------------------------------------------------------------------------------------------------
struct VPrepParams {
  const uint16_t* v; int64_t sb, sh, sn;
  uint8_t* out; int64_t ob, oh, od, o_tile;
  float* v_scale; float* v_mean;  // [B,H,D]; v_mean null = no smoothing
  float* part;                    // [B*H][C][3][D]
  unsigned* count;
  int H, N, C, poll_limit;
  float scale_max;
};

template <int D, bool BF16>
__global__ __launch_bounds__(256) void v_prep_kernel(const VPrepParams p) {
  constexpr int TPR = D / 8, RPP = 256 / TPR, NP = PREP_ROWS / RPP, NPB = 64 / RPP;
  const int bh = blockIdx.x / p.C, c = blockIdx.x % p.C;
  const int b = bh / p.H, h = bh % p.H;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const uint16_t* vbase = p.v + b * p.sb + h * p.sh + tc * 8;
  // one LDS block, used first for the reductions (3 x RPP x (D+1) floats), then as the [d][pos] byte tile
  constexpr int RED_BYTES = 3 * RPP * (D + 1) * 4, TILE_BYTES = D * 80;
  __shared__ __attribute__((aligned(16))) char lds[RED_BYTES > TILE_BYTES ? RED_BYTES : TILE_BYTES];
  __shared__ float stat[3][D];
  __shared__ unsigned flag;
  float (*red)[RPP][D + 1] = reinterpret_cast<float (*)[RPP][D + 1]>(lds);
  const int n16 = (p.N + 15) / 16 * 16;

  // (max, min, sum) of one chunk in the order of v_stats_partial_kernel; result in stat[][]
  uint4 raw[NP];
  auto chunk_stats = [&](const int chunk, auto own_tag) __attribute__((always_inline)) {
    constexpr bool OWN = decltype(own_tag)::value;
    float mx[8], mn[8], sm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mx[j] = -1000000.0f; mn[j] = 1000000.0f; sm[j] = 0.f; }  // fused.cu:345-347
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = chunk * PREP_ROWS + i * RPP + tr;
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tokens in [N, ceil16(N)) count as zeros (fused.cu:335)
      if (row < n16) {
        if (row < p.N) {
          if constexpr (OWN) unpack8<BF16>(raw[i], f);
          else unpack8<BF16>(*reinterpret_cast<const uint4*>(vbase + (int64_t)row * p.sn), f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = fmaxf(mx[j], f[j]); mn[j] = fminf(mn[j], f[j]); sm[j] += f[j]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][tr][tc * 8 + j] = mx[j]; red[1][tr][tc * 8 + j] = mn[j]; red[2][tr][tc * 8 + j] = sm[j]; }
    __syncthreads();
    if (threadIdx.x < D) {
      float a = red[0][0][threadIdx.x], cc = red[1][0][threadIdx.x], e = red[2][0][threadIdx.x];
      for (int r = 1; r < RPP; ++r) {
        a = fmaxf(a, red[0][r][threadIdx.x]); cc = fminf(cc, red[1][r][threadIdx.x]); e += red[2][r][threadIdx.x];
      }
      stat[0][threadIdx.x] = a; stat[1][threadIdx.x] = cc; stat[2][threadIdx.x] = e;
    }
    __syncthreads();
  };

#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = c * PREP_ROWS + i * RPP + tr;
    raw[i] = make_uint4(0, 0, 0, 0);
    if (row < p.N) raw[i] = *reinterpret_cast<const uint4*>(vbase + (int64_t)row * p.sn);
  }
  chunk_stats(c, std::true_type{});
  float* part = p.part + (int64_t)bh * p.C * 3 * D;
  if (threadIdx.x < D) {
    float* o = part + c * 3 * D + threadIdx.x;
    o[0] = stat[0][threadIdx.x]; o[D] = stat[1][threadIdx.x]; o[2 * D] = stat[2][threadIdx.x];
  }
  const bool all_in = cluster_handoff(p.count + bh, (unsigned)p.C, &flag, p.poll_limit);

  // ---- whole-sequence statistics in chunk order (v_stats_final_kernel)
  float a = -1000000.0f, cc = 1000000.0f, e = 0.f;
  if (all_in) {
    if (threadIdx.x < D)
      for (int s = 0; s < p.C; ++s) {
        const float* q = part + s * 3 * D + threadIdx.x;
        a = fmaxf(a, q[0]); cc = fminf(cc, q[D]); e += q[2 * D];
      }
  } else {
    for (int s = 0; s < p.C; ++s) {
      if (s == c) chunk_stats(s, std::true_type{}); else chunk_stats(s, std::false_type{});
      if (threadIdx.x < D) { a = fmaxf(a, stat[0][threadIdx.x]); cc = fminf(cc, stat[1][threadIdx.x]); e += stat[2][threadIdx.x]; }
    }
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const bool smooth = p.v_mean != nullptr;
    const float mean = smooth ? e / (float)n16 : 0.f;  // fused.cu:381
    const float amax = smooth ? fmaxf(fabsf(a - mean), fabsf(cc - mean)) : fmaxf(fabsf(a), fabsf(cc));
    stat[0][threadIdx.x] = mean;
    stat[1][threadIdx.x] = p.scale_max / amax;
    if (c == 0) {
      p.v_scale[(int64_t)bh * D + threadIdx.x] = amax / p.scale_max;
      if (smooth) p.v_mean[(int64_t)bh * D + threadIdx.x] = mean;
    }
  }
  __syncthreads();
  float mean[8], rcp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = stat[0][tc * 8 + j]; rcp[j] = stat[1][tc * 8 + j]; }

  // ---- quantize + transpose, one 64-token block at a time (v_quant_transpose_kernel)
  uint8_t (*tile)[80] = reinterpret_cast<uint8_t (*)[80]>(lds);
  for (int kb = 0; kb < 4; ++kb) {
    const int blk = c * 4 + kb;
    if (blk * 64 >= p.N) break;  // uniform
    __syncthreads();  // tile free
#pragma unroll
    for (int ii = 0; ii < NPB; ++ii) {
      const int i = kb * NPB + ii;
      const int t = ii * RPP + tr;  // token within the block
      const int row = blk * 64 + t;
      float f[8];
      unpack8<BF16>(raw[i], f);
      const int mt = t >> 5, w = t & 31, hh = (w >> 2) & 1, reg = (w & 3) | ((w >> 3) << 2);
      const int pos = 32 * hh + 16 * mt + reg;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float x0 = row < p.N ? (f[j] - mean[j]) * rcp[j] : 0.f;
        const float x1 = row < p.N ? (f[j + 1] - mean[j + 1]) * rcp[j + 1] : 0.f;
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);  // OCP e4m3fn, RNE, saturating
        tile[tc * 8 + j][pos] = (uint8_t)(pk & 0xff);
        tile[tc * 8 + j + 1][pos] = (uint8_t)((pk >> 8) & 0xff);
      }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < D * 4; ch += 256) {
      const int d = ch >> 2, q4 = ch & 3;
      const uint4 u = *reinterpret_cast<const uint4*>(&tile[d][q4 * 16]);
      *reinterpret_cast<uint4*>(p.out + b * p.ob + h * p.oh + (int64_t)d * p.od + blk * p.o_tile + q4 * 16) = u;
    }
  }
}

static bool tensor_ok8(const sage_tensor* t) {
  return t && t->data && aligned16(t->data) && t->stride_b % 8 == 0 && t->stride_h % 8 == 0 && t->stride_n % 8 == 0;
}

}  // namespace sage

namespace sage { int g_prep_poll_limit = 0; }  // SAGE_TUNE_PREP_POLL (tests: 1 = every workgroup takes the self-help path)

using namespace sage;

// workspace: [counters: B*H u32, padded to 256 B][partials]
static size_t prep_counter_bytes(int B, int H) { return ((size_t)B * H * 4 + 255) / 256 * 256; }

extern "C" size_t sage_k_prep_workspace_bytes(int B, int H, int N, int D) {
  const size_t C = (size_t)(N + PREP_ROWS - 1) / PREP_ROWS;
  return prep_counter_bytes(B, H) + (size_t)B * H * C * D * sizeof(float);
}

extern "C" int sage_k_prep(const sage_tensor* k, int dtype, int B, int H, int N, int D, const sage_tensor* out, float* scale,
                           void* km, int gran, int rounding, void* workspace, sage_stream_t stream) {
  if (!tensor_ok8(k) || !tensor_ok8(out) || !scale || !km || !workspace || !aligned16(workspace) || B <= 0 || H <= 0 || N <= 0)
    return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran != SAGE_GRAN_PER_BLOCK && gran != SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;
  if (rounding != SAGE_ROUND_TRITON && rounding != SAGE_ROUND_CUDA) return SAGE_ERR_INVALID_ARGUMENT;
  const int C = (N + PREP_ROWS - 1) / PREP_ROWS;
  if ((int64_t)B * H * C >= ((int64_t)1 << 31)) return SAGE_ERR_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  KPrepParams p;
  p.x = (const uint16_t*)k->data; p.xsb = k->stride_b; p.xsh = k->stride_h; p.xsn = k->stride_n;
  p.out = (int8_t*)out->data; p.osb = out->stride_b; p.osh = out->stride_h; p.osn = out->stride_n; p.o_blk = 64 * out->stride_n;
  const int gpb = gran == SAGE_GRAN_PER_THREAD ? 4 : 1;
  const int nblk = (N + 63) / 64;
  p.scale = scale; p.ss_b = (int64_t)H * nblk * gpb; p.ss_h = (int64_t)nblk * gpb; p.ss_blk = gpb;
  p.km = (uint16_t*)km;
  p.count = (unsigned*)workspace;
  p.part = (float*)((char*)workspace + prep_counter_bytes(B, H));
  p.H = H; p.N = N; p.C = C; p.per_thread = gran == SAGE_GRAN_PER_THREAD; p.rounding = rounding;
  p.poll_limit = g_prep_poll_limit > 0 ? g_prep_poll_limit : PREP_POLL_LIMIT;
  launch_begin();
  if (hipMemsetAsync(p.count, 0, (size_t)B * H * 4, st) != hipSuccess) return SAGE_ERR_LAUNCH;
  const dim3 grid((unsigned)((int64_t)B * H * C));
#define LK(DD, BF) hipLaunchKernelGGL((k_prep_kernel<DD, BF>), grid, dim3(256), 0, st, p)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) LK(64, true); else LK(64, false); } else { if (bf) LK(128, true); else LK(128, false); }
#undef LK
  return launch_status();
}

extern "C" size_t sage_v_prep_fp8_workspace_bytes(int B, int H, int N, int D) {
  const size_t C = (size_t)(N + PREP_ROWS - 1) / PREP_ROWS;
  return prep_counter_bytes(B, H) + (size_t)B * H * C * 3 * D * sizeof(float);
}

extern "C" int sage_v_prep_fp8(const sage_tensor* v, int dtype, int B, int H, int N, int D, const sage_tensor* v_fp8,
                               float* v_scale, float* v_mean, float scale_max, void* workspace, sage_stream_t stream) {
  if (!tensor_ok8(v)) return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_fp8 || !v_fp8->data || !aligned16(v_fp8->data) || v_fp8->stride_b % 16 || v_fp8->stride_h % 16 || v_fp8->stride_n % 16)
    return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_scale || !workspace || !aligned16(workspace) || B <= 0 || H <= 0 || N <= 0 || !(scale_max > 0.f)) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int C = (N + PREP_ROWS - 1) / PREP_ROWS;
  if ((int64_t)B * H * C >= ((int64_t)1 << 31)) return SAGE_ERR_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  VPrepParams p;
  p.v = (const uint16_t*)v->data; p.sb = v->stride_b; p.sh = v->stride_h; p.sn = v->stride_n;
  p.out = (uint8_t*)v_fp8->data; p.ob = v_fp8->stride_b; p.oh = v_fp8->stride_h; p.od = v_fp8->stride_n; p.o_tile = 64;
  p.v_scale = v_scale; p.v_mean = v_mean;
  p.count = (unsigned*)workspace;
  p.part = (float*)((char*)workspace + prep_counter_bytes(B, H));
  p.H = H; p.N = N; p.C = C; p.scale_max = scale_max;
  p.poll_limit = g_prep_poll_limit > 0 ? g_prep_poll_limit : PREP_POLL_LIMIT;
  launch_begin();
  if (hipMemsetAsync(p.count, 0, (size_t)B * H * 4, st) != hipSuccess) return SAGE_ERR_LAUNCH;
  const dim3 grid((unsigned)((int64_t)B * H * C));
#define LV(DD, BF) hipLaunchKernelGGL((v_prep_kernel<DD, BF>), grid, dim3(256), 0, st, p)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) LV(64, true); else LV(64, false); } else { if (bf) LV(128, true); else LV(128, false); }
#undef LV
  return launch_status();
}
