// HBM-bound pre-pass kernels of the SageAttention hot path for gfx950:
//   K0  k_mean            (replaces torch k.mean at sageattention/core.py:612)
//   K1  quant_qk_int8     (replaces QuantInt8Kernel csrc/fused/fused.cu:64-198 and the Triton
//                          quantizers sageattention/triton/quant_per_{block,thread}.py)
//   K2  sub_mean_f16      (replaces SubMeanKernel csrc/fused/fused.cu:200-260)
// Roofline: HBM. Algorithmic traffic per element: 2 B read + 1 B written (K1), 2 B read (K0).
// Every thread moves 16 B per load (8 fp16/bf16), the widest coalesced access on CDNA4.
// Compiled with -ffp-contract=off: the integer outputs must match the oracle bit for bit.
#include "sage_common.h"
#include "sage_fp8_kernels.h"

namespace sage {

// ------------------------------------------------------------------------------------------------
// K0: k_mean, deterministic two-pass reduction
// ------------------------------------------------------------------------------------------------
constexpr int KMEAN_ROWS = 256;  // granule of the pass-1 chunks (rows)
// Rows per pass-1 chunk: a multiple of KMEAN_ROWS chosen so that a sequence never has more than 16 chunks -- the quantizers
// then ALWAYS finish the statistics themselves (16 partial rows per head out of L2), i.e. the K pre-pass is two launches
// and the FP8 operator's K + V pre-pass is two launches at every length (round 3; up to 4096 rows the chunks are the 256
// rows of rounds 1-2, so nothing changes there bit for bit; beyond, the partial sums are taken over longer chunks).
__host__ __device__ __forceinline__ int kmean_chunk_rows(int N) {
  const int c256 = (N + KMEAN_ROWS - 1) / KMEAN_ROWS;
  return KMEAN_ROWS * ((c256 + 15) / 16 > 0 ? (c256 + 15) / 16 : 1);
}

// chunk s (KMEAN_ROWS rows) of head (b, h) of H: column sums -> part[b][h][s][D]; `red`: 256/(D/8) x (D+1) floats of LDS
template <int D, bool BF16>
__device__ __forceinline__ void k_mean_partial_body(const uint16_t* __restrict__ k, int64_t sb, int64_t sh, int64_t sn, int N,
                                                    float* __restrict__ part, int S, int s, int h, int b, int H,
                                                    float (*red)[D + 1]) {
  constexpr int TPR = D / 8;       // threads per row
  constexpr int RPP = 256 / TPR;   // rows per pass
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const uint16_t* base = k + b * sb + h * sh + tc * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int rows = kmean_chunk_rows(N);
  const int r0 = s * rows;
#pragma unroll 4
  for (int i = 0; i < rows / RPP; ++i) {
    const int row = r0 + i * RPP + tr;
    if (row < N) {
      const uint4 u = *reinterpret_cast<const uint4*>(base + (int64_t)row * sn);
      float f[8];
      unpack8<BF16>(u, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tr][tc * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < D) {
    float sum = 0.f;
    for (int r = 0; r < RPP; ++r) sum += red[r][threadIdx.x];  // fixed order
    part[(((int64_t)b * H + h) * S + s) * D + threadIdx.x] = sum;
  }
}

template <int D, bool BF16>
__global__ __launch_bounds__(256) void k_mean_partial_kernel(const uint16_t* __restrict__ k, int64_t sb, int64_t sh,
                                                             int64_t sn, int N, float* __restrict__ part, int S) {
  __shared__ float red[256 / (D / 8)][D + 1];
  k_mean_partial_body<D, BF16>(k, sb, sh, sn, N, part, S, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, red);
}

template <bool BF16>
__global__ void k_mean_final_kernel(const float* __restrict__ part, int S, int D, int N, uint16_t* __restrict__ km) {
  const int64_t bh = blockIdx.x;
  const int d = threadIdx.x;
  if (d >= D) return;
  float sum = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {  // 8 partials in flight, added in chunk order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(bh * S + min(s0 + u, S - 1)) * D + d];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < S) sum += v[u];
  }
  km[bh * D + d] = f32_to_elem_bits<BF16>(sum / (float)N);
}

// ------------------------------------------------------------------------------------------------
// K1: INT8 quantizer, all granularities
// ------------------------------------------------------------------------------------------------
struct QuantParams {
  const uint16_t* x;
  int64_t xsb, xsh, xsn;
  const uint16_t* mean;  // [B,H,D] or null
  int8_t* out;
  int64_t osb, osh, osn;
  float* scale;  // [B,H,G]
  const uint16_t* dot_vec;  // [B,H/dot_group,D] or null
  float* dot_out;           // [B,H,N]
  int dot_group;
  int N, G;          // rows, scales per (b,h)
  int gran, is_key;  // sage_qk_gran, K-side grouping of per_thread
  int warp;          // rows per warp group (16, 32, 64 or 128)
  int warp_shift;    // log2(warp): the group maps run per row and must not cost an integer division
  float mult;
  int rounding;
  const int* cu;  // varlen: sequence b = rows [cu[b], cu[b+1]) of the packed tensor (stride_b unused); N = max length
  // result layout: rows of block `blk` start at blk * o_blk (elements; dense: BLK * osn); scales of (b, h, blk) at
  // b*ss_b + h*ss_h + blk*ss_blk (floats; dense: [B,H,G])
  int64_t o_blk, ss_b, ss_h, ss_blk;
  // mean given as the per-chunk column sums of k_mean_partial_kernel ([B*H][S][D] fp32) instead of `mean`: the kernel
  // finishes the reduction itself (S <= 16: a few KB per workgroup out of L2) and block 0 stores km -- one launch less
  const float* mean_part;
  int S;
  uint16_t* km_out;
};

__device__ __forceinline__ int group_of_row(int lr, int gran, int is_key, int warp_shift) {
  // group-id maps: per_block: all rows of the workgroup; per_warp: lr/warp; per_thread:
  // triton/quant_per_thread.py:27-36 (Q: r%8) and :73-80 (K: (r%8)/2).
  if (gran == SAGE_GRAN_PER_BLOCK) return 0;
  const int w = lr >> warp_shift;
  if (gran == SAGE_GRAN_PER_WARP) return w;
  return is_key ? w * 4 + ((lr & 7) >> 1) : w * 8 + (lr & 7);
}

// block `blk` (BLK rows) of head (b, h) of H.  LDS: gmax = 64 dwords, mpart = 16 x D floats (16-byte aligned)
template <int D, int BLK, bool BF16>
__device__ __forceinline__ void quant_qk_int8_body(const QuantParams& p, const int blk, const int h, const int b, const int H,
                                                   unsigned int* gmax, float (*mpart)[D]) {
  constexpr int TPR = D / 8;
  constexpr int RPP = 256 / TPR;
  constexpr int NP = BLK / RPP;
  static_assert(NP >= 1, "block too small");
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  int N_ = p.N;
  int64_t x_boff = b * p.xsb, o_boff = b * p.osb;
  if (p.cu) {
    const int lo = p.cu[b];
    N_ = p.cu[b + 1] - lo;
    if (blk * BLK >= N_) return;  // uniform for the workgroup
    x_boff = (int64_t)lo * p.xsn;
    o_boff = (int64_t)lo * p.osn;
  }

  if (threadIdx.x < 64) gmax[threadIdx.x] = 0u;

  // the block's rows first: everything below overlaps with this one trip to HBM
  const uint16_t* xbase = p.x + x_boff + h * p.xsh + tc * 8;
  float xf[NP][8];
  uint4 raw[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = blk * BLK + i * RPP + tr;
    raw[i] = make_uint4(0, 0, 0, 0);
    if (row < N_) raw[i] = *reinterpret_cast<const uint4*>(xbase + (int64_t)row * p.xsn);
  }
  // mean_part: thread row s fetches chunk s (ONE round trip to L2 for all S chunks instead of S dependent ones: at 2048
  // rows the quantizer spent more time on these than on its block), LDS hands every thread all chunks
  if (p.mean_part && tr < p.S) {
    const float* pp = p.mean_part + (((int64_t)b * H + h) * p.S + tr) * D + tc * 8;
    *reinterpret_cast<float4*>(&mpart[tr][tc * 8]) = *reinterpret_cast<const float4*>(pp);
    *reinterpret_cast<float4*>(&mpart[tr][tc * 8 + 4]) = *reinterpret_cast<const float4*>(pp + 4);
  }
  __syncthreads();  // gmax zeroed, mpart filled

  float mean_f[8];
  if (p.mean_part) {
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s_ = 0; s_ < p.S; ++s_) {  // chunk order, as k_mean_final_kernel
      const float4 a = *reinterpret_cast<const float4*>(&mpart[s_][tc * 8]), c = *reinterpret_cast<const float4*>(&mpart[s_][tc * 8 + 4]);
      sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w; sum[4] += c.x; sum[5] += c.y; sum[6] += c.z; sum[7] += c.w;
    }
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t bits = f32_to_elem_bits<BF16>(sum[j] / (float)p.N);
      mean_f[j] = elem_to_f32<BF16>(bits);
      if (j & 1) w[j >> 1] |= (uint32_t)bits << 16; else w[j >> 1] = bits;
    }
    if (blk == 0 && tr == 0) *reinterpret_cast<uint4*>(p.km_out + ((int64_t)b * H + h) * D + tc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  } else if (p.mean) {
    // packed sequences share one mean over all tokens ([1,H,D], core.py:461)
    const uint4 um = *reinterpret_cast<const uint4*>(p.mean + ((int64_t)(p.cu ? 0 : b) * H + h) * D + tc * 8);
    unpack8<BF16>(um, mean_f);
  }
  float dvec[8];
  if (p.dot_vec) {
    const int Hk = H / p.dot_group;
    const uint4 ud = *reinterpret_cast<const uint4*>(p.dot_vec + ((int64_t)b * Hk + h / p.dot_group) * D + tc * 8);
    unpack8<BF16>(ud, dvec);
  }

#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int lr = i * RPP + tr;
    const int row = blk * BLK + lr;
    const bool valid = row < N_;
    unpack8<BF16>(raw[i], xf[i]);
    if (p.dot_vec) {
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += xf[i][j] * dvec[j];
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) dot += __shfl_xor(dot, o);
      if (tc == 0 && valid) p.dot_out[((int64_t)b * H + h) * p.N + row] = dot;
    }
    float amax = 0.f;
    const float mult_row = valid ? p.mult : 0.f;  // rows past the end contribute zeros (finite inputs: raw is 0 there)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = xf[i][j];
      if (p.mean || p.mean_part) {
        v = v - mean_f[j];
        if (p.rounding == SAGE_ROUND_TRITON) v = round_to_elem<BF16>(v);  // torch `k - km` in the input dtype
      }
      v = v * mult_row;
      xf[i][j] = v;
      amax = fmaxf(amax, fabsf(v));
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (tc == 0) atomicMax(&gmax[group_of_row(lr, p.gran, p.is_key, p.warp_shift)], __float_as_uint(amax));
  }
  __syncthreads();

  const int groups_per_blk = p.gran == SAGE_GRAN_PER_BLOCK ? 1
                             : p.gran == SAGE_GRAN_PER_WARP ? BLK >> p.warp_shift
                                                            : (BLK >> p.warp_shift) * (p.is_key ? 4 : 8);
  const float eps = (p.gran == SAGE_GRAN_PER_THREAD) ? 0.0000001f : 0.f;
  if (threadIdx.x < groups_per_blk) {
    const float a = __uint_as_float(gmax[threadIdx.x]);
    const float sc = (p.rounding == SAGE_ROUND_TRITON) ? a / 127.f + eps : fmaxf(a, 0.0000001f) / 127.f;
    p.scale[b * p.ss_b + h * p.ss_h + blk * p.ss_blk + threadIdx.x] = sc;
  }

  int8_t* obase = p.out + o_boff + h * p.osh + blk * p.o_blk + tc * 8;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int lr = i * RPP + tr;
    const int row = blk * BLK + lr;
    const float a = __uint_as_float(gmax[group_of_row(lr, p.gran, p.is_key, p.warp_shift)]);
    int q[8];
    if (p.rounding == SAGE_ROUND_TRITON) {
      // q = trunc(x/sc + 0.5*sign) with an IEEE division (quant_per_block.py:42-44).  The division costs ~10 VALU
      // ops per element and made this HBM-bound kernel VALU-bound, so: multiply by the correctly rounded reciprocal
      // (|x*r - x/sc| <= 1.5 ulp <= 2.3e-5 for |x/sc| <= 127, plus <= 7.6e-6 from the +0.5) and fall back to the exact
      // division, for the whole 8-element chunk of the wave, only when some value lands within 2^-14 of a
      // rounding boundary, where the two could differ (~6 % of the chunks).  Bit-exact by construction.
      const float sc = a / 127.f + eps;
      const float r = 1.0f / sc;
      bool near = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = round_half_away_fast(xf[i][j] * r, near);
      if (__builtin_amdgcn_ballot_w64(near || !(fabsf(r) < 3.0e38f)) != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = xf[i][j] / sc;  // IEEE division
          y = y + (y >= 0.f ? 0.5f : -0.5f);
          q[j] = (int)y;  // truncation, as tl `.to(int8)`
        }
      }
    } else {
      const float inv = 127.f / fmaxf(a, 0.0000001f);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (int)rintf(xf[i][j] * inv);  // cvt.rni (fused.cu:176-181)
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = min(max(q[j], -128), 127);
    const uint32_t w0 = pack_i8x4(q[0], q[1], q[2], q[3]), w1 = pack_i8x4(q[4], q[5], q[6], q[7]);
    if (row < N_) *reinterpret_cast<uint2*>(obase + (int64_t)lr * p.osn) = make_uint2(w0, w1);
  }
}

template <int D, int BLK, bool BF16>
__global__ __launch_bounds__(256) void quant_qk_int8_kernel(const QuantParams p) {
  __shared__ unsigned int gmax[64];
  __shared__ __attribute__((aligned(16))) float mpart[16][D];  // chunk sums of the mean (mean_part form: S <= 16 <= RPP)
  quant_qk_int8_body<D, BLK, BF16>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, gmax, mpart);
}

// ------------------------------------------------------------------------------------------------
// Fused K/V pre-pass of the FP8-PV operator for sequences of at most 16 chunks (4096 rows): TWO launches instead of five
// (k_mean_partial, quantizer | v_stats_partial, v_stats_final, v_quant_transpose).  Every kernel below is bandwidth- or
// latency-bound and fills the chip on its own, so what the fusion saves is three launch boundaries and the ramp / tail of
// three kernels, and the VALU-heavy K quantizer shares the CUs with the bandwidth-bound V quantizer.
//   launch A  workgroups [0, S): K column sums per chunk;  [S, 2S): V per-channel max|v| per chunk
//   launch B  workgroups [0, nblk_k): K quantizer (finishes the mean itself);  the rest: V quantizer + transpose (finishes
//             max|v| itself, block 0 stores v_scale)
// Without V smoothing only max|v| is needed, which does not depend on the order of the reduction: bit-identical to
// sage_k_smooth_quant + sage_quant_v_fp8(v_mean = null).
// ------------------------------------------------------------------------------------------------
struct VPrepParams {
  const uint16_t* v;
  int64_t sb, sh, sn;
  uint8_t* out;
  int64_t ob, oh, od, o_tile;
  float* v_scale;      // [B,H,D]
  const float* part;   // [B,H,S,D] max|v| per chunk
  float scale_max;
};

template <int D, bool BF16>
__device__ __forceinline__ void v_amax_partial_body(const uint16_t* __restrict__ v, int64_t sb, int64_t sh, int64_t sn, int N,
                                                    float* __restrict__ part, int S, int s, int h, int b, int H,
                                                    float (*red)[D + 1]) {
  constexpr int TPR = D / 8, RPP = 256 / TPR;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const uint16_t* base = v + b * sb + h * sh + tc * 8;
  float am[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // rows in [N, ceil16(N)) count as zeros (fused.cu:335): the same as starting at 0
  const int rows = kmean_chunk_rows(N);
#pragma unroll 4
  for (int i = 0; i < rows / RPP; ++i) {
    const int row = s * rows + i * RPP + tr;
    if (row < N) {
      float f[8];
      unpack8<BF16>(*reinterpret_cast<const uint4*>(base + (int64_t)row * sn), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) am[j] = fmaxf(am[j], fabsf(f[j]));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tr][tc * 8 + j] = am[j];
  __syncthreads();
  if (threadIdx.x < D) {
    float a = red[0][threadIdx.x];
    for (int r = 1; r < RPP; ++r) a = fmaxf(a, red[r][threadIdx.x]);
    part[(((int64_t)b * H + h) * S + s) * D + threadIdx.x] = a;
  }
}

template <int D, bool BF16>
__global__ __launch_bounds__(256) void kv_partial_kernel(const uint16_t* __restrict__ k, int64_t ksb, int64_t ksh, int64_t ksn,
                                                         const uint16_t* __restrict__ v, int64_t vsb, int64_t vsh, int64_t vsn,
                                                         int N, float* __restrict__ kpart, float* __restrict__ vpart, int S) {
  static_assert(KMEAN_ROWS == VQ_ROWS, "one chunk size for both tensors");
  __shared__ float red[256 / (D / 8)][D + 1];
  const int x = blockIdx.x;
  if (x < S) k_mean_partial_body<D, BF16>(k, ksb, ksh, ksn, N, kpart, S, x, blockIdx.y, blockIdx.z, gridDim.y, red);
  else v_amax_partial_body<D, BF16>(v, vsb, vsh, vsn, N, vpart, S, x - S, blockIdx.y, blockIdx.z, gridDim.y, red);
}

template <int D, bool BF16>
__global__ __launch_bounds__(256, D == 64 ? 8 : 7) void kv_quant_kernel(const QuantParams p, const VPrepParams q, const int nblk_k) {
  using G = VQuantGeom<D>;
  __shared__ unsigned int gmax[64];
  __shared__ __attribute__((aligned(16))) float exch[16][D];  // chunk partials of this head: K sums or V max|v|
  __shared__ __attribute__((aligned(16))) uint32_t tile[G::BLKS * G::IMG];
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  if ((int)blockIdx.x < nblk_k) {
    quant_qk_int8_body<D, 64, BF16>(p, blockIdx.x, h, b, H, gmax, exch);
    return;
  }
  const int bx = blockIdx.x - nblk_k;
  const int tg = threadIdx.x / G::TPR, tc = threadIdx.x % G::TPR;
  if (tg < p.S) {  // one round trip for all chunks (S <= 16 <= token groups per workgroup)
    const float* pp = q.part + (((int64_t)b * H + h) * p.S + tg) * D + tc * 8;
    *reinterpret_cast<float4*>(&exch[tg][tc * 8]) = *reinterpret_cast<const float4*>(pp);
    *reinterpret_cast<float4*>(&exch[tg][tc * 8 + 4]) = *reinterpret_cast<const float4*>(pp + 4);
  }
  __syncthreads();
  float mean[8], rcp[8], vs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = exch[0][tc * 8 + j];
    for (int s_ = 1; s_ < p.S; ++s_) a = fmaxf(a, exch[s_][tc * 8 + j]);
    mean[j] = 0.f;
    rcp[j] = q.scale_max / a;   // v_stats_final_kernel
    vs[j] = a / q.scale_max;
  }
  if (bx == 0 && tg == 0) {
    float* o = q.v_scale + ((int64_t)b * H + h) * D + tc * 8;
    *reinterpret_cast<float4*>(o) = make_float4(vs[0], vs[1], vs[2], vs[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(vs[4], vs[5], vs[6], vs[7]);
  }
  v_quant_transpose_body<D, BF16>(q.v, q.sb, q.sh, q.sn, p.N, mean, rcp, q.out, q.ob, q.oh, q.od, q.o_tile, bx, h, b, tile);
}

// ------------------------------------------------------------------------------------------------
// K2: sub_mean_f16
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void sub_mean_f16_kernel(const uint16_t* __restrict__ v, int64_t sb, int64_t sh,
                                                           int64_t sn, const uint16_t* __restrict__ vm,
                                                           uint16_t* __restrict__ out, int64_t ob, int64_t oh,
                                                           int64_t on, int N, int D) {
  const int TPR = D / 8, RPP = 256 / TPR;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const int row = blockIdx.x * RPP + tr;
  if (row >= N) return;
  float m[8], x[8];
  unpack8<BF16>(*reinterpret_cast<const uint4*>(vm + ((int64_t)b * H + h) * D + tc * 8), m);
  unpack8<BF16>(*reinterpret_cast<const uint4*>(v + b * sb + h * sh + (int64_t)row * sn + tc * 8), x);
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // packed subtraction in the input dtype (fused.cu:233), then fp16 (:235-238)
    const float lo = round_to_elem<BF16>(x[2 * j] - m[2 * j]);
    const float hi = round_to_elem<BF16>(x[2 * j + 1] - m[2 * j + 1]);
    w[j] = (uint32_t)f32_to_elem_bits<false>(lo) | ((uint32_t)f32_to_elem_bits<false>(hi) << 16);
  }
  *reinterpret_cast<uint4*>(out + b * ob + h * oh + (int64_t)row * on + tc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

static bool tensor_ok(const sage_tensor* t, int align_elems) {
  return t && t->data && aligned16(t->data) && t->stride_b % align_elems == 0 && t->stride_h % align_elems == 0 &&
         t->stride_n % align_elems == 0;
}

}  // namespace sage

using namespace sage;

extern "C" size_t sage_k_mean_workspace_bytes(int B, int H, int N, int D) {
  const size_t S = (size_t)(N + KMEAN_ROWS - 1) / KMEAN_ROWS;
  return (size_t)B * H * S * D * sizeof(float);
}

extern "C" int sage_k_mean(const sage_tensor* k, int dtype, int B, int H, int N, int D, void* km, void* workspace,
                           sage_stream_t stream) {
  if (!tensor_ok(k, 8) || !km || !workspace || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int rows = kmean_chunk_rows(N), S = (N + rows - 1) / rows;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  dim3 grid(S, H, B);
  const uint16_t* kp = (const uint16_t*)k->data;
  float* ws = (float*)workspace;
#define LAUNCH(DD, BF)                                                                                         \
  hipLaunchKernelGGL((k_mean_partial_kernel<DD, BF>), grid, dim3(256), 0, st, kp, k->stride_b, k->stride_h, \
                     k->stride_n, N, ws, S)
  if (D == 64) { if (dtype == SAGE_BF16) LAUNCH(64, true); else LAUNCH(64, false); }
  else { if (dtype == SAGE_BF16) LAUNCH(128, true); else LAUNCH(128, false); }
#undef LAUNCH
  if (dtype == SAGE_BF16)
    hipLaunchKernelGGL((k_mean_final_kernel<true>), dim3(B * H), dim3(128), 0, st, ws, S, D, N, (uint16_t*)km);
  else
    hipLaunchKernelGGL((k_mean_final_kernel<false>), dim3(B * H), dim3(128), 0, st, ws, S, D, N, (uint16_t*)km);
  return launch_status();
}

static int quant_impl(const sage_tensor* x, int dtype, int B, int H, int N, int D, const void* mean,
                      const sage_tensor* out, float* scale, int gran, int is_key, int blk, int warp,
                      float mult, int rounding, const void* lse_dot_vec, int dot_group, float* lse_dot,
                      sage_stream_t stream, const int* cu, int64_t out_blk_stride = 0, const int64_t* scale_strides = nullptr,
                      const float* mean_part = nullptr, int S = 0, void* km_out = nullptr, QuantParams* params_only = nullptr) {
  if (!tensor_ok(x, 8) || !tensor_ok(out, 8) || !scale || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran < SAGE_GRAN_PER_BLOCK || gran > SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;
  if (rounding != SAGE_ROUND_TRITON && rounding != SAGE_ROUND_CUDA) return SAGE_ERR_INVALID_ARGUMENT;
  if (blk != 64 && blk != 128) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran == SAGE_GRAN_PER_BLOCK) warp = blk;
  if ((warp != 16 && warp != 32 && warp != 64 && warp != 128) || blk % warp != 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (mean && !aligned16(mean)) return SAGE_ERR_INVALID_ARGUMENT;
  if (mean_part && (S < 1 || S > 16 || !aligned16(mean_part))) return SAGE_ERR_INVALID_ARGUMENT;  // mpart[16][D] in the kernel
  if ((lse_dot_vec != nullptr) != (lse_dot != nullptr)) return SAGE_ERR_INVALID_ARGUMENT;
  if (lse_dot_vec && (dot_group <= 0 || H % dot_group != 0 || !aligned16(lse_dot_vec))) return SAGE_ERR_INVALID_ARGUMENT;
  const int nblk = (N + blk - 1) / blk;
  const int gpb = gran == SAGE_GRAN_PER_BLOCK ? 1 : gran == SAGE_GRAN_PER_WARP ? blk / warp : (blk / warp) * (is_key ? 4 : 8);
  if (gpb > 64) return SAGE_ERR_INVALID_ARGUMENT;
  QuantParams p;
  p.x = (const uint16_t*)x->data; p.xsb = x->stride_b; p.xsh = x->stride_h; p.xsn = x->stride_n;
  p.mean = (const uint16_t*)mean;
  p.out = (int8_t*)out->data; p.osb = out->stride_b; p.osh = out->stride_h; p.osn = out->stride_n;
  p.scale = scale; p.dot_vec = (const uint16_t*)lse_dot_vec; p.dot_out = lse_dot; p.dot_group = dot_group > 0 ? dot_group : 1;
  p.cu = cu;
  p.mean_part = mean_part; p.S = S; p.km_out = (uint16_t*)km_out;
  p.o_blk = out_blk_stride ? out_blk_stride : (int64_t)blk * out->stride_n;
  p.ss_b = scale_strides ? scale_strides[0] : (int64_t)H * nblk * gpb;
  p.ss_h = scale_strides ? scale_strides[1] : (int64_t)nblk * gpb;
  p.ss_blk = scale_strides ? scale_strides[2] : gpb;
  if (out_blk_stride < 0 || (out_blk_stride & 7) || p.ss_blk < gpb) return SAGE_ERR_INVALID_ARGUMENT;
  p.N = N; p.G = nblk * gpb; p.gran = gran; p.is_key = is_key ? 1 : 0; p.warp = warp; p.mult = mult; p.rounding = rounding;
  p.warp_shift = warp == 16 ? 4 : warp == 32 ? 5 : warp == 64 ? 6 : 7;
  if (params_only) { *params_only = p; return SAGE_OK; }  // validated parameters for a fused launch (sage_kv_prepare_fp8)
  dim3 grid(nblk, H, B);
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
#define LAUNCH(DD, BL, BF) hipLaunchKernelGGL((quant_qk_int8_kernel<DD, BL, BF>), grid, dim3(256), 0, st, p)
#define BY_DT(DD, BL) do { if (dtype == SAGE_BF16) LAUNCH(DD, BL, true); else LAUNCH(DD, BL, false); } while (0)
  if (D == 64) { if (blk == 64) BY_DT(64, 64); else BY_DT(64, 128); }
  else { if (blk == 64) BY_DT(128, 64); else BY_DT(128, 128); }
#undef BY_DT
#undef LAUNCH
  return launch_status();
}

extern "C" int sage_quant_qk_int8(const sage_tensor* x, int dtype, int B, int H, int N, int D, const void* mean,
                                  const sage_tensor* out, float* scale, int gran, int is_key, int blk, int warp,
                                  float mult, int rounding, const void* lse_dot_vec, int dot_group, float* lse_dot,
                                  sage_stream_t stream) {
  return quant_impl(x, dtype, B, H, N, D, mean, out, scale, gran, is_key, blk, warp, mult, rounding, lse_dot_vec, dot_group,
                    lse_dot, stream, nullptr);
}

extern "C" int sage_quant_qk_int8_varlen(const sage_tensor* x, int dtype, const int* cu_seqlens, int num_seqs, int H,
                                         int max_seqlen, int D, const void* mean, const sage_tensor* out, float* scale,
                                         int gran, int is_key, int blk, int warp, float mult, int rounding,
                                         sage_stream_t stream) {
  if (!cu_seqlens) return SAGE_ERR_INVALID_ARGUMENT;
  return quant_impl(x, dtype, num_seqs, H, max_seqlen, D, mean, out, scale, gran, is_key, blk, warp, mult, rounding, nullptr,
                    1, nullptr, stream, cu_seqlens);
}

extern "C" int sage_sub_mean_f16(const sage_tensor* v, int dtype, int B, int H, int N, int D, const void* vm,
                                 const sage_tensor* out, sage_stream_t stream) {
  if (!tensor_ok(v, 8) || !tensor_ok(out, 8) || !vm || !aligned16(vm) || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int RPP = 256 / (D / 8);
  dim3 grid((N + RPP - 1) / RPP, H, B);
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  if (dtype == SAGE_BF16)
    hipLaunchKernelGGL((sub_mean_f16_kernel<true>), grid, dim3(256), 0, st, (const uint16_t*)v->data, v->stride_b, v->stride_h,
                       v->stride_n, (const uint16_t*)vm, (uint16_t*)out->data, out->stride_b, out->stride_h, out->stride_n, N, D);
  else
    hipLaunchKernelGGL((sub_mean_f16_kernel<false>), grid, dim3(256), 0, st, (const uint16_t*)v->data, v->stride_b, v->stride_h,
                       v->stride_n, (const uint16_t*)vm, (uint16_t*)out->data, out->stride_b, out->stride_h, out->stride_n, N, D);
  return launch_status();
}

extern "C" int sage_quant_k_int8_kvtiles(const sage_tensor* k, int dtype, int B, int H, int N, int D, const void* mean,
                                         const sage_tensor* out, int64_t out_tile_stride, float* scale,
                                         const int64_t* scale_strides, int gran, int rounding, sage_stream_t stream) {
  if (!scale_strides || out_tile_stride <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran != SAGE_GRAN_PER_BLOCK && gran != SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;  // the K granularities
  return quant_impl(k, dtype, B, H, N, D, mean, out, scale, gran, 1, 64, 64, 1.0f, rounding, nullptr, 1, nullptr, stream, nullptr,
                    out_tile_stride, scale_strides);
}

// K smoothing + quantization as one call: km = mean over the sequence (sage_k_mean) and the INT8 quantization of k - km
// (sage_quant_qk_int8 with is_key = 1, blk 64).  Two launches at every length: the sequence is cut into at most 16 chunks
// (kmean_chunk_rows) and the quantizer finishes the mean itself; bit-identical to the two separate entry points.
extern "C" int sage_k_smooth_quant(const sage_tensor* k, int dtype, int B, int H, int N, int D, const sage_tensor* out,
                                   float* scale, void* km, int gran, int rounding, void* workspace, sage_stream_t stream) {
  if (!km || !workspace) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran != SAGE_GRAN_PER_BLOCK && gran != SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;
  const int rows = kmean_chunk_rows(N), S = (N + rows - 1) / rows;   // <= 16 chunks at every length
  if (!tensor_ok(k, 8) || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const dim3 grid(S, H, B);
  const uint16_t* kp = (const uint16_t*)k->data;
  float* ws = (float*)workspace;
#define LAUNCH(DD, BF)                                                                                         \
  hipLaunchKernelGGL((k_mean_partial_kernel<DD, BF>), grid, dim3(256), 0, st, kp, k->stride_b, k->stride_h, \
                     k->stride_n, N, ws, S)
  if (D == 64) { if (dtype == SAGE_BF16) LAUNCH(64, true); else LAUNCH(64, false); }
  else { if (dtype == SAGE_BF16) LAUNCH(128, true); else LAUNCH(128, false); }
#undef LAUNCH
  if (launch_status() != SAGE_OK) return SAGE_ERR_LAUNCH;
  return quant_impl(k, dtype, B, H, N, D, nullptr, out, scale, gran, 1, 64, 64, 1.0f, rounding, nullptr, 1, nullptr, stream, nullptr,
                    0, nullptr, ws, S, km);
}

extern "C" size_t sage_kv_prepare_fp8_workspace_bytes(int B, int H, int N, int D) {
  // the fused form needs 2 x [B,H,S,D] floats; longer sequences run the separate kernels on the same buffer
  return sage_k_mean_workspace_bytes(B, H, N, D) + sage_quant_v_fp8_workspace_bytes(B, H, N, D);
}

extern "C" int sage_kv_prepare_fp8(const sage_tensor* k, const sage_tensor* v, int dtype, int B, int H, int N, int D,
                                   const sage_tensor* k_int8, float* k_scale, void* km, int gran, int rounding,
                                   const sage_tensor* v_fp8, float* v_scale, float scale_max, void* workspace,
                                   sage_stream_t stream) {
  if (!km || !workspace || !v_scale || !(scale_max > 0.f)) return SAGE_ERR_INVALID_ARGUMENT;
  if (gran != SAGE_GRAN_PER_BLOCK && gran != SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;
  if (!tensor_ok(k, 8) || !tensor_ok(v, 8) || B <= 0 || H <= 0 || N <= 0) return SAGE_ERR_INVALID_ARGUMENT;
  if (!v_fp8 || !v_fp8->data || !aligned16(v_fp8->data) || v_fp8->stride_b % 16 || v_fp8->stride_h % 16 || v_fp8->stride_n % 16)
    return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if (dtype != SAGE_F16 && dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  const int rows = kmean_chunk_rows(N), S = (N + rows - 1) / rows;   // <= 16 chunks at every length
  float* kpart = (float*)workspace;
  float* vpart = kpart + (size_t)B * H * S * D;
  QuantParams p;
  const int st0 = quant_impl(k, dtype, B, H, N, D, nullptr, k_int8, k_scale, gran, 1, 64, 64, 1.0f, rounding, nullptr, 1, nullptr,
                             stream, nullptr, 0, nullptr, kpart, S, km, &p);
  if (st0 != SAGE_OK) return st0;
  VPrepParams q;
  q.v = (const uint16_t*)v->data; q.sb = v->stride_b; q.sh = v->stride_h; q.sn = v->stride_n;
  q.out = (uint8_t*)v_fp8->data; q.ob = v_fp8->stride_b; q.oh = v_fp8->stride_h; q.od = v_fp8->stride_n; q.o_tile = 64;
  q.v_scale = v_scale; q.part = vpart; q.scale_max = scale_max;
  const int nblk_k = (N + 63) / 64;
  const int vq_blks = D == 128 ? 1 : 2;  // 64-token blocks per V workgroup (VQuantGeom<D>::BLKS)
  const int nblk_v = (nblk_k + vq_blks - 1) / vq_blks;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const dim3 ga(2 * S, H, B), gb(nblk_k + nblk_v, H, B);
  const uint16_t* kp = (const uint16_t*)k->data;
#define LA(DD, BF)                                                                                                          \
  hipLaunchKernelGGL((kv_partial_kernel<DD, BF>), ga, dim3(256), 0, st, kp, k->stride_b, k->stride_h, k->stride_n, q.v, q.sb, \
                     q.sh, q.sn, N, kpart, vpart, S)
#define LB(DD, BF) hipLaunchKernelGGL((kv_quant_kernel<DD, BF>), gb, dim3(256), 0, st, p, q, nblk_k)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) LA(64, true); else LA(64, false); } else { if (bf) LA(128, true); else LA(128, false); }
  if (launch_status() != SAGE_OK) return SAGE_ERR_LAUNCH;
  if (D == 64) { if (bf) LB(64, true); else LB(64, false); } else { if (bf) LB(128, true); else LB(128, false); }
#undef LA
#undef LB
  return launch_status();
}
