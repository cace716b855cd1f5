// HBM-bound pre-pass kernels of the SageAttention hot path for gfx950:
//   K0  k_mean            (replaces torch k.mean at sageattention/core.py:612)
//   K1  quant_qk_int8     (replaces QuantInt8Kernel csrc/fused/fused.cu:64-198 and the Triton
//                          quantizers sageattention/triton/quant_per_{block,thread}.py)
//                         one block per workgroup (quant_qk_int8_kernel) or, for dense K, several consecutive blocks of a
//                         head per workgroup with the next block's rows prefetched (k_quant_stream_kernel); with the FP8 V
//                         quantizer in one launch: kv_quant_kernel / kv_quant_stream_kernel
//   K2  sub_mean_f16      (replaces SubMeanKernel csrc/fused/fused.cu:200-260)
// Roofline: HBM. Algorithmic traffic per element: 2 B read + 1 B written (K1), 2 B read (K0).
// Every thread moves 16 B per load (8 fp16/bf16), the widest coalesced access on CDNA4.
// Compiled with -ffp-contract=off: the integer outputs must match the oracle bit for bit.
#include <type_traits>
#include "sage_common.h"
#include "sage_fp8_kernels.h"

#ifndef SAGE_KV_UNITS_CAP  // blocks / units one workgroup of the streaming K + V quantizer walks at most (variant builds sweep it)
#define SAGE_KV_UNITS_CAP 32
#endif

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
  // Whole chunks (all but the last of a sequence) take the loop without a row test: hipcc sinks a conditional load into
  // its branch and then waits for every load before it issues the next one (one 16-B load in flight per thread: the kernel
  // ran at the latency of sixteen dependent round trips); unconditional, the unrolled loop keeps eight in flight.  Same
  // additions in the same order.
  if (r0 + rows <= N) {
#pragma unroll 8
    for (int i = 0; i < rows / RPP; ++i) {
      const uint4 u = *reinterpret_cast<const uint4*>(base + (int64_t)(r0 + i * RPP + tr) * sn);
      float f[8];
      unpack8<BF16>(u, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  } else {
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

// max over the TPR (8 or 16) consecutive lanes that hold one row, on DPP (quad swaps, then the mirrored half rows / rows): every
// lane ends with the row's maximum; 3-4 v_max_f32_dpp instead of as many ds_bpermute round trips.  (A maximum does not depend
// on the order it is taken in: bit-identical to the xor butterfly.)
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int TPR>
__device__ __forceinline__ float row_lanes_max(float a) {
  static_assert(TPR == 8 || TPR == 16, "a row is 8 or 16 lanes");
  a = fmaxf(a, dpp_lane<0xB1>(a));    // quad_perm [1,0,3,2]
  a = fmaxf(a, dpp_lane<0x4E>(a));    // quad_perm [2,3,0,1]
  a = fmaxf(a, dpp_lane<0x141>(a));   // row_half_mirror: the other quad of the 8 lanes
  if constexpr (TPR == 16) a = fmaxf(a, dpp_lane<0x140>(a));  // row_mirror: the other half of the 16 lanes
  return a;
}

__device__ __forceinline__ int group_of_row(int lr, int gran, int is_key, int warp_shift) {
  // group-id maps: per_block: all rows of the workgroup; per_warp: lr/warp; per_thread:
  // triton/quant_per_thread.py:27-36 (Q: r%8) and :73-80 (K: (r%8)/2).
  if (gran == SAGE_GRAN_PER_BLOCK) return 0;
  const int w = lr >> warp_shift;
  if (gran == SAGE_GRAN_PER_WARP) return w;
  return is_key ? w * 4 + ((lr & 7) >> 1) : w * 8 + (lr & 7);
}

// ---- the quantizer in three steps, shared by the one-block-per-workgroup kernels and the streaming K quantizer ----
template <int D, int BLK>
struct QuantGeom {
  static constexpr int TPR = D / 8;        // threads per row (16 B each)
  static constexpr int RPP = 256 / TPR;    // rows per pass of the workgroup
  static constexpr int NP = BLK / RPP;     // passes per block = 16-B loads per thread
  static_assert(NP >= 1, "block too small");
};

// step 1: the rows of block `blk` (rows past the end read as zeros).  xbase: head base + this thread's column offset
template <int D, int BLK>
__device__ __forceinline__ void quant_load_rows(const uint16_t* xbase, const int64_t xsn, const int blk, const int N_,
                                                uint4 (&raw)[QuantGeom<D, BLK>::NP]) {
  using G = QuantGeom<D, BLK>;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // see quant_block
  const int tr = tid / G::TPR;
#pragma unroll
  for (int i = 0; i < G::NP; ++i) {
    const int row = blk * BLK + i * G::RPP + tr;
    raw[i] = make_uint4(0, 0, 0, 0);
    if (row < N_) raw[i] = *reinterpret_cast<const uint4*>(xbase + (int64_t)row * xsn);
  }
}

// step 2 (once per head): the mean as 8 storage-dtype values per thread column.  Holds the workgroup's FIRST barrier (which
// also publishes the zeroed group maxima) and, in the mean_part form, a second one.  mpart: 16 x D floats (16-byte aligned)
template <int D, bool BF16>
__device__ __forceinline__ uint4 quant_mean_bits(const QuantParams& p, const int h, const int b, const int H, const bool store_km,
                                                 float (*mpart)[D]) {
  constexpr int TPR = D / 8;
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  // mean_part: thread row s fetches chunk s (ONE round trip to L2 for all S chunks instead of S dependent ones: at 2048
  // rows the quantizer spent more time on these than on its block), LDS hands the first thread row all chunks
  if (p.mean_part && tr < p.S) {
    const float* pp = p.mean_part + (((int64_t)b * H + h) * p.S + tr) * D + tc * 8;
    *reinterpret_cast<float4*>(&mpart[tr][tc * 8]) = *reinterpret_cast<const float4*>(pp);
    *reinterpret_cast<float4*>(&mpart[tr][tc * 8 + 4]) = *reinterpret_cast<const float4*>(pp + 4);
  }
  __syncthreads();  // gmax zeroed, mpart filled
  // mean_part form: the FIRST thread row finishes the reduction (chunk order, as k_mean_final_kernel; an IEEE division per
  // column) and hands the bits to the other rows through LDS -- every thread doing it for itself cost 8 divisions + 8 S
  // adds per thread, a fifth of the kernel's vector work, and the rows of the block are still on their way from HBM.
  uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
  if (p.mean_part) {
    if (tr == 0) {
      float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int s_ = 0; s_ < p.S; ++s_) {
        const float4 a = *reinterpret_cast<const float4*>(&mpart[s_][tc * 8]), c = *reinterpret_cast<const float4*>(&mpart[s_][tc * 8 + 4]);
        sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w; sum[4] += c.x; sum[5] += c.y; sum[6] += c.z; sum[7] += c.w;
      }
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint16_t bits = f32_to_elem_bits<BF16>(sum[j] / (float)p.N);
        if (j & 1) w[j >> 1] |= (uint32_t)bits << 16; else w[j >> 1] = bits;
      }
      mbits = make_uint4(w[0], w[1], w[2], w[3]);
      // (this thread alone read columns tc*8 .. tc*8+7 of mpart, so it may overwrite them in row 0)
      *reinterpret_cast<uint4*>(&mpart[0][tc * 8]) = mbits;
      if (store_km) *reinterpret_cast<uint4*>(p.km_out + ((int64_t)b * H + h) * D + tc * 8) = mbits;
    }
    __syncthreads();
    mbits = *reinterpret_cast<const uint4*>(&mpart[0][tc * 8]);
  } else if (p.mean) {
    // packed sequences share one mean over all tokens ([1,H,D], core.py:461)
    mbits = *reinterpret_cast<const uint4*>(p.mean + ((int64_t)(p.cu ? 0 : b) * H + h) * D + tc * 8);
  }
  return mbits;
}

// step 3: block `blk` from its rows: group maxima (gmax: 64 zeroed dwords), ONE barrier, scales, rounding, stores.
// zero_next: 64 dwords zeroed behind the barrier (the streaming kernel's group maxima of the block after next), or null.
// after_phase1(): runs when `raw` has been consumed (the streaming kernel reloads it with the next block's rows there)
template <int D, int BLK, bool BF16, bool TRITON, typename AfterPhase1>
__device__ __forceinline__ void quant_block(const QuantParams& p, const int blk, const int h, const int b, const int H, const int N_,
                                            const int64_t o_boff, uint4 (&raw)[QuantGeom<D, BLK>::NP], const uint4 mbits,
                                            unsigned int* gmax, unsigned int* zero_next, AfterPhase1 after_phase1) {
  using G = QuantGeom<D, BLK>;
  constexpr int TPR = G::TPR, RPP = G::RPP, NP = G::NP;
  // (the thread id is made opaque per block: in the streaming kernel's loop hipcc otherwise hoists every row / group /
  //  address term that depends only on it out of the loop and keeps ~40 registers alive for them)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int tr = tid / TPR, tc = tid % TPR;
  float xf[NP][8];
  float dvec[8];
  if (p.dot_vec) {
    const int Hk = H / p.dot_group;
    const uint4 ud = *reinterpret_cast<const uint4*>(p.dot_vec + ((int64_t)b * Hk + h / p.dot_group) * D + tc * 8);
    unpack8<BF16>(ud, dvec);
  }

  // x - mean (no mean: mbits = 0 and x - 0 is x in either form), times the multiplier, and the group maxima
  //   TRITON: `k - km` in the input dtype (torch).   CUDA: in fp32 (fused.cu).
  // fp16, TRITON: the subtraction runs as v_pk_add_f16 -- the correctly rounded fp16 difference, which is what rounding the
  // fp32 difference of two fp16 values gives as well (24 >= 2*11+2 bits: the double rounding is innocuous) -- 4 packed
  // subtractions + 8 converts per 8 elements instead of 8 + 8 + 16.
  {
    float mean_f[8];
    if constexpr (!TRITON || BF16) unpack8<BF16>(mbits, mean_f);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int lr = i * RPP + tr;
      const int row = blk * BLK + lr;
      const bool valid = row < N_;
      if (p.dot_vec) {
        float xr[8];
        unpack8<BF16>(raw[i], xr);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += xr[j] * dvec[j];
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) dot += __shfl_xor(dot, o);
        if (tc == 0 && valid) p.dot_out[((int64_t)b * H + h) * p.N + row] = dot;
      }
      if constexpr (TRITON && !BF16) {
        const uint32_t xw[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w}, mw[4] = {mbits.x, mbits.y, mbits.z, mbits.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const v2h d = __builtin_bit_cast(v2h, xw[w]) - __builtin_bit_cast(v2h, mw[w]);
          xf[i][2 * w] = (float)d[0];
          xf[i][2 * w + 1] = (float)d[1];
        }
      } else {
        unpack8<BF16>(raw[i], xf[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xf[i][j] = xf[i][j] - mean_f[j];
          if constexpr (TRITON) xf[i][j] = round_to_elem<BF16>(xf[i][j]);
        }
      }
      float amax = 0.f;
      const float mult_row = valid ? p.mult : 0.f;  // rows past the end contribute zeros (finite inputs: raw is 0 there)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xf[i][j] = xf[i][j] * mult_row;
        amax = fmaxf(amax, fabsf(xf[i][j]));
      }
      amax = row_lanes_max<TPR>(amax);
      if (tc == 0) atomicMax(&gmax[group_of_row(lr, p.gran, p.is_key, p.warp_shift)], __float_as_uint(amax));
    }
  }
  after_phase1();
  __syncthreads();
  if (zero_next && threadIdx.x < 64) zero_next[threadIdx.x] = 0u;

  const int groups_per_blk = p.gran == SAGE_GRAN_PER_BLOCK ? 1
                             : p.gran == SAGE_GRAN_PER_WARP ? BLK >> p.warp_shift
                                                            : (BLK >> p.warp_shift) * (p.is_key ? 4 : 8);
  const float eps = (p.gran == SAGE_GRAN_PER_THREAD) ? 0.0000001f : 0.f;
  if (threadIdx.x < groups_per_blk) {
    const float a = __uint_as_float(gmax[threadIdx.x]);
    const float sc = TRITON ? a / 127.f + eps : fmaxf(a, 0.0000001f) / 127.f;
    p.scale[b * p.ss_b + h * p.ss_h + blk * p.ss_blk + threadIdx.x] = sc;
  }

  int8_t* obase = p.out + o_boff + h * p.osh + blk * p.o_blk + tc * 8;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int lr = i * RPP + tr;
    const int row = blk * BLK + lr;
    const float a = __uint_as_float(gmax[group_of_row(lr, p.gran, p.is_key, p.warp_shift)]);
    int q[8];
    // No clamp on the two fast paths: |x| <= a (a is the maximum of a group x belongs to), so |x * r| <= 127 (1 + 3 ulp)
    // and rint() of it is at most 127 in magnitude.
    if constexpr (TRITON) {
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
          q[j] = min(max((int)y, -128), 127);  // truncation, as tl `.to(int8)`
        }
      }
    } else {
      const float inv = 127.f / fmaxf(a, 0.0000001f);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (int)rintf(xf[i][j] * inv);  // cvt.rni (fused.cu:176-181)
    }
    const uint32_t w0 = pack_i8x4(q[0], q[1], q[2], q[3]), w1 = pack_i8x4(q[4], q[5], q[6], q[7]);
    if (row < N_) *reinterpret_cast<uint2*>(obase + (int64_t)lr * p.osn) = make_uint2(w0, w1);
  }
}

// block `blk` (BLK rows) of head (b, h) of H.  LDS: gmax = 64 dwords, mpart = 16 x D floats (16-byte aligned)
template <int D, int BLK, bool BF16, bool TRITON>
__device__ __forceinline__ void quant_qk_int8_body(const QuantParams& p, const int blk, const int h, const int b, const int H,
                                                   unsigned int* gmax, float (*mpart)[D]) {
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
  uint4 raw[QuantGeom<D, BLK>::NP];
  quant_load_rows<D, BLK>(p.x + x_boff + h * p.xsh + (threadIdx.x % (D / 8)) * 8, p.xsn, blk, N_, raw);
  const uint4 mbits = quant_mean_bits<D, BF16>(p, h, b, H, blk == 0, mpart);
  quant_block<D, BLK, BF16, TRITON>(p, blk, h, b, H, N_, o_boff, raw, mbits, gmax, nullptr, [] {});
}

// (the rounding flavour is a template parameter: with both flavours in one kernel the register allocation of the shared
//  part suffered -- 80 instead of 65 registers at head_dim 128 -- and every element paid a uniform branch)
template <int D, int BLK, bool BF16, bool TRITON>
__global__ __launch_bounds__(256, BLK * D <= 64 * 128 ? 7 : 4) void quant_qk_int8_kernel(const QuantParams p) {
  __shared__ unsigned int gmax[64];
  __shared__ __attribute__((aligned(16))) float mpart[16][D];  // chunk sums of the mean (mean_part form: S <= 16 <= RPP)
  quant_qk_int8_body<D, BLK, BF16, TRITON>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, gmax, mpart);
}

// Streaming K quantizer (dense K, 64-row blocks): a workgroup walks `per_wg` consecutive blocks of ONE head and loads the
// rows of block i+1 before it works on block i.  With one block per workgroup a launch is one (or two) generations of
// workgroups that all sit in the same phase -- every load of the tensor issued at once, then every store -- and the
// latency chain load -> maxima -> barrier -> rounding -> store is paid once per generation with nothing beside it; here
// the next block's rows fly during the second half of the chain, stores and loads of neighbouring blocks overlap, and the
// mean of the head is finished once per workgroup instead of once per block.  Same arithmetic per block: bit-identical.
// One barrier per block: the group maxima rotate through three buffers (block i accumulates into buffer i % 3 and, behind
// its barrier, zeroes buffer (i + 2) % 3 -- last read by block i-1, whose readers have all passed this barrier, and next
// written by block i+2, behind the barrier of block i+1).
template <int D, bool BF16, bool TRITON>
__device__ __forceinline__ void k_quant_stream_body(const QuantParams& p, const int per_wg, const int wg, const int h, const int b,
                                                    const int H, unsigned int (*gmax)[64], float (*mpart)[D]) {
  constexpr int BLK = 64;
  using G = QuantGeom<D, BLK>;
  const int nblk = (p.N + BLK - 1) / BLK;
  const int blk0 = wg * per_wg, blk1 = min(blk0 + per_wg, nblk);  // the host launches no empty workgroup
  if (threadIdx.x < 192) (&gmax[0][0])[threadIdx.x] = 0u;
  const uint16_t* xbase = p.x + b * p.xsb + h * p.xsh + (threadIdx.x % G::TPR) * 8;
  const int64_t o_boff = b * p.osb;
  uint4 raw[G::NP];
  quant_load_rows<D, BLK>(xbase, p.xsn, blk0, p.N, raw);
  const uint4 mbits = quant_mean_bits<D, BF16>(p, h, b, H, wg == 0, mpart);
  int par = 0;
  for (int blk = blk0; blk < blk1; ++blk) {
    // the rows of the next block are requested as soon as this block's are unpacked (their registers are free then) and
    // fly during the barrier, the rounding and the stores of this block
    quant_block<D, BLK, BF16, TRITON>(p, blk, h, b, H, p.N, o_boff, raw, mbits, gmax[par], gmax[par == 0 ? 2 : par - 1], [&] {
      if (blk + 1 < blk1) quant_load_rows<D, BLK>(xbase, p.xsn, blk + 1, p.N, raw);
    });
    par = par == 2 ? 0 : par + 1;
  }
}

template <int D, bool BF16, bool TRITON>
__global__ __launch_bounds__(256, D == 64 ? 5 : 4) void k_quant_stream_kernel(const QuantParams p, const int per_wg) {
  __shared__ unsigned int gmax[3][64];
  __shared__ __attribute__((aligned(16))) float mpart[16][D];
  k_quant_stream_body<D, BF16, TRITON>(p, per_wg, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, gmax, mpart);
}

// ------------------------------------------------------------------------------------------------
// Fused K/V pre-pass of the FP8-PV operator, at every length (a sequence is at most 16 chunks, kmean_chunk_rows): TWO launches instead of five
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
  if ((s + 1) * rows <= N) {  // whole chunk: unconditional loads, eight in flight (see k_mean_partial_body)
#pragma unroll 8
    for (int i = 0; i < rows / RPP; ++i) {
      float f[8];
      unpack8<BF16>(*reinterpret_cast<const uint4*>(base + (int64_t)(s * rows + i * RPP + tr) * sn), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) am[j] = fmaxf(am[j], fabsf(f[j]));
    }
  } else {
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

template <int D, bool BF16, bool TRITON>
__global__ __launch_bounds__(256, D == 64 ? 8 : 7) void kv_quant_kernel(const QuantParams p, const VPrepParams q, const int nblk_k) {
  using G = VQuantGeom<D>;
  __shared__ unsigned int gmax[64];
  __shared__ __attribute__((aligned(16))) float exch[16][D];  // chunk partials of this head: K sums or V max|v|
  __shared__ __attribute__((aligned(16))) uint32_t tile[G::BLKS * G::IMG];
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  if ((int)blockIdx.x < nblk_k) {
    quant_qk_int8_body<D, 64, BF16, TRITON>(p, blockIdx.x, h, b, H, gmax, exch);
    return;
  }
  const int bx = blockIdx.x - nblk_k;
  const int tg = threadIdx.x / G::TPR, tc = threadIdx.x % G::TPR;
  // the unit's rows first: the statistics below overlap with this trip to HBM
  uint4 raw[4];
  v_quant_load<D>(q.v + b * q.sb + h * q.sh, q.sn, p.N, bx, raw);
  if (tg < p.S) {  // one round trip for all chunks (S <= 16 <= token groups per workgroup)
    const float* pp = q.part + (((int64_t)b * H + h) * p.S + tg) * D + tc * 8;
    *reinterpret_cast<float4*>(&exch[tg][tc * 8]) = *reinterpret_cast<const float4*>(pp);
    *reinterpret_cast<float4*>(&exch[tg][tc * 8 + 4]) = *reinterpret_cast<const float4*>(pp + 4);
  }
  __syncthreads();
  // the per-channel scale is finished by the FIRST token group and handed over through LDS (every thread doing it for itself:
  // 8 S maxima and 16 IEEE divisions per thread for 32 elements of real work)
  if (tg == 0) {  // (this thread alone reads columns tc*8 .. tc*8+7 of exch, so it may overwrite them in row 0)
    float rc[8], vs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = exch[0][tc * 8 + j];
      for (int s_ = 1; s_ < p.S; ++s_) a = fmaxf(a, exch[s_][tc * 8 + j]);
      rc[j] = q.scale_max / a;   // v_stats_final_kernel
      vs[j] = a / q.scale_max;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) exch[0][tc * 8 + j] = rc[j];
    if (bx == 0) {
      float* o = q.v_scale + ((int64_t)b * H + h) * D + tc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(vs[0], vs[1], vs[2], vs[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(vs[4], vs[5], vs[6], vs[7]);
    }
  }
  __syncthreads();
  float mean[8], rcp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = 0.f; rcp[j] = exch[0][tc * 8 + j]; }
  v_quant_to_image<D, BF16>(raw, p.N, bx, mean, rcp, tile);
  __syncthreads();
  v_quant_store_image<D>(q.out, q.ob, q.oh, q.od, q.o_tile, p.N, bx, h, b, tile);
}

// Launch B as a STREAMING kernel for long sequences (round 3; sage_kv_prepare_fp8 picks by the units a workgroup would walk): workgroups [0, nwg_k) walk per_k consecutive K blocks
// of their head (k_quant_stream_body), the others per_v consecutive V units (BLKS x 64 tokens), the next unit's rows
// requested while this one is transposed, the image double-buffered in LDS (one barrier per unit), and the per-channel
// scale -- S maxima and two IEEE divisions per channel -- finished ONCE per workgroup by its first token group instead of by
// every thread for every unit (that was three quarters of the V half's vector work).  Same arithmetic: bit-identical.
template <int D, bool BF16, bool TRITON>
__global__ __launch_bounds__(256, D == 64 ? 5 : 4) void kv_quant_stream_kernel(const QuantParams p, const VPrepParams q, const int per_k,
                                                                               const int nwg_k, const int per_v) {
  using G = VQuantGeom<D>;
  __shared__ unsigned int gmax[3][64];
  __shared__ __attribute__((aligned(16))) float exch[16][D];  // chunk partials of this head: K sums or V max|v|
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][G::BLKS * G::IMG];
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  if ((int)blockIdx.x < nwg_k) {
    k_quant_stream_body<D, BF16, TRITON>(p, per_k, blockIdx.x, h, b, H, gmax, exch);
    return;
  }
  const int wg = blockIdx.x - nwg_k;
  const int nunits = ((p.N + 63) / 64 + G::BLKS - 1) / G::BLKS;
  const int u0 = wg * per_v, u1 = min(u0 + per_v, nunits);
  const uint16_t* vhead = q.v + b * q.sb + h * q.sh;
  uint4 raw[4];
  v_quant_load<D>(vhead, q.sn, p.N, u0, raw);
  const int tg = threadIdx.x / G::TPR, tc = threadIdx.x % G::TPR;
  if (tg < p.S) {  // one round trip for all chunks (S <= 16 <= token groups per workgroup)
    const float* pp = q.part + (((int64_t)b * H + h) * p.S + tg) * D + tc * 8;
    *reinterpret_cast<float4*>(&exch[tg][tc * 8]) = *reinterpret_cast<const float4*>(pp);
    *reinterpret_cast<float4*>(&exch[tg][tc * 8 + 4]) = *reinterpret_cast<const float4*>(pp + 4);
  }
  __syncthreads();
  if (tg == 0) {  // (this thread alone reads columns tc*8 .. tc*8+7 of exch, so it may overwrite them in rows 0 and 1)
    float rc[8], vs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = exch[0][tc * 8 + j];
      for (int s_ = 1; s_ < p.S; ++s_) a = fmaxf(a, exch[s_][tc * 8 + j]);
      rc[j] = q.scale_max / a;   // v_stats_final_kernel
      vs[j] = a / q.scale_max;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { exch[0][tc * 8 + j] = rc[j]; exch[1][tc * 8 + j] = vs[j]; }
    if (wg == 0) {
      float* o = q.v_scale + ((int64_t)b * H + h) * D + tc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(vs[0], vs[1], vs[2], vs[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(vs[4], vs[5], vs[6], vs[7]);
    }
  }
  __syncthreads();
  float mean[8], rcp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = 0.f; rcp[j] = exch[0][tc * 8 + j]; }
  for (int u = u0; u < u1; ++u) {
    uint32_t* img = tile[(u - u0) & 1];
    v_quant_to_image<D, BF16>(raw, p.N, u, mean, rcp, img);
    if (u + 1 < u1) v_quant_load<D>(vhead, q.sn, p.N, u + 1, raw);
    // one barrier per unit: the image buffers alternate, and the readers of this unit's buffer two units ago all passed the
    // previous unit's barrier before anyone writes it again
    __syncthreads();
    v_quant_store_image<D>(q.out, q.ob, q.oh, q.od, q.o_tile, p.N, u, h, b, img);
  }
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
#define LAUNCH(DD, BL, BF)                                                                                              \
  do {                                                                                                                  \
    if (rounding == SAGE_ROUND_TRITON) hipLaunchKernelGGL((quant_qk_int8_kernel<DD, BL, BF, true>), grid, dim3(256), 0, st, p);  \
    else hipLaunchKernelGGL((quant_qk_int8_kernel<DD, BL, BF, false>), grid, dim3(256), 0, st, p);                      \
  } while (0)
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

// Blocks per workgroup of the streaming K quantizer: as many as leave about as many workgroups per CU as its registers allow
// to be resident (five at head_dim 64, four at 128), so that every workgroup is resident from the start and streams its share
// of a head.
static int k_quant_blocks_per_wg(int B, int H, int N, int D) {
  const int64_t nblk = (N + 63) / 64, total = (int64_t)B * H * nblk, target = 256 * (D == 64 ? 5 : 4);
  const int64_t per = (total + target - 1) / target;
  return (int)(per < 1 ? 1 : per > nblk ? nblk : per);
}

static int launch_k_quant(const QuantParams& p, int dtype, int B, int H, int N, int D, hipStream_t st) {
  const int nblk = (N + 63) / 64, per_wg = k_quant_blocks_per_wg(B, H, N, D);
  const dim3 grid((nblk + per_wg - 1) / per_wg, H, B);
  launch_begin();
#define LAUNCH(DD, BF)                                                                                                  \
  do {                                                                                                                  \
    if (p.rounding == SAGE_ROUND_TRITON) hipLaunchKernelGGL((k_quant_stream_kernel<DD, BF, true>), grid, dim3(256), 0, st, p, per_wg);  \
    else hipLaunchKernelGGL((k_quant_stream_kernel<DD, BF, false>), grid, dim3(256), 0, st, p, per_wg);                 \
  } while (0)
  if (D == 64) { if (dtype == SAGE_BF16) LAUNCH(64, true); else LAUNCH(64, false); }
  else { if (dtype == SAGE_BF16) LAUNCH(128, true); else LAUNCH(128, false); }
#undef LAUNCH
  return launch_status();
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
  QuantParams p;
  const int st0 = quant_impl(k, dtype, B, H, N, D, nullptr, out, scale, gran, 1, 64, 64, 1.0f, rounding, nullptr, 1, nullptr, stream,
                             nullptr, 0, nullptr, ws, S, km, &p);
  if (st0 != SAGE_OK) return st0;
  return launch_k_quant(p, dtype, B, H, N, D, st);
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
  const int vq_blks = D == 128 ? 1 : 2;  // 64-token blocks per V unit (VQuantGeom<D>::BLKS)
  const int nunit_v = (nblk_k + vq_blks - 1) / vq_blks;
  // K blocks / V units per workgroup: each half gets about half of the workgroups the chip holds at once (k_quant_blocks_per_wg)
  const int64_t target = 128 * (D == 64 ? 5 : 4);
  // (at most SAGE_KV_UNITS_CAP units per workgroup: beyond, more generations of workgroups measured better than longer walks)
  auto per_wg = [&](int units) { int64_t per = ((int64_t)B * H * units + target - 1) / target; per = per > SAGE_KV_UNITS_CAP ? SAGE_KV_UNITS_CAP : per; return (int)(per < 1 ? 1 : per > units ? units : per); };
  const int per_k = per_wg(nblk_k), per_v = per_wg(nunit_v);
  const int nwg_k = (nblk_k + per_k - 1) / per_k, nwg_v = (nunit_v + per_v - 1) / per_v;
  hipStream_t st = (hipStream_t)stream;
  launch_begin();
  const dim3 ga(2 * S, H, B), gb(nwg_k + nwg_v, H, B);
  const uint16_t* kp = (const uint16_t*)k->data;
#define LA(DD, BF)                                                                                                          \
  hipLaunchKernelGGL((kv_partial_kernel<DD, BF>), ga, dim3(256), 0, st, kp, k->stride_b, k->stride_h, k->stride_n, q.v, q.sb, \
                     q.sh, q.sn, N, kpart, vpart, S)
  // Streaming pays where a workgroup walks enough units to hide its start-up (measured, tools/prepass_ab.py, B*H = 128: head_dim
  // 128: 4-8 units per workgroup +6..10 % slower than one unit per workgroup, 16-32 units 6-12 % faster, 64 even; head_dim 64:
  // 2 units slower, 4 faster); below that the one-unit-per-workgroup kernel (seven or eight workgroups per CU) runs.
  const bool stream_b = per_v >= (D == 64 ? 4 : 12);
  const int nblk_v1 = nunit_v;
  const dim3 gb1(nblk_k + nblk_v1, H, B);
#define LB(DD, BF)                                                                                                      \
  do {                                                                                                                  \
    if (stream_b) {                                                                                                     \
      if (rounding == SAGE_ROUND_TRITON) hipLaunchKernelGGL((kv_quant_stream_kernel<DD, BF, true>), gb, dim3(256), 0, st, p, q, per_k, nwg_k, per_v);  \
      else hipLaunchKernelGGL((kv_quant_stream_kernel<DD, BF, false>), gb, dim3(256), 0, st, p, q, per_k, nwg_k, per_v); \
    } else {                                                                                                            \
      if (rounding == SAGE_ROUND_TRITON) hipLaunchKernelGGL((kv_quant_kernel<DD, BF, true>), gb1, dim3(256), 0, st, p, q, nblk_k);  \
      else hipLaunchKernelGGL((kv_quant_kernel<DD, BF, false>), gb1, dim3(256), 0, st, p, q, nblk_k);                   \
    }                                                                                                                   \
  } while (0)
  const bool bf = dtype == SAGE_BF16;
  if (D == 64) { if (bf) LA(64, true); else LA(64, false); } else { if (bf) LA(128, true); else LA(128, false); }
  if (launch_status() != SAGE_OK) return SAGE_ERR_LAUNCH;
  if (D == 64) { if (bf) LB(64, true); else LB(64, false); } else { if (bf) LB(128, true); else LB(128, false); }
#undef LA
#undef LB
  return launch_status();
}
