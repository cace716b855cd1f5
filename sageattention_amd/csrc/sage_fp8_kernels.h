// Pass 2 of the FP8 V quantizer (sage_fp8.hip) as a device function, shared by the stand-alone kernel and by the fused
// K/V pre-pass of sage_quant.hip.
#pragma once
#include "sage_common.h"

namespace sage {

constexpr int VQ_ROWS = 256;  // tokens per workgroup in pass 1

__host__ __device__ __forceinline__ int mfma_order_token(int pos) {
  const int h = pos >> 5, j = pos & 31;
  return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h;
}

template <int D>
struct VQuantGeom {
  static constexpr int TPR = D / 8;        // threads per token row
  static constexpr int TG = 256 / TPR;     // token groups (of 4 tokens) per workgroup
  static constexpr int BLKS = TG / 16;     // 64-token blocks per workgroup: 1 (head_dim 128) or 2 (64)
  static constexpr int IMG = D * 16 + (D / 8) * 4;  // dwords per block image
};

// Quantize + transpose.  A thread owns 8 channels of FOUR consecutive tokens: in MFMA order (sage_fp8.hip header) tokens
// 4g .. 4g+3 sit at four consecutive positions, so the thread packs them into one dword per channel and the [d][pos]
// image in LDS is written with 8 ds_write_b32 per thread (the first version wrote single bytes, 16-way bank conflicted:
// 0.24 ms at C4 against 0.16 now).  Rows of the image are 64 B; every group of 8 rows is padded by 16 B, which spreads
// the 16 channel groups of a wave over 8 banks (2-way is free for ds_write_b32) and keeps the 16-B reads aligned.
// `mean` / `rcp`: the thread's 8 channels (tc*8 ..); `bx` = workgroup index along the sequence; `tile` = BLKS*IMG dwords.
// The body in three steps (the streaming K/V pre-pass of sage_quant.hip runs them in a loop over several units, with the next
// unit's rows requested while this one is transposed); `bx` = unit index along the sequence (BLKS x 64 tokens).
// step 1: the thread's four token rows (clamped to the last row: the loads are unconditional and issued together -- under
// `if (row < N)` hipcc sinks each into its branch and waits for it before the next one, four dependent round trips per thread)
template <int D>
__device__ __forceinline__ void v_quant_load(const uint16_t* __restrict__ vhead, int64_t sn, int N, int bx, uint4 (&raw)[4]) {
  using G = VQuantGeom<D>;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (opaque per unit: keeps a loop around this from hoisting every address term into registers)
  const int tg = tid / G::TPR, tc = tid % G::TPR;
  const int row0 = (bx * G::BLKS + tg / 16) * 64 + 4 * (tg % 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const uint4*>(vhead + (int64_t)min(row0 + i, N - 1) * sn + tc * 8);
}

// step 2: scale, round to e4m3, write the thread's 8 dwords of the [d][pos] image (tile: BLKS * IMG dwords)
template <int D, bool BF16>
__device__ __forceinline__ void v_quant_to_image(const uint4 (&raw)[4], int N, int bx, const float (&mean)[8], const float (&rcp)[8],
                                                 uint32_t* __restrict__ tile) {
  using G = VQuantGeom<D>;
  constexpr int TPR = G::TPR, IMG = G::IMG;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int tg = tid / TPR, tc = tid % TPR;
  const int bi = tg / 16, t0 = 4 * (tg % 16);
  const int blk = bx * G::BLKS + bi;
  float x[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = blk * 64 + t0 + i;
    float f[8];
    unpack8<BF16>(raw[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[i][j] = row < N ? (f[j] - mean[j]) * rcp[j] : 0.f;  // pad columns are exact zeros
  }
  // token t = 32*mt + (reg&3) + 8*(reg>>2) + 4*hh  ->  pos = 32*hh + 16*mt + reg; tokens t0..t0+3 differ in reg&3 only
  const int mt = t0 >> 5, w0 = t0 & 31, hh = (w0 >> 2) & 1;
  const int pos_dw = 8 * hh + 4 * mt + (w0 >> 3);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int pk;
    asm("" : "=v"(pk));  // (`old` of the first convert: both halves are written, a literal 0 would cost a v_mov_b32)
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0][j], x[1][j], pk, false);  // OCP e4m3fn, RNE, saturating
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2][j], x[3][j], pk, true);
    const int d = tc * 8 + j;
    tile[bi * IMG + d * 16 + 4 * (d >> 3) + pos_dw] = (uint32_t)pk;
  }
}

// step 3 (behind a barrier): D rows x 64 B per block out of the image: 4 x 16 B chunks per row
template <int D>
__device__ __forceinline__ void v_quant_store_image(uint8_t* __restrict__ out, int64_t ob, int64_t oh, int64_t od, int64_t o_tile,
                                                    int N, int bx, int h, int b, const uint32_t* __restrict__ tile) {
  using G = VQuantGeom<D>;
  constexpr int BLKS = G::BLKS, IMG = G::IMG;
  for (int c = threadIdx.x; c < BLKS * D * 4; c += 256) {
    const int bo = c / (D * 4), cc = c % (D * 4), d = cc >> 2, ch = cc & 3;
    const int ob_blk = bx * BLKS + bo;
    if (ob_blk * 64 >= N) continue;
    const uint4 u = *reinterpret_cast<const uint4*>(&tile[bo * IMG + d * 16 + 4 * (d >> 3) + ch * 4]);
    *reinterpret_cast<uint4*>(out + b * ob + h * oh + (int64_t)d * od + ob_blk * o_tile + ch * 16) = u;
  }
}

template <int D, bool BF16>
__device__ __forceinline__ void v_quant_transpose_body(const uint16_t* __restrict__ v, int64_t sb, int64_t sh, int64_t sn,
                                                       int N, const float (&mean)[8], const float (&rcp)[8],
                                                       uint8_t* __restrict__ out, int64_t ob, int64_t oh, int64_t od,
                                                       int64_t o_tile, int bx, int h, int b, uint32_t* __restrict__ tile) {
  uint4 raw[4];
  v_quant_load<D>(v + b * sb + h * sh, sn, N, bx, raw);
  v_quant_to_image<D, BF16>(raw, N, bx, mean, rcp, tile);
  __syncthreads();
  v_quant_store_image<D>(out, ob, oh, od, o_tile, N, bx, h, b, tile);
}

}  // namespace sage
