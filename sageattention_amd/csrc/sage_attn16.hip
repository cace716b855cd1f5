// INT8-QK^T -> online softmax -> FP16/BF16-PV attention on the 16x16 MFMA shapes of gfx950
// (v_mfma_i32_16x16x64_i8, v_mfma_f32_16x16x32_{f16,bf16}).
//
// Same operator, staging and software pipeline as attn_i8_kernel (sage_attn.hip; replaces the tile loop of
// csrc/qattn/qk_int_sv_f16_cuda_sm80.cu:263-355), different fragment family (SURVEY 2.2 rows K5/K6 name both).  Why it
// exists: under this operator the chip is power limited, and bare MFMA loops on random data deliver more FLOP/s from the
// 16x16 shapes than from the 32x32 ones at equal cycles per FLOP (tools/mfma_power.hip: i8 4004 vs 3120, f16 2001 vs 1678
// TFLOP/s; MI355X_MICROARCH.md "DVFS give-back" item 7).  The price is twice the MFMA issue slots.  Kept or deleted by an
// in-process A/B on random data (tools/ab_bench.py --mfma16).
//
// Fragment layout (lane l: c = l & 15, g = l >> 4; a wave owns 32 query rows = query tiles qt 0/1 of 16):
//  * S^T tile (kt, qt) = K[16 keys of key tile kt] . Q^T[16 queries of qt]: A = K rows from LDS (ds_read_b128: row
//    16kt + c, bytes 64ks + 16g), B = Q^T resident in registers (row 16qt + c, same bytes); ONE K fragment feeds the two
//    query tiles, so the LDS traffic per FLOP equals the 32x32 kernel's.  Result register i of lane (c, g): key
//    16kt + 4g + i, query 16qt + c -- a lane owns TWO query rows and 16 of each row's 64 keys; lanes c, c+16, c+32, c+48
//    share a row (row max: 3 permlane swaps + 2 max for both rows).
//  * P^T: the accumulators of key tiles (2s, 2s+1), packed to fp16/bf16, ARE the B operand of k-step s of
//    O^T[16 d][16 queries] += V^T[16 d][32 keys] . P^T (k slot 8g + j <-> key 32s + 16(j >> 2) + 4g + (j & 3)); the matching
//    A operand = two ds_read_b64_tr_b16 of the row-major V tile (4 keys x 16 channels per 16-lane group), again shared by
//    both query tiles.
//  * O^T tile (dt, qt): register i of lane (c, g) = channel 16dt + 4g + i of query 16qt + c.
#include "sage_attn_common.h"
#include "sage_attn_ablate.h"

namespace sage {

template <int D>
__device__ __forceinline__ int k_swz16(int row) {
  // 16-B chunk XOR of the K tile image for the 16x16x64 A-fragment reads (lane group g reads chunk 4ks + g of row c).
  // head_dim 128: the 32x32 kernel's swizzle is conflict free for this access pattern as well; head_dim 64 (four rows
  // per 256-B bank row): rows 8..15 take the other chunk pair.
  if constexpr (D == 128) return (row >> 1) & 7; else return ((row >> 3) & 1) << 1;
}
template <int D>
__device__ __forceinline__ int v_swz16(int row) {
  // 32-B chunk XOR of the fp16 V tile: a 32-lane half of a transposing read takes rows 8n .. 8n+7 of ONE 32-B column chunk
  if constexpr (D == 128) return row & 7; else return (row >> 1) & 3;
}

__device__ __forceinline__ float swap16_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap16_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int D, int NWAVES, bool CAUSAL, bool KTHREAD, bool V_BF16>
__global__ __launch_bounds__(NWAVES * 64, SAGE_MINWAVES)
void attn16_kernel(const AttnParams p) {
  constexpr int T = NWAVES * 64;
  constexpr int QB = NWAVES * 32;
  constexpr int KS = D / 64;          // k-steps of the int8 QK^T MFMA
  constexpr int DT = D / 16;          // 16-wide d tiles of O^T
  constexpr int KBYTES = 64 * D;      // one K tile (int8)
  constexpr int VBYTES = 64 * D * 2;  // one V tile: 16-bit [64][D]
  constexpr int KCH = D / 16;         // 16-B chunks per K row
  constexpr int VCH = D / 8;          // 16-B chunks per V row
  constexpr int KC = (64 * KCH + T - 1) / T;
  constexpr int VC = (64 * VCH) / T;
  static_assert((64 * VCH) % T == 0, "V tile must divide over the workgroup");
  constexpr int RING = 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const k_lds = smem;
  char* const v_lds = smem + RING * KBYTES;

  // ---- block -> (b, h, q block), XCD aware (as attn_i8_kernel)
  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x, xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  }
  int qb = lid % p.nqb;
  const int bh = lid / p.nqb;
  const int h = bh % p.Hq, b = bh / p.Hq;
  if constexpr (CAUSAL) qb = p.nqb - 1 - qb;  // heaviest q-blocks first
  const int hk = h / (p.Hq / p.Hk);
  const int M_ = p.M, N_ = p.N;
  const int64_t q_boff = b * p.qsb, k_boff = b * p.ksb, v_boff = b * p.vsb, o_boff = b * p.osb;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int q0 = qb * QB + wave * 32;
  int c_l = c, g_l = g;  // re-derived after the fast loop (registers)

  // ---- Q^T fragments (B operand), resident: rows q0 + 16qt + c, bytes 64ks + 16g; one q scale per lane (the two rows of
  //      a lane share r % 8 and their 32-row block, so per-thread / per-warp(32) / per-block scales coincide)
  v4i qf[2][KS];
  float qsc;
  auto prepare_q = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int rowc = min(q0 + 16 * qt + c, M_ - 1);
      const int8_t* qp = p.q + q_boff + h * p.qsh + (int64_t)rowc * p.qsn + 16 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = *reinterpret_cast<const v4i*>(qp + 64 * ks);
    }
    const int rowc = min(q0 + c, M_ - 1);
    int qi;  // …sm80.cu:103-117 index maps
    if (p.qgran == SAGE_GRAN_PER_BLOCK) qi = rowc / p.blkq;
    else if (p.qgran == SAGE_GRAN_PER_WARP) qi = rowc / p.warpq;
    else qi = (rowc / p.warpq) * 8 + (rowc & 7);
    qsc = p.q_scale[((int64_t)b * p.Hq + h) * p.gq + qi] * p.logit_mult;
  };
  const float* ksp = p.k_scale + b * p.ks_b + hk * p.ks_h;

  // ---- tile range
  const int kv_end = CAUSAL ? min(N_, (qb + 1) * QB) : N_;
  const int ntiles = (kv_end + 63) >> 6;
  const int wave_tiles = CAUSAL ? min(ntiles, ((q0 + 31) >> 6) + 1) : ntiles;

  // ---- staging (global -> LDS by LDS-DMA; the bank swizzle is applied to the per-lane SOURCE offset)
  const int8_t* kg = p.k + k_boff + hk * p.ksh;
  const uint8_t* vg = p.v + (v_boff + hk * p.vsh) * 2;
  const int k_tile_stride = p.k_tile_bytes, v_tile_stride = p.v_tile_bytes;
  const int last_t = (N_ - 1) >> 6, last_r = (N_ - 1) & 63;
  const unsigned k_bytes = (unsigned)((int64_t)last_t * k_tile_stride + (int64_t)last_r * p.ksn + D);
  const unsigned v_bytes = (unsigned)((int64_t)last_t * v_tile_stride + ((int64_t)last_r * p.vsn + D) * 2);
  const v4i k_rsrc = make_rsrc(kg, k_bytes), v_rsrc_dma = make_rsrc(vg, v_bytes);
  int k_voff[KC], v_voff[VC];
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const int cc = tid + i * T, kr = cc / KCH, pos = cc % KCH;
    k_voff[i] = kr * (int)p.ksn + ((pos ^ k_swz16<D>(kr)) << 4);
  }
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    const int cc = tid + i * T, vr = cc / VCH, pos = cc % VCH;
    v_voff[i] = vr * (int)p.vsn * 2 + ((pos ^ (v_swz16<D>(vr) << 1)) << 4);
  }
  auto dma_k = [&](const int j, const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KC; ++i)
      if (KC * T == 64 * KCH || wave * 64 + i * T < 64 * KCH)
        lds_dma16(k_rsrc, (unsigned)(buf * KBYTES + (wave * 64 + i * T) * 16), k_voff[i], j * k_tile_stride);
  };
  auto load_v = [&](const int j, const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < VC; ++i)
      lds_dma16(v_rsrc_dma, (unsigned)(RING * KBYTES + buf * VBYTES + (wave * 64 + i * T) * 16), v_voff[i], j * v_tile_stride);
  };
  // ---- lane-constant LDS read pointers
  const char* k_rd[KS];  // K A-fragment: row c (+16kt via immediate), chunk 4ks + g
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = k_lds + (c * D + (((4 * ks + g) ^ k_swz16<D>(c)) << 4));
  const char* v_rd[DT];  // V^T fragment via tr-read: key row 4g + q (+32s, +16 via immediate), 32-B chunk dt, 8p bytes in
  {
    const int q4 = c >> 2, p4 = c & 3, rv = 4 * g + q4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) v_rd[dt] = v_lds + (rv * (2 * D) + ((dt ^ v_swz16<D>(rv)) << 5) + 8 * p4);
  }

  // ---- state: two query rows per lane
  v4f acc_o[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = v4f{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-1e30f, -1e30f};
  float l_run[2] = {0.f, 0.f};  // per-lane partial row sums of the unrounded p (fp32, VALU), the lane's 16 keys per tile
#ifdef SAGE16_MROW
  constexpr bool MROW = !V_BF16;  // experiment: row sums of the fp16-ROUNDED P on the matrix pipe (A = ones: every result row is the full sum)
#else
  constexpr bool MROW = false;
#endif
  v4f l4[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
  v8h ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = (_Float16)1.0f;
  if constexpr (MROW) asm volatile("" : "+v"(ones8));
  constexpr int kBiasI = 0x4B400000;      // the int32 accumulator starts at the bit pattern of 1.5 * 2^23 (see sage_attn.hip)
  constexpr float kBiasF = 12582912.0f;
  constexpr int kMaskedI = (int)0xFF800000u;
  v4i bias = {kBiasI, kBiasI, kBiasI, kBiasI};
  asm volatile("" : "+v"(bias));  // resident (4 registers): as a known constant hipcc re-creates it per S chain

  typedef v4i stile[4][2];  // S^T of one tile: [key tile][query tile]

  float qsc_lo = 0.f, qsc_hi = 0.f;  // = (g & 1) ? (0, qsc) : (qsc, 0)
  auto load_kscales = [&](const int j) __attribute__((always_inline)) -> float4 {
    if constexpr (KTHREAD) return uniform_load4(ksp + j * p.ks_t);
    else return make_float4(uniform_load1(ksp + j * p.ks_t), 0.f, 0.f, 0.f);
  };
  // dequantisation scales of the lane: key 16kt + 4g + i has scale index (key % 8) / 2 = 2 (g & 1) + (i >> 1)
  auto scales_from = [&](const float4 kk, float& sc0, float& sc1) __attribute__((always_inline)) {
    if constexpr (KTHREAD) {
      sc0 = __builtin_fmaf(kk.z, qsc_hi, kk.x * qsc_lo);
      sc1 = __builtin_fmaf(kk.w, qsc_hi, kk.y * qsc_lo);
    } else {
      sc0 = sc1 = qsc * kk.x;
    }
  };
  // sequence end / causal diagonal: register i of tile (kt, qt) holds key 64j + 16kt + 4g + i of query q0 + 16qt + c
  auto mask_limit = [&](const int j, stile& s) __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int lim = min(N_ - 1, CAUSAL ? q0 + 16 * qt + c_l : 0x7fffffff) - (j << 6) - 4 * g_l;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[kt][qt][i] = (16 * kt + i <= lim) ? s[kt][qt][i] : kMaskedI;
    }
  };
  // row max of one tile for both query rows of the lane, from the raw integers (positive scales), all four lanes of a row
  auto finish_max = [&](const int (&mxa)[2], const int (&mxb)[2], const float sc0, const float sc1, float (&mx)[2])
      __attribute__((always_inline)) {
    float m0, m1;
    if constexpr (KTHREAD) {
      m0 = max_raw((__int_as_float(mxa[0]) - kBiasF) * sc0, (__int_as_float(mxb[0]) - kBiasF) * sc1);
      m1 = max_raw((__int_as_float(mxa[1]) - kBiasF) * sc0, (__int_as_float(mxb[1]) - kBiasF) * sc1);
    } else {
      m0 = (__int_as_float(max(mxa[0], mxb[0])) - kBiasF) * sc0;
      m1 = (__int_as_float(max(mxa[1], mxb[1])) - kBiasF) * sc0;
    }
    // lanes g even carry query tile 0 onwards, lanes g odd query tile 1: swap(m0, m1) puts the neighbour's value of the
    // lane's OWN tile in the other register
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m0), __float_as_uint(m1), false, false);
    float y = max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));  // g even: tile 0 over {g, g+1}; g odd: tile 1 over {g-1, g}
    y = swap_max(y);                                                  // ... over all four lanes
    auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    mx[0] = __uint_as_float(r2[0]);  // even rows' value everywhere = tile 0
    mx[1] = __uint_as_float(r2[1]);  // odd rows' value everywhere  = tile 1
  };
  auto row_max = [&](const stile& s, const float sc0, const float sc1, float (&mx)[2]) __attribute__((always_inline)) {
    int mxa[2], mxb[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      mxa[qt] = s[0][qt][0]; mxb[qt] = s[0][qt][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i & 2) mxb[qt] = max(mxb[qt], s[kt][qt][i]); else mxa[qt] = max(mxa[qt], s[kt][qt][i]);
        }
    }
    finish_max(mxa, mxb, sc0, sc1, mx);
  };
  // lazy rescale (threshold 2^6, as attn_i8_kernel): both rows of the lane decide together
  constexpr float kLazyThr = 6.0f;
  float m_thr[2] = {m_run[0] + kLazyThr, m_run[1] + kLazyThr};
  auto maybe_rescale = [&](const float (&mx)[2]) __attribute__((always_inline)) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx[0] > m_thr[0] || mx[1] > m_thr[1]) != 0, 0)) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float m_new = fmaxf(m_run[qt], mx[qt]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        m_thr[qt] = m_new + kLazyThr;
        l_run[qt] *= alpha;
        if constexpr (MROW) l4[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
      }
    }
  };
  auto pack_p = [&](const v2f two) __attribute__((always_inline)) -> v2h {
    if constexpr (V_BF16) return __builtin_bit_cast(v2h, __builtin_convertvector(two, v2bf));
    else return __builtin_convertvector(two, v2h);
  };
  auto pv_mfma = [&](const v8h a, const v8h bq, const v4f cc) __attribute__((always_inline)) -> v4f {
    if constexpr (V_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, bq), cc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, cc, 0, 0, 0);
  };

  // ---- software pipeline: as attn_i8_kernel (K(j) / V(j) in ring slot j & 1; S(j+1) is computed while S(j) is
  //      exponentiated and P(j).V(j) accumulated; K(j+2), V(j+1) copied meanwhile and drained in front of the barrier)
  int n_plain = wave_tiles;
  if (N_ & 63) n_plain = min(n_plain, N_ >> 6);
  if constexpr (CAUSAL) n_plain = min(n_plain, max(0, (q0 + 1) >> 6));  // tile j needs no mask iff 64j + 63 <= q0
  const int n_fast = max(0, min(n_plain - 1, wave_tiles - 1));

  dma_k(0, 0);
  load_v(0, 0);
  if (ntiles > 1) dma_k(1, 1);
  prepare_q();
  dma_wait_all();
  qsc_lo = (g & 1) ? 0.f : qsc;
  qsc_hi = (g & 1) ? qsc : 0.f;
  __syncthreads();

  stile s_cur, s_nxt;
  float sc0, sc1, mx_cur[2];
  // S^T of a whole tile out of K slot `kbuf` (prologue only; the loop spreads these MFMAs through its stream)
  auto qk = [&](const int kbuf, stile& s) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const v4i a = *reinterpret_cast<const v4i*>(k_rd[ks] + (kbuf * KBYTES + kt * 16 * D));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          s[kt][qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, qf[qt][ks], ks == 0 ? bias : s[kt][qt], 0, 0, 0);
      }
  };
  qk(0, s_cur);
  __syncthreads();  // iteration 0 re-fills K slot 0: every wave must have read K(0) first (see attn_i8_kernel)
  scales_from(load_kscales(0), sc0, sc1);
  mask_limit(0, s_cur);
  row_max(s_cur, sc0, sc1, mx_cur);

  float4 kk_nxt = load_kscales(min(1, ntiles - 1));
  int k_slot = 0, v_slot = 0;
  // NEXT: what follows tile j for this wave -- 0 a plain tile (compile-time ring slots), 1 a tile that may need masking,
  // 2 nothing (the wave's last tile); 1 and 2 run the same stream with run-time slots (the read pointers are moved).
  auto fast_iter = [&](auto par_tag, auto next_tag, const int j, stile& sa, stile& sb, float& a0, float& a1, float& b0, float& b1)
      __attribute__((always_inline)) {
    constexpr int R = decltype(par_tag)::value;
    constexpr int NEXT = decltype(next_tag)::value;
    constexpr bool DYN = NEXT != 0;
    const int K_RD = DYN ? (j + 1) % RING : (R + 1) % RING, V_RD = DYN ? j % RING : R, K_WR = V_RD,
              V_WR = DYN ? (j + RING - 1) % RING : (R + RING - 1) % RING;
    if constexpr (DYN) {
      if constexpr (NEXT != 2) {
        const int dk = (K_RD - k_slot) * KBYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] += dk;
        k_slot = K_RD;
      }
      const int dv = (V_RD - v_slot) * VBYTES;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) v_rd[dt] += dv;
      v_slot = V_RD;
    }
    const int kb = DYN ? 0 : K_RD * KBYTES, vb = DYN ? 0 : V_RD * VBYTES;
    auto k_frag = [&](const int i) __attribute__((always_inline)) -> v4i {  // i = kt * KS + ks
      return *reinterpret_cast<const v4i*>(k_rd[i % KS] + (kb + (i / KS) * 16 * D));
    };
    // V^T fragment of k-step s (keys 32s + 4g + {0..3} and + 16), channels 16dt + c
    auto v_frag = [&](const int s, const int dt) __attribute__((always_inline)) -> v8h {
      const char* base = v_rd[dt] + (vb + 32 * s * (2 * D));
      const v4s_vs lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base));
      const v4s_vs hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base + 16 * (2 * D)));
      v8h a;
      a.s0123 = __builtin_bit_cast(v4h, lo);
      a.s4567 = __builtin_bit_cast(v4h, hi);
      return a;
    };
    v4i kf = qf[0][0];
    if constexpr (NEXT != 2) {
      scales_from(kk_nxt, b0, b1);
      kk_nxt = load_kscales(min(j + 2, ntiles - 1));
      kf = k_frag(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    maybe_rescale(mx_cur);
    if (j + 2 < ntiles) dma_k(j + 2, K_WR);
    if (!DYN || j + 1 < ntiles) load_v(j + 1, V_WR);
    const float cq[2][2] = {{__builtin_fmaf(-kBiasF, a0, -m_run[0]), __builtin_fmaf(-kBiasF, a1, -m_run[0])},
                            {__builtin_fmaf(-kBiasF, a0, -m_run[1]), __builtin_fmaf(-kBiasF, a1, -m_run[1])}};
    // P pair `idx` (0..7) of half h (key tiles 2h, 2h+1): query tile idx & 1, key tile 2h + ((idx >> 1) & 1), registers
    // 2 (idx >> 2), +1 (one dequantisation scale per pair) -> elements 4 (kt & 1) + 2 (idx >> 2), +1 of the B fragment
    float pend[2];
    auto p_pair = [&](const int hh2, const int idx, v8h (&pf)[2]) __attribute__((always_inline)) {
      const int qt = idx & 1, k2 = (idx >> 1) & 1, hp = idx >> 2, kt = 2 * hh2 + k2, i0 = 2 * hp;
      const float sc = hp ? a1 : a0, cc = cq[qt][hp];
      v2f two = {__builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[kt][qt][i0]), sc, cc)),
                 __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[kt][qt][i0 + 1]), sc, cc))};
      const v2h ph = pack_p(two);
      pf[qt][4 * k2 + i0] = ph[0];
      pf[qt][4 * k2 + i0 + 1] = ph[1];
      pend[0] = two[0]; pend[1] = two[1];
    };
    auto p_sum = [&](const int idx) __attribute__((always_inline)) {  // row sums of pair idx, one step after the pair itself
      if constexpr (!MROW) {
        l_run[idx & 1] += pend[0];
        l_run[idx & 1] += pend[1];
      }
    };
    auto m_sum = [&](const v8h (&pf)[2]) __attribute__((always_inline)) {
      if constexpr (MROW) {
        l4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[0], l4[0], 0, 0, 0);
        l4[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[1], l4[1], 0, 0, 0);
      }
    };
#define SAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
    v8h pf0[2], pf1[2];
    // ---- region A: the 16 S(j+1) MFMAs (each K fragment feeds both query tiles) beside P(half 0)
    constexpr int NKF = 4 * KS;   // K fragments per tile
    constexpr int PPA = 8 / NKF;  // P pairs beside one K fragment (1 at head_dim 128, 2 at 64)
    v8h vf = v_frag(0, 0), vn;
#pragma unroll
    for (int i = 0; i < NKF; ++i) {
      if constexpr (NEXT != 2) {
        const int kt = i / KS, ks = i % KS;
        const v4i a = kf;
        if (i + 1 < NKF) kf = k_frag(i + 1);
        sb[kt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, qf[0][ks], ks == 0 ? bias : sb[kt][0], 0, 0, 0);
        sb[kt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, qf[1][ks], ks == 0 ? bias : sb[kt][1], 0, 0, 0);
      }
#pragma unroll
      for (int pr = i * PPA; pr < (i + 1) * PPA; ++pr) {
        if (pr > 0) p_sum(pr - 1);
        p_pair(0, pr, pf0);
      }
      SAGE_FENCE();
    }
    // ---- region B: P.V of half 0 (one V^T fragment feeds both query tiles) beside P(half 1)
    constexpr int PPB = 8 / DT > 0 ? 8 / DT : 1;  // P pairs per d tile (1 at head_dim 128, 2 at 64)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      vn = dt + 1 < DT ? v_frag(0, dt + 1) : v_frag(1, 0);
      if (dt == 1) m_sum(pf0);
      acc_o[dt][0] = pv_mfma(vf, pf0[0], acc_o[dt][0]);
      acc_o[dt][1] = pv_mfma(vf, pf0[1], acc_o[dt][1]);
#pragma unroll
      for (int pr = dt * PPB; pr < (dt + 1) * PPB && pr < 8; ++pr) {
        p_sum(pr == 0 ? 7 : pr - 1);
        p_pair(1, pr, pf1);
      }
      vf = vn;
      SAGE_FENCE();
    }
    p_sum(7);
    // ---- region C: P.V of half 1 beside the row max of S(j+1)
    if constexpr (NEXT == 1) { if (j + 1 >= n_plain) mask_limit(j + 1, sb); }
    int mxa[2] = {sb[0][0][0], sb[0][1][0]}, mxb[2] = {sb[0][0][2], sb[0][1][2]};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (dt + 1 < DT) vn = v_frag(1, dt + 1);
      if (dt == 1) m_sum(pf1);
      acc_o[dt][0] = pv_mfma(vf, pf1[0], acc_o[dt][0]);
      acc_o[dt][1] = pv_mfma(vf, pf1[1], acc_o[dt][1]);
      if constexpr (NEXT != 2) {
        // 32 scores of the two rows spread over the DT gaps
#pragma unroll
        for (int idx = dt * (32 / DT); idx < (dt + 1) * (32 / DT); ++idx) {
          const int qt = idx >> 4, kt = (idx >> 2) & 3, i = idx & 3;
          if (i & 2) mxb[qt] = max(mxb[qt], sb[kt][qt][i]); else mxa[qt] = max(mxa[qt], sb[kt][qt][i]);
        }
        asm volatile("" : "+v"(mxa[0]), "+v"(mxb[0]), "+v"(mxa[1]), "+v"(mxb[1]));
      }
      vf = vn;
      SAGE_FENCE();
    }
#undef SAGE_FENCE
    if constexpr (NEXT != 2) finish_max(mxa, mxb, b0, b1, mx_cur);
#ifdef SAGE_EXP_FENCE_MAX
    __builtin_amdgcn_sched_barrier(0);
#endif
    dma_wait_all();
    __syncthreads();
  };
  float nsc0 = 0.f, nsc1 = 0.f;
  int j = 0;
  constexpr std::integral_constant<int, 0> kPlainNext{};
  for (; j + 1 < n_fast; j += 2) {
    fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
    fast_iter(std::integral_constant<int, 1>{}, kPlainNext, j + 1, s_nxt, s_cur, nsc0, nsc1, sc0, sc1);
  }
  auto take_next = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) { s_cur[kt][0] = s_nxt[kt][0]; s_cur[kt][1] = s_nxt[kt][1]; }
    sc0 = nsc0; sc1 = nsc1;
  };
  if (j < n_fast) {
    fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
    take_next();
    ++j;
  }
  {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    c_l = ln & 15;
    g_l = ln >> 4;
  }
  for (; j + 1 < wave_tiles; ++j) {
    fast_iter(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
    take_next();
  }
  if (j < wave_tiles) {
    fast_iter(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
    ++j;
  }
  for (; j < ntiles; ++j) {  // causal: this wave is done but still stages tiles for its workgroup
    if (j + 2 < ntiles) dma_k(j + 2, j & 1);
    if (j + 1 < ntiles) load_v(j + 1, (j + 1) & 1);
    dma_wait_all();
    __syncthreads();
  }

  // ---- epilogue: normalise, convert, store; LSE (…sm80.cu:540-668)
  float l_tot[2], inv[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    l_tot[qt] = MROW ? l4[qt][0] : swap_sum(swap16_sum(l_run[qt]));
    inv[qt] = 1.0f / l_tot[qt];
  }
  const float* vmp = p.v_mean ? p.v_mean + ((int64_t)b * p.Hk + hk) * D : nullptr;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int row = q0 + 16 * qt + c_l;
    if (row >= M_) continue;  // (a row and the three other lanes that hold it are inside or outside together)
    uint16_t* op = p.o + o_boff + h * p.osh + (int64_t)row * p.osn;
    auto run4 = [&](const int dt) __attribute__((always_inline)) -> uint2 {
      const int d0 = 16 * dt + 4 * g_l;
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = acc_o[dt][qt][e] * inv[qt];
      if (vmp) {
        const float4 vmv = *reinterpret_cast<const float4*>(vmp + d0);
        x[0] += vmv.x; x[1] += vmv.y; x[2] += vmv.z; x[3] += vmv.w;
      }
      uint2 w;
      if (p.out_bf16) {
        w.x = (uint32_t)f32_to_elem_bits<true>(x[0]) | ((uint32_t)f32_to_elem_bits<true>(x[1]) << 16);
        w.y = (uint32_t)f32_to_elem_bits<true>(x[2]) | ((uint32_t)f32_to_elem_bits<true>(x[3]) << 16);
      } else {
        w.x = (uint32_t)f32_to_elem_bits<false>(x[0]) | ((uint32_t)f32_to_elem_bits<false>(x[1]) << 16);
        w.y = (uint32_t)f32_to_elem_bits<false>(x[2]) | ((uint32_t)f32_to_elem_bits<false>(x[3]) << 16);
      }
      return w;
    };
    // lanes g and g ^ 1 hold neighbouring 4-channel runs of one row: after a half exchange (v_permlane16_swap) the even
    // lane stores 16 contiguous bytes of d tile 2k and the odd lane 16 bytes of d tile 2k + 1
#pragma unroll
    for (int dp = 0; dp < DT; dp += 2) {
      const uint2 wa = run4(dp), wb = run4(dp + 1);
      if (p.o_vec16) {
        const auto sx = __builtin_amdgcn_permlane16_swap(wa.x, wb.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(wa.y, wb.y, false, false);
        *reinterpret_cast<uint4*>(op + 16 * (dp + (g_l & 1)) + 8 * (g_l >> 1)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      } else {
        *reinterpret_cast<uint2*>(op + 16 * dp + 4 * g_l) = wa;
        *reinterpret_cast<uint2*>(op + 16 * (dp + 1) + 4 * g_l) = wb;
      }
    }
    if (p.lse && g_l == 0) p.lse[((int64_t)b * p.Hq + h) * M_ + row] = m_run[qt] + log2f(l_tot[qt]);
  }
}

// ---- launch (called by run_attn, sage_attn.hip, when the calling thread selected SAGE_TUNE_MFMA = 16)
bool attn16_supported(const AttnParams& p, bool pv_fp8) {
  return !pv_fp8 && !p.mask && !p.cu_q && !p.q_f16 && !(p.qgran != SAGE_GRAN_PER_BLOCK && p.warpq < 32);
}

template <int D, int NWAVES>
static int launch16(const AttnParams& p, bool causal, bool kthread, bool v_bf16, hipStream_t st) {
  const size_t smem = 2 * (64 * D + 64 * D * 2);
  const dim3 grid(p.nqb * p.Hq * p.B), block(NWAVES * 64);
  launch_begin();
#define SAGE_LAUNCH16(C, K, V)                                                                                   \
  do {                                                                                                           \
    auto kern = attn16_kernel<D, NWAVES, C, K, V>;                                                               \
    if (smem > 48 * 1024 &&                                                                                      \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
      return SAGE_ERR_LAUNCH;                                                                                    \
    hipLaunchKernelGGL(kern, grid, block, smem, st, p);                                                          \
  } while (0)
#define SAGE_BY_V16(C, K) do { if (v_bf16) SAGE_LAUNCH16(C, K, true); else SAGE_LAUNCH16(C, K, false); } while (0)
#define SAGE_BY_K16(C) do { if (kthread) SAGE_BY_V16(C, true); else SAGE_BY_V16(C, false); } while (0)
  if (causal) SAGE_BY_K16(true); else SAGE_BY_K16(false);
#undef SAGE_BY_K16
#undef SAGE_BY_V16
#undef SAGE_LAUNCH16
  return launch_status();
}

int launch_attn16(const AttnParams& p, int D, int nwaves, bool causal, bool kthread, bool v_bf16, hipStream_t st) {
  if (nwaves == 8) return D == 64 ? launch16<64, 8>(p, causal, kthread, v_bf16, st) : launch16<128, 8>(p, causal, kthread, v_bf16, st);
  return D == 64 ? launch16<64, 4>(p, causal, kthread, v_bf16, st) : launch16<128, 4>(p, causal, kthread, v_bf16, st);
}

}  // namespace sage
