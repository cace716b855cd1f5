// Fused INT8-QK^T -> online softmax -> FP16/FP8-PV attention for gfx950 (MI355X, CDNA4).
//
// Replaces the tile loop of csrc/qattn/qk_int_sv_f16_cuda_sm80.cu:44-671 and
// csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:44-713 (semantics), designed for wave64 + MFMA:
//
//  * one wave owns 32 query rows; a workgroup of NWAVES waves owns NWAVES*32 rows and shares the
//    K/V tiles (64 keys) through an LDS ring (two slots, four for FP8 PV) filled by LDS-DMA (buffer_load ... lds): a
//    copy is issued one to four tiles ahead and drained by a counted s_waitcnt in front of the barrier that publishes it,
//    so HBM/L2 latency hides under the MFMA phases and no VGPR is spent on staging.
//  * S^T = K . Q^T on v_mfma_i32_32x32x32_i8 (A = K tile rows from LDS via ds_read_b128, B = Q^T held in
//    registers for the whole kernel).  With this orientation every lane owns ONE query row
//    (column of S^T): 32 of the 64 scores of its row sit in its own registers, the other 32 in
//    lane^32, so the row max needs one v_permlane32_swap and the row sum none until the end.
//  * P^T stays in registers: the fp32 accumulator layout of S^T is, after a packed fp16 convert,
//    exactly the B operand of O^T += V^T . P^T on v_mfma_f32_32x32x16_f16; V^T fragments come
//    from the row-major V tile in LDS through ds_read_b64_tr_b16 (hardware transpose).
//  * O^T (d in registers, query row on the lane) is rescaled per lane, normalised and stored.
//
// Roofline: MFMA (4*M*N*D flop per (b,h); half int8 at 2x the fp16 rate), VALU/exp2 co-limited.
// Algorithmic HBM bytes per (b,h): M*D (Q) + N*D (K) + 2*N*D (V fp16) + 2*M*D (O) + scales.
#include "sage_attn_common.h"
#include "sage_attn_ablate.h"

namespace sage {

// K/V slots of the LDS tile ring (see the kernel): 4 for the FP8-PV loop where every wave copies a full share of each tile
constexpr int attn_ring_slots(int D, int nwaves, bool pv_fp8) { return (pv_fp8 && nwaves * 64 <= 4 * D) ? 4 : 2; }

// PV_FP8 = false: V fp16 [N][D] row major, PV on v_mfma_f32_32x32x16_f16.  V_BF16: V stays bf16 -- P is packed to bf16
//                 (v_cvt_pk_bf16_f32) and P.V runs on v_mfma_f32_32x32x16_bf16, so the tile is staged and read exactly
//                 like an fp16 one and nothing is converted (the reference converts V to fp16 first, core.py:633).
// PV_FP8 = true : V^T OCP e4m3 [D][Npad] in MFMA token order (sage_fp8.hip), PV on the MX-scaled
//                 v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x the fp16 rate), P in e4m3.
template <int D, int NWAVES, bool CAUSAL, bool KTHREAD, bool V_BF16, bool PV_FP8, bool HAS_MASK>
// (head_dim 64 FP8 PV in its dispatched 4-wave geometry is told to stay within 168 registers = three waves per SIMD: it fits
//  without scratch, but left alone hipcc settles a few registers above the line)
__global__ __launch_bounds__(NWAVES * 64, (D == 64 && PV_FP8 && NWAVES == 4 && SAGE_MINWAVES < 3) ? 3 : SAGE_MINWAVES)
void attn_i8_kernel(const AttnParams p) {
  static_assert(!(PV_FP8 && V_BF16), "fp8 V has no bf16 flavour");
  static_assert(!HAS_MASK || (!CAUSAL && !PV_FP8), "attn_mask: non-causal 16-bit-PV operator");
  constexpr int T = NWAVES * 64;
  constexpr int QB = NWAVES * 32;
  constexpr int KS = D / 32;          // k-steps of the int8 QK^T MFMA
  constexpr int DT = D / 32;          // 32-wide d tiles of O^T
  constexpr int KBYTES = 64 * D;      // one K tile (int8)
  constexpr int VBYTES = PV_FP8 ? 64 * D : 64 * D * 2;  // one V tile: fp16 [64][D] or e4m3 [D][64]
  constexpr int KCH = D / 16;         // 16-B chunks per K row
  constexpr int VCH = PV_FP8 ? 4 : D / 8;       // 16-B chunks per V tile row (fp8: a V^T row holds 64 tokens)
  constexpr int VROWS = PV_FP8 ? D : 64;        // rows of the V tile image
  constexpr int KC = (64 * KCH + T - 1) / T;    // chunks per thread
  constexpr int VC = (VROWS * VCH + T - 1) / T;
  static_assert(PV_FP8 || (64 * VCH) % T == 0, "V tile must divide over the workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const k_lds = smem;
  // K/V tile ring in LDS.  FP16 PV: two slots each (K fetched two tiles ahead, V one; every tile copy has ONE iteration to
  // land and is drained with vmcnt(0) in front of the barrier that publishes it).  FP8 PV: four slots each, K fetched four
  // tiles ahead and V three, and the per-tile wait leaves the copies of the last two iterations in flight (counted vmcnt):
  // a copy has three iterations to land, which takes the L2 round trip off the critical path (measured bound of that
  // latency on the FP8 loop, same-tile ablation: +4.8 %; tiles are half as big, so the ring costs 64 KiB at head_dim 128).
  // (counted waits need every wave to issue the same number of copies per tile: not so when a tile has fewer 16-B
  // chunks than the workgroup has threads -- head_dim 64 with 8 waves, a tuning-only geometry, keeps two slots)
  constexpr int RING = attn_ring_slots(D, NWAVES, PV_FP8);
  char* const v_lds = smem + RING * KBYTES;

  // ---- block -> (b, h, q block), XCD aware: consecutive logical ids (same head) share an L2
  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x, xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  }
  int qb = lid % p.nqb;
  const int bh = lid / p.nqb;
  const int h = bh % p.Hq, b = bh / p.Hq;
  // Causal: heaviest q-blocks of a head first (load balance at the end of the grid).  An XCD holds fewer workgroups than
  // a long head has q-blocks, so a head runs in two generations and the second starts again at key 0 (C4 reads 2.06x its
  // K/V + Q bytes from the HBM side, 1.72x with 8-wave workgroups: profiles/r03_ab/fetch_by_geometry.md; at 0.35 TB/s).
  // The opposite order, meant to let the second generation find the first one's tiles in L2, was measured: 1-2 % slower
  // on every causal shape and FETCH_SIZE went UP (742 vs 604 MB at C4).
  if constexpr (CAUSAL && !abl::kLightFirst) qb = p.nqb - 1 - qb;
  const int hk = h / (p.Hq / p.Hk);
  int M_ = p.M, N_ = p.N;
  int64_t q_boff = b * p.qsb, k_boff = b * p.ksb, v_boff = b * p.vsb, o_boff = b * p.osb;
  if (p.cu_q) {  // packed sequences: wave-uniform, before any barrier
    const int q_lo = p.cu_q[b], k_lo = p.cu_k[b];
    M_ = p.cu_q[b + 1] - q_lo;
    N_ = p.cu_k[b + 1] - k_lo;
    if (qb * QB >= M_) return;
    q_boff = (int64_t)q_lo * p.qsn; o_boff = (int64_t)q_lo * p.osn;
    k_boff = (int64_t)k_lo * p.ksn; v_boff = (int64_t)k_lo * p.vsn;
    if (N_ <= 0) {  // a sequence with queries but no keys: the reference stores zeros (acc = 0, l_i = 1;
                    // attn_qk_int8_block_varlen.py:109-121), it does not leave the rows unwritten
      const int rows = min(QB, M_ - qb * QB);
      uint16_t* ob = p.o + o_boff + h * p.osh + (int64_t)(qb * QB) * p.osn;
      for (int i = threadIdx.x; i < rows * (D / 4); i += T)
        *reinterpret_cast<uint2*>(ob + (int64_t)(i / (D / 4)) * p.osn + (i % (D / 4)) * 4) = make_uint2(0u, 0u);
      return;
    }
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int q0 = qb * QB + wave * 32;
  const int row = q0 + r;
  const int rowc = min(row, M_ - 1);
  // the lane's row / key-half as the masked tiles and the epilogue see them: re-derived from the lane id after the fast loop
  // (below), so that neither they nor the output addresses built from them occupy registers while it runs
  int row_l = row, hh_l = hh;

  // ---- Q^T fragments (B operand), resident for the whole kernel, and the per-row q scale
  // (a lambda: it runs AFTER the first K/V tile copies have been issued, below -- nothing in it depends on them, and with
  //  one workgroup per CU nobody else hides the latency of its loads; the copies and the Q loads then fly together)
  v4i qf[KS];
  float qsc;
  auto prepare_q = [&]() __attribute__((always_inline)) {
  if (p.q_f16 == nullptr) {
    const int8_t* qp = p.q + q_boff + h * p.qsh + (int64_t)rowc * p.qsn + 16 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const v4i*>(qp + 32 * ks);
    int qi;  // …sm80.cu:103-117 index maps, evaluated once per lane
    if (p.qgran == SAGE_GRAN_PER_BLOCK) qi = rowc / p.blkq;
    else if (p.qgran == SAGE_GRAN_PER_WARP) qi = rowc / p.warpq;
    else qi = (rowc / p.warpq) * 8 + (rowc & 7);
    qsc = p.q_scale[((int64_t)b * p.Hq + h) * p.gq + qi] * p.logit_mult;
  } else {
    // Fused Q quantizer (replaces one launch of K1 and the q8 round trip through HBM).  Lane (r, hh) owns the 16-column
    // chunks [32*ks + 16*hh, +16) of its row: exactly the bytes of its B fragments.  Same arithmetic as K1, so q8 and
    // the scales are bit-identical to the stand-alone quantizer; rows >= M are zeros, as there.
    const bool valid = row < M_;
    const uint16_t* qp = p.q_f16 + q_boff + h * p.qsh + (int64_t)rowc * p.qsn + 16 * hh;
    uint4 raw[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      raw[ks][0] = *reinterpret_cast<const uint4*>(qp + 32 * ks);
      raw[ks][1] = *reinterpret_cast<const uint4*>(qp + 32 * ks + 8);
    }
    // (the LSE correction q . k_mean is only computed when the caller asked for the LSE)
    const uint16_t* kmp = (p.km && p.lse) ? p.km + ((int64_t)b * p.Hk + hk) * D + 16 * hh : nullptr;
    // Everything below is instantiated per element type (ONE uniform branch here instead of one per 8-element chunk) and
    // the numerics flavour is the kernel's KTHREAD (per-thread Q scales <=> the Triton quantizer's numerics, run_attn): with
    // both as run-time flags inside the unrolled loops hipcc emitted two scalar branches per ELEMENT -- ~500 scalar
    // instructions and as many taken branches per wave, a third of a short sequence's fixed cost.
    auto fused_q = [&](auto bf16_tag) __attribute__((always_inline)) {
    constexpr bool QBF = decltype(bf16_tag)::value;
    constexpr bool triton = KTHREAD;
    if (!valid) {  // rows >= M are zeros (as in K1): zeroed once here, 4 selects per chunk instead of 8 per pass
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) raw[ks][0] = raw[ks][1] = make_uint4(0u, 0u, 0u, 0u);
    }
    float amax = 0.f, dot = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float f[8];
        unpack8<QBF>(raw[ks][c], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
        if (kmp) {
          float g[8];
          const uint4 uk = *reinterpret_cast<const uint4*>(kmp + 32 * ks + 8 * c);
          unpack8<QBF>(uk, g);
#pragma unroll
          for (int e = 0; e < 8; ++e) dot += f[e] * g[e];
        }
      }
    amax = swap_max(amax);  // the other half of the row
    // LSE correction q . k_mean of the row: parked in the caller's LSE slot of this row until the epilogue (a register that
    // lives through the whole tile loop costs the head_dim-64 FP8 variants their third wave per SIMD)
    const float lse_corr = swap_sum(dot);
    if (p.lse && hh == 0 && valid) p.lse[((int64_t)b * p.Hq + h) * M_ + row] = lse_corr;
    if constexpr (triton) {  // rows with equal r % 8 inside the 32-row block (quant_per_thread.py:27-36)
      amax = fmaxf(amax, __shfl_xor(amax, 8));
      amax = fmaxf(amax, __shfl_xor(amax, 16));
    } else {       // per warp: warpq rows (16 or 32) share a scale (fused.cu:746-750)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
      if (p.warpq == 32) amax = fmaxf(amax, __shfl_xor(amax, 16));
    }
    const float a_c = fmaxf(amax, 0.0000001f);
    const float sc = triton ? amax / 127.f + 0.0000001f : a_c / 127.f;
    const float inv = 127.f / a_c;
    const float rcp_sc = 1.0f / sc;
    // Triton numerics need x/sc correctly rounded before the half-away rounding; as in K1 the reciprocal product is
    // used unless some value of the wave's 8-column chunk lands within 2^-14 of a rounding boundary (then the exact
    // division decides, for that chunk: ~6 % of the chunks; deciding once for the wave's whole 32 x D block sent
    // 40 % of the waves through the slow form)
    const bool rcp_bad = !(fabsf(rcp_sc) < 3.0e38f);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t w[4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float f[8];
        unpack8<QBF>(raw[ks][c], f);
        int qv[8];
        // (no clamp on the two fast forms: |f| <= amax, so |f * r| <= 127 (1 + 3 ulp) and rint() of it is at most 127; as K1)
        if constexpr (triton) {
          bool near = false;
#pragma unroll
          for (int e = 0; e < 8; ++e) qv[e] = round_half_away_fast(f[e] * rcp_sc, near);
          if (__builtin_amdgcn_ballot_w64(near || rcp_bad) != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y = f[e] / sc;  // IEEE division (quant_per_thread.py:41)
              y = y + (y >= 0.f ? 0.5f : -0.5f);
              qv[e] = min(max((int)y, -128), 127);
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) qv[e] = (int)rintf(f[e] * inv);  // cvt.rni (fused.cu:176-181)
        }
        w[2 * c] = pack_i8x4(qv[0], qv[1], qv[2], qv[3]);
        w[2 * c + 1] = pack_i8x4(qv[4], qv[5], qv[6], qv[7]);
      }
      qf[ks][0] = (int)w[0]; qf[ks][1] = (int)w[1]; qf[ks][2] = (int)w[2]; qf[ks][3] = (int)w[3];
    }
    qsc = sc * p.logit_mult;
    };
    if (p.q_bf16) fused_q(std::true_type{}); else fused_q(std::false_type{});
  }
  };
  const float* ksp = p.k_scale + b * p.ks_b + hk * p.ks_h;

  // ---- tile range
  const int kv_end = CAUSAL ? min(N_, (qb + 1) * QB) : N_;
  const int ntiles = (kv_end + 63) >> 6;
  const int wave_tiles = CAUSAL ? min(ntiles, ((q0 + 31) >> 6) + 1) : ntiles;

  // ---- staging (global -> LDS by LDS-DMA).  Buffer addressing: the descriptor holds the (b, h_kv) slice, the
  //      per-thread byte offset is constant for the whole kernel and the tile advance is a scalar offset, so a
  //      tile costs no address VALU; rows >= N fall outside num_records and read as ZERO (V rows beyond the
  //      sequence must be zero: 0 * garbage could be NaN; K rows beyond it are masked in the softmax).
  const int8_t* kg = p.k + k_boff + hk * p.ksh;
  const uint8_t* vg = p.v + (v_boff + hk * p.vsh) * (PV_FP8 ? 1 : 2);
  const int k_tile_stride = p.k_tile_bytes;  // bytes per 64 keys (dense: 64 rows)
  const int v_tile_stride = p.v_tile_bytes;
  // last valid byte + 1 of the (b, h_kv) slice: row N-1 = row (N-1)%64 of tile (N-1)/64 (dense layouts: (N-1)*stride_n)
  const int last_t = (N_ - 1) >> 6, last_r = (N_ - 1) & 63;
  const unsigned k_bytes = (unsigned)((int64_t)last_t * k_tile_stride + (int64_t)last_r * p.ksn + D);
  const unsigned v_bytes = PV_FP8 ? (unsigned)((int64_t)last_t * v_tile_stride + (int64_t)(D - 1) * p.vsn + 64)
                                  : (unsigned)((int64_t)last_t * v_tile_stride + ((int64_t)last_r * p.vsn + D) * 2);
  const v4i k_rsrc = make_rsrc(kg, k_bytes), v_rsrc_dma = make_rsrc(vg, v_bytes);
  // LDS-DMA (buffer_load ... lds): a wave instruction writes 64 x 16 B = 1 KiB of LDS LINEARLY (wave-uniform
  // base + 16*lane), so the bank swizzle of the tile image is applied to the per-lane SOURCE offset instead:
  // LDS chunk position c of a tile holds global chunk (row(c), pos(c) ^ swizzle(row)).  No VGPR staging, no
  // ds_write, and the copy has a whole iteration to land (it is drained by the vmcnt(0) of the next barrier).
  // bf16 V is staged like fp16 V (16-bit elements; the transposing LDS read does not care) and multiplied as bf16.
  // History: rounds 1-2 converted the tile to fp16 on the way (head_dim 128: through registers, head_dim 64: in place in
  // LDS by the wave that copied the slice), which cost 5-8 % at head_dim 128 and 15-18 % at head_dim 64 against fp16 V.
  int k_voff[KC], v_voff[VC];
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const int c = tid + i * T, kr = c / KCH, pos = c % KCH;
    k_voff[i] = kr * (int)p.ksn + ((pos ^ k_swz<D>(kr)) << 4);
  }
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    const int c = tid + i * T, vr = c / VCH, pos = c % VCH;
    if constexpr (PV_FP8) {
      v_voff[i] = vr * (int)p.vsn + ((pos ^ ((vr >> 2) & 3)) << 4);  // V^T row vr (= channel), 16-B chunk swizzle
    } else {
      const int cc = (((pos >> 2) ^ v_win_swz<D>(vr)) << 2) | (pos & 3);
      v_voff[i] = (vr * (int)p.vsn + cc * 8) * 2;
    }
  }
  // K(j) -> K buffer `buf`
  auto dma_k = [&](int j, const int buf) __attribute__((always_inline)) {
    if constexpr (abl::kSameTile) j = 0;
#pragma unroll
    for (int i = 0; i < KC; ++i)
      if (KC * T == 64 * KCH || wave * 64 + i * T < 64 * KCH)
        lds_dma16(k_rsrc, (unsigned)(buf * KBYTES + (wave * 64 + i * T) * 16), k_voff[i], j * k_tile_stride);
  };
  // V(j) -> V buffer `buf`
  auto load_v = [&](int j, const int buf) __attribute__((always_inline)) {
    if constexpr (abl::kSameTile) j = 0;
#pragma unroll
    for (int i = 0; i < VC; ++i) {
      if (!(VC * T == VROWS * VCH || wave * 64 + i * T < VROWS * VCH)) continue;
      lds_dma16(v_rsrc_dma, (unsigned)(RING * KBYTES + buf * VBYTES + (wave * 64 + i * T) * 16), v_voff[i], j * v_tile_stride);
    }
  };
  // ---- lane-constant LDS read offsets
  // (pointers that already include the K / V region base: the fast loops and the generic body then share ONE register per
  // offset; with integer offsets hipcc kept `base + offset` and `offset` as two live values)
  const char* k_rd[KS];  // K A-fragment: row r (+32*mt via immediate), chunk 2*ks+hh
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = k_lds + (r * D + (((2 * ks + hh) ^ k_swz<D>(r)) << 4));
  const char* v_rd8[2];  // fp8: V^T row r (+32*dt immediate), 16-B chunks 2*hh and 2*hh+1
#pragma unroll
  for (int c = 0; c < 2; ++c) v_rd8[c] = v_lds + (r * 64 + (((2 * hh + c) ^ ((r >> 2) & 3)) << 4));
  const char* v_rd[DT];  // fp16: V^T fragment via tr-read: row 4*hh + q4 (+32*mt+16*s(+8) immediate), window dt
  {
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g = (lane >> 4) & 1;
    const int rv = 4 * hh + q4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) v_rd[dt] = v_lds + (rv * (2 * D) + ((dt ^ v_win_swz<D>(rv)) << 6) + 32 * g + 8 * p4);
  }

  // ---- state
  v16f acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_o[dt][e] = 0.f;
  float m_run = -1e30f;
  // per-lane partial row sum of the UNROUNDED p in fp32 on the VALU (the lane's 32 of the row's 64 keys per tile;
  // the two lane halves are added once in the epilogue), as the reference's Triton kernels and its fp8 CUDA kernel
  // (attn_qk_int8_per_block.py:55-60; ComputeUnit::kCudaCore, sm89_*.cu:148).  A ones-row MFMA that sums the
  // rounded P (the reference's fp16 CUDA trick, attn_utils.cuh:543-547) was measured slower in every configuration
  // (head_dim 128 fp16 -3 %, fp8 -0.7 %; head_dim 64 fp16 -9 %, fp8 -7 %): the chip is power limited and an extra MFMA
  // per P operand costs more clock than the 32 v_add_f32 it replaces.
  float l_run = 0.f;
  // FP16 PV at head_dim 64: the row sum runs on the matrix pipe instead -- one v_mfma_f32_16x16x32_f16 per 16-key quarter.
  // B = the quarter's 8 packed fp16 p of the lane (MFMA column l%16, k group l/16: lanes l and l+32 are the two halves of
  // query l%16, lanes l+16 and l+48 those of query l%16+16); A[i][k] = 1 iff (k/8) % 2 == i % 2, so the even result rows hold
  // the full row sum of query l%16 and the odd ones that of query l%16+16: elements 0 / 1 of every lane's fp32 accumulator
  // (C is elementwise, so the running sums stay private to the lane and are rescaled by its own alpha).  4 MFMAs of 16 cycles
  // per tile replace 32 v_add_f32 (128 cycles of the vector issue port) and the final lane-half exchange.  It sums the
  // ROUNDED P -- exactly what the reference's fp16 CUDA kernel does (ComputeUnit::kTensorCore: mma::rowsum_f16f16f32 on the
  // packed half P, attn_utils.cuh:528-548, qk_int_sv_f16_cuda_sm80.cu:318-320,814), where its Triton twin sums the fp32 p.
  // At head_dim 64 the loop is bound by vector issue and the matrix pipe is a third busy: steady state C2 +2.7 %, C2-causal
  // +3.1 %, (4,32,8192,64) +3.0 % against the VALU sums (a first form on v_mfma_f32_4x4x4_16b_f16, 8 per tile, gave +1.5 /
  // +2.3 / +1.4 %), and the unrounded p are dead after the convert (153-156 registers instead of 162-168).  At head_dim 128
  // every gap already holds a P.V MFMA: 4x4x4 -1.5 %, this form +0.2 ... +0.8 % -- not worth giving up the exact fp32 sums there.
  // (bf16 PV keeps the VALU sums of the unrounded p: a sum of bf16-rounded P would cost the LSE three more bits)
  constexpr bool MROW = !PV_FP8 && !V_BF16 && (D == 64 ? !abl::kValuRowSum64 : abl::kMfmaRowSum128);
  v4f l4 = {0.f, 0.f, 0.f, 0.f};
  v8h sel8;
  {
    const _Float16 one = (((lane >> 4) & 1) == (lane & 1)) ? (_Float16)1.0f : (_Float16)0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sel8[e] = one;
    if constexpr (MROW) asm volatile("" : "+v"(sel8));  // resident: as a known value it is re-materialised per use
  }
  auto rowsum8 = [&](const v8h ph) __attribute__((always_inline)) {
    l4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel8, ph, l4, 0, 0, 0);
  };
  // The int32 accumulator of S^T starts at the BIT PATTERN of 1.5*2^23: for |S| < 2^22 (|S| <= 128*127^2) the
  // accumulated integer, reinterpreted as fp32, IS the float 12582912 + S exactly, so the logit needs no
  // v_cvt_f32_i32: t - m = fma(as_float(acc), scale, -(12582912*scale + m)).
  constexpr int kBiasI = 0x4B400000;
  constexpr float kBiasF = 12582912.0f;
  // A masked score is the bit pattern of -inf: as an INTEGER it is below every real score (they are ~0x4B400000), so the
  // integer row max ignores it; as a FLOAT it makes fma(-inf, scale, c) = -inf for any positive scale and exp2 returns
  // exactly 0 -- no per-element zeroing of p, no mask bits to carry from the S tile to the P tile.
  constexpr int kMaskedI = (int)0xFF800000u;
  v16i bias;
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = kBiasI;
  // keep the 16 bias registers resident: as a known constant the compiler re-materialises them with 8 v_mov_b64 per
  // tile (or, in the in-place form below, 16 moves per S chain), and the kernel is bound by the vector issue port (every
  // VALU instruction costs 4 cycles of it).  Not in the attn_mask instantiation, which has no registers to spare.
  // History: until the end of round 2 the head_dim-64 FP8-PV and causal variants gave the pin up to stay within the 168
  // registers of three waves per SIMD (worth more than the moves: fp8 +5 %, causal +7 %).  Since the register diet, the MFMA
  // row sums and the bf16-native P.V every dispatched head_dim-64 variant fits WITH the pin (153-168 registers, no scratch;
  // the build fails otherwise): C2-causal +3.7 % (fp16), +3.1 % (bf16), +4.9 % (FP8 PV); C2-fp8 +3.9 %, (4,32,8192,64)-fp8
  // +4.0 %; bit-identical.
  constexpr bool BIAS_RESIDENT = !HAS_MASK;
  if constexpr (BIAS_RESIDENT) asm volatile("" : "+v"(bias));
  // First k-step of an S^T chain: acc = bias + K.Q^T.  C is either the resident bias tuple or the MFMA's own destination
  // registers initialised in place (C = D), so the chain never needs a second 16-register tuple.
  // History: round 1 blamed nondeterministic rows at head_dim 64 / causal on hipcc re-using a temporary C tuple too early
  // and introduced this form as the fix.  That diagnosis was wrong: the cause was the missing barrier between the
  // prologue S(0) and the first K(2) copy (below; profiles/r02_race_evidence.md).  With the barrier in place the old
  // form (-DSAGE_EXP_CTEMP) is bit-stable too (0 of 100 + 0 of 1000 stressed launches); this form stays because it is
  // what every test and profile of the kernel ran on.
  auto mfma_s_first = [&](const v4i a, const v4i b) __attribute__((always_inline)) -> v16i {
    if constexpr (BIAS_RESIDENT || abl::kCTemp) {
      return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, bias, 0, 0, 0);
    } else {
      v16i acc = bias;
      asm volatile("" : "+v"(acc));  // the 16 moves land in the accumulator itself
      return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    }
  };

  // S^T = K . Q^T for one tile (2 x 32 keys x 32 query rows) out of LDS buffer `kbuf`
  auto qk = [&](const int kbuf, v16i (&s)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if constexpr (abl::kNoQK) {
          s[mt] = bias; s[mt][0] += kbuf + ks;
        } else {
          const v4i a = *reinterpret_cast<const v4i*>(k_rd[ks] + (kbuf * KBYTES + mt * 32 * D));
          s[mt] = ks == 0 ? mfma_s_first(a, qf[ks]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], s[mt], 0, 0, 0);
        }
      }
  };
  // dequantisation scales of tile j: …sm80.cu:131, 4 per 64 keys, index (c%8)/2 = 2*hh + ((reg&3)>>1)
  // wave-uniform address: the scales of a tile come through the scalar cache (s_load_dwordx4, lgkmcnt), not through
  // vmcnt where they would queue behind the tile DMA.  The lane-half select is an fma with a zeroed partner
  // (x*q + z*0 is exactly x*q) instead of two v_mov + v_cndmask per scale: SGPR operands feed the VALU directly.
  float qsc_lo = 0.f, qsc_hi = 0.f;  // = hh ? (0, qsc) : (qsc, 0), set once the Q scale is known (prologue)
  auto load_kscales = [&](const int j) __attribute__((always_inline)) -> float4 {
    if constexpr (KTHREAD) return uniform_load4(ksp + j * p.ks_t);
    else return make_float4(uniform_load1(ksp + j * p.ks_t), 0.f, 0.f, 0.f);
  };
  auto scales_from = [&](const float4 kk, float& sc0, float& sc1) __attribute__((always_inline)) {
    if constexpr (KTHREAD) {
      sc0 = __builtin_fmaf(kk.z, qsc_hi, kk.x * qsc_lo);
      sc1 = __builtin_fmaf(kk.w, qsc_hi, kk.y * qsc_lo);
    } else {
      sc0 = sc1 = qsc * kk.x;
    }
  };
  auto tile_scales = [&](const int j, float& sc0, float& sc1) __attribute__((always_inline)) {
    scales_from(load_kscales(j), sc0, sc1);
  };
  // which of the lane's 32 keys of tile j may be attended: bit 16*mt+e.  Sequence end, causal diagonal and the
  // caller's bool attn_mask (False = masked; the reference adds -1e6, which is the same for every row that keeps
  // at least one key; rows with no allowed key at all are undefined there -- they depend on its tile skipping).
  // attn_mask exists only on the non-causal 16-bit-PV operator (fp16 or bf16 V; the reference converts a bf16 V to fp16
  // for masked calls as well, core.py:289-290)
  constexpr bool CAN_MASK = HAS_MASK;  // separate instantiation: the mask bookkeeping must not cost the main variants registers
  const uint8_t* mrow = nullptr;
  if constexpr (CAN_MASK)
    if (p.mask) mrow = p.mask + ((int64_t)b * p.msb + (int64_t)h * p.msh + (int64_t)rowc * p.msm) * (p.mask_kind == 1 ? 1 : 2);
  struct __attribute__((packed)) u32_unaligned { uint32_t v; };
  struct __attribute__((packed)) u64_unaligned { uint64_t v; };
  auto allow_bits = [&](const int j) __attribute__((always_inline)) -> uint32_t {
    const int n0 = j << 6;
    uint32_t bits = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int kv0 = n0 + 32 * mt + 8 * g4 + 4 * hh;  // the lane's keys come in runs of 4: registers 4*g4 .. 4*g4+3
        uint32_t run = 0;                                 // bit i: key kv0+i allowed
#pragma unroll
        for (int i = 0; i < 4; ++i) run |= (((kv0 + i) < N_ && !(CAUSAL && (kv0 + i) > row)) ? 1u : 0u) << i;
        if constexpr (CAN_MASK) {
          if (p.mask_kind == 1 && run) {
            uint32_t mb = 0;
            if (p.msn == 1 && kv0 + 3 < N_) {  // 4 mask bytes in one (unaligned) dword
              const uint32_t w = reinterpret_cast<const u32_unaligned*>(mrow + kv0)->v;
#pragma unroll
              for (int i = 0; i < 4; ++i) mb |= (((w >> (8 * i)) & 0xffu) ? 1u : 0u) << i;
            } else {
#pragma unroll 1
              for (int i = 0; i < 4; ++i)
                if (kv0 + i < N_) mb |= (mrow[(int64_t)(kv0 + i) * p.msn] ? 1u : 0u) << i;
            }
            run &= mb;
          }
        }
        bits |= run << (16 * mt + 4 * g4);
      }
    return bits;
  };
  // sequence end and causal diagonal (the kernels without attn_mask): register e of block mt holds key
  // 64*j + 32*mt + (e&3) + 8*(e>>2) + 4*hh, allowed iff <= min(N-1, row): one compare against a per-lane limit
  auto mask_limit = [&](const int j, v16i (&s)[2]) __attribute__((always_inline)) {
    const int lim = min(N_ - 1, CAUSAL ? row_l : 0x7fffffff) - (j << 6) - 4 * hh_l;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[mt][e] = (32 * mt + (e & 3) + 8 * (e >> 2) <= lim) ? s[mt][e] : kMaskedI;
  };
  auto mask_scores = [&](const uint32_t bits, v16i (&s)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[mt][e] = ((bits >> (16 * mt + e)) & 1u) ? s[mt][e] : kMaskedI;
  };
  // additive attn_mask: the score registers are rewritten in place with the fp32 base-2 logits
  // t = S*scale + mask (masked / out-of-range keys: -1e6 as in the reference, attn_qk_int8_per_block.py:40-43)
  const bool fmask = CAN_MASK && p.mask_kind >= 2;
  auto to_float_logits = [&](const int j, const uint32_t bits, v16i (&s)[2], const float sc0, const float sc1)
      __attribute__((always_inline)) {
    const int n0 = j << 6;
    const uint16_t* mrow16 = reinterpret_cast<const uint16_t*>(mrow);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int kv0 = n0 + 32 * mt + 8 * g4 + 4 * hh;
        uint16_t raw[4] = {0, 0, 0, 0};
        if (p.msn == 1 && kv0 + 3 < N_) {  // 4 mask values in one (unaligned) 8-byte load
          const uint64_t w = reinterpret_cast<const u64_unaligned*>(mrow16 + kv0)->v;
#pragma unroll
          for (int i = 0; i < 4; ++i) raw[i] = (uint16_t)(w >> (16 * i));
        } else {
#pragma unroll 1
          for (int i = 0; i < 4; ++i)
            if (kv0 + i < N_) raw[i] = mrow16[(int64_t)(kv0 + i) * p.msn];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * g4 + i;
          const float sc = (e & 2) ? sc1 : sc0;
          const float t = __builtin_fmaf(__int_as_float(s[mt][e]), sc, -kBiasF * sc);
          float mv = -1.0e6f;
          if ((bits >> (16 * mt + e)) & 1u) mv = p.mask_kind == 3 ? bf16_bits_to_f32(raw[i]) : f16_bits_to_f32(raw[i]);
          s[mt][e] = __float_as_int(t + mv);
        }
      }
  };
  auto row_max_f = [&](const v16i (&s)[2]) __attribute__((always_inline)) -> float {
    float mx = __int_as_float(s[0][0]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, __int_as_float(s[mt][e]));
    return swap_max(mx);
  };
  // row max of the logits of one tile, from the raw integers (scales are positive): v_max3_i32 + 1 cvt/group
  auto row_max = [&](const v16i (&s)[2], const float sc0, const float sc1) __attribute__((always_inline)) -> float {
    int mxa = s[0][0], mxb = s[0][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e & 2) mxb = max(mxb, s[mt][e]); else mxa = max(mxa, s[mt][e]);
      }
    // the biased integers are the floats 12582912 + S: one v_sub_f32 recovers S exactly (no v_cvt_f32_i32)
    float mx;
    if constexpr (KTHREAD) mx = max_raw((__int_as_float(mxa) - kBiasF) * sc0, (__int_as_float(mxb) - kBiasF) * sc1);
    else mx = (__int_as_float(max(mxa, mxb)) - kBiasF) * sc0;
    return swap_max(mx);
  };
  // lazy rescale (attn_utils.cuh:354-458 rescales every tile; here only when some row's max grew by more than
  // kLazyThr, so p <= 2^kLazyThr -- harmless in fp16/fp32; m_run stays exact for the LSE)
  // fp8: p carries the reference's exponent offset (attn_utils.cuh:30,379: m tracked as t - 8.807 so p <= 448 = e4m3
  // max); the lazy threshold is taken out of that headroom so p <= 2^(kLazyThr + kPOff) = 448 still holds.
  constexpr float kLazyThr = PV_FP8 ? 3.0f : 6.0f;
  constexpr float kPOff = PV_FP8 ? 8.807f - 3.0f : 0.f;
  float m_thr = m_run + kLazyThr;  // kept in a register: the comparison runs every tile, the update almost never
  auto maybe_rescale = [&](const float mx) __attribute__((always_inline)) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > m_thr) != 0, 0)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      m_thr = m_new + kLazyThr;
      l_run *= alpha;
      if constexpr (MROW) l4 *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[dt][e] *= alpha;
    }
  };
  // P pair -> two packed 16-bit values in the element type of V (RNE both), and the P.V MFMA of that type
  auto pack_p = [&](const v2f two) __attribute__((always_inline)) -> v2h {
    if constexpr (V_BF16) return __builtin_bit_cast(v2h, __builtin_convertvector(two, v2bf));  // v_cvt_pk_bf16_f32
    else return __builtin_convertvector(two, v2h);                                              // v_cvt_pk_f16_f32 (fp16_rn)
  };
  auto pv_mfma = [&](const v8h a, const v8h b, const v16f c) __attribute__((always_inline)) -> v16f {
    if constexpr (V_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  };
  // p = exp2(t - m) and O^T += V^T . P^T
  uint32_t bits_cur = 0xffffffffu;  // allow mask of tile j (attn_mask loop only)
  auto softmax_pv = [&](const int j, const int vbuf, const v16i (&s)[2], const float sc0, const float sc1,
                        auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const float c0 = __builtin_fmaf(-kBiasF, sc0, kPOff - m_run), c1 = __builtin_fmaf(-kBiasF, sc1, kPOff - m_run);
    auto prob = [&](const int mt, const int e) __attribute__((always_inline)) -> float {
      const bool g1 = (e & 2) != 0;
      float pv;
      if (MASKED && fmask) {
        pv = __builtin_amdgcn_exp2f(__int_as_float(s[mt][e]) - m_run + kPOff);  // registers hold fp32 logits
      } else {
        pv = __builtin_fmaf(__int_as_float(s[mt][e]), g1 ? sc1 : sc0, g1 ? c1 : c0);
        if constexpr (!abl::kNoExp) pv = __builtin_amdgcn_exp2f(pv);
      }
      return pv;
    };
    if constexpr (!PV_FP8) {
      // quarters of 16 keys, each followed by its PV MFMAs
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
          v8h pf;
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            v2f pp = {prob(mt, 8 * sq + e), prob(mt, 8 * sq + e + 1)};
            const v2h ph = pack_p(pp);
            pf[e] = ph[0];
            pf[e + 1] = ph[1];
            if constexpr (!MROW) {
              l_run += pp[0];
              l_run += pp[1];
            }
          }
          if constexpr (MROW) rowsum8(pf);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const char* base = v_rd[dt] + (vbuf * VBYTES + (32 * mt + 16 * sq) * (2 * D));
            const v4s_vs lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base));
            const v4s_vs hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4s_vs*)(base + 8 * (2 * D)));
            v8h a;
            a.s0123 = __builtin_bit_cast(v4h, lo);
            a.s4567 = __builtin_bit_cast(v4h, hi);
            if constexpr (abl::kNoPV) asm volatile("" ::"v"(a), "v"(pf));
            else acc_o[dt] = pv_mfma(a, pf, acc_o[dt]);
          }
        }
    } else {
      // all 64 keys of the tile in one K=64 MFMA per d tile; P^T bytes in accumulator order j = 16*mt + reg
      v8i pb;
      float psum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const int mt = w >> 2, e0 = 4 * (w & 3);
        const float p0 = prob(mt, e0), p1 = prob(mt, e0 + 1), p2 = prob(mt, e0 + 2), p3 = prob(mt, e0 + 3);
        psum += p0; psum += p1; psum += p2; psum += p3;
        int pk = 0;
        if constexpr (D == 128) asm volatile("" : "=v"(pk));  // (no v_mov for the `old` operand: see the hand-placed stream)
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, pk, false);  // OCP e4m3, RNE (e4m3_rn_satfinite)
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, pk, true);
        pb[w] = pk;
      }
      l_run += psum;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int base = vbuf * VBYTES + dt * 32 * 64;
        const v4i lo = *reinterpret_cast<const v4i*>(v_rd8[0] + base);
        const v4i hi = *reinterpret_cast<const v4i*>(v_rd8[1] + base);
        v8i a;
        a.s0123 = lo;
        a.s4567 = hi;
        // cbsz = blgp = 0: both operands e4m3; E8M0 block scales 127 = 2^0
        acc_o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, pb, acc_o[dt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  };

  // ---- software pipeline.  LDS: K(j) in K buffer j&1, V(j) in V buffer j&1.  During iteration j the wave
  //      computes S(j+1) = K(j+1).Q^T (MFMA) while it exponentiates S(j) (VALU) and accumulates P(j).V(j);
  //      K(j+2) and V(j+1) are copied global -> LDS during the iteration (LDS-DMA, drained in front of the barrier).
  //   [0, n_fast)           tiles j and j+1 both unmasked: branch-free body
  //   [n_fast, wave_tiles)  the wave's last tiles (a successor that may need masking; the final tile): run-time ring slots
  //   [wave_tiles, ntiles)  causal only: this wave is done but still stages tiles for its workgroup
  // Every wave executes the same number of barriers.
  int n_plain = wave_tiles;
  if (N_ & 63) n_plain = min(n_plain, N_ >> 6);
  if constexpr (CAUSAL) n_plain = min(n_plain, max(0, (q0 + 1) >> 6));  // tile j needs no mask iff 64*j+63 <= q0
  const int n_fast = (abl::kAllGeneric || (CAN_MASK && p.mask)) ? 0 : max(0, min(n_plain - 1, wave_tiles - 1));  // attn_mask: all tiles generic

  if constexpr (abl::kPrio >= 0) {  // static priority for the second-dispatched half of the workgroup: measured 0 %
    if (wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(abl::kPrio >= 0 ? abl::kPrio : 0);
  }
  // tile copies a wave issues per iteration (full tiles): the unit of the counted waits of the four-slot ring
  constexpr int NDMA = KC + VC;
  static_assert(RING == 2 || (KC * T == 64 * KCH && VC * T == VROWS * VCH), "four-slot ring: every wave copies full shares");
  const int last_tile = ntiles - 1;
  if constexpr (RING == 2) {
    dma_k(0, 0);
    load_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    prepare_q();
    dma_wait_all();
  } else {
    // K(0..3) and V(0..2), clamped to the last tile so that every wave issues the same number of copies whatever the
    // sequence length (a clamped copy re-loads the last tile into a slot nobody reads any more); only K(0), V(0), K(1)
    // are needed now, the rest keeps flying
    dma_k(0, 0);
    load_v(0, 0);
    dma_k(min(1, last_tile), 1);
    load_v(min(1, last_tile), 1);
    dma_k(min(2, last_tile), 2);
    load_v(min(2, last_tile), 2);
    dma_k(min(3, last_tile), 3);
    prepare_q();
    dma_wait_keep<2 * NDMA>();
  }
  qsc_lo = hh ? 0.f : qsc;
  qsc_hi = hh ? qsc : 0.f;
  __syncthreads();

  if constexpr (HAS_MASK) {
    // attn_mask instantiation: plain serial loop (K and V one tile ahead), every tile through the masked body.
    // A tile in which this wave's 32 rows keep no key is skipped (the reference skips all-False 128x64 tiles,
    // attn_qk_int8_per_block.py:38-39).
    v16i s_cur[2];
    float sc0, sc1;
    for (int j = 0; j < ntiles; ++j) {
      const int buf = j & 1;
      if (j + 1 < ntiles) {
        if (j > 0) dma_k(j + 1, buf ^ 1);  // K(1) was copied by the prologue
        load_v(j + 1, buf ^ 1);
      }
      bits_cur = allow_bits(j);
      if (__builtin_amdgcn_ballot_w64(bits_cur != 0) != 0) {
        tile_scales(j, sc0, sc1);
        qk(buf, s_cur);
        float mx;
        if (fmask) { to_float_logits(j, bits_cur, s_cur, sc0, sc1); mx = row_max_f(s_cur); }
        else { mask_scores(bits_cur, s_cur); mx = row_max(s_cur, sc0, sc1); }
        maybe_rescale(mx);
        softmax_pv(j, buf, s_cur, sc0, sc1, std::true_type{});
      }
      dma_wait_all();
      __syncthreads();
    }
  } else {
  v16i s_cur[2], s_nxt[2];
  float sc0, sc1, mx_cur;
  if constexpr (abl::kDelayWave) {  // mechanism probe: force the interleaving that the missing barrier below allowed
    if (wave == 1) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
  }
  qk(0, s_cur);
  // Every wave reads ALL 64 rows of K buffer 0 for S(0) above, and iteration 0 below re-fills that buffer with K(2)
  // (each wave DMA-writes its own 1 KiB slice).  Later iterations are ordered by the barrier that closes the previous
  // one; this first re-fill needs its own: without it a wave that is held back between the prologue barrier and its
  // K(0) fragment reads (three waves per SIMD: the youngest wave can starve for longer than an L2 round trip) computes
  // S(0) from a mix of K(0) and K(2) rows -- one wrong 32-row wave, the same wrong value every time.
  if constexpr (!abl::kNoPrologueBarrier) __syncthreads();
  tile_scales(0, sc0, sc1);
  // a plain tile 0 needs no mask (64 compare + select instructions per wave); the causal head_dim-64 variants keep it
  // unconditional -- under the branch they need 170 registers, two over the three-waves-per-SIMD line
  if (CAUSAL || n_plain <= 0) mask_limit(0, s_cur);
  mx_cur = row_max(s_cur, sc0, sc1);

  // fast loop, unrolled by two so that S(j) / S(j+1) swap roles without register copies
  float4 kk_nxt = load_kscales(min(1, ntiles - 1));
  int k_slot = 0, v_slot = 0;  // slots the LDS read pointers point at (moved by the run-time-slot tiles only)
  // NEXT (second tag): what follows tile j for this wave.
  //   0  a plain tile: the fast loops.  Ring slots are compile-time (R = j % RING), every LDS offset an immediate.
  //   1  a tile that may need masking (sequence end, causal diagonal), 2  nothing (the wave's last tile): the remaining
  //      tiles of a wave run through the SAME hand-placed stream with run-time ring slots (a few address adds) -- the
  //      compiler-scheduled body they used before was 1.6x slower per tile, 5-7 % of a short causal sequence.
  auto fast_iter = [&](auto par_tag, auto next_tag, const int j, v16i (&sa)[2], v16i (&sb)[2], float& a0, float& a1, float& b0,
                       float& b1) __attribute__((always_inline)) {
    constexpr int R = decltype(par_tag)::value;  // j % RING, static so every LDS offset is an immediate
    constexpr int NEXT = decltype(next_tag)::value;
    constexpr bool DYN = NEXT != 0;
    // slots of K(j+1), V(j), K(j+RING), V(j+RING-1)
    const int K_RD = DYN ? (j + 1) % RING : (R + 1) % RING, V_RD = DYN ? j % RING : R, K_WR = V_RD,
              V_WR = DYN ? (j + RING - 1) % RING : (R + RING - 1) % RING;
    // The first K fragment of S(j+1) is read BEFORE the tile copies are issued: the first S MFMA needs it at once, and the
    // copies are inline asm with a memory clobber, so the compiler cannot hoist the read across them itself (+0.2..0.8 %).
    // scales of tile j+1 were fetched during the previous iteration; those of tile j+2 are fetched first thing here: the
    // scalar load shares lgkmcnt with the LDS reads, so it must be in flight long before the first wait on a K fragment
    // (left to hipcc it is issued right in front of that wait and every iteration pays a scalar-cache round trip)
    if constexpr (DYN) {
      // run-time slots without extra address registers: the read pointers themselves move to the slots of this tile (the
      // fast loops, which need them at the region base, are over) and every offset below stays an immediate
      if constexpr (NEXT != 2) {
        const int dk = (K_RD - k_slot) * KBYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] += dk;
        k_slot = K_RD;
      }
      const int dv = (V_RD - v_slot) * VBYTES;
      if constexpr (PV_FP8) { v_rd8[0] += dv; v_rd8[1] += dv; }
      else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) v_rd[dt] += dv;
      }
      v_slot = V_RD;
    }
    v4i kf_early = qf[0];
    if constexpr (NEXT != 2) {
      scales_from(kk_nxt, b0, b1);
      kk_nxt = load_kscales(min(j + 2, ntiles - 1));
      if constexpr (!abl::kNoLdsK) kf_early = *reinterpret_cast<const v4i*>(k_rd[0] + (DYN ? 0 : K_RD * KBYTES));
    }
    __builtin_amdgcn_sched_barrier(0);
    maybe_rescale(mx_cur);
    if constexpr (!abl::kNoStage) {
      if constexpr (RING == 2) {
        if (j + 2 < ntiles && !(abl::kHalfCopies && (j & 1))) dma_k(j + 2, K_WR);
        if ((!DYN || j + 1 < ntiles) && !(abl::kHalfCopies && (j & 1))) load_v(j + 1, V_WR);
      } else {
        dma_k(min(j + RING, last_tile), K_WR);
        load_v(min(j + RING - 1, last_tile), V_WR);
      }
    }
    constexpr int HAND_PLACED = PV_FP8 ? abl::kHandPlacedF8 : abl::kHandPlacedF16;
    if constexpr (HAND_PLACED == 0) {
      if constexpr (NEXT != 2) qk(DYN ? 0 : K_RD, sb);
      if constexpr (NEXT == 1) { if (j + 1 >= n_plain) mask_limit(j + 1, sb); }
      softmax_pv(j, DYN ? 0 : V_RD, sa, a0, a1, std::false_type{});
      if constexpr (NEXT != 2) mx_cur = row_max(sb, b0, b1);
    } else if constexpr (HAND_PLACED == 2) {
      // Hand-placed stream, FP8 PV.  The K = 64 MFMA consumes the P of the whole tile, so all of P(j) precedes the P.V
      // MFMAs; left alone hipcc emits ~110 softmax VALU instructions with the matrix pipe idle and then the 12 MFMAs in
      // one cluster.  Here: the S(j+1) MFMAs are spread through the computation of the eight P words (4 keys each:
      // 4 fma, 4 exp2, 2 cvt_pk_fp8; the row-sum adds trail by one word), the V^T fragments are read while the last
      // words are computed, and the four P.V MFMAs run beside the row max of S(j+1).
      constexpr int NS = 2 * KS;       // S MFMAs per tile
      constexpr int WPS = 8 / NS;      // P words per S MFMA (1 at head_dim 128, 2 at 64)
      const int kb = DYN ? 0 : K_RD * KBYTES, vb = DYN ? 0 : V_RD * VBYTES;  // slot offsets (DYN: the pointers were moved)
      const float c0 = __builtin_fmaf(-kBiasF, a0, kPOff - m_run), c1 = __builtin_fmaf(-kBiasF, a1, kPOff - m_run);
      auto k_frag = [&](const int i) __attribute__((always_inline)) -> v4i {
        return *reinterpret_cast<const v4i*>(k_rd[i % KS] + (kb + (i / KS) * 32 * D));
      };
      auto s_step = [&](const int i, const v4i a) __attribute__((always_inline)) {
        const int mt = i / KS, ks = i % KS;
        sb[mt] = ks == 0 ? mfma_s_first(a, qf[ks]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], sb[mt], 0, 0, 0);
      };
      auto v_frag8 = [&](const int dt) __attribute__((always_inline)) -> v8i {
        const int base = vb + dt * 32 * 64;
        v8i a;
        a.s0123 = *reinterpret_cast<const v4i*>(v_rd8[0] + base);
        a.s4567 = *reinterpret_cast<const v4i*>(v_rd8[1] + base);
        return a;
      };
      v8i pb;
      float pend[4], psum = 0.f;
      auto p_word = [&](const int w) __attribute__((always_inline)) {
        const int mt = w >> 2, e0 = 4 * (w & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool g1 = ((e0 + i) & 2) != 0;
          pend[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[mt][e0 + i]), g1 ? a1 : a0, g1 ? c1 : c0));
        }
        // (the first convert's `old` operand is a register nobody has written: both halves of the dword are produced by the two
        //  converts, and a literal 0 there costs a v_mov_b32 per P word -- 8 of ~166 vector instructions per tile)
        int pk = 0;
        if constexpr (D == 128) asm volatile("" : "=v"(pk));  // (head_dim 64: the variants at the 168-register line spill with it)
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(pend[0], pend[1], pk, false);  // OCP e4m3, RNE
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(pend[2], pend[3], pk, true);
        pb[w] = pk;
      };
      auto p_sum = [&]() __attribute__((always_inline)) {
        psum += pend[0];
        if constexpr (!abl::kNoRowSumF8) { psum += pend[1]; psum += pend[2]; psum += pend[3]; }
      };
#define SAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
      v4i kf = kf_early;
      v8i vf[DT];
      int si = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        if (NEXT != 2 && w % WPS == 0) {
          s_step(si, kf);
          if (si + 1 < NS) kf = k_frag(si + 1);
          ++si;
        }
        if (w > 0) p_sum();
        p_word(w);
        if (w == 6) vf[0] = v_frag8(0);
        if (w == 7 && DT > 1) vf[1] = v_frag8(1);
        SAGE_FENCE();
      }
      p_sum();
      l_run += psum;
      // P.V beside the row max of S(j+1)
      if constexpr (NEXT == 1) { if (j + 1 >= n_plain) mask_limit(j + 1, sb); }
      int mxa = sb[0][0], mxb = sb[0][2];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (dt + 2 < DT) vf[dt + 2] = v_frag8(dt + 2);
        acc_o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[dt], pb, acc_o[dt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if constexpr (NEXT != 2) {
#pragma unroll
          for (int idx = dt * (32 / DT); idx < (dt + 1) * (32 / DT); ++idx) {
            const int mt = idx >> 4, e = idx & 15;
            if (e & 2) mxb = max(mxb, sb[mt][e]); else mxa = max(mxa, sb[mt][e]);
          }
          asm volatile("" : "+v"(mxa), "+v"(mxb));
        }
        SAGE_FENCE();
      }
#undef SAGE_FENCE
      if constexpr (NEXT != 2) {
        float mx;
        if constexpr (KTHREAD) mx = max_raw((__int_as_float(mxa) - kBiasF) * b0, (__int_as_float(mxb) - kBiasF) * b1);
        else mx = (__int_as_float(max(mxa, mxb)) - kBiasF) * b0;
        mx_cur = swap_max(mx);
      }
    } else {
      // Hand-placed instruction stream (fp16 PV).  The wave issues in order and an MFMA that finds the matrix pipe
      // busy blocks the VALU instructions behind it, so what counts is what sits BETWEEN consecutive MFMAs: about
      // 24 cycles of vector issue hide beside a 32-cycle MFMA (tools/issue_cost.hip).  Left to itself hipcc emits the
      // S(j+1) MFMAs as one burst, P.V MFMAs with nothing but LDS reads between them, and the row sums / row max as a
      // VALU-only tail.  Here the tile is cut into quarters of 16 keys: P of quarter q+1 is computed beside the P.V
      // MFMAs of quarter q, the S(j+1) MFMAs are spread between them, V^T fragments are read one quarter ahead, and
      // the row max of S(j+1) runs beside the last quarter's MFMAs.  sched_barrier(0) pins each group.
      constexpr int NS = 2 * KS, SPR = NS / 4;  // S MFMAs per tile / per region
      constexpr int PPG = 4 / DT;               // P pairs computed beside one P.V MFMA
      const int kb = DYN ? 0 : K_RD * KBYTES, vb = DYN ? 0 : V_RD * VBYTES;  // slot offsets (DYN: the pointers were moved)
      const float c0 = __builtin_fmaf(-kBiasF, a0, kPOff - m_run), c1 = __builtin_fmaf(-kBiasF, a1, kPOff - m_run);
      auto k_frag = [&](const int i) __attribute__((always_inline)) -> v4i {
        if constexpr (abl::kNoLdsK) return qf[i % KS];
        else if constexpr (abl::kHalfReads) { if (i & 1) return qf[i % KS]; else return *reinterpret_cast<const v4i*>(k_rd[i % KS] + (kb + (i / KS) * 32 * D)); }
        else if constexpr (abl::kConstOdd) {
          const v4i f = *reinterpret_cast<const v4i*>(k_rd[i % KS] + (kb + (i / KS) * 32 * D));
          if (i & 1) { asm volatile("" :: "v"(f)); return qf[i % KS]; }
          return f;
        }
        else return *reinterpret_cast<const v4i*>(k_rd[i % KS] + (kb + (i / KS) * 32 * D));
      };
      auto v_frag = [&](const int q, const int dt) __attribute__((always_inline)) -> v8h {
        if constexpr (abl::kNoLdsV) return __builtin_bit_cast(v8h, qf[(q + dt) % KS]);
        if constexpr (abl::kHalfReads) { if ((q + dt) & 1) return __builtin_bit_cast(v8h, qf[(q + dt) % KS]); }
        const char* base = v_rd[dt] + (vb + 16 * q * (2 * D));
        const v4s_vs lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base));
        const v4s_vs hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base + 8 * (2 * D)));
        v8h a;
        a.s0123 = __builtin_bit_cast(v4h, lo);
        a.s4567 = __builtin_bit_cast(v4h, hi);
        if constexpr (abl::kConstOdd) { if ((q + dt) & 1) { asm volatile("" :: "v"(a)); return __builtin_bit_cast(v8h, qf[(q + dt) % KS]); } }
        return a;
      };
      auto s_step = [&](const int i, const v4i a) __attribute__((always_inline)) {
        const int mt = i / KS, ks = i % KS;
        sb[mt] = ks == 0 ? mfma_s_first(a, qf[ks]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], sb[mt], 0, 0, 0);
      };
      float pp[8];  // unrounded p of the quarter in flight (row-sum operands)
      auto p_pair = [&](const int q, const int pr, v8h& pf) __attribute__((always_inline)) {
        const int mt = q >> 1, e = 8 * (q & 1) + 2 * pr;
        const bool g1 = (e & 2) != 0;
        const float sc = g1 ? a1 : a0, cc = g1 ? c1 : c0;
        v2f two = {__builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[mt][e]), sc, cc)),
                   __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[mt][e + 1]), sc, cc))};
        const v2h ph = pack_p(two);
        pf[2 * pr] = ph[0];
        pf[2 * pr + 1] = ph[1];
        if constexpr (!MROW) {
          pp[2 * pr] = two[0];
          pp[2 * pr + 1] = two[1];
        }
      };
      // row sum of pair `pr` of the quarter whose packed P is `vec` (MROW: one MFMA per quarter, after its last pair)
      auto p_sum = [&](const int pr, const v8h& vec) __attribute__((always_inline)) {
        if constexpr (MROW) {
          if (pr == 3) rowsum8(vec);
        } else {
          l_run += pp[2 * pr];
          l_run += pp[2 * pr + 1];
        }
      };
#define SAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
      v4i kf = kf_early;
      // head_dim 64: K fragments are read TWO S MFMAs ahead (two registers in flight; four S MFMAs per tile, one per region,
      // so one-ahead left the read ~70 cycles).  Round 3, long windows, bit-identical: C2 +0.75 %, (4,32,8192,64) +0.8 %,
      // C2-causal +0.3 %; head_dim 128 (eight S MFMAs, two per region): -0.3 % -> not there.
      constexpr bool KPREF2 = D == 64;
      v4i kfq[2] = {kf_early, kf_early};
      if constexpr (KPREF2 && NEXT != 2) kfq[1] = k_frag(1);
      v8h vf[DT], vn[DT], pf, pn;
      // region 0: P(quarter 0) beside the first S MFMAs; V^T fragments of quarter 0
#pragma unroll
      for (int g = 0; g < SPR; ++g) {
        if constexpr (NEXT != 2) {
          if constexpr (KPREF2) {
            s_step(g, kfq[g & 1]);
            if (g + 2 < NS) kfq[g & 1] = k_frag(g + 2);
          } else {
            s_step(g, kf);
            kf = k_frag(g + 1);
          }
        }
#pragma unroll
        for (int dt = g * (DT / SPR); dt < (g + 1) * (DT / SPR); ++dt) vf[dt] = v_frag(0, dt);
#pragma unroll
        for (int pr = g * (4 / SPR); pr < (g + 1) * (4 / SPR); ++pr) {
          if (pr > 0) p_sum(pr - 1, pf);
          p_pair(0, pr, pf);
        }
        SAGE_FENCE();
      }
      // regions 1..3: P.V of quarter q-1 | P(quarter q) | S MFMAs | V^T fragments of quarter q
      int si = SPR;
#pragma unroll
      for (int q = 1; q < 4; ++q) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          acc_o[dt] = pv_mfma(vf[dt], pf, acc_o[dt]);
          vn[dt] = v_frag(q, dt);
#pragma unroll
          for (int pr = dt * PPG; pr < (dt + 1) * PPG; ++pr) {
            p_sum(pr == 0 ? 3 : pr - 1, pr == 0 ? pf : pn);  // the pair computed one step earlier (pair 3 of the previous quarter first)
            p_pair(q, pr, pn);
          }
          SAGE_FENCE();
          if (NEXT != 2 && (dt + 1) % (DT / SPR) == 0) {
            if constexpr (KPREF2) {
              s_step(si, kfq[si & 1]);
              if (si + 2 < NS) kfq[si & 1] = k_frag(si + 2);
            } else {
              s_step(si, kf);
              if (si + 1 < NS) kf = k_frag(si + 1);
            }
            ++si;
            SAGE_FENCE();
          }
        }
        pf = pn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vf[dt] = vn[dt];
      }
      // tail: P.V of quarter 3 beside the row max of S(j+1)
      p_sum(3, pf);
      if constexpr (NEXT == 1) { if (j + 1 >= n_plain) mask_limit(j + 1, sb); }
      int mxa = sb[0][0], mxb = sb[0][2];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        acc_o[dt] = pv_mfma(vf[dt], pf, acc_o[dt]);
        if constexpr (NEXT != 2) {
#pragma unroll
          for (int idx = dt * (32 / DT); idx < (dt + 1) * (32 / DT); ++idx) {
            const int mt = idx >> 4, e = idx & 15;
            if (e & 2) mxb = max(mxb, sb[mt][e]); else mxa = max(mxa, sb[mt][e]);
          }
          asm volatile("" : "+v"(mxa), "+v"(mxb));  // keeps this part of the max chain here (integer max re-associates)
        }
        SAGE_FENCE();
      }
#undef SAGE_FENCE
      if constexpr (NEXT != 2) {
        float mx;
        if constexpr (KTHREAD) mx = max_raw((__int_as_float(mxa) - kBiasF) * b0, (__int_as_float(mxb) - kBiasF) * b1);
        else mx = (__int_as_float(max(mxa, mxb)) - kBiasF) * b0;
        mx_cur = swap_max(mx);
      }
    }
    // keep the cross-lane end of the row max (a dependent chain of ~8 instructions with hazard nops) in FRONT of the
    // tile's wait and barrier, where a wave idles anyway: hipcc sank it below the barrier in one of the two unrolled
    // bodies, i.e. in front of the next tile's first MFMA (round 3, bit-identical: C3 +1.1 %, C3-causal +0.5 %, C2 +0.2 %)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!abl::kNoStage) {
      if constexpr (RING == 2) {
        dma_wait_all();  // two-slot ring: every copy of this wave has landed before the barrier publishes the tiles
      } else if constexpr (DYN) {
        dma_wait_all();  // the last tiles of a wave drain every copy (the counts of the four-slot ring stay constant)
      } else {
        dma_wait_keep<2 * NDMA>();  // K(j+2), V(j+1) and everything older have landed; the last two iterations' copies fly on
      }
    }
    if constexpr (!abl::kNoBar) __syncthreads();
  };
  float nsc0 = 0.f, nsc1 = 0.f;
  int j = 0;
  constexpr std::integral_constant<int, 0> kPlainNext{};
  if constexpr (RING == 4) {
    // four-slot ring: the slot pattern repeats every four tiles
    for (; j + 3 < n_fast; j += 4) {
      fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      fast_iter(std::integral_constant<int, 1>{}, kPlainNext, j + 1, s_nxt, s_cur, nsc0, nsc1, sc0, sc1);
      fast_iter(std::integral_constant<int, 2>{}, kPlainNext, j + 2, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      fast_iter(std::integral_constant<int, 3>{}, kPlainNext, j + 3, s_nxt, s_cur, nsc0, nsc1, sc0, sc1);
    }
    // up to three fast tiles left (j % 4 == 0 here); after an odd number the live scores sit in the other register set
    bool odd = false;
    if (j < n_fast) {
      fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      ++j; odd = true;
      if (j < n_fast) {
        fast_iter(std::integral_constant<int, 1>{}, kPlainNext, j, s_nxt, s_cur, nsc0, nsc1, sc0, sc1);
        ++j; odd = false;
        if (j < n_fast) {
          fast_iter(std::integral_constant<int, 2>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
          ++j; odd = true;
        }
      }
    }
    if (odd) {
      s_cur[0] = s_nxt[0]; s_cur[1] = s_nxt[1];
      sc0 = nsc0; sc1 = nsc1;
    }
  } else {
  for (; j + 1 < n_fast; j += 2) {
    fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
    fast_iter(std::integral_constant<int, 1>{}, kPlainNext, j + 1, s_nxt, s_cur, nsc0, nsc1, sc0, sc1);
  }
  // an odd fast tile left (j is even here): one more fast iteration instead of a generic one (+11 % at C2, where the
  // generic body otherwise takes 2 of 32 tiles).
  if constexpr (!abl::kNoOddFast) {
    if (j < n_fast) {
      fast_iter(std::integral_constant<int, 0>{}, kPlainNext, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      s_cur[0] = s_nxt[0]; s_cur[1] = s_nxt[1];
      sc0 = nsc0; sc1 = nsc1;
      ++j;
    }
  }
  }
  {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    row_l = q0 + (ln & 31);
    hh_l = ln >> 5;
  }
  // staging with run-time slots: the compiler-scheduled tail body below and, for causal waves that are done early, staging-only iterations;
  // every copy drained (vmcnt(0)) -- the four-slot ring keeps its copy COUNT per iteration constant here as well
  auto stage_generic = [&](const int jj) __attribute__((always_inline)) {
    if constexpr (RING == 2) {
      if (jj + 2 < ntiles) dma_k(jj + 2, jj & 1);
      if (jj + 1 < ntiles) load_v(jj + 1, (jj + 1) & 1);
    } else {
      dma_k(min(jj + RING, last_tile), jj % RING);
      load_v(min(jj + RING - 1, last_tile), (jj + RING - 1) % RING);
    }
  };
  // The wave's remaining tiles (a successor that may need masking, then its last one).
  //  * FP16 / BF16 PV and FP8 PV at head_dim 64: the same hand-placed stream with
  //    run-time slots (fast_iter, NEXT = 1 / 2): C2 +0.8 %, C2-fp8 +1.6 %, (8,32,2048,128) causal +3.5 %.
  //  * FP8 PV at head_dim 128: the compiler-scheduled body.  With the stream variants instantiated
  //    their MAIN loop came out 0.3-1 % slower (different register assignment), more than the tail tiles return.
  constexpr bool STREAM_TAIL = !(PV_FP8 && D == 128);
  if constexpr (STREAM_TAIL) {
    for (; j + 1 < wave_tiles; ++j) {
      fast_iter(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      s_cur[0] = s_nxt[0]; s_cur[1] = s_nxt[1];
      sc0 = nsc0; sc1 = nsc1;
    }
    if (j < wave_tiles) {
      fast_iter(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, j, s_cur, s_nxt, sc0, sc1, nsc0, nsc1);
      ++j;
    }
  } else {
    for (; j < wave_tiles; ++j) {
      maybe_rescale(mx_cur);
      stage_generic(j);
      const bool has_next = j + 1 < wave_tiles;
      if (has_next) {
        tile_scales(j + 1, nsc0, nsc1);
        qk((j + 1) % RING, s_nxt);
        if (j + 1 >= n_plain) mask_limit(j + 1, s_nxt);  // a plain last tile (N % 64 == 0, no diagonal) needs none
      }
      softmax_pv(j, j % RING, s_cur, sc0, sc1, std::true_type{});
      if (has_next) mx_cur = row_max(s_nxt, nsc0, nsc1);
      dma_wait_all();
      __syncthreads();
      s_cur[0] = s_nxt[0]; s_cur[1] = s_nxt[1];
      sc0 = nsc0; sc1 = nsc1;
    }
  }
  for (; j < ntiles; ++j) {
    stage_generic(j);
    dma_wait_all();
    __syncthreads();
  }

  }

  // ---- epilogue: normalise, (+ v_mean), convert, store; LSE (…sm80.cu:540-668)
  const float l_tot = MROW ? (((row_l - q0) & 16) ? l4[1] : l4[0]) : swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (row_l < M_) {
    uint16_t* op = p.o + o_boff + h * p.osh + (int64_t)row_l * p.osn;
    // A lane holds runs of 4 output channels (8 B); lane ^ 32 holds the neighbouring run of the same row.  The two halves
    // exchange words (v_permlane32_swap) so that each stores 16 contiguous bytes: 8 global_store_dwordx4 per lane instead
    // of 16 dwordx2 (the store tail of a workgroup is bound by the number of store instructions, not by bytes).
    // (instantiated per output element type and store form, selected by ONE uniform branch: as run-time flags inside the
    //  unrolled loops they cost a scalar branch per 4-channel run)
    auto store_rows = [&](auto has_vm, auto obf_tag, auto vec16_tag) __attribute__((always_inline)) {
      constexpr bool OBF = decltype(obf_tag)::value, VEC16 = decltype(vec16_tag)::value;
      const float* vmp = p.v_mean + ((int64_t)b * p.Hk + hk) * D;
      auto run4 = [&](const int dt, const int g4) __attribute__((always_inline)) -> uint2 {
        const int d0 = 32 * dt + 8 * g4 + 4 * hh_l;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = acc_o[dt][4 * g4 + e] * inv;
        if constexpr (PV_FP8) {  // fuse_v_scale (qk_int_sv_f8_cuda_sm89.cuh:578-626)
          const float4 vs = *reinterpret_cast<const float4*>(p.v_scale + ((int64_t)b * p.Hk + hk) * D + d0);
          x[0] *= vs.x; x[1] *= vs.y; x[2] *= vs.z; x[3] *= vs.w;
        }
        if constexpr (decltype(has_vm)::value) {
          const float4 vmv = *reinterpret_cast<const float4*>(vmp + d0);
          x[0] += vmv.x; x[1] += vmv.y; x[2] += vmv.z; x[3] += vmv.w;
        }
        // o = round16(round32(acc * inv ...)): the fp32 value is made opaque, otherwise hipcc folds the last multiply and the
        // convert into v_fma_mixlo_f16 (one rounding) in SOME instantiations -- more exact by up to one fp16 ulp in ~5e-5 of
        // the elements, but not the arithmetic of the reference epilogue (…sm80.cu:600-640) nor of this library's earlier builds
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(x[e]));
        uint2 w;  // packed converts (round to nearest even, as the scalar ones): v_cvt_pk_{f16,bf16}_f32
        if constexpr (OBF) {
          w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f){x[0], x[1]}, v2bf));
          w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f){x[2], x[3]}, v2bf));
        } else {
          w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f){x[0], x[1]}, v2h));
          w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f){x[2], x[3]}, v2h));
        }
        return w;
      };
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          // runs A (g4 = 2gp) and B (g4 = 2gp+1): the low half-wave keeps both halves of A, the high one both halves of B
          const uint2 wa = run4(dt, 2 * gp), wb = run4(dt, 2 * gp + 1);
          if constexpr (VEC16) {  // (a row and its lane ^ 32 twin are both inside or both outside the `row < M` guard)
            const auto sx = __builtin_amdgcn_permlane32_swap(wa.x, wb.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(wa.y, wb.y, false, false);
            *reinterpret_cast<uint4*>(op + 32 * dt + 16 * gp + 8 * hh_l) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          } else {          // output rows that are only 8-byte aligned: the runs as they are
            *reinterpret_cast<uint2*>(op + 32 * dt + 16 * gp + 4 * hh_l) = wa;
            *reinterpret_cast<uint2*>(op + 32 * dt + 16 * gp + 8 + 4 * hh_l) = wb;
          }
        }
    };
    constexpr std::true_type kT{};
    constexpr std::false_type kF{};
    if (!p.v_mean && p.o_vec16) {  // the common forms
      if (p.out_bf16) store_rows(kF, kT, kT); else store_rows(kF, kF, kT);
    } else if (p.v_mean) {
      if (p.o_vec16) { if (p.out_bf16) store_rows(kT, kT, kT); else store_rows(kT, kF, kT); }
      else { if (p.out_bf16) store_rows(kT, kT, kF); else store_rows(kT, kF, kF); }
    } else {
      if (p.out_bf16) store_rows(kF, kT, kF); else store_rows(kF, kF, kF);
    }
    if (p.lse && hh_l == 0) {
      const float lse2 = m_run + log2f(l_tot) - kPOff;  // base 2, scaled + smoothed logits (…sm80.cu:657-668)
      float* const slot = p.lse + ((int64_t)b * p.Hq + h) * M_ + row_l;
      *slot = p.q_f16 ? lse2 / 1.44269504f + *slot * p.sm_scale : lse2;
    }
  }
}

// dynamic LDS above the 48 KiB default needs the function attribute; its status is part of the launch status
static bool allow_lds(const void* kern, size_t bytes) {
  launch_begin();
  return bytes <= 48 * 1024 ||
         hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

template <int D, int NWAVES, bool PV_FP8>
static int launch_attn(const AttnParams& p, bool causal, bool kthread, bool v_bf16, hipStream_t st) {
  if constexpr (!PV_FP8) {
    if (p.mask) {  // attn_mask variant (non-causal, fp16 V: checked by run_attn)
      const size_t smem_m = 2 * 64 * D + 2 * 64 * D * 2;
      const dim3 grid_m(p.nqb * p.Hq * p.B), block_m(NWAVES * 64);
#define SAGE_LAUNCH_MASKED(K, V)                                                                                   \
  do {                                                                                                             \
    auto kern = attn_i8_kernel<D, NWAVES, false, K, V, false, true>;                                                \
    if (!allow_lds((const void*)kern, smem_m)) return SAGE_ERR_LAUNCH;                                             \
    hipLaunchKernelGGL(kern, grid_m, block_m, smem_m, st, p);                                                      \
  } while (0)
      if (kthread) { if (v_bf16) SAGE_LAUNCH_MASKED(true, true); else SAGE_LAUNCH_MASKED(true, false); }
      else { if (v_bf16) SAGE_LAUNCH_MASKED(false, true); else SAGE_LAUNCH_MASKED(false, false); }
#undef SAGE_LAUNCH_MASKED
      return launch_status();
    }
  }
  const size_t smem = (size_t)attn_ring_slots(D, NWAVES, PV_FP8) * (64 * D + (PV_FP8 ? 64 * D : 64 * D * 2));  // RING x (K tile + V tile)
  const dim3 grid(p.nqb * p.Hq * p.B), block(NWAVES * 64);
#define SAGE_LAUNCH(C, K, V)                                                                                       \
  do {                                                                                                             \
    auto kern = attn_i8_kernel<D, NWAVES, C, K, V, PV_FP8, false>;                                                      \
    if (!allow_lds((const void*)kern, smem)) return SAGE_ERR_LAUNCH;                                                   \
    hipLaunchKernelGGL(kern, grid, block, smem, st, p);                                                            \
  } while (0)
#define SAGE_BY_V(C, K)                                                                                            \
  do {                                                                                                             \
    if constexpr (PV_FP8) SAGE_LAUNCH(C, K, false);                                                                \
    else { if (v_bf16) SAGE_LAUNCH(C, K, true); else SAGE_LAUNCH(C, K, false); }                                   \
  } while (0)
#define SAGE_BY_K(C) do { if (kthread) SAGE_BY_V(C, true); else SAGE_BY_V(C, false); } while (0)
  if (causal) SAGE_BY_K(true); else SAGE_BY_K(false);
#undef SAGE_BY_K
#undef SAGE_BY_V
#undef SAGE_LAUNCH
  return launch_status();
}

static bool t_ok(const sage_tensor* t, int align_elems) {
  return t && t->data && aligned16(t->data) && t->stride_b % align_elems == 0 && t->stride_h % align_elems == 0 &&
         t->stride_n % align_elems == 0;
}

// tuning hook (sage_set_tuning): per host thread, so a test or tool that pins the geometry for its own calls cannot change
// the launches of another thread; 0 = the measured default below
static thread_local int g_nwaves_override = 0;


// shared argument handling of the two attention entry points
static int run_attn(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, bool pv_fp8, int v_dtype,
                    const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale, const float* v_scale,
                    const float* v_mean, float* lse, int B, int Hq, int Hk, int M, int N, int D, int is_causal, int qk_gran,
                    int blkq, int warpq, float sm_scale, int logit_mult_is_one, hipStream_t st,
                    const int* cu_q = nullptr, const int* cu_k = nullptr, int q_dtype = -1, const void* km = nullptr,
                    const void* mask = nullptr, int mask_kind = 0, const int64_t* mask_strides = nullptr,
                    const sage_kv_layout* kvl = nullptr) {
  if (mask && (mask_kind < 1 || mask_kind > 3 || !mask_strides || is_causal || pv_fp8 || cu_q)) return SAGE_ERR_INVALID_ARGUMENT;
  const bool fusedq = q_dtype >= 0;  // q8 is then the fp16/bf16 query tensor
  if ((cu_q == nullptr) != (cu_k == nullptr)) return SAGE_ERR_INVALID_ARGUMENT;
  if (fusedq) {
    if (q_dtype != SAGE_F16 && q_dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
    if (qk_gran == SAGE_GRAN_PER_BLOCK || cu_q || (km && !aligned16(km))) return SAGE_ERR_UNSUPPORTED;
    if (!t_ok(q8, 8) || !k_scale) return SAGE_ERR_INVALID_ARGUMENT;
    q_scale = k_scale;  // unused placeholder so the shared checks below pass
  }
  if (cu_q && (pv_fp8 || lse)) return SAGE_ERR_UNSUPPORTED;  // packed sequences: fp16 PV, no LSE (as the reference)
  if (!t_ok(q8, fusedq ? 8 : 16) || !t_ok(k8, 16) || !t_ok(v, pv_fp8 ? 16 : 8) || !t_ok(o, 4) || !q_scale || !k_scale) return SAGE_ERR_INVALID_ARGUMENT;
  if (pv_fp8 && !v_scale) return SAGE_ERR_INVALID_ARGUMENT;
  if (B <= 0 || Hq <= 0 || Hk <= 0 || M <= 0 || N <= 0 || Hq % Hk != 0) return SAGE_ERR_INVALID_ARGUMENT;
  // the integer row max and the -inf mask pattern rely on a positive, finite dequantisation scale
  if (!logit_mult_is_one && !(sm_scale > 0.f && sm_scale < 1.0e30f)) return SAGE_ERR_INVALID_ARGUMENT;
  if (D != 64 && D != 128) return SAGE_ERR_UNSUPPORTED_HEAD_DIM;
  if ((v_dtype != SAGE_F16 && v_dtype != SAGE_BF16) || (o_dtype != SAGE_F16 && o_dtype != SAGE_BF16)) return SAGE_ERR_INVALID_ARGUMENT;
  if (qk_gran < SAGE_GRAN_PER_BLOCK || qk_gran > SAGE_GRAN_PER_THREAD) return SAGE_ERR_INVALID_ARGUMENT;
  if (blkq != 64 && blkq != 128) return SAGE_ERR_INVALID_ARGUMENT;
  if (qk_gran == SAGE_GRAN_PER_BLOCK) warpq = blkq;
  if ((warpq != 16 && warpq != 32 && warpq != 64 && warpq != 128) || blkq % warpq != 0) return SAGE_ERR_INVALID_ARGUMENT;
  if ((v_mean && !aligned16(v_mean)) || (v_scale && !aligned16(v_scale))) return SAGE_ERR_INVALID_ARGUMENT;
  // KV tile layout: dense by default (a tile = 64 consecutive rows of the sage_tensor), explicit for tile-major buffers
  const int64_t ntile = ((int64_t)N + 63) >> 6;
  int64_t k_tile = 64 * k8->stride_n, v_tile = pv_fp8 ? 64 : 128 * v->stride_n;  // bytes
  const int per_tile = qk_gran == SAGE_GRAN_PER_THREAD ? 4 : 1;
  int64_t ks_h = ntile * per_tile, ks_b = ks_h * Hk, ks_t = per_tile;
  bool tiled = false;
  if (kvl) {
    if (cu_q || mask || kvl->k_tile_stride < 0 || kvl->v_tile_stride < 0) return SAGE_ERR_INVALID_ARGUMENT;
    if (kvl->k_tile_stride) { k_tile = kvl->k_tile_stride; tiled = true; }
    if (kvl->v_tile_stride) { v_tile = pv_fp8 ? kvl->v_tile_stride : 2 * kvl->v_tile_stride; tiled = true; }
    if (kvl->ks_stride_b || kvl->ks_stride_h || kvl->ks_stride_tile) {
      if (kvl->ks_stride_tile < per_tile || kvl->ks_stride_h < 0 || kvl->ks_stride_b < 0) return SAGE_ERR_INVALID_ARGUMENT;
      ks_b = kvl->ks_stride_b; ks_h = kvl->ks_stride_h; ks_t = kvl->ks_stride_tile;
      if (per_tile == 4 && ((ks_b | ks_h | ks_t) & 3)) return SAGE_ERR_INVALID_ARGUMENT;  // 16-B scalar loads
    }
    if ((k_tile & 15) || (v_tile & 15)) return SAGE_ERR_INVALID_ARGUMENT;
  }
  // the K/V slices of one (b, h_kv) are addressed with 32-bit buffer offsets
  const int64_t lim = (int64_t)1 << 31;
  if (ntile * k_tile + 64 * k8->stride_n + D >= lim) return SAGE_ERR_TOO_LARGE;
  if (pv_fp8 ? (ntile * v_tile + (int64_t)D * v->stride_n + 64 >= lim) : (ntile * v_tile + (64 * v->stride_n + D) * 2 >= lim)) return SAGE_ERR_TOO_LARGE;
  if (ks_t * ntile >= lim) return SAGE_ERR_TOO_LARGE;
  AttnParams p;
  p.q = (const int8_t*)q8->data; p.qsb = q8->stride_b; p.qsh = q8->stride_h; p.qsn = q8->stride_n;
  p.k = (const int8_t*)k8->data; p.ksb = k8->stride_b; p.ksh = k8->stride_h; p.ksn = k8->stride_n;
  p.v = (const uint8_t*)v->data; p.vsb = v->stride_b; p.vsh = v->stride_h; p.vsn = v->stride_n;
  p.o = (uint16_t*)o->data; p.osb = o->stride_b; p.osh = o->stride_h; p.osn = o->stride_n;
  p.q_scale = q_scale; p.k_scale = k_scale; p.v_scale = v_scale; p.v_mean = v_mean; p.lse = lse;
  p.B = B; p.Hq = Hq; p.Hk = Hk; p.M = M; p.N = N;
  const int nblkq = (M + blkq - 1) / blkq;
  p.gq = qk_gran == SAGE_GRAN_PER_BLOCK ? nblkq : qk_gran == SAGE_GRAN_PER_WARP ? nblkq * (blkq / warpq) : nblkq * (blkq / warpq) * 8;
  const int nblkk = (N + 63) / 64;
  p.gk = qk_gran == SAGE_GRAN_PER_THREAD ? nblkk * 4 : nblkk;
  p.qgran = qk_gran; p.blkq = blkq; p.warpq = warpq;
  p.logit_mult = logit_mult_is_one ? 1.0f : sm_scale * kLog2e;
  p.out_bf16 = o_dtype == SAGE_BF16;
  p.cu_q = cu_q; p.cu_k = cu_k;
  p.mask = (const uint8_t*)mask; p.mask_kind = mask ? mask_kind : 0;
  p.msb = mask ? mask_strides[0] : 0; p.msh = mask ? mask_strides[1] : 0; p.msm = mask ? mask_strides[2] : 0; p.msn = mask ? mask_strides[3] : 0;
  p.k_tile_bytes = (int)k_tile; p.v_tile_bytes = (int)v_tile; p.ks_b = ks_b; p.ks_h = ks_h; p.ks_t = (int)ks_t;
  p.kv_tiled = tiled ? 1 : 0;
  p.o_vec16 = (o->stride_b % 8 == 0 && o->stride_h % 8 == 0 && o->stride_n % 8 == 0) ? 1 : 0;  // 16-byte aligned output rows
  p.q_f16 = fusedq ? (const uint16_t*)q8->data : nullptr;
  p.km = (const uint16_t*)km; p.q_bf16 = q_dtype == SAGE_BF16; p.sm_scale = sm_scale;
  const bool kthread = qk_gran == SAGE_GRAN_PER_THREAD, vb = v_dtype == SAGE_BF16;
  // measured on MI355X: D=128 fp16 PV -> one 8-wave workgroup per CU (4-wave: -3 %); D=128 fp8 PV -> two 4-wave
  // workgroups per CU (+3.6 % non-causal, +5.7 % causal); D=64 (<= 168 VGPRs) -> 4-wave workgroups, 3 per CU
  // ... except for short key sequences (few tiles per workgroup, so prologue and epilogue weigh more and two smaller
  // workgroups per CU overlap them better).  Re-measured at the end of round 2 (the prologue now issues its tile copies
  // first): 4-wave +10 % at 1024 keys, +7 % at 1536, +6 % at 2048, 0 % at 3072-4096, -2 % from 6144; causal (a row attends
  // half the keys on average) +21 % at 2048, +13 % at 4096, 0 % at 8192, -1 % at 16384 -> 4 waves up to 2048 keys per row.
  // Round 3, end to end with the Q quantizer in the prologue (tools/nw_e2e.py, profiles/r03_ab/geometry_end_to_end.log): the
  // heavier prologue moves the crossover out -- 4-wave 0.959x the 8-wave time at 2048 keys, 0.979x at 3072, 0.998x at 4096,
  // 1.02x from 6144; causal 0.993x at 8192 (4096 keys per row on average) -> 4 waves up to 3072 keys per row.
  const int keys_per_row = is_causal ? N / 2 : N;
  // FP8 PV at head_dim 128, end of round 3 (tools/ab_bench.py --pv fp8 lib@4 lib@8, non-causal): 4-wave workgroups +2.2 % at 8K
  // keys, +0.7 % at 16K, -1.1 % at 32K, -1.3 % at 64K.  The two co-resident 4-wave workgroups of a CU drift apart on a long
  // stream (the older one wins the issue arbitration) until they no longer share K/V tiles in L2: FETCH_SIZE of one rank's launch
  // of the 64K-key gather schedule (8192 rows x 65536 keys) is 1.90x the algorithmic bytes with 4 waves and 1.00x with 8
  // (profiles/r03_ab/fetch_by_geometry.md) -> 8 waves beyond 24K keys per row.
  const int nw = g_nwaves_override ? g_nwaves_override
                                   : ((D == 64 || (pv_fp8 && keys_per_row <= 24576) || keys_per_row <= 3072) ? 4 : 8);
  p.nqb = (M + nw * 32 - 1) / (nw * 32);
#define SAGE_GO(DD, NW) (pv_fp8 ? launch_attn<DD, NW, true>(p, is_causal, kthread, false, st) : launch_attn<DD, NW, false>(p, is_causal, kthread, vb, st))
  if (nw == 8) return D == 64 ? SAGE_GO(64, 8) : SAGE_GO(128, 8);
  return D == 64 ? SAGE_GO(64, 4) : SAGE_GO(128, 4);
#undef SAGE_GO
}

}  // namespace sage

using namespace sage;

extern "C" int sage_set_tuning(int key, int value) {
  if (key == SAGE_TUNE_NWAVES) {
    if (value != 0 && value != 4 && value != 8) return SAGE_ERR_INVALID_ARGUMENT;
    g_nwaves_override = value;
    return SAGE_OK;
  }

  return SAGE_ERR_INVALID_ARGUMENT;
}

extern "C" int sage_get_tuning(int key) {
  if (key == SAGE_TUNE_NWAVES) return g_nwaves_override;
  return -1;
}

extern "C" int sage_attn_qk_int8_pv_f16(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                        const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                        const float* v_mean, float* lse, int B, int Hq, int Hk, int M, int N, int D,
                                        int is_causal, int qk_gran, int blkq, int warpq, float sm_scale,
                                        int logit_mult_is_one, sage_stream_t stream) {
  return run_attn(q8, k8, v, false, v_dtype, o, o_dtype, q_scale, k_scale, nullptr, v_mean, lse, B, Hq, Hk, M, N, D, is_causal,
                  qk_gran, blkq, warpq, sm_scale, logit_mult_is_one, (hipStream_t)stream);
}

extern "C" int sage_attn_qk_int8_pv_f8(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v_fp8,
                                       const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                       const float* v_scale, const float* v_mean, float* lse, int B, int Hq, int Hk, int M,
                                       int N, int D, int is_causal, int qk_gran, int blkq, int warpq, float sm_scale,
                                       int logit_mult_is_one, sage_stream_t stream) {
  return run_attn(q8, k8, v_fp8, true, SAGE_F16, o, o_dtype, q_scale, k_scale, v_scale, v_mean, lse, B, Hq, Hk, M, N, D,
                  is_causal, qk_gran, blkq, warpq, sm_scale, logit_mult_is_one, (hipStream_t)stream);
}

extern "C" int sage_attn_qk_int8_pv_f16_varlen(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                               const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                               const int* cu_seqlens_q, const int* cu_seqlens_k, int num_seqs, int Hq, int Hk,
                                               int max_seqlen_q, int max_seqlen_k, int D, int is_causal, int qk_gran, int blkq,
                                               int warpq, float sm_scale, int logit_mult_is_one, sage_stream_t stream) {
  if (!cu_seqlens_q || !cu_seqlens_k) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q8, k8, v, false, v_dtype, o, o_dtype, q_scale, k_scale, nullptr, nullptr, nullptr, num_seqs, Hq, Hk,
                  max_seqlen_q, max_seqlen_k, D, is_causal, qk_gran, blkq, warpq, sm_scale, logit_mult_is_one,
                  (hipStream_t)stream, cu_seqlens_q, cu_seqlens_k);
}

extern "C" int sage_attn_fusedq_pv_f16(const sage_tensor* q, int q_dtype, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                       const sage_tensor* o, int o_dtype, const float* k_scale, const void* km,
                                       const float* v_mean, float* lse, int B, int Hq, int Hk, int M, int N, int D,
                                       int is_causal, int qk_gran, int warpq, float sm_scale, sage_stream_t stream) {
  if (q_dtype != SAGE_F16 && q_dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q, k8, v, false, v_dtype, o, o_dtype, nullptr, k_scale, nullptr, v_mean, lse, B, Hq, Hk, M, N, D, is_causal,
                  qk_gran, 128, warpq, sm_scale, 0, (hipStream_t)stream, nullptr, nullptr, q_dtype, km);
}

extern "C" int sage_attn_fusedq_pv_f8(const sage_tensor* q, int q_dtype, const sage_tensor* k8, const sage_tensor* v_fp8,
                                      const sage_tensor* o, int o_dtype, const float* k_scale, const void* km,
                                      const float* v_scale, const float* v_mean, float* lse, int B, int Hq, int Hk, int M,
                                      int N, int D, int is_causal, int qk_gran, int warpq, float sm_scale,
                                      sage_stream_t stream) {
  if (q_dtype != SAGE_F16 && q_dtype != SAGE_BF16) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q, k8, v_fp8, true, SAGE_F16, o, o_dtype, nullptr, k_scale, v_scale, v_mean, lse, B, Hq, Hk, M, N, D,
                  is_causal, qk_gran, 128, warpq, sm_scale, 0, (hipStream_t)stream, nullptr, nullptr, q_dtype, km);
}

extern "C" int sage_attn_qk_int8_pv_f16_masked(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                               const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                               const void* attn_mask, int mask_kind, const int64_t* mask_strides, float* lse,
                                               int B, int Hq, int Hk, int M, int N, int D, int qk_gran, int blkq, int warpq,
                                               float sm_scale, int logit_mult_is_one, sage_stream_t stream) {
  if (!attn_mask) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q8, k8, v, false, v_dtype, o, o_dtype, q_scale, k_scale, nullptr, nullptr, lse, B, Hq, Hk, M, N, D, 0, qk_gran,
                  blkq, warpq, sm_scale, logit_mult_is_one, (hipStream_t)stream, nullptr, nullptr, -1, nullptr, attn_mask,
                  mask_kind, mask_strides);
}

extern "C" int sage_attn_qk_int8_pv_f16_kvtiles(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                                const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                                const sage_kv_layout* kv_layout, float* lse, int B, int Hq, int Hk, int M,
                                                int N, int D, int is_causal, int qk_gran, int blkq, int warpq, float sm_scale,
                                                sage_stream_t stream) {
  if (!kv_layout) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q8, k8, v, false, v_dtype, o, o_dtype, q_scale, k_scale, nullptr, nullptr, lse, B, Hq, Hk, M, N, D, is_causal,
                  qk_gran, blkq, warpq, sm_scale, 0, (hipStream_t)stream, nullptr, nullptr, -1, nullptr, nullptr, 0, nullptr,
                  kv_layout);
}

extern "C" int sage_attn_qk_int8_pv_f8_kvtiles(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v_fp8,
                                               const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                               const float* v_scale, const sage_kv_layout* kv_layout, float* lse, int B,
                                               int Hq, int Hk, int M, int N, int D, int is_causal, int qk_gran, int blkq,
                                               int warpq, float sm_scale, sage_stream_t stream) {
  if (!kv_layout) return SAGE_ERR_INVALID_ARGUMENT;
  return run_attn(q8, k8, v_fp8, true, SAGE_F16, o, o_dtype, q_scale, k_scale, v_scale, nullptr, lse, B, Hq, Hk, M, N, D,
                  is_causal, qk_gran, blkq, warpq, sm_scale, 0, (hipStream_t)stream, nullptr, nullptr, -1, nullptr, nullptr, 0,
                  nullptr, kv_layout);
}
