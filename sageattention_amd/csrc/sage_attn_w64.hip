// Hot-shape specialisation of the fused INT8-QK^T / FP16-PV attention kernel: ONE WAVE PER SIMD, 64 QUERY ROWS PER WAVE.
//
// Why: beside an MFMA every LDS fragment read costs the issuing wave 7-9 cycles (tools/issue_cost.hip; removing the
// reads from attn_i8_kernel, timing only, lifts C3 from 1.38 to 1.94 PFLOP/s).  A wave that owns TWO 32-row query
// sub-tiles uses every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) for two MFMAs, halving
// LDS instructions and LDS->VGPR traffic per flop.  The price is registers: two sets of O^T accumulators (128) and Q
// fragments (32) -- they live in the ACCUMULATOR half of the 512-entry unified file, which only MFMAs can address, so
// the MFMAs are issued through inline asm with "a" operands (hipcc would otherwise park the S tiles there and copy
// them out with 64 v_accvgpr_read per tile).  The arch VGPRs hold S(j), S(j+1) and the softmax.
// With one wave per SIMD nothing else fills the gaps between MFMAs: the stream is hand-placed (same scheme as the
// fast path of attn_i8_kernel): per quarter of 16 keys and per d-tile {V^T fragment read, 2 P.V MFMAs (one per
// sub-tile), the fma/exp2/cvt of one P pair of the NEXT quarter and the row-sum adds of the previous pair} -- 6 VALU
// per MFMA, i.e. about 32 issue cycles beside a 32-cycle MFMA -- with the S(j+1) MFMAs spread between them.
// Arithmetic, LDS images, LDS-DMA staging and the (m, l, O) bookkeeping are those of attn_i8_kernel; results are
// bit-identical to it.  Covers D = 128, fp16 V, int8 q (per_warp / per_thread scales), causal or not, ragged M/N,
// GQA, LSE, dense [B,H,N,D] tensors with any strides.  Everything else stays on attn_i8_kernel.
#include "sage_attn_common.h"

namespace sage {

// ---- MFMAs with operands pinned to register classes (C and D of an MFMA share one class: ACC_CD)
// S^T(first k-step) = K.Q^T + bias: D, C (bias), A (K fragment) in VGPRs, B (Q fragment) in AGPRs
// The accumulator is initialised IN PLACE (C = D): a C operand in a temporary tuple would be dead to hipcc as soon as
// the statement is issued and could be overwritten while the MFMA still reads it (see attn_i8_kernel, mfma_s_first).
// s_nop 1: the moves that write the bias are VALU results feeding an MFMA source.
__device__ __forceinline__ void mfma_s_first(v16i& d, const v4i& a, const v4i& b, const v16i& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  d = c;
  asm("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
#endif
}
__device__ __forceinline__ void mfma_s_acc(v16i& d, const v4i& a, const v4i& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
#endif
}
// O^T += V^T.P^T: accumulator in AGPRs, fragments in VGPRs.  hipcc pads no hazards around asm: callers keep at least
// two instructions between the VALU write of an operand and the MFMA (cdna guide 5.7), see the stream below.
__device__ __forceinline__ void mfma_pv(v16f& acc, const v8h& a, const v8h& b) {
#ifdef SAGE_W64_ASM_PV
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
#else
  // builtin: with the arch VGPRs full of S tiles hipcc selects the AGPR form by itself, and (unlike the tied asm
  // operand) coalesces the accumulator across the two unrolled loop halves without copies
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#endif
}
// MFMA result -> VALU / v_accvgpr_read: up to 18 wait states after the last MFMA of the chain.  The registers pass
// THROUGH the statement so that no reader can be scheduled above it.
__device__ __forceinline__ void drain_o(v16f (&o)[2][4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]),
                 "+a"(o[1][3]));
#endif
}
__device__ __forceinline__ void drain_s(v16i (&s)[2][2]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
#endif
}
// no instruction: keeps the eight O^T accumulators in the accumulator register class at this point, so that the merge
// after the (rare) rescale branch is AGPR with AGPR and costs the common path no copies
__device__ __forceinline__ void pin_o(v16f (&o)[2][4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(""
               : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]),
                 "+a"(o[1][3]));
#endif
}
// a just-written VALU result -> MFMA operand: 2 wait states
__device__ __forceinline__ void settle_p(v8h (&pf)[2]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 1" : "+v"(pf[0]), "+v"(pf[1]));
#endif
}

template <int D, bool CAUSAL, bool KTHREAD>
__global__ __launch_bounds__(256, 1) void attn_i8_w64_kernel(const AttnParams p) {
  constexpr int NWAVES = 4, T = 256, QT = 2, QB = NWAVES * 32 * QT;
  constexpr int KS = D / 32, DT = D / 32;
  constexpr int KBYTES = 64 * D, VBYTES = 64 * D * 2;
  constexpr int KCH = D / 16, VCH = D / 8;
  constexpr int KC = (64 * KCH) / T, VC = (64 * VCH) / T;
  static_assert((64 * KCH) % T == 0 && (64 * VCH) % T == 0, "tiles must divide over the workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const k_lds = smem;
  char* const v_lds = smem + 2 * KBYTES;

  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x, xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  }
  int qb = lid % p.nqb;
  const int bh = lid / p.nqb;
  const int h = bh % p.Hq, b = bh / p.Hq;
  if constexpr (CAUSAL) qb = p.nqb - 1 - qb;
  const int hk = h / (p.Hq / p.Hk);
  const int M_ = p.M, N_ = p.N;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int q0 = qb * QB + wave * 64;  // first row of the wave; sub-tile qt covers rows q0 + 32*qt + [0,32)
  int row[QT], rowc[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { row[qt] = q0 + 32 * qt + r; rowc[qt] = min(row[qt], M_ - 1); }

  v4i qf[QT][KS];
  float qsc[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int8_t* qp = p.q + b * p.qsb + h * p.qsh + (int64_t)rowc[qt] * p.qsn + 16 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = *reinterpret_cast<const v4i*>(qp + 32 * ks);
    int qi;
    if (p.qgran == SAGE_GRAN_PER_BLOCK) qi = rowc[qt] / p.blkq;
    else if (p.qgran == SAGE_GRAN_PER_WARP) qi = rowc[qt] / p.warpq;
    else qi = (rowc[qt] / p.warpq) * 8 + (rowc[qt] & 7);
    qsc[qt] = p.q_scale[((int64_t)b * p.Hq + h) * p.gq + qi] * p.logit_mult;
  }
  const float* ksp = p.k_scale + ((int64_t)b * p.Hk + hk) * p.gk;

  const int kv_end = CAUSAL ? min(N_, (qb + 1) * QB) : N_;
  const int ntiles = (kv_end + 63) >> 6;
  const int wave_tiles = CAUSAL ? min(ntiles, ((q0 + 63) >> 6) + 1) : ntiles;

  // ---- staging: LDS-DMA, swizzle on the source offset (same images as the general kernel)
  const int8_t* kg = p.k + b * p.ksb + hk * p.ksh;
  const uint8_t* vg = p.v + (b * p.vsb + hk * p.vsh) * 2;
  const v4i k_rsrc = make_rsrc(kg, (unsigned)((int64_t)(N_ - 1) * p.ksn + D));
  const v4i v_rsrc = make_rsrc(vg, (unsigned)(((int64_t)(N_ - 1) * p.vsn + D) * 2));
  const int k_tile_stride = 64 * (int)p.ksn, v_tile_stride = 128 * (int)p.vsn;
  // chunk c = tid + i*T of a tile: row c / CH, position c % CH.  T / KCH = 32 and T / VCH = 16 rows per step i leave the
  // swizzle term unchanged ((row >> 1) & 7, row & 3), so ONE per-lane offset serves every i and the row step goes
  // through the scalar offset operand of the buffer load -- registers are the scarce resource of this kernel
  static_assert((T / KCH) % 16 == 0 && (T / VCH) % 4 == 0, "row step must preserve the swizzle");
  int k_voff, v_voff;
  {
    const int kr = tid / KCH, kpos = tid % KCH;
    k_voff = kr * (int)p.ksn + ((kpos ^ k_swz<D>(kr)) << 4);
    const int vr = tid / VCH, vpos = tid % VCH;
    const int cc = (((vpos >> 2) ^ v_win_swz<D>(vr)) << 2) | (vpos & 3);
    v_voff = (vr * (int)p.vsn + cc * 8) * 2;
  }
  const int k_step = (T / KCH) * (int)p.ksn, v_step = (T / VCH) * (int)p.vsn * 2;
  auto dma_k = [&](const int j, const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KC; ++i)
      lds_dma16(k_rsrc, (unsigned)(buf * KBYTES + (wave * 64 + i * T) * 16), k_voff, j * k_tile_stride + i * k_step);
  };
  auto dma_v = [&](const int j, const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < VC; ++i)
      lds_dma16(v_rsrc, (unsigned)(2 * KBYTES + buf * VBYTES + (wave * 64 + i * T) * 16), v_voff, j * v_tile_stride + i * v_step);
  };

  int k_rd[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * D + (((2 * ks + hh) ^ k_swz<D>(r)) << 4);
  int v_rd[DT];
  {
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g = (lane >> 4) & 1;
    const int rv = 4 * hh + q4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) v_rd[dt] = rv * (2 * D) + ((dt ^ v_win_swz<D>(rv)) << 6) + 32 * g + 8 * p4;
  }


  v16f acc_o[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc_o[qt][dt][e] = 0.f;
  constexpr float kLazyThr = 6.0f;
  float m_run[QT], l_run[QT], m_thr[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -1e30f; l_run[qt] = 0.f; m_thr[qt] = m_run[qt] + kLazyThr; }
  constexpr int kBiasI = 0x4B400000;  // see attn_i8_kernel: the int32 accumulator doubles as the float 12582912 + S
  constexpr float kBiasF = 12582912.0f;
  constexpr int kMaskedI = kBiasI - (1 << 22);
  v16i bias;
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = kBiasI;  // not pinned: the arch-VGPR half has no 16 registers to spare here

  // S^T(qt) = K . Q(qt)^T: every K fragment read once, used by both query sub-tiles (generic loop form)
  auto qk = [&](const int kbuf, v16i (&s)[QT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const v4i a = *reinterpret_cast<const v4i*>(k_lds + kbuf * KBYTES + mt * 32 * D + k_rd[ks]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if (ks == 0) mfma_s_first(s[qt][mt], a, qf[qt][ks], bias); else mfma_s_acc(s[qt][mt], a, qf[qt][ks]);
        }
      }
  };
  float qsc_lo[QT], qsc_hi[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { qsc_lo[qt] = hh ? 0.f : qsc[qt]; qsc_hi[qt] = hh ? qsc[qt] : 0.f; }
  auto load_kscales = [&](const int j) __attribute__((always_inline)) -> float4 {
    if constexpr (KTHREAD) return uniform_load4(ksp + j * 4);
    else return make_float4(uniform_load1(ksp + j), 0.f, 0.f, 0.f);
  };
  auto scales_from = [&](const float4 kk, float (&sc)[QT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if constexpr (KTHREAD) {
        sc[qt][0] = __builtin_fmaf(kk.z, qsc_hi[qt], kk.x * qsc_lo[qt]);
        sc[qt][1] = __builtin_fmaf(kk.w, qsc_hi[qt], kk.y * qsc_lo[qt]);
      } else {
        sc[qt][0] = sc[qt][1] = qsc[qt] * kk.x;
      }
    }
  };
  auto allow_bits = [&](const int j, const int qt) __attribute__((always_inline)) -> uint32_t {
    const int n0 = j << 6;
    uint32_t bits = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kv = n0 + 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * hh;
        bits |= (((kv < N_) && !(CAUSAL && kv > row[qt])) ? 1u : 0u) << (16 * mt + e);
      }
    return bits;
  };
  auto mask_scores = [&](const uint32_t bits, v16i (&s)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[mt][e] = ((bits >> (16 * mt + e)) & 1u) ? s[mt][e] : kMaskedI;
  };
  auto finish_max = [&](const int mxa, const int mxb, const float sc0, const float sc1) __attribute__((always_inline)) -> float {
    float mx;
    if constexpr (KTHREAD) mx = max_raw((__int_as_float(mxa) - kBiasF) * sc0, (__int_as_float(mxb) - kBiasF) * sc1);
    else mx = (__int_as_float(max(mxa, mxb)) - kBiasF) * sc0;
    return swap_max(mx);
  };
  auto row_max = [&](const v16i (&s)[2], const float sc0, const float sc1) __attribute__((always_inline)) -> float {
    int mxa = s[0][0], mxb = s[0][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e & 2) mxb = max(mxb, s[mt][e]); else mxa = max(mxa, s[mt][e]);
      }
    return finish_max(mxa, mxb, sc0, sc1);
  };
  auto maybe_rescale = [&](const float (&mx)[QT]) __attribute__((always_inline)) {
    bool need = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) need |= mx[qt] > m_thr[qt];
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0, 0)) {
      drain_o(acc_o);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const float m_new = fmaxf(m_run[qt], mx[qt]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        m_thr[qt] = m_new + kLazyThr;
        l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc_o[qt][dt][e] *= alpha;
      }
      pin_o(acc_o);
    }
  };
  auto v_frag_at = [&](const char* vb, const int q, const int dt) __attribute__((always_inline)) -> v8h {
    const char* base = vb + 16 * q * (2 * D) + v_rd[dt];
    const v4s_vs lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base));
    const v4s_vs hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_vs*)(base + 8 * (2 * D)));
    v8h a;
    a.s0123 = __builtin_bit_cast(v4h, lo);
    a.s4567 = __builtin_bit_cast(v4h, hi);
    return a;
  };
  uint32_t bits_cur[QT] = {0xffffffffu, 0xffffffffu}, bits_nxt[QT] = {0xffffffffu, 0xffffffffu};
  // generic (masked / last) tiles: p = exp2(t - m) in quarters of 16 keys for both sub-tiles, then the quarter's MFMAs
  auto softmax_pv = [&](const int vbuf, const v16i (&s)[QT][2], const float (&sc)[QT][2]) __attribute__((always_inline)) {
    float c[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      c[qt][0] = __builtin_fmaf(-kBiasF, sc[qt][0], -m_run[qt]);
      c[qt][1] = __builtin_fmaf(-kBiasF, sc[qt][1], -m_run[qt]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mt = q >> 1, sq = q & 1;
      v8h pf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float pp[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ee = 8 * sq + e + u;
            const bool g1 = (ee & 2) != 0;
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(s[qt][mt][ee]), g1 ? sc[qt][1] : sc[qt][0],
                                                             g1 ? c[qt][1] : c[qt][0]));
            pp[u] = ((bits_cur[qt] >> (16 * mt + ee)) & 1u) ? pv : 0.f;
          }
          l_run[qt] += pp[0];
          l_run[qt] += pp[1];
          v2f p2 = {pp[0], pp[1]};
          const v2h ph = __builtin_convertvector(p2, v2h);
          pf[qt][e] = ph[0];
          pf[qt][e + 1] = ph[1];
        }
      settle_p(pf);  // v_cvt_pk_f16_f32 -> MFMA operand
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const v8h a = v_frag_at(v_lds + vbuf * VBYTES, q, dt);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mfma_pv(acc_o[qt][dt], a, pf[qt]);
      }
    }
  };

  // ---- pipeline: K two tiles ahead, V one (as attn_i8_kernel); fast loop unrolled by two
  int n_plain = wave_tiles;
  if (N_ & 63) n_plain = min(n_plain, N_ >> 6);
  if constexpr (CAUSAL) n_plain = min(n_plain, max(0, (q0 + 1) >> 6));
  const int n_fast = max(0, min(n_plain - 1, wave_tiles - 1));

  dma_k(0, 0);
  dma_v(0, 0);
  if (ntiles > 1) dma_k(1, 1);
  dma_wait_all();
  __syncthreads();

  v16i s_cur[QT][2], s_nxt[QT][2];
  float sc_cur[QT][2], sc_nxt[QT][2], mx_cur[QT];
  qk(0, s_cur);
  drain_s(s_cur);
  __syncthreads();  // all waves are done reading K buffer 0 before iteration 0 re-fills it with K(2) (see sage_attn.hip)
  scales_from(load_kscales(0), sc_cur);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    bits_cur[qt] = allow_bits(0, qt);
    mask_scores(bits_cur[qt], s_cur[qt]);
    mx_cur[qt] = row_max(s_cur[qt], sc_cur[qt][0], sc_cur[qt][1]);
  }
  float4 kk_nxt = load_kscales(min(1, ntiles - 1));

  // ---- hand-placed fast iteration (both tiles j, j+1 unmasked)
#define SAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto fast_iter = [&](auto par_tag, const int j, v16i (&sa)[QT][2], v16i (&sb)[QT][2], float (&sca)[QT][2],
                       float (&scb)[QT][2]) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    pin_o(acc_o);
    maybe_rescale(mx_cur);
    pin_o(acc_o);
    if (j + 2 < ntiles) dma_k(j + 2, PAR);
    dma_v(j + 1, PAR ^ 1);
    const char* const kb = k_lds + (PAR ^ 1) * KBYTES;
    const char* const vb = v_lds + PAR * VBYTES;
    float c[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      c[qt][0] = __builtin_fmaf(-kBiasF, sca[qt][0], -m_run[qt]);
      c[qt][1] = __builtin_fmaf(-kBiasF, sca[qt][1], -m_run[qt]);
    }
    auto k_frag = [&](const int i) __attribute__((always_inline)) -> v4i {  // fragment i = (mt, ks) = (i / KS, i % KS)
      return *reinterpret_cast<const v4i*>(kb + (i / KS) * 32 * D + k_rd[i % KS]);
    };
    auto s_step = [&](const int i, const int qt, const v4i a) __attribute__((always_inline)) {
      const int mt = i / KS, ks = i % KS;
      if (ks == 0) mfma_s_first(sb[qt][mt], a, qf[qt][ks], bias); else mfma_s_acc(sb[qt][mt], a, qf[qt][ks]);
    };
    float pp[QT][2];  // the P pair whose row-sum adds are still pending, per sub-tile
    auto p_pair = [&](const int q, const int pr, const int qt, v8h& pf) __attribute__((always_inline)) {
      const int mt = q >> 1, e = 8 * (q & 1) + 2 * pr;
      const bool g1 = (e & 2) != 0;
      const float sc = g1 ? sca[qt][1] : sca[qt][0], cc = g1 ? c[qt][1] : c[qt][0];
      v2f two = {__builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[qt][mt][e]), sc, cc)),
                 __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(sa[qt][mt][e + 1]), sc, cc))};
      const v2h ph = __builtin_convertvector(two, v2h);
      pf[2 * pr] = ph[0];
      pf[2 * pr + 1] = ph[1];
      pp[qt][0] = two[0];
      pp[qt][1] = two[1];
    };
    auto p_sum = [&](const int qt) __attribute__((always_inline)) { l_run[qt] += pp[qt][0]; l_run[qt] += pp[qt][1]; };

    constexpr int NKF = 2 * KS;  // K fragments per tile; each feeds QT MFMAs
    v4i kf = k_frag(0);
    v8h va = v_frag_at(vb, 0, 0), vbn;
    v8h pf[QT], pn[QT];
    // region 0: P(quarter 0) of both sub-tiles beside the first NKF/4 K fragments' S MFMAs
    {
      int fi = 0;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        if (pr % 2 == 0) {  // one K fragment, two S MFMAs
          const v4i a = kf;
          kf = k_frag(fi + 1);
          s_step(fi, 0, a);
          s_step(fi, 1, a);
          ++fi;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if (pr > 0) p_sum(qt);
          p_pair(0, pr, qt, pf[qt]);
          SAGE_FENCE();
        }
      }
      // regions 1..3: per d-tile {V^T fragment of the next step, 2 P.V MFMAs, one P pair of quarter q per sub-tile}
#pragma unroll
      for (int q = 1; q < 4; ++q) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          vbn = (dt + 1 < DT) ? v_frag_at(vb, q - 1, dt + 1) : v_frag_at(vb, q, 0);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            mfma_pv(acc_o[qt][dt], va, pf[qt]);
            p_sum(qt);                 // pair computed one step earlier (pair 3 of the previous quarter when dt == 0)
            p_pair(q, dt, qt, pn[qt]);
            SAGE_FENCE();
          }
          va = vbn;
          if (dt % 2 == 1) {          // two K fragments per region: S MFMAs of both sub-tiles
            const v4i a = kf;
            if (fi + 1 < NKF) kf = k_frag(fi + 1);
            s_step(fi, 0, a);
            s_step(fi, 1, a);
            ++fi;
            SAGE_FENCE();
          }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) pf[qt] = pn[qt];
      }
      // remaining K fragments (NKF - 2 - 3*2 = 0 for D = 128) would go here
      static_assert(NKF == 8, "stream laid out for D = 128");
    }
    // tail: P.V of quarter 3 beside the row max of S(j+1)
    int mxa[QT], mxb[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { p_sum(qt); mxa[qt] = sb[qt][0][0]; mxb[qt] = sb[qt][0][2]; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (dt + 1 < DT) vbn = v_frag_at(vb, 3, dt + 1);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        mfma_pv(acc_o[qt][dt], va, pf[qt]);
#pragma unroll
        for (int idx = dt * (32 / DT); idx < (dt + 1) * (32 / DT); ++idx) {
          const int mt = idx >> 4, e = idx & 15;
          if (e & 2) mxb[qt] = max(mxb[qt], sb[qt][mt][e]); else mxa[qt] = max(mxa[qt], sb[qt][mt][e]);
        }
        asm volatile("" : "+v"(mxa[qt]), "+v"(mxb[qt]));
        SAGE_FENCE();
      }
      va = vbn;
    }
    // scales of tile j+1 (fetched during the previous iteration) are formed only now, to keep them out of the
    // register budget of the stream above; those of tile j+2 are requested for the next iteration
    scales_from(kk_nxt, scb);
    kk_nxt = load_kscales(min(j + 2, ntiles - 1));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mx_cur[qt] = finish_max(mxa[qt], mxb[qt], scb[qt][0], scb[qt][1]);
    dma_wait_all();
    __syncthreads();
  };
#undef SAGE_FENCE
  int j = 0;
  for (; j + 1 < n_fast; j += 2) {
    fast_iter(std::integral_constant<int, 0>{}, j, s_cur, s_nxt, sc_cur, sc_nxt);
    fast_iter(std::integral_constant<int, 1>{}, j + 1, s_nxt, s_cur, sc_nxt, sc_cur);
  }
  if (j > 0 && j < wave_tiles) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) bits_cur[qt] = allow_bits(j, qt);
  }
  for (; j < wave_tiles; ++j) {
    maybe_rescale(mx_cur);
    if (j + 2 < ntiles) dma_k(j + 2, j & 1);
    if (j + 1 < ntiles) dma_v(j + 1, (j + 1) & 1);
    const bool has_next = j + 1 < wave_tiles;
    if (has_next) {
      scales_from(load_kscales(j + 1), sc_nxt);
      qk((j + 1) & 1, s_nxt);
      drain_s(s_nxt);  // S MFMA results -> VALU
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        bits_nxt[qt] = allow_bits(j + 1, qt);
        mask_scores(bits_nxt[qt], s_nxt[qt]);
      }
    }
    softmax_pv(j & 1, s_cur, sc_cur);
    if (has_next) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) mx_cur[qt] = row_max(s_nxt[qt], sc_nxt[qt][0], sc_nxt[qt][1]);
    }
    dma_wait_all();
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      bits_cur[qt] = bits_nxt[qt];
      s_cur[qt][0] = s_nxt[qt][0]; s_cur[qt][1] = s_nxt[qt][1];
      sc_cur[qt][0] = sc_nxt[qt][0]; sc_cur[qt][1] = sc_nxt[qt][1];
    }
  }
  for (; j < ntiles; ++j) {
    if (j + 2 < ntiles) dma_k(j + 2, j & 1);
    if (j + 1 < ntiles) dma_v(j + 1, (j + 1) & 1);
    dma_wait_all();
    __syncthreads();
  }

  // ---- epilogue
  drain_o(acc_o);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l_tot = swap_sum(l_run[qt]);
    const float inv = 1.0f / l_tot;
    if (row[qt] < M_) {
      uint16_t* op = p.o + b * p.osb + h * p.osh + (int64_t)row[qt] * p.osn;
      const float* vmp = p.v_mean ? p.v_mean + ((int64_t)b * p.Hk + hk) * D : nullptr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dt + 8 * g4 + 4 * hh;
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = acc_o[qt][dt][4 * g4 + e] * inv;
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
          *reinterpret_cast<uint2*>(op + d0) = w;
        }
      if (p.lse && hh == 0) p.lse[((int64_t)b * p.Hq + h) * M_ + row[qt]] = m_run[qt] + log2f(l_tot);
    }
  }
}

int launch_attn_w64(const AttnParams& p_in, int D, bool causal, bool kthread, bool pv_fp8, hipStream_t st) {
  if (D != 128 || pv_fp8 || p_in.cu_q || p_in.mask || p_in.q_f16) return SAGE_ERR_UNSUPPORTED;
  AttnParams p = p_in;
  p.nqb = (p.M + 255) / 256;
  const size_t smem = 2 * 64 * D + 2 * 64 * D * 2;
  const dim3 grid(p.nqb * p.Hq * p.B), block(256);
#define SAGE_W64(C, K)                                                                                              \
  do {                                                                                                              \
    auto kern = attn_i8_w64_kernel<128, C, K>;                                                                      \
    launch_begin();                                                                                              \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return SAGE_ERR_LAUNCH; \
    hipLaunchKernelGGL(kern, grid, block, smem, st, p);                                                             \
  } while (0)
  if (causal) { if (kthread) SAGE_W64(true, true); else SAGE_W64(true, false); }
  else { if (kthread) SAGE_W64(false, true); else SAGE_W64(false, false); }
#undef SAGE_W64
  return launch_status();
}

}  // namespace sage
