// Device helpers and the kernel parameter block of the attention kernel (sage_attn.hip).
#pragma once
#include <type_traits>
#include "sage_common.h"

namespace sage {

struct AttnParams {
  const int8_t* q; int64_t qsb, qsh, qsn;
  const int8_t* k; int64_t ksb, ksh, ksn;
  const uint8_t* v; int64_t vsb, vsh, vsn;  // byte pointer; strides in ELEMENTS of v's dtype
  uint16_t* o; int64_t osb, osh, osn;
  const float* q_scale; const float* k_scale; const float* v_scale; const float* v_mean;
  float* lse;
  int B, Hq, Hk, M, N;
  int nqb;      // query blocks per (b,h)
  int gq, gk;   // scales per (b,h)
  int qgran, blkq, warpq;
  float logit_mult;
  int out_bf16;
  // varlen (packed sequences, core.py:363-477): when cu_q/cu_k are set, "batch" b is sequence b, its rows are
  // [cu[b], cu[b+1]) of the packed [total, H, D] tensors (stride_b unused) and p.M / p.N are the maximum lengths
  const int* cu_q;
  const int* cu_k;
  // fused Q quantizer: when q_f16 is set, q/q_scale are ignored and every wave quantizes its own 32 query rows in the
  // prologue (per_warp: CUDA numerics, per_thread: Triton numerics -- the pairings of core.py:621-624); km (optional,
  // [B,Hk,D] in q's dtype) yields the LSE correction q.km and lse then receives the FINAL natural-log LSE (core.py:651)
  const uint16_t* q_f16;
  const uint16_t* km;
  int q_bf16;
  float sm_scale;
  // attn_mask of sageattn_qk_int8_pv_fp16_triton (core.py:306-318; kernels attn_qk_int8_per_block.py:33-52):
  // [B,H,M,N] view with element strides (0 = broadcast); kind 1 = bool (False -> -1e6), 2 = fp16, 3 = bf16 (added to
  // the base-2 logits, exactly as the reference adds it after its sm_scale*log2e scaling)
  const uint8_t* mask;
  int64_t msb, msh, msm, msn;
  int mask_kind;
  // KV tile layout (sage_kv_layout, include/sageattn_hip.h): byte distance between consecutive 64-key tiles of one
  // (b, h_kv) in k8 and in v (always set by run_attn; the dense defaults are 64 rows), and the strides of k_scale
  // (floats) per batch, kv head and 64-key tile.  Sequence-parallel exchange buffers are tile-major: tile j of every
  // (b, h) is a contiguous block and the tiles of all ranks form one sequence.
  int k_tile_bytes, v_tile_bytes;
  int64_t ks_b, ks_h;
  int ks_t;
  int kv_tiled;  // non-default tile strides
  int o_vec16;   // every output row starts on a 16-byte boundary (all strides multiples of 8 elements): 16-byte stores
};

// v_max_f32 on values that are never signalling NaNs: fmaxf() makes hipcc canonicalise both operands first
// (v_max_f32 x, x, x), two extra instructions on a kernel bound by the vector issue port
__device__ __forceinline__ float max_raw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a > b ? a : b;
#endif
}
__device__ __forceinline__ float swap_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 16 bytes per lane, buffer -> LDS without passing through VGPRs (buffer_load_dwordx4 ... offen lds).
// Issued through inline asm ON PURPOSE: hipcc would otherwise treat the copy as a store that may alias every later
// LDS read and drain it with s_waitcnt vmcnt(0) a few instructions after issue.  Here nothing waits for it until
// dma_wait_all() in front of the workgroup barrier that publishes the tile, a whole iteration later.
// `lds_off` = byte offset of the wave's 1 KiB destination inside the workgroup's LDS (wave-uniform; lane l lands at
// +16*l), `rsrc` = buffer descriptor (4 uniform dwords), `voffset` per lane, `soffset` uniform.
__device__ __forceinline__ void lds_dma16(v4i rsrc, unsigned lds_off, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  // M0 (the LDS base of the copy) is declared clobbered instead of saved and restored around the instruction: two scalar
  // moves less per copy (+0.2 ... +2 %, most on the short head_dim-64 tiles)
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voffset), "s"(rsrc), "s"(lds_off), "s"(soffset)
      : "memory", "m0");
#endif
}
// counted form: wait until at most N of the wave's vector-memory operations are outstanding (they complete in order)
template <int N>
__device__ __forceinline__ void dma_wait_keep() {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void dma_wait_all() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// raw buffer descriptor as 4 provably wave-uniform dwords (gfx950: word3 = 0x00020000)
__device__ __forceinline__ v4i make_rsrc(const void* base, unsigned num_bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)num_bytes);
  r[3] = 0x00020000;
  return r;
}

// Loads through the scalar cache (s_load, lgkmcnt) for wave-uniform addresses of read-only data: the constant
// address space cast is what lets hipcc pick SMEM; hidden from the host pass.
__device__ __forceinline__ float4 uniform_load4(const float* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const v4f v = *(const __attribute__((address_space(4))) v4f*)(ptr);
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ float uniform_load1(const float* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(4))) float*)(ptr);
#else
  return 0.f;
#endif
}

template <int D>
__device__ __forceinline__ int k_swz(int row) {
  // 16-B chunk XOR that makes the ds_read_b128 A-fragment reads conflict free
  if constexpr (D == 128) return (row >> 1) & 7; else return (row >> 2) & 3;
}
template <int D>
__device__ __forceinline__ int v_win_swz(int row) {
  // 64-B window XOR for the fp16 V tile so that the 4 rows of a tr-read land on 4 windows
  if constexpr (D == 128) return row & 3; else return (row >> 1) & 1;
}

}  // namespace sage
