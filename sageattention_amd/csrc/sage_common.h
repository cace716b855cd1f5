// Internal helpers shared by the gfx950 kernels.  Not part of the C ABI (include/sageattn_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sageattn_hip.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "this library is written for gfx950 (MI355X / CDNA4) only"
#endif

namespace sage {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v4s_vs __attribute__((__vector_size__(8)));  // operand type of ds_read_tr16_b64

constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) {
  return (float)__builtin_bit_cast(_Float16, b);
}
template <bool BF16>
__device__ __forceinline__ float elem_to_f32(uint16_t b) {
  if constexpr (BF16) return bf16_bits_to_f32(b); else return f16_bits_to_f32(b);
}
// round-to-nearest-even to the storage dtype, result as fp32 again
template <bool BF16>
__device__ __forceinline__ float round_to_elem(float x) {
  if constexpr (BF16) return (float)(__bf16)x; else return (float)(_Float16)x;
}
template <bool BF16>
__device__ __forceinline__ uint16_t f32_to_elem_bits(float x) {
  if constexpr (BF16) return __builtin_bit_cast(uint16_t, (__bf16)x);
  else return __builtin_bit_cast(uint16_t, (_Float16)x);
}

// unpack 8 fp16/bf16 held in a uint4 into 8 floats
template <bool BF16>
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = elem_to_f32<BF16>((uint16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = elem_to_f32<BF16>((uint16_t)(w[i] >> 16));
  }
}

// Four int32 already clamped to [-128, 127] -> their low bytes in one dword: 3 v_perm_b32 instead of 4 masks + 3 shift-ors.
// v_perm_b32 D, S0, S1, sel: selector bytes 0-3 pick bytes of S1, 4-7 bytes of S0, 0x0c a zero byte.
__device__ __forceinline__ uint32_t pack_i8x4(int a, int b, int c, int d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);
  const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x0c0c0400u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
#else
  return ((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8) | (((uint32_t)c & 0xffu) << 16) | (((uint32_t)d & 0xffu) << 24);
#endif
}

// Half-away-from-zero rounding of y = x / scale the cheap way (the Triton quantizers, quant_per_block.py:42-44): q = rint(y)
// agrees with trunc(y + 0.5 sign y) unless y sits on a rounding boundary; `near` reports |y - rint(y)| > 0.5 - 2^-14, the
// band in which (a) the two roundings or (b) the reciprocal product and the IEEE division the reference uses could
// differ (|x*r - x/sc| <= 2.3e-5 for |y| <= 127, plus 7.6e-6 from the reference's rounded y + 0.5): the caller then redoes
// the chunk with the exact division.  4 VALU per element (v_rndne, v_sub, v_cmp, v_cvt) against 7 for fract-based forms.
__device__ __forceinline__ int round_half_away_fast(float y, bool& near) {
  const float rn = rintf(y);
  near |= fabsf(y - rn) > 0.5f - 6.1035156e-5f;
  return (int)rn;
}

// hipGetLastError() is sticky per host thread: every entry point clears it before its first launch (launch_begin), so
// that launch_status() reports THIS call's launches and not a stale error of an earlier, unrelated runtime call.
__host__ inline void launch_begin() { (void)hipGetLastError(); }
__host__ inline int launch_status() {
  return hipGetLastError() == hipSuccess ? SAGE_OK : SAGE_ERR_LAUNCH;
}

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace sage
