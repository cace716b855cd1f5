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

// hipGetLastError() is sticky per host thread: every entry point clears it before its first launch (launch_begin), so
// that launch_status() reports THIS call's launches and not a stale error of an earlier, unrelated runtime call.
__host__ inline void launch_begin() { (void)hipGetLastError(); }
__host__ inline int launch_status() {
  return hipGetLastError() == hipSuccess ? SAGE_OK : SAGE_ERR_LAUNCH;
}

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace sage
