// Power-limited MFMA throughput by instruction shape (all CUs busy, random operands in registers, wall clock):
// which shapes should the attention kernel use when the chip, not the schedule, sets the clock?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_power.hip -o ab_libs/mfma_power && ab_libs/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define REP 4096

template <int KIND>
__global__ __launch_bounds__(512) void k_loop(const int* __restrict__ src, float* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  v4i a4[2], b4[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 4; ++e) { a4[i][e] = src[(t * 16 + i * 4 + e) & 0xfffff]; b4[i][e] = src[(t * 16 + 8 + i * 4 + e) & 0xfffff]; }
  float r = 0.f;
  if (KIND == 0) {  // i8 32x32x32 (asm: integer accumulation is associative and hipcc folds a builtin loop)
    v16i acc[2] = {};
    for (int i = 0; i < REP; ++i) {
      asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a4[0]), "v"(b4[0]));
      asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a4[1]), "v"(b4[1]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    r = (float)(acc[0][0] + acc[1][3]);
  } else if (KIND == 1) {  // i8 16x16x64: half the work per instruction -> twice the instructions
    v4i acc[4] = {};
    for (int i = 0; i < REP; ++i) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a4[0]), "v"(b4[0]));
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a4[1]), "v"(b4[1]));
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(a4[0]), "v"(b4[1]));
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(a4[1]), "v"(b4[0]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    r = (float)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
  } else if (KIND == 2) {  // f16 32x32x16
    v16f acc[2] = {};
    const v8h a0 = __builtin_bit_cast(v8h, a4[0]), a1 = __builtin_bit_cast(v8h, a4[1]);
    const v8h b0 = __builtin_bit_cast(v8h, b4[0]), b1 = __builtin_bit_cast(v8h, b4[1]);
    for (int i = 0; i < REP; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1], 0, 0, 0);
    }
    r = acc[0][0] + acc[1][3];
  } else {  // f16 16x16x32
    v4f acc[4] = {};
    const v8h a0 = __builtin_bit_cast(v8h, a4[0]), a1 = __builtin_bit_cast(v8h, a4[1]);
    const v8h b0 = __builtin_bit_cast(v8h, b4[0]), b1 = __builtin_bit_cast(v8h, b4[1]);
    for (int i = 0; i < REP; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[3], 0, 0, 0);
    }
    r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  }
  if (r == 12345.678f) out[t] = r;
}

template <int KIND>
static void run(const char* name, const int* d_src, float* d_out, double flop_per_wave_iter) {
  const int blocks = 256 * 4, threads = 512;  // 2 waves per SIMD, 4 workgroups queued per CU
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_loop<KIND>, dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int launches = 8;
  for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(k_loop<KIND>, dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * REP * flop_per_wave_iter * launches;
  printf("%-22s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';
  std::vector<int> h(1 << 20);
  srand(1);
  for (auto& x : h) x = zeros ? 0 : ((rand() & 0xffff) | (rand() << 16));
  if (!zeros)  // keep fp16 bit patterns finite: clear the top exponent bit of each half
    for (auto& x : h) x &= 0xbfffbfff;
  int* d_src; float* d_out;
  (void)hipMalloc(&d_src, h.size() * 4);
  (void)hipMalloc(&d_out, 256 * 4 * 512 * 4);
  (void)hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("operands: %s\n", zeros ? "zeros" : "random");
  run<0>("i8  32x32x32 (x2)", d_src, d_out, 2.0 * 2 * 32 * 32 * 32);
  run<1>("i8  16x16x64 (x4)", d_src, d_out, 4.0 * 2 * 16 * 16 * 64);
  run<2>("f16 32x32x16 (x2)", d_src, d_out, 2.0 * 2 * 32 * 32 * 16);
  run<3>("f16 16x16x32 (x4)", d_src, d_out, 4.0 * 2 * 16 * 16 * 32);
  return 0;
}
