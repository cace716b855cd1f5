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


// Attention-like instruction mix per loop iteration (1/8 of a 32x64 wave-tile of the kernel): MFMAs of the given shape
// family + 18 VALU (4 v_exp_f32, 4 v_fma_f32, 4 v_add_f32, 2 v_cvt_pk_f16_f32, 2 v_max3_i32, 2 v_mul_f32) + 5 LDS reads
// (1 ds_read_b128, 4 ds_read_b64_tr_b16), random data.  SMALL = false: 1 x i8 32x32x32 + 2 x f16 32x32x16;
// SMALL = true: 2 x i8 16x16x64 + 4 x f16 16x16x32 (same flops).  What matters is the wall-clock ratio of the two.
template <bool SMALL, int NTR = 4, int NB128 = 1, int NEXP = 4>
__global__ __launch_bounds__(512) void k_mix(const int* __restrict__ src, float* out) {
  __shared__ __attribute__((aligned(16))) int lds[8192];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(t + i * 7) & 0xfffff];
  __syncthreads();
  v4i a4[2], b4[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 4; ++e) { a4[i][e] = src[(t * 16 + i * 4 + e) & 0xfffff]; b4[i][e] = src[(t * 16 + 8 + i * 4 + e) & 0xfffff]; }
  v8h h0 = __builtin_bit_cast(v8h, a4[0]), h1 = __builtin_bit_cast(v8h, b4[1]);
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -0.001f * (float)((src[(t + i) & 0xfffff] & 0xffff) + 1);
  float sc = 0.999f, cc = -0.5f, sum = 0.f;
  int mx = 0;
  const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  v16i si = {}; v16f of[2] = {};
  v4i s4[2] = {}; v4f o4[4] = {};
  v4i kk; typedef int v2i_t __attribute__((ext_vector_type(2))); v2i_t tt[4];
  for (int it = 0; it < REP; ++it) {
    if (NB128 > 0) asm volatile("ds_read_b128 %0, %1" : "=v"(kk) : "v"(addr));
    if (NTR > 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(tt[0]) : "v"(addr));
    if (NTR > 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(tt[1]) : "v"(addr));
    if (SMALL) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(s4[0]) : "v"(a4[0]), "v"(b4[0]));
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(s4[1]) : "v"(a4[1]), "v"(b4[0]));
    } else {
      asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(si) : "v"(a4[0]), "v"(b4[0]));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e0, e1; int pk;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[2 * u]), "v"(sc), "v"(cc));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[2 * u + 1]), "v"(sc), "v"(cc));
      if (NEXP > 2 * u) asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
      if (NEXP > 2 * u + 1) asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(e0));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(e1));
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(e0), "v"(e1));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(mx) : "v"(a4[u][0]), "v"(b4[u][1]));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[4 + u]) : "v"(sc));
      if (SMALL) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(o4[2 * u]) : "v"(h0), "v"(h1));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(o4[2 * u + 1]) : "v"(h1), "v"(h0));
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(of[u]) : "v"(h0), "v"(h1));
      }
      if (u == 0) {
        if (NTR > 2) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:24576" : "=v"(tt[2]) : "v"(addr));
        if (NTR > 3) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:28672" : "=v"(tt[3]) : "v"(addr));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  for (int i = 0; i < 4; ++i) if (i >= NTR) { tt[i][0] = 0; }
  if (NB128 == 0) kk[0] = 0;
  float r = sum + (float)mx + x[4] + x[5] + (float)si[0] + of[0][0] + of[1][1] + (float)s4[0][0] + (float)s4[1][1] + o4[0][0] + o4[1][0] +
            o4[2][0] + o4[3][0] + (float)kk[0] + (float)tt[0][0] + (float)tt[1][0] + (float)tt[2][0] + (float)tt[3][0];
  if (r == 12345.678f) out[t] = r;
}

template <bool SMALL, int NTR = 4, int NB128 = 1, int NEXP = 4>
static void run_mix(const char* name, const int* d_src, float* d_out) {
  const int blocks = 256 * 4, threads = 512;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mix<SMALL, NTR, NB128, NEXP>), dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int launches = 8;
  for (int w = 0; w < launches; ++w) hipLaunchKernelGGL((k_mix<SMALL, NTR, NB128, NEXP>), dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * REP * (2.0 * 32 * 32 * 32 + 2 * 2.0 * 32 * 32 * 16) * launches;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}


// FP8-PV instruction mix (1/4 of a 32x64 wave-tile): 2 x i8 32x32x32 + 1 x MX-scaled fp8 32x32x64 (unit scales) +
// 36 VALU (8 v_exp_f32, 8 v_fma_f32, 8 v_add_f32, 4 v_cvt_pk_fp8_f32, 4 v_max3_i32, 4 v_mul_f32) + 4 ds_read_b128.
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k_mix8(const int* __restrict__ src, float* out) {
  __shared__ __attribute__((aligned(16))) int lds[8192];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(t + i * 7) & 0xfffff];
  __syncthreads();
  v4i a4[2], b4[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 4; ++e) { a4[i][e] = src[(t * 16 + i * 4 + e) & 0xfffff]; b4[i][e] = src[(t * 16 + 8 + i * 4 + e) & 0xfffff]; }
  v8i f8a, f8b;
  for (int e = 0; e < 8; ++e) { f8a[e] = src[(t * 8 + e) & 0xfffff] & 0x77777777; f8b[e] = src[(t * 8 + e + 64) & 0xfffff] & 0x77777777; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -0.001f * (float)((src[(t + i) & 0xfffff] & 0xffff) + 1);
  float sc = 0.999f, cc = -0.5f, sum = 0.f;
  int mx = 0;
  const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  v16i si[2] = {}; v16f of = {};
  v4i kk[4];
  for (int it = 0; it < REP; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kk[u]) : "v"(addr), "n"(u * 8192));
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(si[0]) : "v"(a4[0]), "v"(b4[0]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float e0, e1; int pk = 0;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[2 * (u & 1)]), "v"(sc), "v"(cc));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[2 * (u & 1) + 1]), "v"(sc), "v"(cc));
      asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
      asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(e0));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(e1));
      asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(pk) : "v"(e0), "v"(e1));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(mx) : "v"(a4[u & 1][0]), "v"(pk));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[4 + (u & 1)]) : "v"(sc));
      if (u == 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(si[1]) : "v"(a4[1]), "v"(b4[1]));
    }
    of = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f8a, f8b, of, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float r = sum + (float)mx + x[4] + x[5] + (float)si[0][0] + (float)si[1][1] + of[0] + (float)(kk[0][0] + kk[1][0] + kk[2][0] + kk[3][0]);
  if (r == 12345.678f) out[t] = r;
}
static void run_mix8(const int* d_src, float* d_out) {
  const int blocks = 256 * 4, threads = 512;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_mix8, dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int launches = 8;
  for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(k_mix8, dim3(blocks), dim3(threads), 0, 0, d_src, d_out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * REP * (2 * 2.0 * 32 * 32 * 32 + 2.0 * 32 * 32 * 64) * launches;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", "FP8-PV mix (i8 + MX fp8 32x32x64)", ms, flops / (ms * 1e-3) / 1e12);
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
  run_mix<false>("attention-like mix, 32x32 MFMAs", d_src, d_out);
  run_mix<true>("attention-like mix, 16x16 MFMAs", d_src, d_out);
  run_mix8(d_src, d_out);
  run_mix<false, 2, 1, 4>("mix 32x32, half the V^T fragment reads", d_src, d_out);
  run_mix<false, 0, 0, 4>("mix 32x32, no LDS reads", d_src, d_out);
  run_mix<false, 4, 1, 2>("mix 32x32, half the v_exp_f32", d_src, d_out);
  run_mix<false, 4, 1, 0>("mix 32x32, no v_exp_f32", d_src, d_out);
  return 0;
}
