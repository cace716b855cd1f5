// Issue-cost microbenchmark for gfx950 vector instructions, alone and as fillers between MFMAs.
//   hipcc --offload-arch=gfx950 -O2 tools/issue_cost.hip -o gpurun_out/issue_cost && gpurun_out/issue_cost
// One workgroup per CU-sized launch; WAVES_PER_SIMD = 1 or 2.  Each case runs REP iterations of a block of independent
// instructions; cycles = s_memtime delta (shader clock) / instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define REP 2000

#define X8(op)                                                                                          \
  op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define X16(op) X8(op) X8(op)

// each OP_* is one instruction writing %[rN]
#define CASE_BODY(NAME, ASMSTR8)                                                                                  \
  __global__ void NAME(long long* out, float seed) {                                                             \
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6,    \
          r7 = seed + 7;                                                                                          \
    float a = seed * 0.5f, b = seed * 0.25f;                                                                      \
    long long t0 = clock64();                                                                                     \
    for (int i = 0; i < REP; ++i) {                                                                               \
      asm volatile(ASMSTR8 ASMSTR8 ASMSTR8 ASMSTR8                                                                \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)               \
                   : "v"(a), "v"(b));                                                                             \
    }                                                                                                             \
    long long t1 = clock64();                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                    \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1] = 1;                                          \
  }

#define I8(fmt_pre, fmt_post)                                                                                     \
  fmt_pre "%0" fmt_post "\n" fmt_pre "%1" fmt_post "\n" fmt_pre "%2" fmt_post "\n" fmt_pre "%3" fmt_post "\n"     \
  fmt_pre "%4" fmt_post "\n" fmt_pre "%5" fmt_post "\n" fmt_pre "%6" fmt_post "\n" fmt_pre "%7" fmt_post "\n"

CASE_BODY(k_add, I8("v_add_f32 ", ", %8, %9"))
CASE_BODY(k_fma, I8("v_fma_f32 ", ", %8, %9, %9"))
CASE_BODY(k_exp, I8("v_exp_f32 ", ", %8"))
CASE_BODY(k_dot2, I8("v_dot2_f32_f16 ", ", %8, %9, %9"))
CASE_BODY(k_dot2c, I8("v_dot2c_f32_f16 ", ", %8, %9"))
CASE_BODY(k_pkaddh, I8("v_pk_add_f16 ", ", %8, %9"))
CASE_BODY(k_max3, I8("v_max3_i32 ", ", %8, %9, %9"))
CASE_BODY(k_cvtpk, I8("v_cvt_pk_f16_f32 ", ", %8, %9"))
CASE_BODY(k_mov, I8("v_mov_b32 ", ", %8"))
CASE_BODY(k_expf16, I8("v_exp_f16 ", ", %8"))
CASE_BODY(k_pkmaxi16, I8("v_pk_max_i16 ", ", %8, %9"))
CASE_BODY(k_pkmulh, I8("v_pk_mul_f16 ", ", %8, %9"))

// MFMA + fillers: 1 MFMA (32x32x16 f16, 4 independent accumulators round robin) followed by NF fillers
template <int KIND, int NF>
__global__ void k_mfma_fill(long long* out, float seed) {
  v16f acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = seed;
  v8h a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed; b[e] = (_Float16)(seed + e); }
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = seed + i;
  float x = seed * 0.5f, y = seed * 0.25f;
  long long t0 = clock64();
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "+v"(r[f & 7]) : "v"(x), "v"(y));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "+v"(r[f & 7]) : "v"(x));
        if (KIND == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r[f & 7]) : "v"(x), "v"(y));
        if (KIND == 3) asm volatile("v_pk_add_f16 %0, %1, %2" : "+v"(r[f & 7]) : "v"(x), "v"(y));
        if (KIND == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r[f & 7]) : "v"(x), "v"(y));
        if (KIND == 5) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(r[f & 7]) : "v"(x), "v"(y));
      }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += r[i];
  if (s == 12345.678f) out[1] = 1;
}

// MFMA + NV v_add_f32 + NK ds_read_b128 + NT ds_read_b64_tr_b16 per gap (reads are never waited for inside the loop)
template <int NV, int NK, int NT>
__global__ void k_mfma_lds(long long* out, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed;
  __syncthreads();
  v16f acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = seed;
  v8h a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed; b[e] = (_Float16)(seed + e); }
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = seed + i;
  float x = seed * 0.5f, y = seed * 0.25f;
  const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;  // conflict-free linear image
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  v4i_t kk[3];
  v2i_t tt[3];
  long long t0 = clock64();
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NK; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kk[f]) : "v"(addr), "n"(f * 8192));
#pragma unroll
      for (int f = 0; f < NT; ++f) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tt[f]) : "v"(addr), "n"(f * 8192 + 32768));
#pragma unroll
      for (int f = 0; f < NV; ++f) asm volatile("v_add_f32 %0, %1, %2" : "+v"(r[f & 7]) : "v"(x), "v"(y));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += r[i];
  if (s == 12345.678f) out[1] = 1;
}

template <typename F>
static double run(F kern, int threads, long long* d_out, double per) {
  long long h = 0;
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_out, 1.0f);
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_out, 1.0f);
  hipDeviceSynchronize();
  hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  return (double)h / per;
}

int main() {
  long long* d_out;
  hipMalloc(&d_out, 16);
  hipMemset(d_out, 0, 16);
  // s_memtime ticks at a fixed 100 MHz on this part? print both waves/SIMD settings; compare RATIOS to v_add_f32
  struct C { const char* name; void (*k)(long long*, float); };
  C cases[] = {{"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_exp_f32", k_exp}, {"v_dot2_f32_f16", k_dot2},
               {"v_dot2c_f32_f16", k_dot2c}, {"v_pk_add_f16", k_pkaddh}, {"v_max3_i32", k_max3},
               {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_mov_b32", k_mov}, {"v_exp_f16", k_expf16},
               {"v_pk_max_i16", k_pkmaxi16}, {"v_pk_mul_f16", k_pkmulh}};
  for (int wps = 1; wps <= 2; ++wps) {
    printf("== %d wave(s) per SIMD: ticks per instruction per wave (ratio to v_add_f32)\n", wps);
    double base = 0;
    for (auto& c : cases) {
      double t = run(c.k, 256 * wps, d_out, (double)REP * 32);
      if (base == 0) base = t;
      printf("  %-20s %8.3f  x%.2f\n", c.name, t, t / base);
    }
  }
#define MF(KIND, NF, NAME)                                                                  \
  {                                                                                         \
    double t1 = run(k_mfma_fill<KIND, NF>, 256, d_out, (double)REP * 4);                    \
    double t2 = run(k_mfma_fill<KIND, NF>, 512, d_out, (double)REP * 4);                    \
    printf("  %-16s fillers=%d  ticks/MFMA: 1 wave %8.3f   2 waves %8.3f\n", NAME, NF, t1, t2); \
  }
  printf("== one v_mfma_f32_32x32x16_f16 followed by N fillers (ticks per MFMA per wave)\n");
  MF(0, 0, "none")
  MF(0, 2, "v_add_f32") MF(0, 4, "v_add_f32") MF(0, 6, "v_add_f32") MF(0, 8, "v_add_f32")
  MF(1, 2, "v_exp_f32") MF(1, 4, "v_exp_f32")
  MF(2, 2, "v_dot2_f32_f16") MF(2, 4, "v_dot2_f32_f16") MF(2, 6, "v_dot2_f32_f16") MF(2, 8, "v_dot2_f32_f16")
  MF(3, 4, "v_pk_add_f16") MF(3, 8, "v_pk_add_f16")
  MF(4, 4, "v_dot2c_f32_f16") MF(4, 8, "v_dot2c_f32_f16")
  MF(5, 4, "v_max3_i32") MF(5, 8, "v_max3_i32")
#define ML(NV, NK, NT)                                                                        \
  {                                                                                           \
    double t1 = run(k_mfma_lds<NV, NK, NT>, 256, d_out, (double)REP * 4);                     \
    double t2 = run(k_mfma_lds<NV, NK, NT>, 512, d_out, (double)REP * 4);                     \
    printf("  v_add=%d b128=%d tr_b64=%d  ticks/MFMA: 1 wave %8.3f   2 waves %8.3f\n", NV, NK, NT, t1, t2); \
  }
  printf("== one MFMA + NV v_add_f32 + NK ds_read_b128 + NT ds_read_b64_tr_b16 per gap\n");
  ML(0, 0, 0) ML(0, 1, 0) ML(0, 2, 0) ML(0, 0, 2) ML(0, 0, 3)
  ML(4, 0, 0) ML(4, 1, 0) ML(4, 2, 0) ML(4, 0, 2) ML(4, 0, 3) ML(4, 1, 2)
  ML(6, 0, 0) ML(6, 1, 0) ML(6, 0, 2) ML(6, 1, 2)
  return 0;
}
