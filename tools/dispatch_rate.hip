// How fast does the chip START workgroups?  Empty kernels (one predicated-off store) of G workgroups x T threads, with and
// without an LDS allocation and a register footprint: microseconds per launch (back-to-back launches, HIP events).
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_rate.hip -o gpurun_out/dispatch_rate && gpurun_out/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VGPRS>
__global__ void empty_kernel(float* out, int never) {
  if constexpr (VGPRS > 0) {
    float acc[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) acc[i] = (float)(threadIdx.x + i);
    if (never) {
#pragma unroll
      for (int i = 0; i < VGPRS; ++i) out[threadIdx.x * VGPRS + i] = acc[i];
    }
  } else {
    if (never) out[threadIdx.x] = 1.f;
  }
}

template <int VGPRS>
__global__ void lds_kernel(float* out, int never) {
  extern __shared__ float sm[];
  if (never) { sm[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = sm[threadIdx.x ^ 1]; }
}

template <typename F>
static double time_us(F launch, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / n;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  const int Ts[] = {64, 256, 512, 1024};
  const int Gs[] = {1, 256, 512, 1024, 2048, 4096, 8192, 16384};
  printf("empty kernel, us per launch (back to back)\n%8s", "G \\ T");
  for (int T : Ts) printf("%10d", T);
  printf("\n");
  for (int G : Gs) {
    printf("%8d", G);
    for (int T : Ts) printf("%10.2f", time_us([&] { hipLaunchKernelGGL(empty_kernel<0>, dim3(G), dim3(T), 0, 0, d, 0); }, 200));
    printf("\n");
  }
  printf("same with 64 KB of LDS per workgroup (at most 2 resident per CU)\n");
  hipFuncSetAttribute((const void*)lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int G : Gs) {
    printf("%8d", G);
    for (int T : Ts) printf("%10.2f", time_us([&] { hipLaunchKernelGGL(lds_kernel<0>, dim3(G), dim3(T), 65536, 0, d, 0); }, 200));
    printf("\n");
  }
  return 0;
}
