// Probe: achievable fp32-MFMA rate on gfx950 (tools/probe; not part of libarx).  Every wave issues
// back-to-back v_mfma_f32_32x32x2_f32 with NACC independent accumulators, no memory traffic; 1, 2 or
// 4 waves per SIMD.  Prints TFLOP/s per configuration -- the ceiling the scorer GEMMs are priced
// against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.f) out[0] = s;
}
template <int NACC>
static void run(int wgs_per_cu, int cus, int threads) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * (threads / 64) * iters * 8.0 * NACC * 4096.0;
  printf("NACC=%d wg/CU=%d threads=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, threads, ms, flop / ms / 1e9);
  hipFree(d);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("CUs %d clock %d kHz\n", cus, p.clockRate);
  run<1>(1, cus, 256); run<2>(1, cus, 256); run<4>(1, cus, 256);
  run<2>(2, cus, 256); run<4>(2, cus, 256); run<2>(1, cus, 512);
  // short launch (like one 50 us GEMM): 40 iterations
  return 0;
}
