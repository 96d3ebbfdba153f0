// Probe of v_mfma_f32_4x4x1_16b_f32's operand / result layout (tools/probe; not part of libarx).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  // A value encodes (lane), B value encodes (lane): a = 1 + lane, b = 100 + lane (asymmetric)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(100 + lane), acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[lane * 4 + v] = acc[v];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: lane l = 4*blk + j; D[blk][i = v][j] = A[blk][i] * B[blk][j] with A lane 4*blk+i, B lane 4*blk+j
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const int blk = l / 4, j = l % 4;
    const float exp = (float)(1 + 4 * blk + v) * (float)(100 + 4 * blk + j);
    if (h[l * 4 + v] != exp) { if (bad < 8) printf("lane %d v %d got %g exp %g\n", l, v, h[l*4+v], exp); ++bad; }
  }
  printf("mismatches: %d\n", bad);
  printf("lane0: %g %g %g %g ; lane1: %g %g %g %g ; lane5: %g %g %g %g\n", h[0],h[1],h[2],h[3],h[4],h[5],h[6],h[7],h[20],h[21],h[22],h[23]);
  return 0;
}
