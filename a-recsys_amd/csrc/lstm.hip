// lstm.hip -- K9: LSTM sequence encoder (lstm/seqModel.py:99-103,477).
//
// tf.contrib.rnn LSTMCell(h): z = [x_t, h_{t-1}] . W + b, split i,j,f,o;
// c = sigmoid(f + forget_bias)*c_prev + sigmoid(i)*tanh(j); h = sigmoid(o)*tanh(c);
// zero initial state; static_rnn runs all L steps (padding is masked by the
// loss weights only).
//
// The recurrence is latency-bound (L dependent [B,(d+h)]x[(d+h),4h] products),
// so ONE persistent launch walks all L steps: a workgroup owns 16 batch rows
// (the M of v_mfma_f32_16x16x4_f32), its 4 waves own h/4 units each -- all four
// gates of those units, so the cell update is register-local; h_{t-1} lives in
// LDS (double-buffered, stride h+2 => conflict-free operand reads), c_{t-1} in
// registers, x_{t+1} is prefetched into LDS while step t computes.  W streams
// from L2 as the MFMA B operand (it is read-only and tiny: 128 KB at d=h=64).
#include <stdlib.h>

#include "common.h"

namespace arx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate non-linearities on the transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each), no libm call and no IEEE
// division: ~6 / ~14 instructions instead of ~15 / ~45.  tanh: (1 - e) / (1 + e) with e = exp(-2|x|) loses the
// small-|x| digits to 1 - e, so |x| < 1/16 takes the odd series (relative error < 1e-10 there); both within ~5e-7
// relative of tanh over the whole range (tests/test_kernels_gpu.py::test_lstm_fwd_bwd compares the kernels' gates,
// cells and outputs with the float64 oracle).  FORWARD kernels only: k_lstm_fwd_r4 83.8 -> 75.2 us at L = 50,
// B = 1024; in the backward kernels the same replacement measured SLOWER (k_lstm_bwd_r4 46.8 -> 56.8 us), they keep
// libm's tanhf.  Also measured and dropped: the x half of step t + 1 issued between the gate arithmetic of step t
// (82.5 us compiler-scheduled, 93 us with a sched_group_barrier interleave that needs > 256 registers).
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);
  const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  const float x2 = ax * ax;
  const float small = ax * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  return copysignf(ax < 0.0625f ? small : big, x);
}

constexpr int kRows = 16;  // batch rows per workgroup (MFMA M)

// TPG = 16-unit tiles per gate per wave = h / 64
template <int TPG>
__global__ __launch_bounds__(256) void k_lstm_fwd(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
    int64_t L, int64_t B, int din, int h, float forget_bias, float* __restrict__ hs,
    float* __restrict__ cs, float* __restrict__ gates) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SX = din + 2, SH = h + 2;
  float* xbuf = smem;                         // [2][16][SX]
  float* hbuf = smem + 2 * kRows * SX;        // [2][16][SH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int UPW = h / 4;                      // units per wave
  const int ubase = wave * UPW;
  const int H4 = 4 * h;

  // zero h_{-1}, stage x_0
  for (int i = threadIdx.x; i < kRows * SH; i += 256) hbuf[i] = 0.f;
  for (int i = threadIdx.x; i < kRows * din; i += 256) {
    const int r = i / din, c = i % din;
    const int64_t gr = row0 + r;
    xbuf[r * SX + c] = (gr < B) ? x[gr * din + c] : 0.f;
  }
  float cprev[TPG][4];
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp)
#pragma unroll
    for (int r = 0; r < 4; ++r) cprev[tp][r] = 0.f;
  __syncthreads();

  for (int64_t t = 0; t < L; ++t) {
    const int cur = (int)(t & 1), nxt = cur ^ 1;
    const float* xb = xbuf + cur * kRows * SX;
    const float* hb = hbuf + cur * kRows * SH;
    // prefetch x_{t+1}
    if (t + 1 < L) {
      float* xn = xbuf + nxt * kRows * SX;
      const float* xs = x + (t + 1) * B * din;
      for (int i = threadIdx.x; i < kRows * din; i += 256) {
        const int r = i / din, c = i % din;
        const int64_t gr = row0 + r;
        xn[r * SX + c] = (gr < B) ? xs[gr * din + c] : 0.f;
      }
    }
    f32x4 acc[4][TPG];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int tp = 0; tp < TPG; ++tp) {
        const float bv = bias[g * h + ubase + tp * 16 + l15];
        acc[g][tp] = (f32x4){bv, bv, bv, bv};
      }
    // x part
    for (int kk = 0; kk < din; kk += 4) {
      const float a = xb[l15 * SX + kk + lq];
      const float* wrow = W + (int64_t)(kk + lq) * H4 + ubase + l15;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tp = 0; tp < TPG; ++tp)
          acc[g][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[g * h + tp * 16], acc[g][tp], 0, 0, 0);
    }
    // h part
    for (int kk = 0; kk < h; kk += 4) {
      const float a = hb[l15 * SH + kk + lq];
      const float* wrow = W + (int64_t)(din + kk + lq) * H4 + ubase + l15;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tp = 0; tp < TPG; ++tp)
          acc[g][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[g * h + tp * 16], acc[g][tp], 0, 0, 0);
    }
    // cell update; C/D map of 16x16: col = lane&15 (unit), row = (lane>>4)*4 + reg
    float* hn = hbuf + nxt * kRows * SH;
#pragma unroll
    for (int tp = 0; tp < TPG; ++tp) {
      const int unit = ubase + tp * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = lq * 4 + r;
        const int64_t gr = row0 + lrow;
        const float gi = sigmoidf_(acc[0][tp][r]);
        const float gj = tanhf_(acc[1][tp][r]);
        const float gf = sigmoidf_(acc[2][tp][r] + forget_bias);
        const float go = sigmoidf_(acc[3][tp][r]);
        const float c = gf * cprev[tp][r] + gi * gj;
        const float hh = go * tanhf_(c);
        cprev[tp][r] = c;
        hn[lrow * SH + unit] = hh;
        if (gr < B) {
          const int64_t o = (t * B + gr);
          hs[o * h + unit] = hh;
          cs[o * h + unit] = c;
          float* gp = gates + o * H4 + unit;
          gp[0] = gi;
          gp[h] = gj;
          gp[2 * h] = gf;
          gp[3 * h] = go;
        }
      }
    }
    __syncthreads();
  }
}

template <int TPG>
__global__ __launch_bounds__(256) void k_lstm_bwd(
    const float* __restrict__ W, const float* __restrict__ cs, const float* __restrict__ gates,
    const float* __restrict__ dhs, int64_t L, int64_t B, int din, int h,
    float* __restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H4 = 4 * h, SZ = H4 + 2;
  float* zbuf = smem;  // [16][SZ]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int UPW = h / 4;
  const int ubase = wave * UPW;
  float dh_rec[TPG][4], dc[TPG][4];
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh_rec[tp][r] = dc[tp][r] = 0.f;

  for (int64_t t = L - 1; t >= 0; --t) {
#pragma unroll
    for (int tp = 0; tp < TPG; ++tp) {
      const int unit = ubase + tp * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = lq * 4 + r;
        const int64_t gr = row0 + lrow;
        float zi = 0.f, zj = 0.f, zf = 0.f, zo = 0.f;
        if (gr < B) {
          const int64_t o = t * B + gr;
          const float* gp = gates + o * H4 + unit;
          const float gi = gp[0], gj = gp[h], gf = gp[2 * h], go = gp[3 * h];
          const float c = cs[o * h + unit];
          const float cp = (t > 0) ? cs[(o - B) * h + unit] : 0.f;
          const float dh = dhs[o * h + unit] + dh_rec[tp][r];
          const float tc = tanhf(c);
          const float d_o = dh * tc;
          const float dcc = dc[tp][r] + dh * go * (1.f - tc * tc);
          zi = dcc * gj * gi * (1.f - gi);
          zj = dcc * gi * (1.f - gj * gj);
          zf = dcc * cp * gf * (1.f - gf);
          zo = d_o * go * (1.f - go);
          dc[tp][r] = dcc * gf;
          float* zp = dz + o * H4 + unit;
          zp[0] = zi;
          zp[h] = zj;
          zp[2 * h] = zf;
          zp[3 * h] = zo;
        }
        float* zb = zbuf + lrow * SZ + unit;
        zb[0] = zi;
        zb[h] = zj;
        zb[2 * h] = zf;
        zb[3 * h] = zo;
      }
    }
    __syncthreads();
    if (t > 0) {
      // dh_{t-1}[row, unit] = sum_k dz[row, k] * W[din + unit, k]
      f32x4 acc[TPG];
#pragma unroll
      for (int tp = 0; tp < TPG; ++tp) acc[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < H4; kk += 4) {
        const float a = zbuf[l15 * SZ + kk + lq];
#pragma unroll
        for (int tp = 0; tp < TPG; ++tp) {
          const float b = W[(int64_t)(din + ubase + tp * 16 + l15) * H4 + kk + lq];
          acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tp], 0, 0, 0);
        }
      }
#pragma unroll
      for (int tp = 0; tp < TPG; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh_rec[tp][r] = acc[tp][r];
    }
    __syncthreads();
  }
}

// ---- d = h = 64 (config C4): W register-resident --------------------------------------
// The streaming kernels above re-read their W slice from L2 every time step (4 dependent-ish
// global loads per k-step in a rolled loop: ~8 us per step measured, 8 TF).  At h = 64 a
// wave's slice is (din + h) x 64 gate columns = 32 KB = 128 VGPRs per lane, so it is loaded
// ONCE and the step is 32 LDS operand reads + 128 back-to-back MFMAs; bias lives in
// registers too.  Backward: W_h^T slice = 64 VGPRs; the next step's gates / cells / dh are
// prefetched into registers while the current step's dh GEMM runs.
template <int DIN>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_wreg(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
    int64_t L, int64_t B, float forget_bias, float* __restrict__ hs, float* __restrict__ cs,
    float* __restrict__ gates) {
  constexpr int H = 64, H4 = 256, SX = DIN + 2, SH = H + 2;
  constexpr int NKX = DIN / 4, NKH = H / 4;
  __shared__ __attribute__((aligned(16))) float xbuf[2][kRows * SX];
  __shared__ __attribute__((aligned(16))) float hbuf[2][kRows * SH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int ubase = wave * 16;
  // W slice: k = 4*ks + lq, columns g*64 + ubase + l15
  float wx[NKX][4], wh[NKH][4], bv[4];
#pragma unroll
  for (int ks = 0; ks < NKX; ++ks)
#pragma unroll
    for (int g = 0; g < 4; ++g) wx[ks][g] = W[(int64_t)(4 * ks + lq) * H4 + g * H + ubase + l15];
#pragma unroll
  for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
    for (int g = 0; g < 4; ++g) wh[ks][g] = W[(int64_t)(DIN + 4 * ks + lq) * H4 + g * H + ubase + l15];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = bias[g * H + ubase + l15];
  for (int i = threadIdx.x; i < kRows * SH; i += 256) hbuf[0][i] = 0.f;
  for (int i = threadIdx.x; i < kRows * DIN; i += 256) {
    const int r = i / DIN, c = i % DIN;
    const int64_t gr = row0 + r;
    xbuf[0][r * SX + c] = (gr < B) ? x[gr * DIN + c] : 0.f;
  }
  float cprev[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int64_t t = 0; t < L; ++t) {
    const int cur = (int)(t & 1), nxt = cur ^ 1;
    const float* xb = xbuf[cur];
    const float* hb = hbuf[cur];
    float xn[kRows * DIN / 256];                 // x_{t+1}: loads fly under the MFMAs
    if (t + 1 < L) {
      const float* xs = x + (t + 1) * B * DIN;
#pragma unroll
      for (int q = 0; q < kRows * DIN / 256; ++q) {
        const int i = threadIdx.x + q * 256;
        const int64_t gr = row0 + i / DIN;
        xn[q] = xs[min(gr, B - 1) * DIN + i % DIN];
      }
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){bv[g], bv[g], bv[g], bv[g]};
#pragma unroll
    for (int ks = 0; ks < NKX; ++ks) {
      const float a = xb[l15 * SX + 4 * ks + lq];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wx[ks][g], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < NKH; ++ks) {
      const float a = hb[l15 * SH + 4 * ks + lq];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wh[ks][g], acc[g], 0, 0, 0);
    }
    // cell update; C/D map of 16x16: col = lane&15 (unit), row = (lane>>4)*4 + reg
    float* hn = hbuf[nxt];
    const int unit = ubase + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = lq * 4 + r;
      const int64_t gr = row0 + lrow;
      const float gi = sigmoidf_(acc[0][r]);
      const float gj = tanhf_(acc[1][r]);
      const float gf = sigmoidf_(acc[2][r] + forget_bias);
      const float go = sigmoidf_(acc[3][r]);
      const float c = gf * cprev[r] + gi * gj;
      const float hh = go * tanhf_(c);
      cprev[r] = c;
      hn[lrow * SH + unit] = hh;
      if (gr < B) {
        const int64_t o = (t * B + gr);
        hs[o * H + unit] = hh;
        cs[o * H + unit] = c;
        float* gp = gates + o * H4 + unit;
        gp[0] = gi;
        gp[H] = gj;
        gp[2 * H] = gf;
        gp[3 * H] = go;
      }
    }
    if (t + 1 < L) {
      float* xw = xbuf[nxt];
#pragma unroll
      for (int q = 0; q < kRows * DIN / 256; ++q) {
        const int i = threadIdx.x + q * 256;
        const int64_t gr = row0 + i / DIN;
        xw[(i / DIN) * SX + i % DIN] = (gr < B) ? xn[q] : 0.f;
      }
    }
    __syncthreads();
  }
}

template <int DIN>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_wreg(
    const float* __restrict__ W, const float* __restrict__ cs, const float* __restrict__ gates,
    const float* __restrict__ dhs, int64_t L, int64_t B, float* __restrict__ dz) {
  constexpr int H = 64, H4 = 256, SZ = H4 + 2, NK = H4 / 4;
  __shared__ __attribute__((aligned(16))) float zbuf[kRows * SZ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int unit = wave * 16 + l15;
  // W_h^T slice: B operand element (k = 4*ks + lq, unit) = W[DIN + unit][k]
  float wt[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) wt[ks] = W[(int64_t)(DIN + unit) * H4 + 4 * ks + lq];
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
  // step-t inputs of this lane's 4 rows: gates (4), c, c_prev, dh
  float pg[4][4], pc[4], pcp[4], pdh[4];
  auto prefetch = [&](int64_t t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gr = min(row0 + lq * 4 + r, B - 1);
      const int64_t o = t * B + gr;
      const float* gp = gates + o * H4 + unit;
      pg[r][0] = gp[0];
      pg[r][1] = gp[H];
      pg[r][2] = gp[2 * H];
      pg[r][3] = gp[3 * H];
      pc[r] = cs[o * H + unit];
      pcp[r] = (t > 0) ? cs[(o - B) * H + unit] : 0.f;
      pdh[r] = dhs[o * H + unit];
    }
  };
  if (L > 0) prefetch(L - 1);
  for (int64_t t = L - 1; t >= 0; --t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = lq * 4 + r;
      const int64_t gr = row0 + lrow;
      float zi = 0.f, zj = 0.f, zf = 0.f, zo = 0.f;
      if (gr < B) {
        const float gi = pg[r][0], gj = pg[r][1], gf = pg[r][2], go = pg[r][3];
        const float dh = pdh[r] + dh_rec[r];
        const float tc = tanhf(pc[r]);
        const float d_o = dh * tc;
        const float dcc = dc[r] + dh * go * (1.f - tc * tc);
        zi = dcc * gj * gi * (1.f - gi);
        zj = dcc * gi * (1.f - gj * gj);
        zf = dcc * pcp[r] * gf * (1.f - gf);
        zo = d_o * go * (1.f - go);
        dc[r] = dcc * gf;
        float* zp = dz + (t * B + gr) * H4 + unit;
        zp[0] = zi;
        zp[H] = zj;
        zp[2 * H] = zf;
        zp[3 * H] = zo;
      }
      float* zb = zbuf + lrow * SZ + unit;
      zb[0] = zi;
      zb[H] = zj;
      zb[2 * H] = zf;
      zb[3 * H] = zo;
    }
    __syncthreads();
    if (t > 0) {
      prefetch(t - 1);                          // in flight under the dh GEMM
      // dh_{t-1}[row, unit] = sum_k dz[row, k] * W[DIN + unit, k]
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;   // two chains: MFMA latency
#pragma unroll
      for (int ks = 0; ks < NK; ks += 2) {
        const float a0 = zbuf[l15 * SZ + 4 * ks + lq];
        const float a1 = zbuf[l15 * SZ + 4 * (ks + 1) + lq];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wt[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wt[ks + 1], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[r] = acc0[r] + acc1[r];
    }
    __syncthreads();
  }
}

// ---- d = h = 64, FOUR batch rows per workgroup (v_mfma_f32_4x4x1_16b_f32) ----------------------
// The 16-row kernels above put B / 16 workgroups on the chip: 64 of 256 CUs at the C4 batch
// (B = 1024).  The 16-block 4x4x1 MFMA has the same flop rate (512 flop / 8 cycles per SIMD) with a
// 4-row A operand broadcast to its 16 blocks, so a workgroup can own FOUR complete batch rows --
// B / 4 workgroups, every CU busy at B = 1024, no exchange between workgroups -- at unchanged MFMA
// efficiency per row.  Operand / result layout (probed on gfx950, tools/probe/mfma4x4.hip):
// lane l = 4 * block + j supplies A[block][i = l & 3] and B[block][j]; D[block][i][j] is VGPR i of
// lane 4 * block + j.
//   forward : block = one of the wave's 16 units, j = gate (i, j, f, o) => a wave forms the four
//             gate pre-activations of 16 units x 4 rows per instruction chain over K = d + h = 128;
//             W slice 128 VGPRs; a quad (the 4 gate lanes of a unit) exchanges its values with 16
//             width-4 shuffles and lane j finishes row j.
//   backward: thread (row, unit) does the element-wise part; for dh_{t-1} = dz_t . W_h^T the four
//             waves split K = 4h (wave = gate), blocks x j = the 64 output units, partial sums meet
//             in LDS and are added in wave order (fixed => deterministic).
constexpr int kR4 = 4;

// lane g of every quad to its four lanes: DPP quad_perm [g, g, g, g] (the control word is an immediate)
__device__ __forceinline__ float quad_bcast(float x, int g) {
  const int v = __float_as_int(x);
  switch (g) {
    case 0: return __int_as_float(__builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true));
    case 1: return __int_as_float(__builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true));
    case 2: return __int_as_float(__builtin_amdgcn_mov_dpp(v, 0xAA, 0xf, 0xf, true));
    default: return __int_as_float(__builtin_amdgcn_mov_dpp(v, 0xFF, 0xf, 0xf, true));
  }
}

template <int DIN>
__global__ __launch_bounds__(256, 2) void k_lstm_fwd_r4(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
    int64_t L, int64_t B, float forget_bias, float* __restrict__ hs, float* __restrict__ cs,
    float* __restrict__ gates) {
  constexpr int H = 64, H4 = 256, SX = DIN + 4, SH = H + 4, NK = DIN + H;
  static_assert(DIN == 64, "one x value per thread and step");
  __shared__ __attribute__((aligned(16))) float xbuf[2][kR4 * SX];
  __shared__ __attribute__((aligned(16))) float hbuf[2][kR4 * SH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = lane >> 2, j = lane & 3;
  const int64_t row0 = (int64_t)blockIdx.x * kR4;
  const int unit = wave * 16 + blk;
  float wreg[NK];                                   // W[k][gate j of this unit]
#pragma unroll
  for (int k = 0; k < NK; ++k) wreg[k] = W[(int64_t)k * H4 + j * H + unit];
  const float bv = bias[j * H + unit];
  for (int i = threadIdx.x; i < kR4 * SH; i += 256) hbuf[0][i] = 0.f;
  {
    const int r = threadIdx.x / DIN, c = threadIdx.x % DIN;
    const int64_t gr = row0 + r;
    xbuf[0][r * SX + c] = (gr < B) ? x[gr * DIN + c] : 0.f;
  }
  float cprev = 0.f;                                // cell state of (row j, unit)
  __syncthreads();
  for (int64_t t = 0; t < L; ++t) {
    const int cur = (int)(t & 1), nxt = cur ^ 1;
    const float* xb = xbuf[cur] + j * SX;           // this lane supplies row i = j of the A operand
    const float* hb = hbuf[cur] + j * SH;
    float xn = 0.f;                                 // x_{t+1}: the load flies under the MFMAs
    const int xr = threadIdx.x / DIN, xc = threadIdx.x % DIN;
    if (t + 1 < L) xn = x[((t + 1) * B + min(row0 + xr, B - 1)) * DIN + xc];
    f32x4 a0 = (f32x4){bv, bv, bv, bv}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
    // 32 operand groups (16 of x_t, 16 of h_{t-1}), four MFMAs each; the LDS read of group g + 4 is issued behind the
    // MFMAs of group g (a ring of four registers).  Left to the compiler the read of group g + 1 sat under the 32
    // cycles of group g -- less than an LDS round trip: 1 990 cycles for the 1 024 of the 128 MFMAs (cycle stamps,
    // tools/lstm_trace.py).
    constexpr int NG = (DIN + H) / 4;
    float4 q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = *reinterpret_cast<const float4*>((g < DIN / 4 ? xb + 4 * g : hb + 4 * (g - DIN / 4)));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 v = q[g & 3];
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.x, wreg[4 * g + 0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, wreg[4 * g + 1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.z, wreg[4 * g + 2], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.w, wreg[4 * g + 3], a3, 0, 0, 0);
      if (g + 4 < NG) {
        const int gn = g + 4;
        q[g & 3] = *reinterpret_cast<const float4*>((gn < DIN / 4 ? xb + 4 * gn : hb + 4 * (gn - DIN / 4)));
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // z[i] = pre-activation of gate j, unit, row i.  Quad exchange: lane j takes row j's four gates.
    float zg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float pick = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float zi = (a0[i] + a1[i]) + (a2[i] + a3[i]);
        // row i, gate g of this unit: lane g of the quad, as a DPP quad_perm broadcast (a VALU move; __shfl went
        // through the LDS crossbar: 16 ds_bpermute round trips per step)
        const float got = quad_bcast(zi, g);
        pick = (i == j) ? got : pick;
      }
      zg[g] = pick;
    }
    const float gi = sigmoidf_(zg[0]);
    const float gj = tanhf_(zg[1]);
    const float gf = sigmoidf_(zg[2] + forget_bias);
    const float go = sigmoidf_(zg[3]);
    const float c = gf * cprev + gi * gj;
    const float hh = go * tanhf_(c);
    cprev = c;
    hbuf[nxt][j * SH + unit] = hh;
    // (x_{t+1} goes to LDS BEFORE the step's global stores are issued: loads and stores share the vmcnt counter, and
    // behind the stores the wait for this one load was a wait for seven store acknowledgements per step)
    if (t + 1 < L) xbuf[nxt][xr * SX + xc] = (row0 + xr < B) ? xn : 0.f;
    const int64_t gr = row0 + j;
    if (gr < B) {
      const int64_t o = t * B + gr;
      hs[o * H + unit] = hh;
      cs[o * H + unit] = c;
      float* gp = gates + o * H4 + unit;
      gp[0] = gi;
      gp[H] = gj;
      gp[2 * H] = gf;
      gp[3 * H] = go;
    }
    __syncthreads();
  }
}

template <int DIN>
__global__ __launch_bounds__(256, 2) void k_lstm_bwd_r4(
    const float* __restrict__ W, const float* __restrict__ cs, const float* __restrict__ gates,
    const float* __restrict__ dhs, int64_t L, int64_t B, float* __restrict__ dz,
    float* __restrict__ wxt) {
  constexpr int H = 64, H4 = 256, SZ = H4 + 4;
  __shared__ __attribute__((aligned(16))) float zbuf[kR4 * SZ];
  __shared__ __attribute__((aligned(16))) float part[4][kR4 * H];
  // W_x^T [4h, din] for the caller's dx product rides along (a 16 k-element transpose that was a launch
  // of its own on the step's serial chain): a few elements per workgroup, ahead of the recurrence
  if (wxt)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < DIN * H4; e += gridDim.x * 256)
      wxt[(e % H4) * DIN + e / H4] = W[e];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = lane >> 2, j = lane & 3;
  const int64_t row0 = (int64_t)blockIdx.x * kR4;
  // element-wise role: thread = (row = wave, unit = lane)
  const int64_t gr = row0 + wave;
  const bool live = gr < B;
  const int64_t grc = live ? gr : B - 1;
  // MFMA role: wave = K quarter (gate `wave` of dz), output unit 4 * blk + j
  float wt[H];
#pragma unroll
  for (int kk = 0; kk < H; ++kk) wt[kk] = W[(int64_t)(DIN + 4 * blk + j) * H4 + wave * H + kk];
  float dh_rec = 0.f, dc = 0.f;
  float pg0, pg1, pg2, pg3, pc, pcp, pdh;
  auto prefetch = [&](int64_t t) {
    const int64_t o = t * B + grc;
    const float* gp = gates + o * H4 + lane;
    pg0 = gp[0]; pg1 = gp[H]; pg2 = gp[2 * H]; pg3 = gp[3 * H];
    pc = cs[o * H + lane];
    pcp = (t > 0) ? cs[(o - B) * H + lane] : 0.f;
    pdh = dhs[o * H + lane];
  };
  if (L > 0) prefetch(L - 1);
  for (int64_t t = L - 1; t >= 0; --t) {
    float zi = 0.f, zj = 0.f, zf = 0.f, zo = 0.f;
    if (live) {
      const float gi = pg0, gj = pg1, gf = pg2, go = pg3;
      const float dh = pdh + dh_rec;
      const float tc = tanhf(pc);
      const float d_o = dh * tc;
      const float dcc = dc + dh * go * (1.f - tc * tc);
      zi = dcc * gj * gi * (1.f - gi);
      zj = dcc * gi * (1.f - gj * gj);
      zf = dcc * pcp * gf * (1.f - gf);
      zo = d_o * go * (1.f - go);
      dc = dcc * gf;
      float* zp = dz + (t * B + gr) * H4 + lane;
      zp[0] = zi;
      zp[H] = zj;
      zp[2 * H] = zf;
      zp[3 * H] = zo;
    }
    float* zb = zbuf + wave * SZ + lane;
    zb[0] = zi;
    zb[H] = zj;
    zb[2 * H] = zf;
    zb[3 * H] = zo;
    __syncthreads();
    if (t > 0) {
      prefetch(t - 1);                              // in flight under the dh product
      const float* ar = zbuf + j * SZ + wave * H;   // A operand: row i = j, this wave's K quarter
      f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
      for (int ks = 0; ks < H / 4; ++ks) {
        const float4 v = *reinterpret_cast<const float4*>(ar + 4 * ks);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.x, wt[4 * ks + 0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, wt[4 * ks + 1], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.z, wt[4 * ks + 2], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.w, wt[4 * ks + 3], a3, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][i * H + 4 * blk + j] = (a0[i] + a1[i]) + (a2[i] + a3[i]);
      __syncthreads();
      dh_rec = (part[0][wave * H + lane] + part[1][wave * H + lane]) +
               (part[2][wave * H + lane] + part[3][wave * H + lane]);
    }
  }
}

// ---- generic fallback (any din, h): one workgroup per batch row -------------
__global__ __launch_bounds__(256) void k_lstm_fwd_generic(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
    int64_t L, int64_t B, int din, int h, float forget_bias, float* __restrict__ hs,
    float* __restrict__ cs, float* __restrict__ gates) {
  extern __shared__ float smem[];
  float* in = smem;            // [din + h]
  float* z = smem + din + h;   // [4h]
  float* cst = z + 4 * h;      // [h]
  const int64_t b = blockIdx.x;
  const int K = din + h, H4 = 4 * h;
  for (int i = threadIdx.x; i < h; i += 256) { in[din + i] = 0.f; cst[i] = 0.f; }
  for (int64_t t = 0; t < L; ++t) {
    for (int i = threadIdx.x; i < din; i += 256) in[i] = x[(t * B + b) * din + i];
    __syncthreads();
    for (int c = threadIdx.x; c < H4; c += 256) {
      float s = bias[c];
      for (int k = 0; k < K; ++k) s = fmaf(in[k], W[(int64_t)k * H4 + c], s);
      z[c] = s;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < h; u += 256) {
      const float gi = sigmoidf_(z[u]), gj = tanhf_(z[h + u]);
      const float gf = sigmoidf_(z[2 * h + u] + forget_bias), go = sigmoidf_(z[3 * h + u]);
      const float c = gf * cst[u] + gi * gj;
      const float hh = go * tanhf_(c);
      cst[u] = c;
      in[din + u] = hh;
      const int64_t o = t * B + b;
      hs[o * h + u] = hh;
      cs[o * h + u] = c;
      gates[o * H4 + u] = gi;
      gates[o * H4 + h + u] = gj;
      gates[o * H4 + 2 * h + u] = gf;
      gates[o * H4 + 3 * h + u] = go;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_lstm_bwd_generic(
    const float* __restrict__ W, const float* __restrict__ cs, const float* __restrict__ gates,
    const float* __restrict__ dhs, int64_t L, int64_t B, int din, int h,
    float* __restrict__ dz) {
  extern __shared__ float smem[];
  float* z = smem;           // [4h]
  float* dhr = smem + 4 * h; // [h]
  float* dcs = dhr + h;      // [h]
  const int64_t b = blockIdx.x;
  const int H4 = 4 * h;
  for (int i = threadIdx.x; i < h; i += 256) { dhr[i] = 0.f; dcs[i] = 0.f; }
  __syncthreads();
  for (int64_t t = L - 1; t >= 0; --t) {
    const int64_t o = t * B + b;
    for (int u = threadIdx.x; u < h; u += 256) {
      const float gi = gates[o * H4 + u], gj = gates[o * H4 + h + u];
      const float gf = gates[o * H4 + 2 * h + u], go = gates[o * H4 + 3 * h + u];
      const float c = cs[o * h + u];
      const float cp = (t > 0) ? cs[(o - B) * h + u] : 0.f;
      const float dh = dhs[o * h + u] + dhr[u];
      const float tc = tanhf(c);
      const float dcc = dcs[u] + dh * go * (1.f - tc * tc);
      const float zi = dcc * gj * gi * (1.f - gi), zj = dcc * gi * (1.f - gj * gj);
      const float zf = dcc * cp * gf * (1.f - gf), zo = dh * tc * go * (1.f - go);
      dcs[u] = dcc * gf;
      z[u] = zi; z[h + u] = zj; z[2 * h + u] = zf; z[3 * h + u] = zo;
      dz[o * H4 + u] = zi; dz[o * H4 + h + u] = zj;
      dz[o * H4 + 2 * h + u] = zf; dz[o * H4 + 3 * h + u] = zo;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < h; u += 256) {
      float s = 0.f;
      const float* wr = W + (int64_t)(din + u) * H4;
      for (int k = 0; k < H4; ++k) s = fmaf(z[k], wr[k], s);
      dhr[u] = s;
    }
    __syncthreads();
  }
}

}  // namespace arx

using namespace arx;

extern "C" {

int arx_lstm_fwd(const float* x, const float* W, const float* b, int64_t L, int64_t B, int din,
                 int h, float forget_bias, float* hs, float* cs, float* gates, void* stream) {
  ARX_CHECK_ARG(x && W && b && hs && cs && gates, "arx_lstm_fwd: null pointer");
  ARX_CHECK_ARG(L >= 0 && B >= 0 && din > 0 && h > 0, "arx_lstm_fwd: bad size");
  if (L == 0 || B == 0) return ARX_OK;
  hipStream_t s = as_stream(stream);
  const bool mfma_ok = (h % 64 == 0) && (h <= 128) && (din % 4 == 0);
  // four-row workgroups while they fill the chip better than sixteen-row ones (B / 16 < 2 per CU)
  const bool r4 = h == 64 && din == 64 && ceil_div(B, kRows) < 2 * (int64_t)cu_count();
  if (r4) {
    k_lstm_fwd_r4<64><<<(int)ceil_div(B, kR4), 256, 0, s>>>(x, W, b, L, B, forget_bias, hs, cs, gates);
  } else if (h == 64 && din == 64) {
    k_lstm_fwd_wreg<64><<<(int)ceil_div(B, kRows), 256, 0, s>>>(x, W, b, L, B, forget_bias, hs, cs,
                                                                   gates);
  } else if (mfma_ok) {
    const size_t lds = (size_t)(2 * kRows * (din + 2) + 2 * kRows * (h + 2)) * sizeof(float);
    const int grid = (int)ceil_div(B, kRows);
    if (h == 64) k_lstm_fwd<1><<<grid, 256, lds, s>>>(x, W, b, L, B, din, h, forget_bias, hs, cs, gates);
    else k_lstm_fwd<2><<<grid, 256, lds, s>>>(x, W, b, L, B, din, h, forget_bias, hs, cs, gates);
  } else {
    const size_t lds = (size_t)(din + h + 4 * h + h) * sizeof(float);
    ARX_CHECK_ARG(lds <= 160 * 1024, "arx_lstm_fwd: sizes exceed LDS");
    k_lstm_fwd_generic<<<(int)B, 256, lds, s>>>(x, W, b, L, B, din, h, forget_bias, hs, cs, gates);
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_lstm_bwd(const float* W, const float* hs, const float* cs, const float* gates,
                 const float* dhs, int64_t L, int64_t B, int din, int h, float* dz,
                 void* stream) {
  return arx_lstm_bwd_wxt(W, hs, cs, gates, dhs, L, B, din, h, dz, nullptr, stream);
}

int arx_lstm_bwd_wxt(const float* W, const float* hs, const float* cs, const float* gates,
                     const float* dhs, int64_t L, int64_t B, int din, int h, float* dz, float* wxt,
                     void* stream) {
  (void)hs;
  ARX_CHECK_ARG(W && cs && gates && dhs && dz, "arx_lstm_bwd: null pointer");
  ARX_CHECK_ARG(L >= 0 && B >= 0 && din > 0 && h > 0, "arx_lstm_bwd: bad size");
  hipStream_t s = as_stream(stream);
  const bool mfma_ok = (h % 64 == 0) && (h <= 128);
  const bool r4 = h == 64 && din == 64 && ceil_div(B, kRows) < 2 * (int64_t)cu_count();
  if (wxt && !(r4 && L > 0 && B > 0)) {          // the other kernels leave the transpose to its own launch
    const int rc = arx_transpose_f32(W, 4 * (int64_t)h, din, 4 * (int64_t)h, wxt, din, stream);
    if (rc) return rc;
  }
  if (L == 0 || B == 0) return ARX_OK;
  if (r4) {
    k_lstm_bwd_r4<64><<<(int)ceil_div(B, kR4), 256, 0, s>>>(W, cs, gates, dhs, L, B, dz, wxt);
  } else if (h == 64 && din == 64) {
    k_lstm_bwd_wreg<64><<<(int)ceil_div(B, kRows), 256, 0, s>>>(W, cs, gates, dhs, L, B, dz);
  } else if (mfma_ok) {
    const size_t lds = (size_t)(kRows * (4 * h + 2)) * sizeof(float);
    const int grid = (int)ceil_div(B, kRows);
    if (h == 64) k_lstm_bwd<1><<<grid, 256, lds, s>>>(W, cs, gates, dhs, L, B, din, h, dz);
    else k_lstm_bwd<2><<<grid, 256, lds, s>>>(W, cs, gates, dhs, L, B, din, h, dz);
  } else {
    const size_t lds = (size_t)(4 * h + 2 * h) * sizeof(float);
    k_lstm_bwd_generic<<<(int)B, 256, lds, s>>>(W, cs, gates, dhs, L, B, din, h, dz);
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
