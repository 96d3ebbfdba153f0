// gemm_dma.hip -- the two backward scorer GEMMs with LDS-DMA operand streaming:
//   dU[B,d] = dL[B,S] . I[S,d]      (NN: A k-contiguous, B n-contiguous;  embed_attribute.py:171 bwd)
//   dI[S,d] = dL[B,S]^T . U[B,d]    (TN: A m-contiguous, B n-contiguous)
// Both have N = d (<= 128 columns, one tile wide) and a long K, so dL -- the only large
// operand (B*S*4 bytes) -- is read from HBM exactly ONCE (the 64x64-tiled kernel in gemm.hip
// read it twice for dI: PMC FETCH 140 MB vs 67 MB algorithmic at B=16384).
//
// Tile BM x 128 x 32, 4 waves as 2 (M) x 2 (N); operand tiles arrive by
// global_load_lds_dwordx4 into two LDS stages (separate __shared__ arrays: hipcc must see
// that the DMA destination and the stage being read cannot alias, or it drains vmcnt(0)
// ahead of the MFMA block); tile t+1 is in flight while tile t is multiplied.
//   k-contiguous A  : image [BM][32], 16-B chunk c of row r at slot c ^ ((r>>1)&7) (source-
//                     side swizzle, DMA writes lane-linear); the operand is read with one
//                     ds_read_b128 per 4 MFMAs (lane supplies k = 8s + 4*(lane/32) + j);
//   m/n-contiguous  : image [32][rows], linear; operand reads are 32 consecutive floats.
// Deterministic split-K (partials in the caller's workspace, fixed-order reduce by
// k_splitk_reduce in gemm.hip); optional row sums of op(A) (the bias gradient of dI).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace arx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kBK = 32;
constexpr int kBNMax = 128;

// LDS-DMA piece issued from inline asm: hipcc's waitcnt pass loses its per-buffer LDS-DMA
// tracking at the loop header and drains vmcnt(0) before the first ds_read of every loop
// iteration when the builtin is used in a 3-stage ring; issued this way the copy is invisible
// to it and ALL ordering is explicit (counted s_waitcnt vmcnt(N) + s_barrier below).  M0 (the
// wave-uniform LDS base) is saved and restored inside the statement.
template <bool NT = false>
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds_wave_base);
  uint32_t keep;
  if constexpr (NT) {       // streamed operand (read once per launch): do not keep it in L2
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(dst)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(dst)
        : "memory");
  }
}

// Second B operand of the paired TN product (arx_gemm_f32_tn_pair): columns [n1, N) of op(B)
// come from B2, whose row k - shift sits under row k of A (rows k < shift read zeros).
struct PairB {
  const float* B2;
  int64_t ldb2, shift;
  int n1;
};
struct NoPair {};
__device__ __attribute__((aligned(16))) float g_zero4[4];      // what a row ahead of the shift reads

template <bool A_KC, int BM, int kBN, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void k_gemm_dma(
    int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float beta, float* __restrict__ C, int64_t ldc,
    const float* __restrict__ col_bias, float* __restrict__ partial, int64_t kchunk,
    float* __restrict__ a_rowsum, float* __restrict__ rowsum_partial,
    typename std::conditional<PAIR, PairB, NoPair>::type pair) {
  constexpr int FM = BM / 64;                 // A fragments per wave (wave covers BM/2 rows)
  constexpr int NPA = BM * kBK / 4 / 256;     // DMA pieces per thread, A tile
  constexpr int NPB = kBK * kBN / 4 / 256;    // 4 (BN=128) or 2 (BN=64)
  constexpr int FN = kBN / 64;                // B fragments per wave (wave covers BN/2 columns)
  // BM=64: three LDS stages (72 KB): two tiles of DMA in flight behind the one being
  // multiplied -- one tile of MFMA work is ~0.85 us, shorter than an HBM round trip.
  // BM=128: two stages (64 KB) so that two workgroups still fit a CU.
  constexpr int STAGES = (BM == 64) ? 3 : 2;
  constexpr int PD = STAGES - 1;              // prefetch distance (tiles)
  __shared__ __attribute__((aligned(1024))) float sA0[BM * kBK];
  __shared__ __attribute__((aligned(1024))) float sA1[BM * kBK];
  __shared__ __attribute__((aligned(1024))) float sA2[STAGES == 3 ? BM * kBK : 4];
  __shared__ __attribute__((aligned(1024))) float sB0[kBK * kBN];
  __shared__ __attribute__((aligned(1024))) float sB1[kBK * kBN];
  __shared__ __attribute__((aligned(1024))) float sB2[STAGES == 3 ? kBK * kBN : 4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // (M tile, K split) of this block.  Blocks are dealt round-robin to the 8 XCDs (linear id %
  // 8); with split-K the M tiles of one K chunk all read the same B rows (dI: the U rows of the
  // chunk, re-read by every M tile -- PMC showed 134 MB fetched for 75 MB algorithmic), so they
  // are re-indexed to sit on ONE XCD and share those rows through its L2.
  int bx = blockIdx.x, bz = blockIdx.z;
  {
    const int nx = gridDim.x, nz = gridDim.z;
    if (nz > 1 && nz % 8 == 0) {
      const int lin = bx + nx * bz;
      const int xcd = lin & 7, slot = lin >> 3;      // slot < nx * nz / 8
      bz = xcd * (nz / 8) + slot / nx;
      bx = slot % nx;
    }
  }
  const int64_t m0 = (int64_t)bx * BM;
  const int64_t kbeg = (int64_t)bz * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);
  const int64_t nt = (kend - kbeg) / kBK;      // K % 32 == 0 and kchunk % 32 == 0 (host)

  auto dma = [&](int64_t t, float* imgA, float* imgB) {
    const int64_t k0 = kbeg + t * kBK;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int f = threadIdx.x + i * 256;
      const float* src;
      if (A_KC) {                               // [BM][32]: row r, chunk c (swizzled)
        const int r = f / 8, c = (f % 8) ^ ((r >> 1) & 7);
        src = A + min(m0 + r, M - 1) * lda + k0 + c * 4;
      } else {                                  // [32][BM]: k row, 4 consecutive m
        const int k = f / (BM / 4), mq = f % (BM / 4);
        src = A + (k0 + k) * lda + min(m0 + mq * 4, M - 4);
      }
      glds16<true>(src, imgA + (f - lane) * 4);     // A = dlogits: streamed once
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int f = threadIdx.x + i * 256;
      const int k = f / (kBN / 4), nq = f % (kBN / 4);
      const int64_t col = min((int64_t)nq * 4, N - 4);
      const float* src = B + (k0 + k) * ldb + col;
      if constexpr (PAIR) {
        if (col >= pair.n1) {
          const int64_t r2 = k0 + k - pair.shift;
          src = r2 >= 0 ? pair.B2 + r2 * pair.ldb2 + (col - pair.n1) : g_zero4;
        }
      }
      glds16(src, imgB + (f - lane) * 4);
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // row sums of op(A) ride along in registers: every lane already holds its row's elements
  // for its half of the k indices (no extra LDS reads, no divergent block: an LDS read inside
  // a conditional block made hipcc drain the DMA queue with vmcnt(0))
  const bool want_rs = a_rowsum != nullptr;
  float rsl[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) rsl[i] = 0.f;

  auto tile = [&](int64_t t, const float* cA, const float* cB, float* nA, float* nB) {
#ifndef ARX_ABL_NODMA
    if (t + PD < nt) dma(t + PD, nA, nB);
#endif
    // operands of step s+1 are read from LDS before the MFMAs of step s issue (pinned with
    // sched_barrier: hipcc otherwise sinks the reads to just in front of their use and the wave
    // sits in s_waitcnt lgkmcnt for an LDS round trip per step -- PMC SQ_WAIT_ANY 27-37%)
    auto load_ops = [&](int s, float (&av)[FM][4], float (&bv)[FN][4]) {
#ifdef ARX_ABL_NOLDS
      for (int i = 0; i < FM; ++i) for (int j = 0; j < 4; ++j) av[i][j] = (float)(s + lane);
      for (int i = 0; i < FN; ++i) for (int j = 0; j < 4; ++j) bv[i][j] = (float)(s - lane);
      return;
#endif
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + l31;
        if (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(
              cA + r * kBK + (((2 * s + lhi) ^ ((r >> 1) & 7)) * 4));
          av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) av[i][j] = cA[(8 * s + 4 * lhi + j) * BM + r];
        }
      }
#pragma unroll
      for (int jn = 0; jn < FN; ++jn)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bv[jn][j] = cB[(8 * s + 4 * lhi + j) * kBN + wn * (kBN / 2) + jn * 32 + l31];
    };
    float av[FM][4], bv[FN][4];
    load_ops(0, av, bv);
#pragma unroll
    for (int s = 0; s < kBK / 8; ++s) {
      float nav[FM][4], nbv[FN][4];
      if (s + 1 < kBK / 8) load_ops(s + 1, nav, nbv);
      __builtin_amdgcn_sched_barrier(0);
      if (want_rs) {
#pragma unroll
        for (int i = 0; i < FM; ++i) rsl[i] += (av[i][0] + av[i][1]) + (av[i][2] + av[i][3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int jn = 0; jn < FN; ++jn)
#ifdef ARX_ABL_NOMFMA
            acc[i][jn][0] += av[i][j] * bv[jn][j];
#else
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], bv[jn][j], acc[i][jn], 0, 0, 0);
#endif
      if (s + 1 < kBK / 8) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) av[i][j] = nav[i][j];
#pragma unroll
        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[jn][j] = nbv[jn][j];
      }
    }
    // this wave's pieces of tile t+1 have landed once at most the NP pieces of tile t+2 are
    // outstanding (vmcnt retires in order); the barrier then covers every wave's pieces and
    // frees the stage just read.  (__syncthreads() would drain vmcnt(0): no overlap.)
    if (PD > 1 && t + PD < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * (NPA + NPB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef ARX_ABL_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
  };

  if (nt > 0) {
    dma(0, sA0, sB0);
    if (STAGES == 3 && nt > 1) {
      dma(1, sA1, sB1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA + NPB) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (STAGES == 3) {
      for (int64_t t = 0; t < nt; t += 3) {
        tile(t, sA0, sB0, sA2, sB2);
        if (t + 1 < nt) tile(t + 1, sA1, sB1, sA0, sB0);
        if (t + 2 < nt) tile(t + 2, sA2, sB2, sA1, sB1);
      }
    } else {
      for (int64_t t = 0; t < nt; t += 2) {
        tile(t, sA0, sB0, sA1, sB1);
        if (t + 1 < nt) tile(t + 1, sA1, sB1, sA0, sB0);
      }
    }
  }

  if (want_rs) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const float tot = rsl[i] + __shfl_xor(rsl[i], 32, 64);     // the two k halves of the row
      const int64_t row = m0 + wm * (BM / 2) + i * 32 + l31;
      if (wn == 0 && lhi == 0 && row < M) {
        if (rowsum_partial) rowsum_partial[(int64_t)bz * M + row] = tot;
        else a_rowsum[row] = tot;
      }
    }
  }
  // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      const int64_t col = wn * (kBN / 2) + jn * 32 + l31;
      if (col >= N) continue;
      const float bias = (col_bias && !partial) ? col_bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (row >= M) continue;
        float v = acc[i][jn][e];
        if (partial) {
          partial[((int64_t)bz * M + row) * N + col] = v;
        } else {
          v *= alpha;
          if (beta != 0.f) v += beta * C[row * ldc + col];
          C[row * ldc + col] = v + bias;
        }
      }
    }
}

}  // namespace

// Shapes: op(B) n-contiguous (transB == 0), 4 <= N <= 128, N % 4 == 0, M % 4 == 0 (M >= 4),
// K % 32 == 0, 16-byte aligned operands.  Anything else: ARX_EUNSUPPORTED (caller falls back).
bool gemm_dma_supported(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                        int64_t lda, const float* B, int64_t ldb) {
  if (transB) return false;
  if (N <= 32 || N > kBNMax || (N % 4) || M < 64 || (M % 4) || K < 64 || (K % kBK)) return false;   // N <= 32: half the 64-wide tile would be padding
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (lda % 4) || (ldb % 4))
    return false;
  (void)transA;
  return true;
}

void gemm_dma_plan(int64_t M, int64_t N, int64_t K, int* bm, int* splits, int64_t* kchunk) {
  (void)N;
  const int cus = cu_count();
  const int64_t tiles128 = ceil_div(M, (int64_t)128);
  if (tiles128 >= cus) {
    *bm = 128;
    *splits = 1;
    *kchunk = K;
    return;
  }
  *bm = 64;
  const int64_t tiles = ceil_div(M, (int64_t)64);
  // one workgroup per CU: the kernels are bound by the per-CU LDS-DMA rate (~12 B/clk), a
  // second resident workgroup only adds split-K partial traffic (measured: dI at B=16384
  // 55 us with 32 splits, 50 us with 16; dU 52 us with 2 splits, 48 us with none)
  int64_t s = ceil_div((int64_t)cus, tiles);
  const int64_t maxs = K / 256 > 0 ? K / 256 : 1;       // keep >= 256 of K per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int64_t chunk = ceil_div(ceil_div(K, s), (int64_t)kBK) * kBK;
  *kchunk = chunk;
  *splits = (int)ceil_div(K, chunk);
}

int gemm_dma_launch(int transA, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                    int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc,
                    const float* col_bias, float* partial, int bm, int splits, int64_t kchunk,
                    float* a_rowsum, float* rowsum_partial, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(M, (int64_t)bm), 1, (unsigned)splits);
  float* part = splits > 1 ? partial : nullptr;
  float* rsp = splits > 1 ? rowsum_partial : nullptr;
#define ARX_DMA_GO(AKC, BM_, BN_)                                                              \
  k_gemm_dma<AKC, BM_, BN_><<<grid, 256, 0, s>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                                 col_bias, part, kchunk, a_rowsum, rsp, NoPair{})
  const bool narrow = N <= 64;                   // LSTM / d=64 shapes: 64-column tile
  if (!transA) {
    if (bm == 128) { if (narrow) ARX_DMA_GO(true, 128, 64); else ARX_DMA_GO(true, 128, 128); }
    else { if (narrow) ARX_DMA_GO(true, 64, 64); else ARX_DMA_GO(true, 64, 128); }
  } else {
    if (bm == 128) { if (narrow) ARX_DMA_GO(false, 128, 64); else ARX_DMA_GO(false, 128, 128); }
    else { if (narrow) ARX_DMA_GO(false, 64, 64); else ARX_DMA_GO(false, 64, 128); }
  }
#undef ARX_DMA_GO
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

// The paired TN product: always through partials (the reduce writes the transposed result).
int gemm_dma_launch_tn_pair(int64_t M, int64_t N, int n1, int64_t K, const float* A, int64_t lda,
                            const float* B1, int64_t ldb1, const float* B2, int64_t ldb2,
                            int64_t shift, float* partial, int bm, int splits, int64_t kchunk,
                            float* a_rowsum, float* rowsum_partial, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(M, (int64_t)bm), 1, (unsigned)splits);
  const PairB pb{B2, ldb2, shift, n1};
#define ARX_DMA_GO(BM_, BN_)                                                                   \
  k_gemm_dma<false, BM_, BN_, true><<<grid, 256, 0, s>>>(M, N, K, 1.f, A, lda, B1, ldb1, 0.f,  \
                                                         nullptr, 0, nullptr, partial, kchunk, \
                                                         a_rowsum, rowsum_partial, pb)
  const bool narrow = N <= 64;
  if (bm == 128) { if (narrow) ARX_DMA_GO(128, 64); else ARX_DMA_GO(128, 128); }
  else { if (narrow) ARX_DMA_GO(64, 64); else ARX_DMA_GO(64, 128); }
#undef ARX_DMA_GO
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // namespace arx
