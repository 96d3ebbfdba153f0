// gemm.hip -- K4: fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// The scorer of A-RecSys (embed_attribute.py:171,188-193,205) in embedding-space
// form is three small GEMMs per step: logits = U.I^T + b (NT), dU = dL.I (NN),
// dI = dL^T.U (TN).  fp32 in / fp32 accumulate is required (parity 1e-4 against
// the TF fp32 graph; gfx950 has no TF32).  The f32 MFMA runs at the fp32 vector
// peak (157 TF, 64 cycles per 32x32x2), so feeding it from LDS is cheap; what
// matters for these shapes is filling 256 CUs: 128x128 tiles when the output is
// large, otherwise 64x64 tiles with deterministic split-K (partials in a caller
// workspace, summed in a fixed order by a second kernel).
//
// LDS images (per operand, chosen by how the operand is stored in HBM):
//   K-contiguous storage  ([rows][k]) : X[rows][BK+1]  -- odd stride => the MFMA
//        operand read (32 rows at fixed k) hits 32 distinct banks;
//   MN-contiguous storage ([k][rows]) : X[BK][rows+4]  -- the operand read is 32
//        consecutive floats, conflict-free; stores are aligned ds_write_b128.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace arx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROWS, int BK, bool KC>
struct LdsImg {
  static constexpr int kStride = KC ? (BK + 1) : (ROWS + 4);
  static constexpr int kSize = KC ? ROWS * (BK + 1) : BK * (ROWS + 4);
  __device__ static __forceinline__ int idx(int r, int k) {
    return KC ? r * kStride + k : k * kStride + r;
  }
};

// Loads 4 consecutive floats from p (elements [0,4) valid where i < nvalid).
__device__ __forceinline__ float4 load4_guard(const float* p, int nvalid, bool vec_ok) {
  if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) v.x = p[0];
  if (nvalid > 1) v.y = p[1];
  if (nvalid > 2) v.z = p[2];
  if (nvalid > 3) v.w = p[3];
  return v;
}

// Stage one operand tile (ROWS x BK) from HBM into registers.
//   KC : element (r,k) at X[(r0+r)*ld + k0+k]   -> float4 along k
//   !KC: element (r,k) at X[(k0+k)*ld + r0+r]   -> float4 along r
template <int ROWS, int BK, bool KC, int NL, bool FAST>
__device__ __forceinline__ void tile_load(const float* __restrict__ X, int64_t ld, int64_t r0,
                                          int64_t R, int64_t k0, int64_t kend, bool vec_ok,
                                          float4 (&reg)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int f = threadIdx.x + i * 256;
    if (FAST) {
      // interior tile, aligned: plain 16-B loads with no select on the result, so the
      // compiler can leave them in flight across the MFMA block (the guarded form forces
      // s_waitcnt vmcnt(0) right behind the loads: 3-4x slower, measured)
      if (KC) {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        reg[i] = *reinterpret_cast<const float4*>(X + (r0 + r) * ld + k0 + kq * 4);
      } else {
        const int k = f / (ROWS / 4), rq = f % (ROWS / 4);
        reg[i] = *reinterpret_cast<const float4*>(X + (k0 + k) * ld + r0 + rq * 4);
      }
    } else if (KC) {
      const int r = f / (BK / 4), kq = f % (BK / 4);
      const int64_t gr = r0 + r, gk = k0 + kq * 4;
      int nv = (gr < R) ? (int)min((int64_t)4, kend - gk) : 0;
      reg[i] = (nv > 0) ? load4_guard(X + gr * ld + gk, nv, vec_ok) : make_float4(0, 0, 0, 0);
    } else {
      const int k = f / (ROWS / 4), rq = f % (ROWS / 4);
      const int64_t gk = k0 + k, gr = r0 + rq * 4;
      int nv = (gk < kend) ? (int)min((int64_t)4, R - gr) : 0;
      reg[i] = (nv > 0) ? load4_guard(X + gk * ld + gr, nv, vec_ok) : make_float4(0, 0, 0, 0);
    }
  }
}

template <int ROWS, int BK, bool KC, int NL>
__device__ __forceinline__ void tile_store(float* __restrict__ S, const float4 (&reg)[NL]) {
  using Img = LdsImg<ROWS, BK, KC>;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int f = threadIdx.x + i * 256;
    if (KC) {
      const int r = f / (BK / 4), kq = f % (BK / 4);
      float* p = S + Img::idx(r, kq * 4);
      p[0] = reg[i].x;
      p[1] = reg[i].y;
      p[2] = reg[i].z;
      p[3] = reg[i].w;
    } else {
      const int k = f / (ROWS / 4), rq = f % (ROWS / 4);
      *reinterpret_cast<float4*>(S + Img::idx(rq * 4, k)) = reg[i];
    }
  }
}

// C (or split-K partial) = op(A).op(B) over this block's k range.
template <int BM, int BN, int BK, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm_f32(
    int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float beta, float* __restrict__ C, int64_t ldc,
    const float* __restrict__ col_bias, float* __restrict__ partial, int64_t kchunk,
    int vec_a, int vec_b, int tiles_n, float* __restrict__ a_rowsum,
    float* __restrict__ rowsum_partial) {
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 32, FN = TN / 32;
  constexpr int NLA = BM * BK / 4 / 256, NLB = BN * BK / 4 / 256;
  static_assert(NLA >= 1 && NLB >= 1, "tile too small for 256 threads");
  using ImgA = LdsImg<BM, BK, A_KC>;
  using ImgB = LdsImg<BN, BK, B_KC>;
  __shared__ __attribute__((aligned(16))) float sA[2][ImgA::kSize];
  __shared__ __attribute__((aligned(16))) float sB[2][ImgB::kSize];

  // XCD-aware tile order: consecutive blocks of one XCD (b % 8) walk tiles that
  // share the same A row-panel, so the panel stays in that XCD's L2.
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool want_rs = (a_rowsum != nullptr) && (tn == 0) && ((int)threadIdx.x < BM);
  float rs = 0.f;
  const int64_t nt = (kend > kbeg) ? ceil_div(kend - kbeg, (int64_t)BK) : 0;
  const bool fast = vec_a && vec_b && (m0 + BM <= M) && (n0 + BN <= N) &&
                    (kend > kbeg) && ((kend - kbeg) % BK == 0);
  auto mainloop = [&](auto fast_c) {
  constexpr bool FAST = decltype(fast_c)::value;
  float4 ra[NLA], rb[NLB];
  if (nt > 0) {
    tile_load<BM, BK, A_KC, NLA, FAST>(A, lda, m0, M, kbeg, kend, vec_a, ra);
    tile_load<BN, BK, B_KC, NLB, FAST>(B, ldb, n0, N, kbeg, kend, vec_b, rb);
    tile_store<BM, BK, A_KC, NLA>(sA[0], ra);
    tile_store<BN, BK, B_KC, NLB>(sB[0], rb);
  }
  __syncthreads();
  for (int64_t t = 0; t < nt; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < nt) {
      const int64_t k0 = kbeg + (t + 1) * BK;
      tile_load<BM, BK, A_KC, NLA, FAST>(A, lda, m0, M, k0, kend, vec_a, ra);
      tile_load<BN, BK, B_KC, NLB, FAST>(B, ldb, n0, N, k0, kend, vec_b, rb);
    }
    const float* a_s = sA[cur];
    const float* b_s = sB[cur];
    if (want_rs) {   // sum_k op(A)[m,k] rides along: the tile is already in LDS
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) rs += a_s[ImgA::idx(threadIdx.x, kk)];
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[FM], bv[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) av[i] = a_s[ImgA::idx(wm * TM + i * 32 + l31, kk + lhi)];
#pragma unroll
      for (int j = 0; j < FN; ++j) bv[j] = b_s[ImgB::idx(wn * TN + j * 32 + l31, kk + lhi)];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      tile_store<BM, BK, A_KC, NLA>(sA[cur ^ 1], ra);
      tile_store<BN, BK, B_KC, NLB>(sB[cur ^ 1], rb);
    }
    __syncthreads();
  }
  };
  if (fast) mainloop(std::true_type{});
  else mainloop(std::false_type{});

  if (want_rs && m0 + threadIdx.x < M) {
    if (rowsum_partial) rowsum_partial[(int64_t)blockIdx.z * M + m0 + threadIdx.x] = rs;
    else a_rowsum[m0 + threadIdx.x] = rs;
  }

  // epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int64_t col = n0 + wn * TN + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (row >= M) continue;
        float v = acc[i][j][e];
        if (partial) {
          partial[((int64_t)blockIdx.z * M + row) * N + col] = v;
        } else {
          v *= alpha;
          if (beta != 0.f) v += beta * C[row * ldc + col];
          if (col_bias) v += col_bias[col];
          C[row * ldc + col] = v;
        }
      }
    }
}

// fixed-order reduction of split-K partials + epilogue
__global__ __launch_bounds__(256) void k_splitk_reduce(
    const float* __restrict__ partial, int splits, int64_t M, int64_t N, float alpha, float beta,
    float* __restrict__ C, int64_t ldc, const float* __restrict__ col_bias,
    const float* __restrict__ rowsum_partial, float* __restrict__ a_rowsum, int tr) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t total = M * N;
  if (a_rowsum) {
    for (int64_t m = i; m < M; m += stride) {
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += rowsum_partial[(int64_t)z * M + m];
      a_rowsum[m] = s;
    }
  }
  for (; i < total; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * total + i];
    const int64_t r = i / N, c = i % N;
    float* dst = tr ? C + c * ldc + r : C + r * ldc + c;      // tr: the result is stored as C^T [N, M]
    float v = alpha * s;
    if (beta != 0.f) v += beta * *dst;
    if (col_bias) v += col_bias[c];
    *dst = v;
  }
}

// Same reduction with the splits spread over 16 lanes per output element group: every thread has
// its (<= 4 at 64 splits) loads in flight at once instead of walking the splits one L2 round
// trip at a time (64 splits x [256 x 64]: 24 us -> a few us).  Fixed combine order => deterministic.
// Blocks [0, nb_main): 16 float4 of C each; blocks [nb_main, ...): 16 row sums each.
__global__ __launch_bounds__(256) void k_splitk_reduce_lanes(
    const float* __restrict__ partial, int splits, int64_t M, int64_t N, float alpha, float beta,
    float* __restrict__ C, int64_t ldc, const float* __restrict__ col_bias,
    const float* __restrict__ rowsum_partial, float* __restrict__ a_rowsum, int nb_main, int tr) {
  __shared__ float4 sh[16][17];
  const int cx = threadIdx.x & 15, sy = threadIdx.x >> 4;
  const int64_t total = M * N;
  if ((int)blockIdx.x < nb_main) {
    const int64_t i4 = (int64_t)blockIdx.x * 16 + cx;       // float4 index into [M*N]
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 * 4 < total) {
      const float4* p4 = reinterpret_cast<const float4*>(partial) + i4;
      const int64_t st4 = total >> 2;
      for (int z = sy; z < splits; z += 16) {
        const float4 v = p4[(int64_t)z * st4];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    sh[sy][cx] = a;
    __syncthreads();
    if (sy == 0 && i4 * 4 < total) {
      float4 t = sh[0][cx];
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = sh[k][cx];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      const int64_t e = i4 * 4, r = e / N, c = e % N;
      float4 o = make_float4(alpha * t.x, alpha * t.y, alpha * t.z, alpha * t.w);
      if (tr) {               // C^T [N, M]: four rows of the transposed result (beta = 0, no bias: host)
        float* d = C + c * ldc + r;
        d[0] = o.x; d[ldc] = o.y; d[2 * ldc] = o.z; d[3 * ldc] = o.w;
        return;
      }
      float4* dst = reinterpret_cast<float4*>(C + r * ldc + c);
      if (beta != 0.f) {
        const float4 old = *dst;
        o.x += beta * old.x; o.y += beta * old.y; o.z += beta * old.z; o.w += beta * old.w;
      }
      if (col_bias) {
        const float4 cb = *reinterpret_cast<const float4*>(col_bias + c);
        o.x += cb.x; o.y += cb.y; o.z += cb.z; o.w += cb.w;
      }
      *dst = o;
    }
  } else {
    float* shf = reinterpret_cast<float*>(sh);
    const int64_t m = (int64_t)(blockIdx.x - nb_main) * 16 + cx;
    float a = 0.f;
    if (m < M)
      for (int z = sy; z < splits; z += 16) a += rowsum_partial[(int64_t)z * M + m];
    shf[sy * 17 + cx] = a;
    __syncthreads();
    if (sy == 0 && m < M) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += shf[k * 17 + cx];
      a_rowsum[m] = t;
    }
  }
}

// picks the lane-parallel variant when the layout allows float4 traffic
static inline void launch_splitk_reduce(const float* partial, int splits, int64_t M, int64_t N,
                                        float alpha, float beta, float* C, int64_t ldc,
                                        const float* col_bias, const float* rowsum_partial,
                                        float* a_rowsum, hipStream_t s, int tr = 0) {
  const int64_t total = M * N;
  const bool v4 = (N % 4 == 0) && (tr || ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(partial) & 15) == 0) &&
                  (!col_bias || (reinterpret_cast<uintptr_t>(col_bias) & 15) == 0) && splits >= 8 &&
                  total / 64 + M / 16 + 2 < 0x7fffffff;
  if (v4) {
    const int nb_main = (int)ceil_div(total, (int64_t)64);
    const int nb_rs = a_rowsum ? (int)ceil_div(M, (int64_t)16) : 0;
    k_splitk_reduce_lanes<<<nb_main + nb_rs, 256, 0, s>>>(partial, splits, M, N, alpha, beta, C, ldc,
                                                          col_bias, rowsum_partial, a_rowsum, nb_main, tr);
    return;
  }
  int64_t g = ceil_div(total, 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_splitk_reduce<<<(int)g, 256, 0, s>>>(partial, splits, M, N, alpha, beta, C, ldc, col_bias,
                                         rowsum_partial, a_rowsum, tr);
}

struct GemmPlan {
  bool big;
  int splits;
  int64_t kchunk;
};

static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K) {
  GemmPlan p;
  const int cus = cu_count();
  const int64_t tiles_big = ceil_div(M, 128) * ceil_div(N, 128);
  p.big = tiles_big * 4 >= (int64_t)cus * 3;
  p.splits = 1;
  p.kchunk = K > 0 ? K : 1;
  if (!p.big) {
    const int64_t tiles = ceil_div(M, 64) * ceil_div(N, 64);
    int64_t want = ceil_div((int64_t)cus * 2, tiles);
    int64_t maxs = K / 128;  // keep >= 128 of K per split
    if (maxs < 1) maxs = 1;
    int64_t s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    int64_t chunk = ceil_div(ceil_div(K, s), 64) * 64;
    if (chunk < 64) chunk = 64;
    p.kchunk = chunk;
    p.splits = (int)ceil_div(K > 0 ? K : 1, chunk);
  }
  return p;
}

template <int BM, int BN, int BK>
static int launch_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                       const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                       float* C, int64_t ldc, const float* col_bias, float* partial,
                       const GemmPlan& p, hipStream_t s, float* a_rowsum, float* rowsum_partial) {
  const int tiles_m = (int)ceil_div(M, BM), tiles_n = (int)ceil_div(N, BN);
  dim3 grid(tiles_m * tiles_n, 1, p.splits);
  const int vec_a = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
  const int vec_b = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);
  const bool akc = (transA == 0), bkc = (transB != 0);
#define ARX_GEMM_LAUNCH(AKC, BKC)                                                         \
  k_gemm_f32<BM, BN, BK, AKC, BKC><<<grid, 256, 0, s>>>(M, N, K, alpha, A, lda, B, ldb,   \
                                                        beta, C, ldc, col_bias, partial,  \
                                                        p.kchunk, vec_a, vec_b, tiles_n,  \
                                                        a_rowsum, rowsum_partial)
  if (akc && bkc) ARX_GEMM_LAUNCH(true, true);
  else if (akc && !bkc) ARX_GEMM_LAUNCH(true, false);
  else if (!akc && bkc) ARX_GEMM_LAUNCH(false, true);
  else ARX_GEMM_LAUNCH(false, false);
#undef ARX_GEMM_LAUNCH
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0) return 0;
  GemmPlan p = plan_gemm(M, N, K);
  size_t need = 0;
  if (p.splits > 1) need = (size_t)p.splits * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
  if (N <= 128 && K % 32 == 0) {   // the DMA kernels (gemm_dma.hip) plan their own split-K
    int bm, sp;
    int64_t kc;
    gemm_dma_plan(M, N, K, &bm, &sp, &kc);
    if (sp > 1) {
      const size_t nd = (size_t)sp * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
      if (nd > need) need = nd;
    }
  }
  return need;
}

int arx_gemm_f32_rowsum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                        const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                        float* C, int64_t ldc, const float* col_bias, float* a_rowsum,
                        void* workspace, size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "arx_gemm_f32: negative dimension");
  if (M == 0 || N == 0) return ARX_OK;
  ARX_CHECK_ARG(A && B && C, "arx_gemm_f32: null pointer");
  ARX_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                "arx_gemm_f32: leading dimension too small");
  hipStream_t s = as_stream(stream);
  if (!transA && transB && beta == 0.f && !a_rowsum) {   // the scorer shape: K = embedding width
    const int rc_nt = gemm_nt_smallk(M, N, K, alpha, A, lda, B, ldb, C, ldc, col_bias, s);
    if (rc_nt != ARX_EUNSUPPORTED) return rc_nt;
  }
  if (gemm_dma_supported(transA, transB, M, N, K, A, lda, B, ldb)) {
    int bm, sp;
    int64_t kc;
    gemm_dma_plan(M, N, K, &bm, &sp, &kc);
    float* part = nullptr;
    float* rsp = nullptr;
    if (sp > 1) {
      const size_t need = (size_t)sp * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
      if (!workspace || workspace_bytes < need) {
        set_error("arx_gemm_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
        return ARX_EWORKSPACE;
      }
      part = reinterpret_cast<float*>(workspace);
      if (a_rowsum) rsp = part + (size_t)sp * (size_t)M * (size_t)N;
    }
    int rc_d = gemm_dma_launch(transA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, col_bias, part,
                               bm, sp, kc, a_rowsum, rsp, s);
    if (rc_d) return rc_d;
    if (part) {
      launch_splitk_reduce(part, sp, M, N, alpha, beta, C, ldc, col_bias, rsp, a_rowsum, s);
      ARX_CHECK_LAUNCH();
    }
    return ARX_OK;
  }
  GemmPlan p = plan_gemm(M, N, K);
  float* partial = nullptr;
  float* rs_partial = nullptr;
  if (p.splits > 1) {
    size_t need = arx_gemm_f32_workspace_bytes(M, N, K);
    if (!workspace || workspace_bytes < need) {
      set_error("arx_gemm_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
      return ARX_EWORKSPACE;
    }
    partial = reinterpret_cast<float*>(workspace);
    if (a_rowsum) rs_partial = partial + (size_t)p.splits * (size_t)M * (size_t)N;
  }
  int rc;
#define ARX_GO(BM_, BN_, BK_)                                                                   \
  rc = launch_gemm<BM_, BN_, BK_>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                  col_bias, partial, p, s, a_rowsum, rs_partial)
  if (p.big) {
    ARX_GO(128, 128, 16);
  } else {
    ARX_GO(64, 64, 16);
  }
#undef ARX_GO
  if (rc) return rc;
  if (partial) {
    launch_splitk_reduce(partial, p.splits, M, N, alpha, beta, C, ldc, col_bias, rs_partial, a_rowsum, s);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

int arx_gemm_f32_steps_tn(int64_t steps, int64_t M, int64_t N, int64_t Kb, const float* A,
                          int64_t lda, const float* B, int64_t ldb, float* C_steps,
                          float* rowsum_steps, float beta, float* C_sum, int64_t ldc,
                          float* rowsum_sum, void* stream) {
  ARX_CHECK_ARG(steps > 0 && M > 0 && N > 0 && Kb > 0, "arx_gemm_f32_steps_tn: bad size");
  ARX_CHECK_ARG(A && B && C_steps, "arx_gemm_f32_steps_tn: null pointer");
  ARX_CHECK_ARG(Kb % 16 == 0, "arx_gemm_f32_steps_tn: rows per step must be a multiple of 16");
  ARX_CHECK_ARG(lda >= M && ldb >= N, "arx_gemm_f32_steps_tn: leading dimension too small");
  ARX_CHECK_ARG(!(rowsum_sum && !rowsum_steps), "arx_gemm_f32_steps_tn: rowsum_sum needs rowsum_steps");
  hipStream_t s = as_stream(stream);
  GemmPlan p;
  p.big = false;
  p.splits = (int)steps;
  p.kchunk = Kb;
  int rc;
  if (steps > 1 && Kb % 32 == 0 &&
      gemm_dma_supported(1, 0, M, N, steps * Kb, A, lda, B, ldb)) {
    // one split-K slice per step on the LDS-DMA kernel: the "partials" ARE the per-step products
    rc = gemm_dma_launch(1, M, N, steps * Kb, 1.f, A, lda, B, ldb, 0.f, C_steps, N, nullptr, C_steps,
                         (M % 128 != 0) ? 64 : 128, (int)steps, Kb, rowsum_steps, rowsum_steps, s);
  } else
  rc = launch_gemm<64, 64, 16>(1, 0, M, N, steps * Kb, 1.f, A, lda, B, ldb, 0.f, C_steps, N,
                               nullptr, C_steps, p, s, rowsum_steps ? rowsum_steps : nullptr,
                               rowsum_steps);
  if (rc) return rc;
  if (C_sum) {
    ARX_CHECK_ARG(ldc >= N, "arx_gemm_f32_steps_tn: ldc too small");
    launch_splitk_reduce(C_steps, (int)steps, M, N, 1.f, beta, C_sum, ldc, nullptr, rowsum_steps,
                         rowsum_sum, s);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

size_t arx_gemm_f32_tn_pair_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int bm, sp;
  int64_t kc;
  gemm_dma_plan(M, N, K, &bm, &sp, &kc);
  return (size_t)sp * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
}

int arx_gemm_f32_tn_pair(int64_t M, int64_t N1, int64_t N2, int64_t K, const float* A, int64_t lda,
                         const float* B1, int64_t ldb1, const float* B2, int64_t ldb2,
                         int64_t shift, float* Ct, int64_t ldct, float* a_rowsum, void* workspace,
                         size_t workspace_bytes, void* stream) {
  const int64_t N = N1 + N2;
  ARX_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0 && K > 0 && shift >= 0, "arx_gemm_f32_tn_pair: bad size");
  ARX_CHECK_ARG(A && B1 && B2 && Ct, "arx_gemm_f32_tn_pair: null pointer");
  ARX_CHECK_ARG(lda >= M && ldb1 >= N1 && ldb2 >= N2 && ldct >= M,
                "arx_gemm_f32_tn_pair: leading dimension too small");
  // the LDS-DMA kernel's shape class, with 16-byte pieces that never straddle the two operands
  if (!gemm_dma_supported(1, 0, M, N, K, A, lda, B1, ldb1) || (N1 % 4) || (N2 % 4) || (ldb2 % 4) ||
      (reinterpret_cast<uintptr_t>(B2) & 15)) {
    set_error("arx_gemm_f32_tn_pair: unsupported shape (M=%lld N=%lld+%lld K=%lld)", (long long)M,
              (long long)N1, (long long)N2, (long long)K);
    return ARX_EUNSUPPORTED;
  }
  int bm, sp;
  int64_t kc;
  gemm_dma_plan(M, N, K, &bm, &sp, &kc);
  const size_t need = (size_t)sp * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_error("arx_gemm_f32_tn_pair: workspace too small (%zu < %zu)", workspace_bytes, need);
    return ARX_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  float* rsp = a_rowsum ? part + (size_t)sp * (size_t)M * (size_t)N : nullptr;
  const int rc = gemm_dma_launch_tn_pair(M, N, (int)N1, K, A, lda, B1, ldb1, B2, ldb2, shift, part, bm,
                                         sp, kc, a_rowsum, rsp, s);
  if (rc) return rc;
  launch_splitk_reduce(part, sp, M, N, 1.f, 0.f, Ct, ldct, nullptr, rsp, a_rowsum, s, /*tr=*/1);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                 int64_t ldc, const float* col_bias, void* workspace, size_t workspace_bytes,
                 void* stream) {
  return arx_gemm_f32_rowsum(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                             col_bias, nullptr, workspace, workspace_bytes, stream);
}

}  // extern "C"
