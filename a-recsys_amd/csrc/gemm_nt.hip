// gemm_nt.hip -- the scorer GEMM  logits[M,N] = U[M,K] . I[N,K]^T + b[N]  for K = d <= 128
// (embed_attribute.py:171,188-193: tf.matmul(user, item_pool, transpose_b=True) + bias).
//
// K is the embedding width (32..128), so a generic K-loop kernel spends most of its time
// in prologue / epilogue: at M=16384, N=1024, K=128 the tiled kernel in gemm.hip reaches
// 59 TF (37% of the fp32 MFMA peak).  Here the whole K extent of the A operand lives in
// REGISTERS for the lifetime of a workgroup:
//   * a workgroup owns 128 rows of U (4 waves x 32 rows) and a range of item-pool tiles;
//   * each lane keeps its row's K/2 values in K/8 float4 registers, laid out so that one
//     16-byte LDS read of the B operand feeds four v_mfma_f32_32x32x2_f32 (the k index a
//     lane supplies is arbitrary as long as A and B agree: lane l holds k = 8s + 4*(l/32) + j
//     for MFMA j of step s);
//   * pool tiles (64 items x K) stream into a double-buffered LDS image by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write pass, and -- unlike
//     register-staged loads, which hipcc drained with s_waitcnt vmcnt(0) AHEAD of the MFMA
//     block because the accumulators wanted the same registers -- the copy of tile t+1
//     stays in flight while tile t is multiplied);
//   * the DMA writes lane-linear (wave base + 16 B x lane), so the image is unpadded and
//     bank conflicts are avoided by an XOR swizzle applied to the SOURCE chunk index and
//     again on the ds_read_b128 side: 16 B chunk c of pool row r lives at slot c ^ swz(r);
//   * 64 KB of LDS and ~200 VGPRs => two workgroups per CU.
#include "common.h"

namespace arx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kNtBM = 128;   // rows per workgroup
constexpr int kNtBN = 64;    // pool rows (output columns) per tile

template <int KT>
__device__ __forceinline__ int nt_swz(int r) {
  // chunks per row: 32 (K=128), 16 (K=64): 16 consecutive rows must hit 16 distinct 16-B
  // slots of the 256-B bank row; 8 (K=32): rows are 128 B, pairs of rows share a bank row.
  return (KT >= 64) ? (r & 15) : ((r >> 1) & 7);
}

// FILTER (round 5, the recommend path: hmf_model.py:154 tf.nn.top_k over the FULL vocabulary): no C.  A logit leaves
// the kernel only if it beats its row's threshold thr[row] (the k-th best of the columns scored so far): it is appended
// to the row's candidate segment of this workgroup's column range, cand[row][part * capp ..), in ascending column
// order -- a wave owns its 32 rows, keeps their 16 + 16 counters in registers and places a tile's survivors by ballot
// + prefix count, so the lists are deterministic, need no atomics, and POSITION order is COLUMN order (what
// arx_topk_chunk's tie rule needs).  A full segment raises *overflow (the caller falls back to the chunked path).
// lse_part (nullable): also the log-sum-exp of the row's logits over the workgroup's column range, lse_part[row * ldl +
// part] -- the softmax normaliser of seqModel.py:514-517 top_k(softmax(logits)) without the logits: every lane keeps a
// running (max, sum exp(v - max)) per row over ITS columns, the 32 lanes of a row meet once at the end.
struct NtFilter {
  const float* thr; int64_t ldthr;
  float* cand_v; int32_t* cand_i; int64_t ldcand;
  int capp; int32_t col_base;
  int* overflow;
  float* lse_part; int64_t ldl;
  // the evaluation losses of a sampled-loss model over the FULL vocabulary (hmf_model.py:130,144, seqModel.py:510)
  // without the logits: relu_part[row * ldl + part] = sum over the range of relu(v - tsc[row] + 1) (WMRB, :605-618);
  // thr == nullptr: no candidate lists at all (arx_gemm_nt_eval_parts)
  const float* tsc; float* relu_part;
};

// MODE (FILTER kernels): bit 0 candidate lists (thr), bit 1 lse_part, bit 2 relu_part -- a template parameter, not a
// run-time switch: every feature is 32 registers of per-row state and the plain kernel already holds 246
constexpr int kNtTopk = 1, kNtLse = 2, kNtRelu = 4;

template <int KT, int MODE = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_areg(
    int64_t M, int64_t N, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
    int64_t ldb, float alpha, float* __restrict__ C, int64_t ldc,
    const float* __restrict__ col_bias, int tiles_per_block, int nsplit, NtFilter flt) {
  constexpr bool FILTER = MODE != 0;
  constexpr bool want_topk = (MODE & kNtTopk) != 0, want_lse = (MODE & kNtLse) != 0, want_relu = (MODE & kNtRelu) != 0;
  constexpr int NS = KT / 8;                  // steps of 4 MFMAs
  constexpr int CPR = KT / 4;                 // 16-B chunks per pool row
  constexpr int NLB = kNtBN * CPR / 256;      // DMA pieces per thread per tile
  static_assert(NLB >= 1, "tile too small");
  // two separate arrays (not sB[2][..]): the compiler must see that the DMA destination and
  // the buffer being read never alias, or it drains the DMA (vmcnt(0)) before the first ds_read
  __shared__ __attribute__((aligned(1024))) float sB0[kNtBN * KT];
  __shared__ __attribute__((aligned(1024))) float sB1[kNtBN * KT];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware order: the nsplit blocks that share one A panel get consecutive ids on one XCD
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int panel = bid / nsplit, part = bid % nsplit;
  const int64_t m0 = (int64_t)panel * kNtBM;
  const int64_t tiles_n = ceil_div(N, (int64_t)kNtBN);
  const int64_t t_beg = (int64_t)part * tiles_per_block;
  const int64_t t_end = min(tiles_n, t_beg + tiles_per_block);
  if (t_beg >= t_end) return;

  // LDS-DMA of one pool tile: piece f = threadIdx.x + 256 i lands at chunk f of the image;
  // it fetches chunk (f % CPR) ^ swz(row) of pool row n0 + f / CPR (rows past N clamped:
  // their columns are never stored).
  auto dma_b = [&](int64_t t, float* img) {
    const int64_t n0 = t * kNtBN;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int f = threadIdx.x + i * 256;
      const int r = f / CPR, slot = f % CPR;
      const int c = slot ^ nt_swz<KT>(r);
      const int64_t gr = min(n0 + r, N - 1);
      const float* src = B + gr * ldb + c * 4;
      float* dst = img + (f - lane) * 4;           // wave-uniform base; HW adds 16 B x lane
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  dma_b(t_beg, sB0);

  // ---- A operand: this lane's row, k = 8s + 4*lhi + {0..3} ----
  float4 a[NS];
  {
    const int64_t row = m0 + wave * 32 + l31;
    const bool ok = row < M;
    const float* ap = A + (ok ? row : 0) * lda + 4 * lhi;
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const float4*>(ap + 8 * s);
    if (!ok) {
#pragma unroll
      for (int s = 0; s < NS; ++s) a[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // FILTER: thresholds and list lengths of the lane's 16 rows (row = rbase + (e & 3) + 8 (e >> 2))
  // per-row CONSTANTS (thresholds, target scores) and the list lengths live in LDS -- read four rows at a time per
  // tile, the lengths only when a tile has a survivor; only the running sums are registers (the plain kernel holds
  // 246 VGPRs: with everything in registers the top-k form spilled 476 bytes per lane)
  __shared__ float s_th[want_topk ? kNtBM : 1];
  __shared__ float s_tq[want_relu ? kNtBM : 1];
  __shared__ int s_cnt[want_topk ? kNtBM : 1];
  float lm[want_lse ? 16 : 1], ls[want_lse ? 16 : 1], rsum[want_relu ? 16 : 1];
  bool ovf = false;
  if (FILTER) {
    if (threadIdx.x < kNtBM) {
      const int64_t row = m0 + threadIdx.x;
      if (want_topk) {
        s_th[threadIdx.x] = row < M ? flt.thr[row * flt.ldthr] : __builtin_inff();
        s_cnt[threadIdx.x] = 0;
      }
      if (want_relu) s_tq[threadIdx.x] = row < M ? flt.tsc[row] - 1.f : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (want_lse) {
        lm[e] = -__builtin_inff();
        ls[e] = 0.f;
      }
      if (want_relu) rsum[e] = 0.f;
    }
  }
  // bias of the first tile (later ones are fetched one tile ahead, behind the DMA issue)
  float bias0 = 0.f, bias1 = 0.f;
  if (col_bias) {
    const int64_t c0 = min(t_beg * kNtBN + l31, N - 1), c1 = min(t_beg * kNtBN + 32 + l31, N - 1);
    bias0 = col_bias[c0];
    bias1 = col_bias[c1];
  }
  __syncthreads();

  // read-side swizzle: chunk 2s + lhi of pool row (l31 [+32]); swz(l31) == swz(l31 + 32)
  const int sw = nt_swz<KT>(l31);
  auto tile = [&](int64_t t, const float* cur_img, float* nxt_img) {
    float nb0 = 0.f, nb1 = 0.f;
    if (t + 1 < t_end) {
      if (col_bias) {
        const int64_t c0 = min((t + 1) * kNtBN + l31, N - 1);
        const int64_t c1 = min((t + 1) * kNtBN + 32 + l31, N - 1);
        nb0 = col_bias[c0];
        nb1 = col_bias[c1];
      }
      dma_b(t + 1, nxt_img);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc0[e] = 0.f;
      acc1[e] = 0.f;
    }
    const float* b0p = cur_img + l31 * KT;
    const float* b1p = b0p + 32 * KT;
    // operands of step s+1 are fetched from LDS before the 8 MFMAs of step s issue (an LDS
    // round trip is ~128 cycles; issued right in front of their use they stalled ~20% of the
    // wave's cycles in s_waitcnt lgkmcnt -- PMC SQ_WAIT_ANY)
    float4 b0 = *reinterpret_cast<const float4*>(b0p + ((lhi ^ sw) * 4));
    float4 b1 = *reinterpret_cast<const float4*>(b1p + ((lhi ^ sw) * 4));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float4 n0 = b0, n1 = b1;
      if (s + 1 < NS) {
        const int off = ((2 * (s + 1) + lhi) ^ sw) * 4;
        n0 = *reinterpret_cast<const float4*>(b0p + off);
        n1 = *reinterpret_cast<const float4*>(b1p + off);
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the fetch ahead of this step's MFMAs (hipcc sinks it otherwise)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].w, b1.w, acc1, 0, 0, 0);
      b0 = n0;
      b1 = n1;
    }
    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int64_t n0 = t * kNtBN;
    const int64_t rbase = m0 + wave * 32 + 4 * lhi;
    if (FILTER) {
      const int rl0 = wave * 32 + 4 * lhi;                          // the lane's rows: rl0 + (e & 3) + 8 (e >> 2)
      float th[want_topk ? 16 : 1], tq[want_relu ? 16 : 1];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (want_topk) {
          const float4 t4 = *reinterpret_cast<const float4*>(&s_th[rl0 + 8 * g4]);
          th[4 * g4] = t4.x; th[4 * g4 + 1] = t4.y; th[4 * g4 + 2] = t4.z; th[4 * g4 + 3] = t4.w;
        }
        if (want_relu) {
          const float4 t4 = *reinterpret_cast<const float4*>(&s_tq[rl0 + 8 * g4]);
          tq[4 * g4] = t4.x; tq[4 * g4 + 1] = t4.y; tq[4 * g4 + 2] = t4.z; tq[4 * g4 + 3] = t4.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + j * 32 + l31;
        const float bias = (j == 0) ? bias0 : bias1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = alpha * (j == 0 ? acc0[e] : acc1[e]) + bias;
          if (want_relu && col < N) rsum[e] += fmaxf(v - tq[e], 0.f);
          if (want_lse && col < N) {                                // online (max, sum): ONE exp per logit
            const float dd = v - lm[e];
            const float ex = __expf(-fabsf(dd));                    // (first logit: dd = +inf, ex = 0: ls = 0 * 0 + 1)
            ls[e] = dd > 0.f ? ls[e] * ex + 1.f : ls[e] + ex;
            lm[e] = fmaxf(lm[e], v);
          }
          if (!want_topk) continue;
          const bool pred = col < N && v > th[e];
          const unsigned long long m = __ballot(pred);
          if (m == 0ull) continue;                                  // (the common case by far)
          const uint32_t mh = lhi ? (uint32_t)(m >> 32) : (uint32_t)m;
          const int rl = rl0 + (e & 3) + 8 * (e >> 2);
          // (the wave owns its rows: no other writer.  volatile: lane l31 == 0 stores the new length, the other lanes of
          // the half read it in the next (j, e) iteration -- a plain access would let the compiler carry a lane's
          // earlier load across a store that lane did not execute; advisor, round 5)
          const int c0 = *const_cast<volatile int*>(&s_cnt[rl]);
          const int pos = c0 + __popc(mh & ((1u << l31) - 1u));
          if (pred) {
            if (pos < flt.capp) {
              const int64_t at = (rbase + (e & 3) + 8 * (e >> 2)) * flt.ldcand + (int64_t)part * flt.capp + pos;
              flt.cand_v[at] = v;
              flt.cand_i[at] = flt.col_base + (int32_t)col;
            } else {
              ovf = true;
            }
          }
          if (l31 == 0 && mh) *const_cast<volatile int*>(&s_cnt[rl]) = c0 + __popc(mh);
        }
      }
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + j * 32 + l31;
      if (col < N) {
        const float bias = (j == 0) ? bias0 : bias1;
        float* cp = C + rbase * ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ro = (e & 3) + 8 * (e >> 2);
          if (rbase + ro < M)
            __builtin_nontemporal_store(alpha * (j == 0 ? acc0[e] : acc1[e]) + bias, cp + (int64_t)ro * ldc);
        }
      }
    }
    bias0 = nb0;
    bias1 = nb1;
    __syncthreads();          // tile t+1 landed (vmcnt(0) rides on the barrier); buffer cur free
  };
  for (int64_t t = t_beg; t < t_end; t += 2) {
    tile(t, sB0, sB1);
    if (t + 1 < t_end) tile(t + 1, sB1, sB0);
  }
  if (want_topk && ovf) *flt.overflow = 1;
  if (want_lse) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float mx = lm[e], sm = ls[e];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {                            // the row's 32 lanes (one half wave)
        const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(sm, o, 64);
        const float mn = fmaxf(mx, m2);
        sm = (mx == -__builtin_inff() ? 0.f : sm * __expf(mx - mn)) + (m2 == -__builtin_inff() ? 0.f : s2 * __expf(m2 - mn));
        mx = mn;
      }
      const int64_t row = m0 + wave * 32 + 4 * lhi + (e & 3) + 8 * (e >> 2);
      if (l31 == 0 && row < M) flt.lse_part[row * flt.ldl + part] = mx + __logf(sm);
    }
  }
  if (want_relu) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float sm = rsum[e];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
      const int64_t row = m0 + wave * 32 + 4 * lhi + (e & 3) + 8 * (e >> 2);
      if (l31 == 0 && row < M) flt.relu_part[row * flt.ldl + part] = sm;
    }
  }
}

}  // namespace

// Returns ARX_EUNSUPPORTED when the shape is not this kernel's (caller falls back to the
// tiled kernel): K in {32, 64, 128}, 16-byte aligned operands.
int gemm_nt_smallk(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                   const float* B, int64_t ldb, float* C, int64_t ldc, const float* col_bias,
                   hipStream_t s) {
  if (!(K == 32 || K == 64 || K == 128)) return ARX_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (lda % 4) || (ldb % 4))
    return ARX_EUNSUPPORTED;
  const int64_t panels = ceil_div(M, (int64_t)kNtBM);
  const int64_t tiles_n = ceil_div(N, (int64_t)kNtBN);
  // enough workgroups for two per CU; each keeps >= 1 tile
  int64_t nsplit = ceil_div((int64_t)cu_count() * 2, panels);
  if (nsplit > tiles_n) nsplit = tiles_n;
  if (nsplit < 1) nsplit = 1;
  const int64_t tpb = ceil_div(tiles_n, nsplit);
  nsplit = ceil_div(tiles_n, tpb);
  const int64_t grid = panels * nsplit;
  if (grid > 0x7fffffff) return ARX_EUNSUPPORTED;
  if (K == 128)
    k_gemm_nt_areg<128><<<(int)grid, 256, 0, s>>>(M, N, A, lda, B, ldb, alpha, C, ldc, col_bias,
                                                   (int)tpb, (int)nsplit, NtFilter{});
  else if (K == 64)
    k_gemm_nt_areg<64><<<(int)grid, 256, 0, s>>>(M, N, A, lda, B, ldb, alpha, C, ldc, col_bias,
                                                  (int)tpb, (int)nsplit, NtFilter{});
  else
    k_gemm_nt_areg<32><<<(int)grid, 256, 0, s>>>(M, N, A, lda, B, ldb, alpha, C, ldc, col_bias,
                                                  (int)tpb, (int)nsplit, NtFilter{});
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

static void nt_split(int64_t M, int64_t N, int64_t* tpb, int64_t* nsplit) {
  const int64_t panels = ceil_div(M, (int64_t)kNtBM);
  const int64_t tiles_n = ceil_div(N, (int64_t)kNtBN);
  int64_t ns = ceil_div((int64_t)cu_count() * 2, panels);
  if (ns > tiles_n) ns = tiles_n;
  if (ns < 1) ns = 1;
  *tpb = ceil_div(tiles_n, ns);
  *nsplit = ceil_div(tiles_n, *tpb);
}

template <int MODE>
static void nt_launch_mode(int64_t K, int64_t grid, hipStream_t s, int64_t M, int64_t N, const float* A, int64_t lda,
                           const float* Bm, int64_t ldb, const float* col_bias, int tpb, int ns, const NtFilter& f) {
  if (K == 128)
    k_gemm_nt_areg<128, MODE><<<(int)grid, 256, 0, s>>>(M, N, A, lda, Bm, ldb, 1.f, nullptr, 0, col_bias, tpb, ns, f);
  else if (K == 64)
    k_gemm_nt_areg<64, MODE><<<(int)grid, 256, 0, s>>>(M, N, A, lda, Bm, ldb, 1.f, nullptr, 0, col_bias, tpb, ns, f);
  else
    k_gemm_nt_areg<32, MODE><<<(int)grid, 256, 0, s>>>(M, N, A, lda, Bm, ldb, 1.f, nullptr, 0, col_bias, tpb, ns, f);
}

}  // namespace arx

using namespace arx;

extern "C" {

int arx_gemm_nt_topk_parts(int64_t M, int64_t N, int* parts) {
  ARX_CHECK_ARG(M > 0 && N > 0 && parts, "arx_gemm_nt_topk_parts: bad argument");
  int64_t tpb, ns;
  nt_split(M, N, &tpb, &ns);
  *parts = (int)ns;
  return ARX_OK;
}

int arx_gemm_nt_topk_filter(const float* A, int64_t lda, int64_t M, const float* Bm, int64_t ldb, int64_t N, int64_t K,
                            const float* col_bias, const float* thr, int64_t ldthr, int32_t col_base, float* cand_v,
                            int32_t* cand_i, int64_t ldcand, int capp, int* overflow, float* lse_part, int64_t ldl,
                            void* stream) {
  ARX_CHECK_ARG(A && Bm && thr && cand_v && cand_i && overflow && M > 0 && N > 0 && capp > 0,
                "arx_gemm_nt_topk_filter: bad argument");
  ARX_CHECK_ARG(K == 32 || K == 64 || K == 128, "arx_gemm_nt_topk_filter: K must be 32, 64 or 128");
  ARX_CHECK_ARG(!((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15) && lda % 4 == 0 &&
                    ldb % 4 == 0,
                "arx_gemm_nt_topk_filter: operands must be 16-byte aligned");
  int64_t tpb, ns;
  nt_split(M, N, &tpb, &ns);
  ARX_CHECK_ARG(ns * capp <= ldcand, "arx_gemm_nt_topk_filter: candidate rows too short (parts * capp > ldcand)");
  const int64_t grid = ceil_div(M, (int64_t)kNtBM) * ns;
  ARX_CHECK_ARG(grid <= 0x7fffffff, "arx_gemm_nt_topk_filter: grid too large");
  ARX_CHECK_ARG(!lse_part || ldl >= ns, "arx_gemm_nt_topk_filter: lse_part rows too short (ldl < parts)");
  const NtFilter f{thr, ldthr, cand_v, cand_i, ldcand, capp, col_base, overflow, lse_part, ldl, nullptr, nullptr};
  hipStream_t s = as_stream(stream);
  if (lse_part)
    nt_launch_mode<kNtTopk | kNtLse>(K, grid, s, M, N, A, lda, Bm, ldb, col_bias, (int)tpb, (int)ns, f);
  else
    nt_launch_mode<kNtTopk>(K, grid, s, M, N, A, lda, Bm, ldb, col_bias, (int)tpb, (int)ns, f);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gemm_nt_eval_parts(const float* A, int64_t lda, int64_t M, const float* Bm, int64_t ldb, int64_t N, int64_t K,
                           const float* col_bias, const float* tscore, float* lse_part, float* relu_part, int64_t ldl,
                           void* stream) {
  ARX_CHECK_ARG(A && Bm && M > 0 && N > 0 && (lse_part || relu_part), "arx_gemm_nt_eval_parts: bad argument");
  ARX_CHECK_ARG(!relu_part || tscore, "arx_gemm_nt_eval_parts: the margin sums need the target scores");
  ARX_CHECK_ARG(K == 32 || K == 64 || K == 128, "arx_gemm_nt_eval_parts: K must be 32, 64 or 128");
  ARX_CHECK_ARG(!((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15) && lda % 4 == 0 &&
                    ldb % 4 == 0,
                "arx_gemm_nt_eval_parts: operands must be 16-byte aligned");
  int64_t tpb, ns;
  nt_split(M, N, &tpb, &ns);
  ARX_CHECK_ARG(ldl >= ns, "arx_gemm_nt_eval_parts: part rows too short (ldl < parts)");
  const int64_t grid = ceil_div(M, (int64_t)kNtBM) * ns;
  ARX_CHECK_ARG(grid <= 0x7fffffff, "arx_gemm_nt_eval_parts: grid too large");
  const NtFilter f{nullptr, 0, nullptr, nullptr, 0, 0, 0, nullptr, lse_part, ldl, tscore, relu_part};
  hipStream_t s = as_stream(stream);
  if (lse_part && relu_part)
    nt_launch_mode<kNtLse | kNtRelu>(K, grid, s, M, N, A, lda, Bm, ldb, col_bias, (int)tpb, (int)ns, f);
  else if (lse_part)
    nt_launch_mode<kNtLse>(K, grid, s, M, N, A, lda, Bm, ldb, col_bias, (int)tpb, (int)ns, f);
  else
    nt_launch_mode<kNtRelu>(K, grid, s, M, N, A, lda, Bm, ldb, col_bias, (int)tpb, (int)ns, f);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
