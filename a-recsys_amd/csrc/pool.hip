// pool.hip -- score-space pooling over the bag of a multi-hot output feature:
// embed_attribute.py:194-200, get_prediction with output_feat 2 (tf.segment_max) and 3
// (score_max + log(1 + unsorted_segment_sum(exp(score - score_max)))).
//
// Unlike output_feat 1 (a mean, linear: the bag can be averaged in embedding space before the
// scorer GEMM) these poolings are taken over the per-TOKEN scores, so the pool's tokens are
// scored first (scores[r, q] = latent_r . E[tok_q] + b[tok_q]: the scorer GEMM over the gathered
// token rows) and reduced per bag here.  HBM-bound streaming kernels: one pass over [B, n_tok].
#include <math.h>

#include "common.h"

namespace arx {

namespace {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one thread per (row, bag); bags are contiguous column ranges [offs[j], offs[j+1])
__global__ __launch_bounds__(256) void k_segpool_fwd(const float* __restrict__ scores, int64_t lds,
                                                     const int32_t* __restrict__ offs, int64_t W,
                                                     int mode, const float* __restrict__ gmax,
                                                     float* __restrict__ out, int64_t ldo) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (j >= W) return;
  const int q0 = offs[j], q1 = offs[j + 1];
  const float* x = scores + r * lds;
  float res;
  if (mode == 2) {
    res = -INFINITY;
    for (int q = q0; q < q1; ++q) res = fmaxf(res, x[q]);
  } else {
    const float M = *gmax;
    float s = 0.f;
    for (int q = q0; q < q1; ++q) s += expf(x[q] - M);
    res = M + logf(1.f + s);
  }
  out[r * ldo + j] = res;
}

// one workgroup per row: d scores for every bag, pad columns zeroed; mode 3 also leaves the row's
// share of the gradient that flows through score_max (dout * (1 - s / (1 + s)) summed over bags)
__global__ __launch_bounds__(256) void k_segpool_bwd(const float* __restrict__ scores, int64_t lds,
                                                     const int32_t* __restrict__ offs, int64_t W,
                                                     int64_t cap, int mode, const float* __restrict__ gmax,
                                                     const float* __restrict__ out, int64_t ldo,
                                                     const float* __restrict__ dout, int64_t ldd,
                                                     float* __restrict__ dscores, int64_t ldds,
                                                     float* __restrict__ resid_rows) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* x = scores + r * lds;
  float* dx = dscores + r * ldds;
  float resid = 0.f;
  for (int64_t j = threadIdx.x; j < W; j += 256) {
    const int q0 = offs[j], q1 = offs[j + 1];
    const float o = out[r * ldo + j], g = dout[r * ldd + j];
    if (mode == 2) {          // tf.segment_max: the gradient goes to the arg-max entries, ties share it
      int cnt = 0;
      for (int q = q0; q < q1; ++q) cnt += (x[q] == o);
      const float share = cnt > 0 ? g / (float)cnt : 0.f;
      for (int q = q0; q < q1; ++q) dx[q] = (x[q] == o) ? share : 0.f;
    } else {
      const float M = *gmax;
      const float inv = expf(M - o);                 // 1 / (1 + s)
      for (int q = q0; q < q1; ++q) dx[q] = g * expf(x[q] - M) * inv;
      resid += g * inv;                              // 1 - s / (1 + s) = 1 / (1 + s)
    }
  }
  const int total = offs[W];
  for (int64_t q = total + threadIdx.x; q < cap; q += 256) dx[q] = 0.f;
  if (resid_rows) {
    resid = wsum64(resid);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = resid;
    __syncthreads();
    if (threadIdx.x == 0) resid_rows[r] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}

// running (max, first arg-max) of a [rows, cols] block: per-block candidates in fixed slots, the
// last-arriving block folds them in slot order into best[0] (value) / best_idx[0..1] (row, col)
constexpr int kMaxBlocks = 256;
// block partials {value [kMaxBlocks], position [kMaxBlocks]} and the arrival ticket: the caller's reduce scratch

__global__ __launch_bounds__(256) void k_max_argmax(const float* __restrict__ x, int64_t rows, int64_t cols,
                                                    int64_t ld, int64_t col_base, int first,
                                                    float* __restrict__ best, int32_t* __restrict__ best_idx,
                                                    float* mx_val, long long* mx_pos, unsigned int* mx_ticket) {
  __shared__ float sv[256];
  __shared__ long long sp[256];
  __shared__ bool s_last;
  const int64_t n = rows * cols;
  float bv = -INFINITY;
  long long bp = -1;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols, c = i % cols;
    const float v = x[r * ld + c];
    // order of preference: larger value, then smaller (global col, row) -- the table-major order of
    // the reference's [Vf, mb] innerp matrix
    const long long pos = (long long)(col_base + c) * rows + r;
    if (v > bv || (v == bv && pos < bp)) { bv = v; bp = pos; }
  }
  sv[threadIdx.x] = bv;
  sp[threadIdx.x] = bp;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v = sv[threadIdx.x + o];
      const long long p = sp[threadIdx.x + o];
      if (p >= 0 && (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && (sp[threadIdx.x] < 0 || p < sp[threadIdx.x])))) {
        sv[threadIdx.x] = v;
        sp[threadIdx.x] = p;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&mx_val[blockIdx.x], sv[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mx_pos[blockIdx.x], sp[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    s_last = (__hip_atomic_fetch_add(mx_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  float v0 = first ? -INFINITY : best[0];
  long long p0 = first ? -1 : ((long long)best_idx[1] * rows + best_idx[0]);
  for (int b = 0; b < (int)gridDim.x; ++b) {
    const float v = __hip_atomic_load(&mx_val[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long p = __hip_atomic_load(&mx_pos[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p >= 0 && (v > v0 || (v == v0 && (p0 < 0 || p < p0)))) { v0 = v; p0 = p; }
  }
  best[0] = v0;
  best_idx[0] = (int32_t)(p0 % rows);
  best_idx[1] = (int32_t)(p0 / rows);
  __hip_atomic_store(mx_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the gradient that flows through score_max = reduce_max(innerp) lands on the arg-max element
// (row r*, table row v*): d row(v*) += resid * U[r*], d bias(v*) += resid, dU[r*] += resid * E[v*]
__global__ void k_gmax_residual(const float* __restrict__ resid, const int32_t* __restrict__ idx,
                                const float* __restrict__ U, int64_t ldu, const float* __restrict__ Ev,
                                int d, float* __restrict__ row_grad, float* __restrict__ bias_grad,
                                float* __restrict__ dU, int64_t lddu) {
  const int c = threadIdx.x;
  const float g = *resid;
  const int64_t r = idx[0];
  if (c < d) {
    row_grad[c] = g * U[r * ldu + c];
    if (dU) dU[r * lddu + c] += g * Ev[c];
  }
  if (c == 0 && bias_grad) bias_grad[0] = g;
}

// arx_gmax_norm_corr: workgroup t, 256 threads.  M = the merged gradient row of table row v = vrows[t] (sum over the key
// list's entries naming v, in list order within a thread group, groups combined in fixed order), R = the rank-one rows
// that land on v (per_step: RG[t] alone; summed: every RG[t'] with vrows[t'] == v, and only the FIRST such t does the
// work).  corr[t] = 2 <M, R> + |R|^2 (+ the d = 1 analogue on the bias gradients).
__global__ __launch_bounds__(256) void k_gmax_norm_corr(const int32_t* __restrict__ keys, const int32_t* __restrict__ src,
                                                        const float* __restrict__ coef, int64_t n,
                                                        const float* __restrict__ X, int64_t ldx, int d, int per_step,
                                                        int64_t step_stride, const float* __restrict__ Xb,
                                                        int per_step_b, int64_t stepb_stride,
                                                        const int32_t* __restrict__ vrows, const float* __restrict__ RG,
                                                        int64_t ldrg, const float* __restrict__ RGb, int L,
                                                        float* __restrict__ corr) {
  __shared__ float sm[256];
  __shared__ float sr[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int v = vrows[t];
  float total = 0.f;
  for (int pass = 0; pass < 2; ++pass) {              // 0: the embedding rows (width d), 1: the bias cells (width 1)
    const float* Xp = pass == 0 ? X : Xb;
    const float* Rp = pass == 0 ? RG : RGb;
    if (!Xp || !Rp) continue;
    const int w = pass == 0 ? d : 1;
    const bool steps = pass == 0 ? per_step != 0 : per_step_b != 0;
    const int64_t xstep = pass == 0 ? step_stride : stepb_stride;
    const int64_t xld = pass == 0 ? ldx : 1, rld = pass == 0 ? ldrg : 1;
    bool first = true;                                // summed form: only the first step naming v works
    if (!steps)
      for (int t2 = 0; t2 < t; ++t2) first = first && vrows[t2] != v;
    float acc = 0.f;
    if (first) {
      const float* Xt = Xp + (steps ? (int64_t)t * xstep : 0);
      for (int c0 = 0; c0 < w; c0 += 256) {           // column block (d <= 1024)
        const int ng = w - c0 >= 256 ? 1 : 256 / (w - c0 < 1 ? 1 : (w - c0));   // thread groups over the key list
        const int wc = w - c0 >= 256 ? 256 : w - c0;
        const int col = tid % wc, g = tid / wc;
        float m = 0.f;
        if (g < ng)
          for (int64_t k = g; k < n; k += ng)
            if (keys[k] == v) m += coef[k] * Xt[(int64_t)src[k] * xld + c0 + col];
        sm[tid] = g < ng ? m : 0.f;
        float r = 0.f;
        if (g == 0) {
          if (steps) r = Rp[(int64_t)t * rld + c0 + col];
          else
            for (int t2 = t; t2 < L; ++t2)
              if (vrows[t2] == v) r += Rp[(int64_t)t2 * rld + c0 + col];
        }
        sr[tid] = r;
        __syncthreads();
        if (g == 0) {
          float mm = 0.f;
          for (int gg = 0; gg < ng; ++gg) mm += sm[gg * wc + col];
          sm[tid] = 2.f * mm * r + r * r;
        }
        __syncthreads();
        if (tid == 0)
          for (int cc = 0; cc < wc; ++cc) acc += sm[cc];
        __syncthreads();
      }
    }
    total += acc;
  }
  if (tid == 0) corr[t] = total;
}

}  // namespace

}  // namespace arx

using namespace arx;

extern "C" {

int arx_segment_pool_fwd(const float* scores, int64_t lds, const int32_t* offs, int64_t B, int64_t W,
                         int mode, const float* gmax_dev, float* out, int64_t ldo, void* stream) {
  ARX_CHECK_ARG(scores && offs && out, "arx_segment_pool_fwd: null pointer");
  ARX_CHECK_ARG(mode == 2 || (mode == 3 && gmax_dev), "arx_segment_pool_fwd: mode 2 (max) or 3 (log-sum-exp, needs gmax)");
  if (B <= 0 || W <= 0) return ARX_OK;
  ARX_CHECK_ARG(B <= 65535, "arx_segment_pool_fwd: at most 65535 rows per call");
  dim3 grid((unsigned)ceil_div(W, 256), (unsigned)B);
  k_segpool_fwd<<<grid, 256, 0, as_stream(stream)>>>(scores, lds, offs, W, mode, gmax_dev, out, ldo);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_segment_pool_bwd(const float* scores, int64_t lds, const int32_t* offs, int64_t B, int64_t W,
                         int64_t cap, int mode, const float* gmax_dev, const float* out, int64_t ldo,
                         const float* dout, int64_t ldd, float* dscores, int64_t ldds, float* resid_rows,
                         void* stream) {
  ARX_CHECK_ARG(scores && offs && out && dout && dscores, "arx_segment_pool_bwd: null pointer");
  ARX_CHECK_ARG(mode == 2 || (mode == 3 && gmax_dev), "arx_segment_pool_bwd: mode 2 or 3 (needs gmax)");
  if (B <= 0) return ARX_OK;
  k_segpool_bwd<<<(int)B, 256, 0, as_stream(stream)>>>(scores, lds, offs, W, cap, mode, gmax_dev, out, ldo,
                                                       dout, ldd, dscores, ldds, mode == 3 ? resid_rows : nullptr);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_max_argmax(const float* x, int64_t rows, int64_t cols, int64_t ld, int64_t col_base, int first,
                   float* best, int32_t* best_idx, void* scratch, void* stream) {
  ARX_CHECK_ARG(x && best && best_idx && rows > 0 && cols > 0, "arx_max_argmax: bad argument");
  ARX_CHECK_ARG(scratch, "arx_max_argmax: reduce scratch is NULL (arx_reduce_scratch_bytes)");
  static_assert(64 + kMaxBlocks * 12 <= kReduceScratchBytes, "reduce scratch too small for k_max_argmax");
  float* mx_val = reduce_scratch_f32(scratch);
  long long* mx_pos = reinterpret_cast<long long*>(mx_val + kMaxBlocks);
  int64_t g = ceil_div(rows * cols, (int64_t)256 * 8);
  if (g > kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  k_max_argmax<<<(int)g, 256, 0, as_stream(stream)>>>(x, rows, cols, ld, col_base, first, best, best_idx, mx_val,
                                                      mx_pos, reduce_scratch_ticket(scratch));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gmax_residual_bwd(const float* resid_dev, const int32_t* idx_dev, const float* U, int64_t ldu,
                          const float* E_row, int d, float* row_grad, float* bias_grad, float* dU,
                          int64_t lddu, void* stream) {
  ARX_CHECK_ARG(resid_dev && idx_dev && U && E_row && row_grad && d > 0 && d <= 1024,
                "arx_gmax_residual_bwd: bad argument");
  k_gmax_residual<<<1, 1024, 0, as_stream(stream)>>>(resid_dev, idx_dev, U, ldu, E_row, d, row_grad, bias_grad,
                                                      dU, lddu);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gmax_norm_corr(const int32_t* keys, const int32_t* src, const float* coef, int64_t n, const float* X,
                       int64_t ldx, int d, int per_step, int64_t step_stride, const float* Xb, int per_step_b,
                       int64_t stepb_stride, const int32_t* vrows, const float* RG, int64_t ldrg, const float* RGb,
                       int L, float* corr, void* stream) {
  ARX_CHECK_ARG(keys && src && coef && vrows && corr && n >= 0 && L > 0, "arx_gmax_norm_corr: bad argument");
  ARX_CHECK_ARG((X && RG && d > 0 && d <= 1024 && ldx >= d && ldrg >= d) || (Xb && RGb),
                "arx_gmax_norm_corr: neither the embedding nor the bias gradients given");
  k_gmax_norm_corr<<<L, 256, 0, as_stream(stream)>>>(keys, src, coef, n, X, ldx, d, per_step, step_stride, Xb,
                                                      per_step_b, stepb_stride, vrows, RG, ldrg, RGb, L, corr);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
