// misc.hip -- top-k over logits rows (recommend path) and sequence-loss weights.
#include "common.h"

namespace arx {

// tf.nn.top_k(sorted=True) semantics: descending value, ties -> lower index.
// One workgroup per row; k rounds of a block-wide arg-max over the row with the
// previously selected (value, index) as an exclusive upper bound -- no scratch,
// no mutation of the logits.  O(k*V) per row: adequate for the recommend path
// at ML-1m scale; a radix-select version is the planned replacement.
__global__ __launch_bounds__(256) void k_topk(const float* __restrict__ logits, int64_t ld,
                                              int64_t V, int k, float* __restrict__ values,
                                              int32_t* __restrict__ indices) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float bv;
  __shared__ int bi;
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ld;
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int64_t c = threadIdx.x; c < V; c += 256) {
      const float v = x[c];
      // candidate must come strictly after (prev_v, prev_i) in (desc value, asc index) order
      const bool after = (v < prev_v) || (v == prev_v && (int)c > prev_i);
      if (after && (v > best || (v == best && (int)c < besti))) {
        best = v;
        besti = (int)c;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ov > best || (ov == best && oi < besti)) {
        best = ov;
        besti = oi;
      }
    }
    if ((threadIdx.x & 63) == 0) {
      sv[threadIdx.x >> 6] = best;
      si[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float b = sv[0];
      int bidx = si[0];
      for (int w = 1; w < 4; ++w)
        if (sv[w] > b || (sv[w] == b && si[w] < bidx)) {
          b = sv[w];
          bidx = si[w];
        }
      bv = b;
      bi = bidx;
      if (values) values[r * k + j] = b;
      indices[r * k + j] = bidx;
    }
    __syncthreads();
    prev_v = bv;
    prev_i = bi;
  }
}

// 256 threads = 32 sequences x 8 time lanes; fixed-order combine through LDS
__global__ __launch_bounds__(256) void k_seq_weights(const float* __restrict__ w, int64_t L, int64_t B,
                                                     float* __restrict__ out) {
  __shared__ float part[8][33];
  const int cx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t b = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (b < B)
    for (int64_t t = ty; t < L; t += 8) s += w[t * B + b];   // add_n over time steps
  part[ty][cx] = s;
  __syncthreads();
  float tot = 1e-12f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += part[k][cx];
  if (b < B)
    for (int64_t t = ty; t < L; t += 8) out[t * B + b] = w[t * B + b] / tot;
}

// single workgroup, fixed order: *out = scale * sum_i x[i]*y[i]; 4 independent loads in flight
__global__ __launch_bounds__(1024) void k_dot_scaled(const float* __restrict__ x,
                                                      const float* __restrict__ y, int64_t n,
                                                      float scale, float* __restrict__ out) {
  __shared__ float part[16];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {       // 16 loads in flight: one workgroup, latency-bound
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = x[i + u * 1024]; b[u] = y[i + u * 1024]; }
    s0 = fmaf(a[0], b[0], s0); s1 = fmaf(a[1], b[1], s1); s2 = fmaf(a[2], b[2], s2); s3 = fmaf(a[3], b[3], s3);
    s0 = fmaf(a[4], b[4], s0); s1 = fmaf(a[5], b[5], s1); s2 = fmaf(a[6], b[6], s2); s3 = fmaf(a[7], b[7], s3);
  }
  for (; i + 3 * 1024 < n; i += 4 * 1024) {
    const float a0 = x[i], a1 = x[i + 1024], a2 = x[i + 2048], a3 = x[i + 3072];
    const float b0 = y[i], b1 = y[i + 1024], b2 = y[i + 2048], b3 = y[i + 3072];
    s0 = fmaf(a0, b0, s0);
    s1 = fmaf(a1, b1, s1);
    s2 = fmaf(a2, b2, s2);
    s3 = fmaf(a3, b3, s3);
  }
  for (; i < n; i += 1024) s0 = fmaf(x[i], y[i], s0);
  float s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += part[k];
    *out = t * scale;
  }
}

__global__ void k_inv_len_scale(const int32_t* __restrict__ lens, const int32_t* __restrict__ ids,
                                int64_t n, float c, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = c / (float)lens[ids ? ids[i] : (int32_t)i];
}

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_reduce_scratch_bytes(void) { return arx::kReduceScratchBytes; }


int arx_topk(const float* logits, int64_t ld, int64_t B, int64_t V, int k, float* values,
             int32_t* indices, void* stream) {
  ARX_CHECK_ARG(logits && indices, "arx_topk: null pointer");
  ARX_CHECK_ARG(k > 0 && k <= V, "arx_topk: need 0 < k <= V");
  if (B <= 0) return ARX_OK;
  ARX_CHECK_ARG(k <= 1024, "arx_topk: k <= 1024");
  // k passes over the row (k_topk) only pay for tiny problems; otherwise radix select (topk.hip):
  // four passes whatever k is
  if (values && k > 4 && V >= 2048)
    return topk_select_launch(logits, ld, B, V, k, 0, values, indices, as_stream(stream));
  k_topk<<<(int)B, 256, 0, as_stream(stream)>>>(logits, ld, V, k, values, indices);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dot_scaled(const float* x, const float* y, int64_t n, float scale, float* out,
                   void* stream) {
  ARX_CHECK_ARG(x && y && out, "arx_dot_scaled: null pointer");
  k_dot_scaled<<<1, 1024, 0, as_stream(stream)>>>(x, y, n, scale, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_inv_len_scale(const int32_t* lens, const int32_t* ids, int64_t n, float c, float* out,
                      void* stream) {
  ARX_CHECK_ARG(lens && out, "arx_inv_len_scale: null pointer");
  if (n <= 0) return ARX_OK;
  k_inv_len_scale<<<(int)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(lens, ids, n, c, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_seq_weights(const float* w, int64_t L, int64_t B, float* out, void* stream) {
  ARX_CHECK_ARG(w && out, "arx_seq_weights: null pointer");
  if (B <= 0 || L <= 0) return ARX_OK;
  k_seq_weights<<<(int)ceil_div(B, 32), 256, 0, as_stream(stream)>>>(w, L, B, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
