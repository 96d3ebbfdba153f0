// topk.hip -- tf.nn.top_k(logits, k, sorted=True) for the recommend / eval path
// (hmf_model.py:154, seqModel.py:514-517) over rows of up to ~10^8 columns, and the merge step
// of the streaming full-vocabulary scorer.
//
// k_topk in misc.hip makes k passes over the row (k = 100, V = 1 M: 10^8 reads per row).
// Here one workgroup per row does a 3-level radix SELECT on the order-preserving integer image
// of the floats (11 + 11 + 10 bits, histogram in LDS) to find the k-th largest value T, then
// one collecting pass: everything > T, plus as many == T as still needed, lowest column first
// (tf.nn.top_k's tie rule); the <= 1024 winners are rank-sorted in LDS.  Four passes over the
// row in total, whatever k is.
#include "common.h"

namespace arx {

namespace {

constexpr int kSelThreads = 256;
constexpr int kSelBins = 2048;
constexpr int kMaxK = 1024;

// larger float <-> larger unsigned (NaNs sort above +inf; -0 < +0)
__device__ __forceinline__ uint32_t ord_key(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_val(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// From a histogram over `bins` digit values (larger digit = larger key): the digit D that holds
// the need-th largest key, and how many keys sit in digits above D.  All threads get both.
__device__ __forceinline__ void pick_digit(const int* hist, int bins, int need, int* s_scan,
                                           int& digit, int& above) {
  // suffix counts per thread chunk (8 digits per thread at 2048 bins)
  const int per = bins / kSelThreads;
  const int t = threadIdx.x;
  int mine = 0;
  for (int u = 0; u < per; ++u) mine += hist[t * per + u];
  s_scan[t] = mine;
  __syncthreads();
  // exclusive suffix sum over threads (256 entries: serial by thread 0 is 256 adds -- fine)
  if (t == 0) {
    int run = 0;
    for (int i = kSelThreads - 1; i >= 0; --i) {
      const int v = s_scan[i];
      s_scan[i] = run;          // keys in chunks above chunk i
      run += v;
    }
  }
  __syncthreads();
  const int ab = s_scan[t];
  __shared__ int s_digit, s_above;
  if (ab < need && ab + mine >= need) {   // exactly one thread: the need-th largest is in my chunk
    int run = ab;
    for (int u = per - 1; u >= 0; --u) {
      const int c = hist[t * per + u];
      if (run + c >= need) {
        s_digit = t * per + u;
        s_above = run;
        break;
      }
      run += c;
    }
  }
  __syncthreads();
  digit = s_digit;
  above = s_above;
  __syncthreads();
}

__global__ __launch_bounds__(kSelThreads) void k_topk_select(const float* __restrict__ logits,
                                                             int64_t ld, int64_t V, int k,
                                                             int32_t idx_base,
                                                             float* __restrict__ values,
                                                             int32_t* __restrict__ indices) {
  __shared__ int hist[kSelBins];
  __shared__ int s_scan[kSelThreads];
  __shared__ uint32_t c_key[kMaxK];
  __shared__ int32_t c_idx[kMaxK];
  __shared__ int s_cnt, s_tie;
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ld;
  const int t = threadIdx.x;
  // ---- level 1: top 11 bits ----
  for (int b = t; b < kSelBins; b += kSelThreads) hist[b] = 0;
  __syncthreads();
  for (int64_t c = t; c < V; c += kSelThreads) atomicAdd(&hist[ord_key(x[c]) >> 21], 1);
  __syncthreads();
  int d1, above1;
  pick_digit(hist, kSelBins, k, s_scan, d1, above1);
  // ---- level 2: next 11 bits among keys whose top digit is d1 ----
  for (int b = t; b < kSelBins; b += kSelThreads) hist[b] = 0;
  __syncthreads();
  for (int64_t c = t; c < V; c += kSelThreads) {
    const uint32_t key = ord_key(x[c]);
    if ((int)(key >> 21) == d1) atomicAdd(&hist[(key >> 10) & 2047u], 1);
  }
  __syncthreads();
  int d2, above2;
  pick_digit(hist, kSelBins, k - above1, s_scan, d2, above2);
  // ---- level 3: low 10 bits among keys with prefix (d1, d2) ----
  for (int b = t; b < kSelBins; b += kSelThreads) hist[b] = 0;
  __syncthreads();
  const uint32_t pre = ((uint32_t)d1 << 11) | (uint32_t)d2;
  for (int64_t c = t; c < V; c += kSelThreads) {
    const uint32_t key = ord_key(x[c]);
    if ((key >> 10) == pre) atomicAdd(&hist[key & 1023u], 1);
  }
  __syncthreads();
  int d3, above3;
  pick_digit(hist, 1024, k - above1 - above2, s_scan, d3, above3);
  const uint32_t T = (pre << 10) | (uint32_t)d3;               // the k-th largest key
  const int n_gt = above1 + above2 + above3;                    // keys strictly above T
  const int need_tie = k - n_gt;                                // keys == T to take, lowest column first
  // ---- collect ----
  if (t == 0) {
    s_cnt = 0;
    s_tie = 0;
  }
  __syncthreads();
  for (int64_t c = t; c < V; c += kSelThreads) {
    const uint32_t key = ord_key(x[c]);
    if (key > T) {
      const int p = atomicAdd(&s_cnt, 1);
      c_key[p] = key;
      c_idx[p] = (int32_t)c;
    }
  }
  __syncthreads();
  // ties in column order: ordered compaction, 256 columns per round, until need_tie are taken
  for (int64_t c0 = 0; c0 < V; c0 += kSelThreads) {
    if (s_tie >= need_tie) break;                               // uniform: read after a barrier below
    const int64_t c = c0 + t;
    const bool is = (c < V) && (ord_key(x[c]) == T);
    const unsigned long long bal = __ballot(is);
    const int lane = t & 63, wv = t >> 6;
    s_scan[wv] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w) before += s_scan[w];
    const int tot = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    const int pos = s_tie + before + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    if (is && pos < need_tie) {
      c_key[n_gt + pos] = T;
      c_idx[n_gt + pos] = (int32_t)c;
    }
    __syncthreads();
    if (t == 0) s_tie += tot;
    __syncthreads();
  }
  __syncthreads();
  // ---- rank sort the k winners: descending key, ascending column ----
  for (int i = t; i < k; i += kSelThreads) {
    const uint32_t ki = c_key[i];
    const int32_t ci = c_idx[i];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const uint32_t kj = c_key[j];
      rank += (kj > ki) || (kj == ki && c_idx[j] < ci);
    }
    if (values) values[r * k + rank] = ord_val(ki);
    indices[r * k + rank] = ci + idx_base;
  }
}

// merge two per-row top-k lists (each sorted: descending value, ascending index) into one;
// on equal values list A wins (A holds the lower column indices: earlier vocabulary chunks)
__global__ void k_topk_merge(const float* __restrict__ va, const int32_t* __restrict__ ia,
                             const float* __restrict__ vb, const int32_t* __restrict__ ib, int64_t B,
                             int ka, int kb, int k, float* __restrict__ vo, int32_t* __restrict__ io) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= B) return;
  int a = 0, b = 0;
  for (int j = 0; j < k; ++j) {
    const bool has_a = a < ka, has_b = b < kb;
    bool take_a;
    if (has_a && has_b) {
      const float xa = va[r * ka + a], xb = vb[r * kb + b];
      take_a = (xa > xb) || (xa == xb && ia[r * ka + a] <= ib[r * kb + b]);
    } else {
      take_a = has_a;
    }
    if (take_a) {
      vo[r * k + j] = va[r * ka + a];
      io[r * k + j] = ia[r * ka + a];
      ++a;
    } else {
      vo[r * k + j] = vb[r * kb + b];
      io[r * k + j] = ib[r * kb + b];
      ++b;
    }
  }
}

// out[r][j] = table[r][pos[r][j]]: positions inside a row's candidate list -> the candidates' column indices
__global__ void k_take_rows_i32(const int32_t* __restrict__ table, int64_t ld, const int32_t* __restrict__ pos,
                                int64_t ldp, int64_t B, int k, int32_t* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * k) return;
  const int64_t r = i / k;
  const int j = (int)(i % k);
  out[r * ldo + j] = table[r * ld + pos[r * ldp + j]];
}

}  // namespace

int topk_select_launch(const float* logits, int64_t ld, int64_t B, int64_t V, int k, int32_t idx_base,
                       float* values, int32_t* indices, hipStream_t s) {
  k_topk_select<<<(int)B, kSelThreads, 0, s>>>(logits, ld, V, k, idx_base, values, indices);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

extern "C" {

int arx_topk_chunk(const float* logits, int64_t ld, int64_t B, int64_t V, int k, int32_t idx_base,
                   float* values, int32_t* indices, void* stream) {
  ARX_CHECK_ARG(logits && values && indices, "arx_topk_chunk: null pointer");
  ARX_CHECK_ARG(k > 0 && k <= V && k <= kMaxK, "arx_topk_chunk: need 0 < k <= min(V, 1024)");
  if (B <= 0) return ARX_OK;
  return topk_select_launch(logits, ld, B, V, k, idx_base, values, indices, as_stream(stream));
}

int arx_topk_merge(const float* va, const int32_t* ia, const float* vb, const int32_t* ib, int64_t B,
                   int ka, int kb, int k, float* vo, int32_t* io, void* stream) {
  ARX_CHECK_ARG(va && ia && vb && ib && vo && io, "arx_topk_merge: null pointer");
  ARX_CHECK_ARG(ka >= 0 && kb >= 0 && k > 0 && k <= ka + kb, "arx_topk_merge: need 0 < k <= ka + kb");
  if (B <= 0) return ARX_OK;
  k_topk_merge<<<(int)ceil_div(B, 64), 64, 0, as_stream(stream)>>>(va, ia, vb, ib, B, ka, kb, k, vo, io);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_take_rows_i32(const int32_t* table, int64_t ld, const int32_t* pos, int64_t ldp, int64_t B, int k,
                      int32_t* out, int64_t ldo, void* stream) {
  ARX_CHECK_ARG(table && pos && out && k > 0, "arx_take_rows_i32: bad argument");
  if (B <= 0) return ARX_OK;
  k_take_rows_i32<<<(int)ceil_div(B * k, 256), 256, 0, as_stream(stream)>>>(table, ld, pos, ldp, B, k, out, ldo);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
