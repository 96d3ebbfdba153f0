// sampler.hip -- on-device negative-pool sampler (replaces the host call
// utils/prepare_train.py:7-17  np.random.choice(items, S, replace=False, p=p_item)).
//
// Weighted sampling WITHOUT replacement as an exponential race (Efraimidis-Spirakis):
// item i draws key_i = -ln(u_i) / w_i with u_i ~ U(0,1); the S smallest keys, in ascending
// order, have exactly the law of S sequential draws without replacement with probabilities
// proportional to the remaining weights -- the law of np.random.choice(replace=False, p).
// (Same distribution, not the same random stream: numpy's legacy generator is sequential.)
//   k_race_keys : one pass over the weights, counter-based RNG (seed, draw counter, item)
//   radix sort  : the K7 sort (radix_sort.hip) on the float bit patterns (positive floats
//                 order like their bits), payload = item index, stable => ties broken by
//                 index => the sample is a deterministic function of (seed, counter);
//                 zero-weight items get +inf and are dropped by the sort's first pass.
// ~170 us for 1 M items, every n_resample (50) steps; the host call it replaces takes
// ~10 ms at 1 M items.
#include "common.h"

namespace arx {

namespace {

__device__ __forceinline__ uint32_t mix32s(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

constexpr uint32_t kInfBits = 0x7f800000u;

__global__ __launch_bounds__(256) void k_race_keys(const float* __restrict__ w, int64_t n,
                                                   uint64_t seed, uint64_t counter, float key_cap,
                                                   int32_t* __restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t base = (seed * 0x9E3779B97F4A7C15ull) ^ (counter * 0xD1B54A32D192ED03ull);
  for (; i < n; i += stride) {
    const float wi = w[i];
    uint32_t bits = kInfBits;                       // zero / negative / NaN weight: never drawn
    if (wi > 0.f) {
      const uint32_t r = mix32s(base + (uint64_t)i * 0x100000001b3ull);
      const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
      const float key = -__logf(u) / wi;
      bits = __float_as_uint(key);
      if (bits >= kInfBits) bits = kInfBits - 1;    // overflow of a tiny weight: last, not dropped
      if (key_cap > 0.f && key > key_cap) bits = kInfBits;   // cannot be among the S smallest (see arx.h)
    }
    keys[i] = (int32_t)bits;
  }
}

__global__ void k_take_first(const int32_t* __restrict__ src, const uint32_t* __restrict__ keys,
                             const int32_t* __restrict__ n_live, int64_t S, int32_t* __restrict__ out,
                             float* __restrict__ out_keys) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= S) return;
  const bool live = i < *n_live;
  out[i] = live ? src[i] : -1;
  if (out_keys) out_keys[i] = live ? __uint_as_float(keys[i]) : __uint_as_float(kInfBits);
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// item_frequency on device (utils/prepare_train.py:19-35): counts by integer atomics (the result
// does not depend on the arrival order), then p ~ (count / total)^power, un-normalised -- the
// sampler does not need the normalisation.
__global__ __launch_bounds__(256) void k_count_ids(const int32_t* __restrict__ ids, int64_t n,
                                                   int64_t n_items, int32_t* __restrict__ counts) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int32_t v = ids[i];
    if (v >= 0 && v < n_items) atomicAdd(&counts[v], 1);
  }
}

__global__ __launch_bounds__(256) void k_power_weights(const int32_t* __restrict__ counts, int64_t n_items,
                                                       double inv_total, float power,
                                                       float* __restrict__ w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_items; i += stride) {
    const int32_t c = counts[i];
    w[i] = c > 0 ? (float)pow((double)c * inv_total, (double)power) : 0.f;
  }
}

}  // namespace

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_sample_wor_workspace_bytes(int64_t n) {
  const size_t ni = align256((size_t)(n > 0 ? n : 1) * 4);
  // raw keys, 2 x (keys, src, coef) ping-pong, histograms, live count
  return ni * 7 + radix_sort_hist_bytes() + 256;
}

int arx_item_frequency(const int32_t* item_ids, int64_t n, int64_t n_items, int64_t total, float power,
                       int32_t* counts, float* weights, void* stream) {
  ARX_CHECK_ARG(counts && (item_ids || n == 0), "arx_item_frequency: null pointer");
  ARX_CHECK_ARG(n >= 0 && n_items > 0, "arx_item_frequency: bad size");
  hipStream_t s = as_stream(stream);
  const int64_t cap = (int64_t)cu_count() * 16;
  if (n > 0) {
    int64_t g = ceil_div(n, 256);
    if (g > cap) g = cap;
    k_count_ids<<<(int)g, 256, 0, s>>>(item_ids, n, n_items, counts);
    ARX_CHECK_LAUNCH();
  }
  if (weights) {
    ARX_CHECK_ARG(total > 0, "arx_item_frequency: total interactions must be positive");
    int64_t g = ceil_div(n_items, 256);
    if (g > cap) g = cap;
    k_power_weights<<<(int)g, 256, 0, s>>>(counts, n_items, 1.0 / (double)total, power, weights);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

int arx_sample_wor(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                   int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
  return arx_sample_wor_capped(weights, n, S, seed, counter, 0.f, out_idx, workspace, workspace_bytes, stream);
}

int arx_sample_wor_capped(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                          float key_cap, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return arx_sample_wor_keys(weights, n, S, seed, counter, key_cap, out_idx, nullptr, workspace, workspace_bytes,
                             stream);
}

int arx_sample_wor_keys(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                        float key_cap, int32_t* out_idx, float* out_keys, void* workspace,
                        size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(weights && out_idx, "arx_sample_wor: null pointer");
  ARX_CHECK_ARG(n > 0 && n < (int64_t)0x7fffffff && S > 0 && S <= n, "arx_sample_wor: need 0 < S <= n < 2^31");
  const size_t need = arx_sample_wor_workspace_bytes(n);
  if (!workspace || workspace_bytes < need) {
    set_error("arx_sample_wor: workspace too small (%zu < %zu)", workspace_bytes, need);
    return ARX_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  const size_t ni = align256((size_t)n * 4);
  char* base = reinterpret_cast<char*>(workspace);
  int32_t* keys_raw = reinterpret_cast<int32_t*>(base);
  uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(base + ni);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + 2 * ni);
  int32_t* src_tmp = reinterpret_cast<int32_t*>(base + 3 * ni);
  int32_t* src_out = reinterpret_cast<int32_t*>(base + 4 * ni);
  float* coef_tmp = reinterpret_cast<float*>(base + 5 * ni);
  float* coef_out = reinterpret_cast<float*>(base + 6 * ni);
  int32_t* hist = reinterpret_cast<int32_t*>(base + 7 * ni);
  int32_t* n_live = reinterpret_cast<int32_t*>(base + 7 * ni + radix_sort_hist_bytes());
  {
    int64_t g = ceil_div(n, 256);
    const int64_t cap = (int64_t)cu_count() * 16;
    if (g > cap) g = cap;
    k_race_keys<<<(int)g, 256, 0, s>>>(weights, n, seed, counter, key_cap, keys_raw);
    ARX_CHECK_LAUNCH();
  }
  // 31 key bits (positive floats below +inf); +inf (= sentinel) entries are dropped
  int rc = launch_radix_sort(keys_raw, nullptr, nullptr, n, kInfBits, 31, keys_tmp, keys_out, src_tmp,
                             src_out, coef_tmp, coef_out, hist, nullptr, n_live, s);
  if (rc) return rc;
  k_take_first<<<(int)ceil_div(S, 256), 256, 0, s>>>(src_out, keys_out, n_live, S, out_idx, out_keys);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
