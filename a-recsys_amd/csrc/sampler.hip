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
// Capped race (key_cap > 0: item sets of 10^7..10^8 rows, where all but ~8 S keys are known to
// lose): the keys are never stored -- k_race_compact appends the few survivors (key, item) to a
// list, k_rank_take ranks them as 64-bit (key, item) words (every workgroup holds the list in LDS) and
// writes the S smallest in order.  Same keys, same (key, item) order as the sort path => the same draw;
// 100 M items: one 400 MB read of the weights instead of that plus two passes over 100 M keys.
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

__device__ __forceinline__ uint64_t race_base(uint64_t seed, uint64_t counter) {
  return (seed * 0x9E3779B97F4A7C15ull) ^ (counter * 0xD1B54A32D192ED03ull);
}

// key bits of item i (kInfBits: never drawn / above the cap)
__device__ __forceinline__ uint32_t race_bits(float wi, int64_t i, uint64_t base, float key_cap) {
  uint32_t bits = kInfBits;                       // zero / negative / NaN weight: never drawn
  if (wi > 0.f) {
    const uint32_t r = mix32s(base + (uint64_t)i * 0x100000001b3ull);
    const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float key = -__logf(u) / wi;
    bits = __float_as_uint(key);
    if (bits >= kInfBits) bits = kInfBits - 1;    // overflow of a tiny weight: last, not dropped
    if (key_cap > 0.f && key > key_cap) bits = kInfBits;   // cannot be among the S smallest (see arx.h)
  }
  return bits;
}

__global__ __launch_bounds__(256) void k_race_keys(const float* __restrict__ w, int64_t n,
                                                   uint64_t seed, uint64_t counter, float key_cap,
                                                   int32_t* __restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t base = race_base(seed, counter);
  for (; i < n; i += stride) keys[i] = (int32_t)race_bits(w[i], i, base, key_cap);
}

// Capped race without the key array: survivors (a few thousand of up to 2^31 items) are appended to
// `list` as (key bits, item); order of arrival, sorted next.
constexpr int kCompactCap = 16384;                 // list capacity = what a workgroup holds in LDS (128 KB)

// Survivors are staged per workgroup in LDS and leave with ONE global atomic per workgroup (a per-wave atomic on the
// one list cursor serialised at ~10 ns each: 72 us for ~7 k survivors of 1 M items); a workgroup whose stage is
// full appends the overflow directly.  Order of arrival either way: the sort that follows orders (key, item).
constexpr int kRaceStage = 1024;
__device__ __forceinline__ void race_stage(bool live, uint32_t bits, int64_t i, uint2* s_list, int* s_n,
                                           uint2* __restrict__ list, int32_t* __restrict__ count) {
  const uint64_t m = __ballot(live);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  const int first = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(s_n, __popcll(m));
  base = __shfl(base, first);
  if (live) {
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (pos < kRaceStage) s_list[pos] = make_uint2(bits, (uint32_t)i);
    else {                                                     // (stage full: rare, straight to the list)
      const int g = atomicAdd(count, 1);
      if (g < kCompactCap) list[g] = make_uint2(bits, (uint32_t)i);
    }
  }
}

__global__ __launch_bounds__(256) void k_race_compact(const float* __restrict__ w, int64_t n,
                                                      uint64_t seed, uint64_t counter, float key_cap,
                                                      uint2* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ uint2 s_list[kRaceStage];
  __shared__ int s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t base = race_base(seed, counter);
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (n4 + stride - 1) / stride;          // wave-uniform trip count (ballots inside)
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (int64_t r = 0; r < rounds; ++r, q += stride) {
    const bool in = q < n4;
    const float4 w4 = in ? reinterpret_cast<const float4*>(w)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t i = q << 2;
    const uint32_t b0 = race_bits(w4.x, i, base, key_cap), b1 = race_bits(w4.y, i + 1, base, key_cap);
    const uint32_t b2 = race_bits(w4.z, i + 2, base, key_cap), b3 = race_bits(w4.w, i + 3, base, key_cap);
    if (__any(b0 < kInfBits || b1 < kInfBits || b2 < kInfBits || b3 < kInfBits)) {
      race_stage(b0 < kInfBits, b0, i, s_list, &s_n, list, count);
      race_stage(b1 < kInfBits, b1, i + 1, s_list, &s_n, list, count);
      race_stage(b2 < kInfBits, b2, i + 2, s_list, &s_n, list, count);
      race_stage(b3 < kInfBits, b3, i + 3, s_list, &s_n, list, count);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {                   // the last n % 4 items
    const int64_t i = (n4 << 2) + threadIdx.x;
    const bool in = threadIdx.x < (n & 3);
    const uint32_t b = in ? race_bits(w[i], i, base, key_cap) : kInfBits;
    race_stage(b < kInfBits, b, i, s_list, &s_n, list, count);
  }
  __syncthreads();
  const int mine = s_n < kRaceStage ? s_n : kRaceStage;
  if (threadIdx.x == 0) s_base = mine > 0 ? atomicAdd(count, mine) : 0;
  __syncthreads();
  for (int k = threadIdx.x; k < mine; k += blockDim.x)
    if (s_base + k < kCompactCap) list[s_base + k] = s_list[k];
}

// The S smallest (key, item) words of the survivors by RANK, chip-wide in one launch: every workgroup keeps the whole
// list in LDS (<= 128 KB), 16 lanes count the predecessors of one element -- its rank is its output position (the
// words are distinct: item ids are).  The same result as the one-workgroup bitonic sort it replaces (37 us for ~3 k
// survivors, 74 for ~7.7 k: 78 / 91 barrier-separated stages) in a few microseconds.  Fewer survivors than S: the
// tail is -1 / +inf; more than the list holds: every output is -1.
__global__ __launch_bounds__(256) void k_rank_take(const uint2* __restrict__ list, const int32_t* __restrict__ count,
                                                   int64_t S, int32_t* __restrict__ out, float* __restrict__ out_keys) {
  extern __shared__ uint64_t sk[];
  const int tid = threadIdx.x;
  const int c = *count;
  const int m = c > kCompactCap ? 0 : c;
  if (blockIdx.x == 0)
    for (int64_t i = m + tid; i < S; i += 256) {                 // (the positions no survivor takes)
      out[i] = -1;
      if (out_keys) out_keys[i] = __uint_as_float(kInfBits);
    }
  const int e0 = blockIdx.x * 16;
  if (e0 >= m) return;
  const int m2 = (m + 1) & ~1;
  for (int i = tid; i < m2; i += 256) sk[i] = i < m ? ((uint64_t)list[i].x << 32) | (uint64_t)list[i].y : ~0ull;
  __syncthreads();
  const int e = e0 + (tid >> 4), part = tid & 15;
  const uint64_t key = e < m ? sk[e] : 0ull;
  int chunk = ((m2 / 2 + 15) / 16) * 2;                          // entries per lane, even (16-byte LDS reads)
  if ((chunk & 31) == 0) chunk += 2;                             // (slices 256 B apart would share a bank row)
  const int jb = min(m2, part * chunk), je = min(m2, jb + chunk);
  int rank = 0;
  for (int j = jb; j < je; j += 2) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&sk[j]);
    rank += (v.x < key) + (v.y < key);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) rank += __shfl_xor(rank, o, 16);
  if (part == 0 && e < m && rank < S) {
    out[rank] = (int32_t)(uint32_t)key;
    if (out_keys) out_keys[rank] = __uint_as_float((uint32_t)(key >> 32));
  }
}

// The S smallest of m (key, id) pairs in (key, position) order -- the merge of the ranks' sorted race lists of ONE
// pool draw over a sharded item set (arx.dist.draw_global_pool: position = rank-major, so ties go to the lower rank,
// then to the earlier entry): the rank selection of k_rank_take over 64-bit (key bits, position) words, m <= 16 384
// pairs in LDS.  Keys are non-negative floats or +inf (their bit patterns order like the values).
__global__ __launch_bounds__(256) void k_merge_take(const float* __restrict__ keys, const int32_t* __restrict__ ids, int m,
                                                    int64_t S, int32_t* __restrict__ out) {
  extern __shared__ uint64_t sk[];
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * 16;
  if (e0 >= m) return;
  const int m2 = (m + 1) & ~1;
  for (int i = tid; i < m2; i += 256)
    sk[i] = i < m ? ((uint64_t)__float_as_uint(keys[i]) << 32) | (uint64_t)(uint32_t)i : ~0ull;
  __syncthreads();
  const int e = e0 + (tid >> 4), part = tid & 15;
  const uint64_t key = e < m ? sk[e] : 0ull;
  int chunk = ((m2 / 2 + 15) / 16) * 2;
  if ((chunk & 31) == 0) chunk += 2;
  const int jb = min(m2, part * chunk), je = min(m2, jb + chunk);
  int rank = 0;
  for (int j = jb; j < je; j += 2) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&sk[j]);
    rank += (v.x < key) + (v.y < key);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) rank += __shfl_xor(rank, o, 16);
  if (part == 0 && e < m && rank < S) out[rank] = ids[e];
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

// capped draws of up to kCompactCap / 8 items: survivor list + its count
static bool compact_path(int64_t S, float key_cap) { return key_cap > 0.f && S * 8 <= kCompactCap; }

size_t arx_sample_wor_keys_workspace_bytes(int64_t n, int64_t S, float key_cap) {
  if (compact_path(S, key_cap)) return (size_t)kCompactCap * 8 + 256;
  return arx_sample_wor_workspace_bytes(n);
}

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

int arx_merge_keyed_take(const float* keys, const int32_t* ids, int64_t m, int64_t S, int32_t* out, void* stream) {
  ARX_CHECK_ARG(keys && ids && out, "arx_merge_keyed_take: null pointer");
  ARX_CHECK_ARG(m > 0 && m <= kCompactCap && S > 0 && S <= m, "arx_merge_keyed_take: need 0 < S <= m <= 16384");
  const size_t lds = (size_t)((m + 1) & ~1ll) * 8;
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_take),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kCompactCap * 8));
  k_merge_take<<<(int)ceil_div(m, 16), 256, lds, as_stream(stream)>>>(keys, ids, (int)m, S, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_sample_wor_keys(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                        float key_cap, int32_t* out_idx, float* out_keys, void* workspace,
                        size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(weights && out_idx, "arx_sample_wor: null pointer");
  ARX_CHECK_ARG(n > 0 && n < (int64_t)0x7fffffff && S > 0 && S <= n, "arx_sample_wor: need 0 < S <= n < 2^31");
  const size_t need = arx_sample_wor_keys_workspace_bytes(n, S, key_cap);
  if (!workspace || workspace_bytes < need) {
    set_error("arx_sample_wor: workspace too small (%zu < %zu)", workspace_bytes, need);
    return ARX_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  if (compact_path(S, key_cap)) {
    ARX_CHECK_ARG((reinterpret_cast<uintptr_t>(weights) & 15) == 0, "arx_sample_wor: weights must be 16-byte aligned");
    uint2* list = reinterpret_cast<uint2*>(workspace);
    int32_t* count = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + (size_t)kCompactCap * 8);
    ARX_CHECK_HIP(hipMemsetAsync(count, 0, 4, s));
    int64_t g = ceil_div(ceil_div(n, 4), 256);
    const int64_t cap = (int64_t)cu_count() * 2;                // (few workgroups: one list atomic each)
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    k_race_compact<<<(int)g, 256, 0, s>>>(weights, n, seed, counter, key_cap, list, count);
    ARX_CHECK_LAUNCH();
    // per call: the attribute is per device and the call is cheap (a process-wide flag broke the second GPU
    // of a multi-device process -- advisor, round 3)
    ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_take),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kCompactCap * 8));
    k_rank_take<<<kCompactCap / 16, 256, (size_t)kCompactCap * 8, s>>>(list, count, S, out_idx, out_keys);
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
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
