// gather.hip -- attribute-embedding lookups for gfx950 (HBM-bound kernels).
//
//  K1  arx_gather_mulhot_mean_fwd : CSR expand + row gather + segment-mean fused
//  K2  arx_gather_onehot_fwd      : one-hot row gather (+bias)
//  K5  arx_dot_score_fwd/bwd      : per-row dot product (target score)
//      arx_csr_expand             : batch_slice2 / batch_segids2 (integer, bit-exact)
//
// Layout: a table row is d fp32 (d=128 -> 512 B).  A row is read by a
// sub-group of LPR = pow2ceil(d/4) lanes, one float4 (16 B) per lane, so a
// wave64 covers 64/LPR rows per instruction (2 rows at d=128) with fully
// coalesced 16 B/lane accesses -- the coalescing sweet spot on CDNA4.
#include "common.h"

namespace arx {

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// ---------------------------------------------------------------------------
// K2: one-hot gather
// ---------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_onehot(
    const float* __restrict__ E, const float* __restrict__ bias,
    const int32_t* __restrict__ cat_map, const int32_t* __restrict__ ids, int64_t B, int d,
    float scale, int accumulate, float* __restrict__ out, int64_t ldo,
    float* bias_out, int64_t ldb) {      // bias_out[r * ldb]; may be a column of `out` (packed rows)
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave * GPW + gid; r < B; r += nwave * GPW) {
    int id = ids[r];
    int row = cat_map ? cat_map[id] : id;
    if (col < d) {
      float4 v = *reinterpret_cast<const float4*>(E + (int64_t)row * d + col);
      float4* op = reinterpret_cast<float4*>(out + r * ldo + col);
      float4 o = accumulate ? *op : make_float4(0.f, 0.f, 0.f, 0.f);
      o.x += scale * v.x;
      o.y += scale * v.y;
      o.z += scale * v.z;
      o.w += scale * v.w;
      *op = o;
    }
    if (bias_out && lig == 0) {
      float b = scale * bias[row];
      bias_out[r * ldb] = accumulate ? bias_out[r * ldb] + b : b;
    }
  }
}

// ---------------------------------------------------------------------------
// K1: multi-hot gather + segment-mean.  One LPR-lane sub-group per bag.  The
// bag's token ids are fetched coalesced (lane j of the sub-group loads token
// j), then broadcast with ds_bpermute so every lane issues the row loads of
// up to 4 tokens back-to-back (4 x 16 B in flight per lane) before summing.
// ---------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_mulhot(
    const float* __restrict__ E, const float* __restrict__ bias,
    const int32_t* __restrict__ vals, const int32_t* __restrict__ starts,
    const int32_t* __restrict__ lens, const int32_t* __restrict__ ids, int64_t B, int d,
    float scale, int accumulate, float* __restrict__ out, int64_t ldo,
    float* __restrict__ bias_out, const float* __restrict__ E1, const float* __restrict__ bias1,
    const int32_t* __restrict__ cat_map1) {
  // E1 (nullable): a one-hot feature of the same entity (its id row), added to the bag mean before
  // the common scale -- the two lookups of an (id + multi-hot attribute) item in one launch
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave * GPW + gid; r < B; r += nwave * GPW) {
    const int id = ids[r];
    const int st = starts[id];
    const int len = lens[id];
    float4 one = make_float4(0.f, 0.f, 0.f, 0.f);
    float one_b = 0.f;
    if (E1) {
      const int row1 = cat_map1 ? cat_map1[id] : id;
      if (colok) one = *reinterpret_cast<const float4*>(E1 + (int64_t)row1 * d + col);
      if (bias1 && lig == 0) one_b = bias1[row1];
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    float bacc = 0.f;
    for (int j0 = 0; j0 < len; j0 += LPR) {
      const int myj = j0 + lig;
      const int mytok = (myj < len) ? vals[st + myj] : 0;
      if (bias && myj < len) bacc += bias[mytok];
      const int cnt = min(LPR, len - j0);
      int t = 0;
      for (; t + 4 <= cnt; t += 4) {
        const int t0 = __shfl(mytok, t, LPR);
        const int t1 = __shfl(mytok, t + 1, LPR);
        const int t2 = __shfl(mytok, t + 2, LPR);
        const int t3 = __shfl(mytok, t + 3, LPR);
        if (colok) {
          float4 v0 = *reinterpret_cast<const float4*>(E + (int64_t)t0 * d + col);
          float4 v1 = *reinterpret_cast<const float4*>(E + (int64_t)t1 * d + col);
          float4 v2 = *reinterpret_cast<const float4*>(E + (int64_t)t2 * d + col);
          float4 v3 = *reinterpret_cast<const float4*>(E + (int64_t)t3 * d + col);
          a0 = f4_add(a0, v0);
          a1 = f4_add(a1, v1);
          a2 = f4_add(a2, v2);
          a3 = f4_add(a3, v3);
        }
      }
      for (; t < cnt; ++t) {
        const int t0 = __shfl(mytok, t, LPR);
        if (colok) a0 = f4_add(a0, *reinterpret_cast<const float4*>(E + (int64_t)t0 * d + col));
      }
    }
    a0 = f4_add(f4_add(a0, a1), f4_add(a2, a3));
    const float flen = (float)len;
    if (colok) {
      float4* op = reinterpret_cast<float4*>(out + r * ldo + col);
      float4 o = accumulate ? *op : make_float4(0.f, 0.f, 0.f, 0.f);
      o.x += scale * (a0.x / flen + one.x);
      o.y += scale * (a0.y / flen + one.y);
      o.z += scale * (a0.z / flen + one.z);
      o.w += scale * (a0.w / flen + one.w);
      *op = o;
    }
    if (bias_out) {
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) bacc += __shfl_xor(bacc, o, LPR);
      if (lig == 0) {
        float b = scale * (bacc / flen + one_b);
        bias_out[r] = accumulate ? bias_out[r] + b : b;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K5: dot score
// ---------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_dot_fwd(const float* __restrict__ U, int64_t ldu,
                                                 const float* __restrict__ T, int64_t ldt,
                                                 const float* __restrict__ tbias, int64_t B,
                                                 int d, float* __restrict__ score) {
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave * GPW + gid; r < B; r += nwave * GPW) {
    float s = 0.f;
    if (col < d) {
      float4 u = *reinterpret_cast<const float4*>(U + r * ldu + col);
      float4 t = *reinterpret_cast<const float4*>(T + r * ldt + col);
      s = u.x * t.x + u.y * t.y + u.z * t.z + u.w * t.w;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    if (lig == 0) score[r] = s + (tbias ? tbias[r] : 0.f);
  }
}

__global__ __launch_bounds__(256) void k_dot_bwd(const float* __restrict__ U, int64_t ldu,
                                                 const float* __restrict__ T, int64_t ldt,
                                                 const float* __restrict__ ds, int64_t B, int d4,
                                                 float* __restrict__ dU, int64_t lddu, int acc_dU,
                                                 float* __restrict__ dT, int64_t lddt) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t total = B * d4;
  for (; i < total; i += stride) {
    const int64_t r = i / d4;
    const int c = (int)(i % d4) * 4;
    const float g = ds[r];
    const float4 t = *reinterpret_cast<const float4*>(T + r * ldt + c);
    float4* up = reinterpret_cast<float4*>(dU + r * lddu + c);
    float4 o = acc_dU ? *up : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += g * t.x;
    o.y += g * t.y;
    o.z += g * t.z;
    o.w += g * t.w;
    *up = o;
    if (dT) {
      const float4 u = *reinterpret_cast<const float4*>(U + r * ldu + c);
      *reinterpret_cast<float4*>(dT + r * lddt + c) =
          make_float4(g * u.x, g * u.y, g * u.z, g * u.w);
    }
  }
}

// ---------------------------------------------------------------------------
// csr_expand: three launches (block-local scan, scan of block totals, emit).
// ---------------------------------------------------------------------------
constexpr int kExpBlock = 256;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

// returns exclusive prefix of v within the block, *block_total gets the sum
__device__ __forceinline__ int block_excl_scan(int v, int* wsum /*[4]*/, int* block_total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += wsum[i];
  *block_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

__global__ __launch_bounds__(kExpBlock) void k_expand_local(
    const int32_t* __restrict__ lens, const int32_t* __restrict__ row_ids, int64_t B,
    int32_t* __restrict__ offsets, int32_t* __restrict__ block_tot) {
  __shared__ int wsum[4];
  const int64_t r = blockIdx.x * (int64_t)kExpBlock + threadIdx.x;
  int len = 0;
  if (r < B) len = lens[row_ids ? row_ids[r] : (int32_t)r];
  int tot;
  int ex = block_excl_scan(len, wsum, &tot);
  if (r < B) offsets[r] = ex;
  if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kExpBlock) void k_expand_scan_blocks(
    int32_t* __restrict__ block_tot, int nb, int32_t* __restrict__ offsets_B,
    int32_t* __restrict__ total_out) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kExpBlock) {
    const int i = base + threadIdx.x;
    int v = (i < nb) ? block_tot[i] : 0;
    int tot;
    int ex = block_excl_scan(v, wsum, &tot);
    const int carry = carry_s;
    if (i < nb) block_tot[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *offsets_B = carry_s;
    if (total_out) *total_out = carry_s;
  }
}

// final offsets: offsets[r] += base of its 256-bag block (one thread per bag)
__global__ __launch_bounds__(kExpBlock) void k_expand_fix(int32_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ block_base,
                                                          int64_t B) {
  const int64_t r = blockIdx.x * (int64_t)kExpBlock + threadIdx.x;
  if (r < B) offsets[r] += block_base[blockIdx.x];
}

// token-parallel emit: one thread per OUTPUT position (capacity of them) finds its bag
// by binary search in the L2-resident offsets[0..B] (the per-block form ran on B/256
// workgroups only: 23 us at B = 4096, 16 CUs busy).
__global__ __launch_bounds__(256) void k_expand_emit(
    const int32_t* __restrict__ vals, const int32_t* __restrict__ starts,
    const int32_t* __restrict__ lens, const int32_t* __restrict__ row_ids, int64_t B,
    const int32_t* __restrict__ offsets, int32_t* __restrict__ token_ids,
    int32_t* __restrict__ segids, int64_t capacity, int32_t pad_token, int32_t pad_seg,
    int32_t seg_base, float coef_scale, float* __restrict__ coef_out) {
  const int64_t total = offsets[B];
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; q < capacity; q += stride) {
    if (q >= total) {
      token_ids[q] = pad_token;
      if (segids) segids[q] = pad_seg;
      if (coef_out) coef_out[q] = 0.f;
      continue;
    }
    int64_t lo = 0, hi = B - 1;          // last bag with offsets[bag] <= q (skips empty bags)
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (offsets[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int row = row_ids ? row_ids[lo] : (int32_t)lo;
    token_ids[q] = vals[starts[row] + (int32_t)(q - offsets[lo])];
    if (segids) segids[q] = seg_base + (int32_t)lo;
    if (coef_out) coef_out[q] = coef_scale / (float)lens[row];
  }
}

__global__ void k_site_onehot(const int32_t* __restrict__ cat_map, const int32_t* __restrict__ ids,
                              int64_t n, int32_t row_base, float coef,
                              int32_t* __restrict__ keys_out, int32_t* __restrict__ src_out,
                              float* __restrict__ coef_out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int id = ids[i];
    keys_out[i] = id < 0 ? ARX_KEY_NONE : (cat_map ? cat_map[id] : id);     // (id < 0: an empty pool slot, no update)
    if (src_out) src_out[i] = row_base + (int32_t)i;
    if (coef_out) coef_out[i] = coef;
  }
}

// row-sharded table routing (mod-N striping): owner = id % world, local row = id / world
// Block layout of a pool whose items are striped over `world` owners (owner = id % world): the owner's
// items, in slot order, are rows [0, count_g) of its block.  One workgroup, any S, world <= 256: the slots go
// by in rounds of 1024; in a round every wave ranks its slots among the same-owner slots of the wave (one ballot
// per owner bit, as the radix scatter does), the waves' counts meet in LDS and a per-owner running total carries
// over the rounds.  (Round 4's form kept all owners in LDS and counted earlier slots one by one: S <= 4096,
// world <= 64, O(S^2 / 1024).)  A negative id has no owner: it is counted in counts[world] (the caller rejects the
// pool), gets gidx = -1 and takes no row of any block.
__global__ __launch_bounds__(1024) void k_pool_blocks(const int32_t* __restrict__ ids, int S, int world, int rank,
                                                      int32_t zero_row, int cap, int32_t* __restrict__ counts,
                                                      int32_t* __restrict__ gidx, int32_t* __restrict__ my_slots,
                                                      int32_t* __restrict__ pool_rows) {
  __shared__ int wc[16][256];          // a round's per-wave counts, then the waves' starting positions
  __shared__ int tot[256];             // slots of each owner in the rounds before
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 256) tot[tid] = 0;
  if (tid == 0) bad = 0;
  int obits = 1;
  while ((1 << obits) < world) ++obits;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r0 = 0; r0 < S; r0 += 1024) {
    for (int q = tid; q < 16 * 256; q += 1024) (&wc[0][0])[q] = 0;
    __syncthreads();
    const int i = r0 + tid;
    const int id = i < S ? ids[i] : -1;
    const bool live = i < S && id >= 0;
    if (i < S && id < 0) atomicAdd(&bad, 1);
    const int o = live ? id % world : 0;
    unsigned long long peers = __ballot(live);
    for (int bit = 0; bit < obits; ++bit) {
      const bool one = (o >> bit) & 1;
      const unsigned long long bb = __ballot(one);
      peers &= one ? bb : ~bb;
    }
    const int rank_in = __popcll(peers & lt);
    if (live && rank_in == 0) wc[w][o] = __popcll(peers);
    __syncthreads();
    if (tid < world) {                 // the waves' starting positions for owner tid, the running total
      int run = tot[tid];
      for (int q = 0; q < 16; ++q) {
        const int c = wc[q][tid];
        wc[q][tid] = run;
        run += c;
      }
      tot[tid] = run;
    }
    __syncthreads();
    if (i < S && cap > 0) {
      if (live) {
        const int pos = wc[w][o] + rank_in;
        gidx[i] = o * cap + pos;
        if (o == rank) {
          my_slots[pos] = i;
          pool_rows[pos] = id / world;
        }
      } else {
        gidx[i] = -1;
      }
    }
    __syncthreads();
  }
  for (int g = tid; g < world; g += 1024) counts[g] = tot[g];
  if (tid == 0) counts[world] = bad;
  if (cap > 0)
    for (int i = tot[rank] + tid; i < S; i += 1024) {     // the tail of this rank's lists: padding
      my_slots[i] = S;
      pool_rows[i] = zero_row;
    }
}

__global__ void k_shard_route(const int32_t* __restrict__ ids, int64_t n, int world, int rank,
                              int32_t zero_row, int32_t* __restrict__ rows_out,
                              int32_t* __restrict__ keys_out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int id = ids[i];
    const bool own = id >= 0 && (id % world) == rank;     // (id < 0: an EMPTY slot -- the zero row, no update; advisor r5)
    if (rows_out) rows_out[i] = own ? id / world : zero_row;
    if (keys_out) keys_out[i] = own ? id / world : ARX_KEY_NONE;
  }
}

// dst[r, 0:cols] = src[r, 0:cols] with independent leading dimensions (float4 per thread)
__global__ void k_copy_2d(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                          int64_t ldd, int64_t rows, int c4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t total = rows * c4;
  for (; i < total; i += stride) {
    const int64_t r = i / c4;
    const int c = (int)(i % c4) * 4;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + r * lds + c);
  }
}

static inline int grid_waves(int64_t nwaves) {
  int64_t g = ceil_div(nwaves, 4);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


// Several one-hot lookups (user ids, target items, the sampled pool, ...) in ONE launch: the step's
// lookups are independent leaves of the graph and each costs a ~5 us launch slot on its own.
// ---------------------------------------------------------------------------
// Several lookups of a step in ONE launch, one-hot and / or multi-hot features per lookup
// (embed_attribute.py:371-407 for users, target items and the sampled pool: three dependent-free
// launches of ~6-15 us each, issued back to back, cost more in launch latency than in rows).
// Site s: out[r] = scale * ( E_id[map[id]] (if E_id) + mean over the bag of id of E_tok rows (if E_tok) ).
// Every workgroup belongs to one site (wave-uniform pointers: scalar loads of the kernel arguments);
// a sub-group of LPR lanes takes kLookupRows rows, as in k_gather_mulhot.
// ---------------------------------------------------------------------------
struct LookupSites {
  const float* E_id[kMaxSites];
  const float* bias_id[kMaxSites];
  const int32_t* cat_map[kMaxSites];
  const float* E_tok[kMaxSites];
  const float* bias_tok[kMaxSites];
  const int32_t* vals[kMaxSites];
  const int32_t* starts[kMaxSites];
  const int32_t* lens[kMaxSites];
  const int32_t* ids[kMaxSites];
  float* out[kMaxSites];
  float* bias_out[kMaxSites];
  int64_t ldo[kMaxSites];
  int64_t n[kMaxSites];
  int32_t blk_end[kMaxSites];
  float scale[kMaxSites];
  int count;
};

#ifndef ARX_LOOKUP_ROWS
#define ARX_LOOKUP_ROWS 1
#endif
constexpr int kLookupRows = ARX_LOOKUP_ROWS;      // rows per sub-group (1: the bag chains are latency-bound, more sub-groups in flight win)

template <int LPR>
__global__ __launch_bounds__(256) void k_lookup_multi(LookupSites ls, int d) {
  constexpr int GPW = 64 / LPR;
  constexpr int RPB = 4 * GPW * kLookupRows;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kMaxSites - 1; ++k) s += (k < ls.count - 1 && (int)blockIdx.x >= ls.blk_end[k]) ? 1 : 0;
  const int blk0 = s ? ls.blk_end[s - 1] : 0;
  const int64_t n = ls.n[s];
  const float* __restrict__ E1 = ls.E_id[s];
  const float* __restrict__ bias1 = ls.bias_id[s];
  const int32_t* __restrict__ cat_map1 = ls.cat_map[s];
  const float* __restrict__ E = ls.E_tok[s];
  const float* __restrict__ bias = ls.bias_tok[s];
  const int32_t* __restrict__ vals = ls.vals[s];
  const int32_t* __restrict__ starts = ls.starts[s];
  const int32_t* __restrict__ lens = ls.lens[s];
  const int32_t* __restrict__ ids = ls.ids[s];
  float* __restrict__ out = ls.out[s];
  float* __restrict__ bias_out = ls.bias_out[s];
  const int64_t ldo = ls.ldo[s];
  const float scale = ls.scale[s];
  const int64_t base = (int64_t)((int)blockIdx.x - blk0) * RPB + wv * GPW + gid;
#pragma unroll
  for (int u = 0; u < kLookupRows; ++u) {
    const int64_t r = base + u * (4 * GPW);
    if (r >= n) continue;
    const int id = ids[r];
    if (id < 0) {
      // an EMPTY pool slot: DeviceSampler.sample() leaves id -1 where a (37-sigma rare) short capped draw could not
      // fill a position.  It looks nothing up -- a zero row -- and K7's key builders drop it (site_key), instead of
      // reading cat_map[-1] / E[-1] (advisor, round 4).  Round 6 (advisor, round 5): where the lookup has a bias
      // output the slot's bias is kEmptySlotBias = -1e30, so its logit lies 1e30 below every target score: no hinge
      // activity, exp() = 0 -- the slot is OUT of the sampled losses ('mw' rank weighting, 'mce' sum) and of their
      // gradients instead of scoring as a real negative with logit 0.  (Finite on purpose: -inf would read as a
      // poisoned pool row in k_sc_prep's non-finite check.)  Sub-group-uniform branch.
      if (colok) *reinterpret_cast<float4*>(out + r * ldo + col) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias_out && lig == 0) bias_out[r] = kEmptySlotBias;
      continue;
    }
    float4 one = make_float4(0.f, 0.f, 0.f, 0.f);
    float one_b = 0.f;
    if (E1) {
      const int row1 = cat_map1 ? cat_map1[id] : id;
      if (colok) one = *reinterpret_cast<const float4*>(E1 + (int64_t)row1 * d + col);
      if (bias_out && lig == 0) one_b = bias1[row1];
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    float bacc = 0.f;
    int len = 1;
    if (E) {
      const int st = starts[id];
      len = lens[id];
      for (int j0 = 0; j0 < len; j0 += LPR) {
        const int myj = j0 + lig;
        const int mytok = (myj < len) ? vals[st + myj] : 0;
        if (bias_out && myj < len) bacc += bias[mytok];
        const int cnt = min(LPR, len - j0);
        int t = 0;
        for (; t + 4 <= cnt; t += 4) {
          const int t0 = __shfl(mytok, t, LPR);
          const int t1 = __shfl(mytok, t + 1, LPR);
          const int t2 = __shfl(mytok, t + 2, LPR);
          const int t3 = __shfl(mytok, t + 3, LPR);
          if (colok) {
            float4 v0 = *reinterpret_cast<const float4*>(E + (int64_t)t0 * d + col);
            float4 v1 = *reinterpret_cast<const float4*>(E + (int64_t)t1 * d + col);
            float4 v2 = *reinterpret_cast<const float4*>(E + (int64_t)t2 * d + col);
            float4 v3 = *reinterpret_cast<const float4*>(E + (int64_t)t3 * d + col);
            a0 = f4_add(a0, v0);
            a1 = f4_add(a1, v1);
            a2 = f4_add(a2, v2);
            a3 = f4_add(a3, v3);
          }
        }
        for (; t < cnt; ++t) {
          const int t0 = __shfl(mytok, t, LPR);
          if (colok) a0 = f4_add(a0, *reinterpret_cast<const float4*>(E + (int64_t)t0 * d + col));
        }
      }
      a0 = f4_add(f4_add(a0, a1), f4_add(a2, a3));
    }
    // same arithmetic as k_gather_mulhot / k_gather_onehot: sum / len (a true divide), + id row, * scale
    const float flen = (float)len;
    if (colok) {
      float4 o;
      if (E) {
        o.x = scale * (a0.x / flen + one.x);
        o.y = scale * (a0.y / flen + one.y);
        o.z = scale * (a0.z / flen + one.z);
        o.w = scale * (a0.w / flen + one.w);
      } else {
        o.x = scale * one.x; o.y = scale * one.y; o.z = scale * one.z; o.w = scale * one.w;
      }
      *reinterpret_cast<float4*>(out + r * ldo + col) = o;
    }
    if (bias_out) {
      if (E) {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) bacc += __shfl_xor(bacc, o, LPR);
      }
      if (lig == 0) bias_out[r] = E ? scale * (bacc / flen + one_b) : scale * one_b;
    }
  }
}

struct GatherSites {
  const float* E[kMaxSites];
  const float* bias[kMaxSites];
  const int32_t* cat_map[kMaxSites];
  const int32_t* ids[kMaxSites];
  float* out[kMaxSites];
  float* bias_out[kMaxSites];
  int64_t ldo[kMaxSites];
  int64_t ldb[kMaxSites];          // bias_out[r * ldb]: 1, or ldo for a column of `out` (packed rows)
  int64_t ldbi[kMaxSites];         // bias[row * ldbi]: 1, or the row stride of a packed table
  int64_t n[kMaxSites];            // rows of each site
  int32_t blk_end[kMaxSites];      // exclusive prefix ends over the sites' workgroups
  float scale[kMaxSites];
  int count;
};

constexpr int kGatherUnroll = 4;    // rows per sub-group, all in flight together

// Every workgroup belongs to ONE site, so the site's pointers are wave-uniform (scalar loads of
// the kernel arguments).  With the site picked per row, each of E[s], ids[s], out[s]... was a
// vector load from the argument block behind a serial search over the row prefix: six to seven
// dependent round trips per row (SQ_WAVE_CYCLES: 9 us per wave for two 512-byte rows).
// A sub-group takes kGatherUnroll rows: ids together, then table rows together, then stores.
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_onehot_multi(GatherSites gs, int d) {
  constexpr int GPW = 64 / LPR;
  constexpr int UN = kGatherUnroll;
  constexpr int RPB = 4 * GPW * UN;                // rows per workgroup
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kMaxSites - 1; ++k) s += (k < gs.count - 1 && (int)blockIdx.x >= gs.blk_end[k]) ? 1 : 0;
  const int blk0 = s ? gs.blk_end[s - 1] : 0;
  const int64_t n = gs.n[s];
  const float* __restrict__ E = gs.E[s];
  const float* __restrict__ bias = gs.bias[s];
  const int32_t* __restrict__ cat_map = gs.cat_map[s];
  const int32_t* __restrict__ ids = gs.ids[s];
  float* __restrict__ out = gs.out[s];
  float* __restrict__ bias_out = gs.bias_out[s];
  const int64_t ldo = gs.ldo[s];
  const int64_t ldb = gs.ldb[s];
  const int64_t ldbi = gs.ldbi[s];
  const float sc = gs.scale[s];
  const int64_t base = (int64_t)((int)blockIdx.x - blk0) * RPB + wv * GPW + gid;
  int64_t r[UN];
  int row[UN];
  bool ok[UN], hit[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    r[u] = base + u * (4 * GPW);                   // neighbouring sub-groups take neighbouring rows
    ok[u] = r[u] < n;
    row[u] = ok[u] ? ids[r[u]] : 0;
    hit[u] = ok[u] && row[u] >= 0;                 // (id < 0: an EMPTY pool slot -- a zero row, k_lookup_multi)
  }
  if (cat_map) {
#pragma unroll
    for (int u = 0; u < UN; ++u) row[u] = hit[u] ? cat_map[row[u]] : 0;
  }
  float4 v[UN];
  float bv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    v[u] = (hit[u] && col < d) ? *reinterpret_cast<const float4*>(E + (int64_t)row[u] * d + col)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[u] = (hit[u] && bias_out && lig == 0) ? bias[(int64_t)row[u] * ldbi] : 0.f;
    if (ok[u] && !hit[u] && sc > 0.f) bv[u] = kEmptySlotBias / sc;     // an EMPTY slot is out of the sampled losses (k_lookup_multi)
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    if (!ok[u]) continue;
    if (col < d)
      *reinterpret_cast<float4*>(out + r[u] * ldo + col) =
          make_float4(sc * v[u].x, sc * v[u].y, sc * v[u].z, sc * v[u].w);
    if (bias_out && lig == 0) bias_out[r[u] * ldb] = sc * bv[u];
  }
}

// Padded (un-compacted) bag expansion for the gradient scatter: slot r * max_len + j holds token j
// of bag r, slots past the bag's length hold pad_token.  No prefix sums -- K7's sort drops the
// pads in its first pass anyway, so compacting them first (arx_csr_expand: four launches) buys
// nothing there.  Order of the live entries = (r, j), the same as the compact form.
__global__ __launch_bounds__(256) void k_bag_expand_padded(
    const int32_t* __restrict__ vals, const int32_t* __restrict__ starts,
    const int32_t* __restrict__ lens, const int32_t* __restrict__ row_ids, int64_t B, int max_len,
    int32_t pad_token, int32_t seg_base, float coef_scale, int32_t* __restrict__ token_ids,
    int32_t* __restrict__ segids, float* __restrict__ coef_out) {
  const int64_t total = B * (int64_t)max_len;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = q / max_len;
    const int j = (int)(q - r * max_len);
    const int32_t row = row_ids ? row_ids[r] : (int32_t)r;
    const int32_t len = lens[row];
    const bool live = j < len;
    token_ids[q] = live ? vals[(int64_t)starts[row] + j] : pad_token;
    segids[q] = seg_base + (int32_t)r;
    if (coef_out) coef_out[q] = live ? coef_scale / (float)len : 0.f;
  }
}

// one column of a packed [rows, ld] matrix <-> a dense vector (bias / bias-gradient column of the
// sharded exchanges' packed rows)
__global__ __launch_bounds__(256) void k_copy_strided(const float* __restrict__ src, int64_t ss,
                                                      float* __restrict__ dst, int64_t ds, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i * ds] = src[i * ss];
}

}  // namespace arx

using namespace arx;

#define ARX_DISPATCH_LPR(lpr, CALL)                  \
  switch (lpr) {                                     \
    case 1: { constexpr int LPR = 1; CALL; } break;  \
    case 2: { constexpr int LPR = 2; CALL; } break;  \
    case 4: { constexpr int LPR = 4; CALL; } break;  \
    case 8: { constexpr int LPR = 8; CALL; } break;  \
    case 16: { constexpr int LPR = 16; CALL; } break;\
    case 32: { constexpr int LPR = 32; CALL; } break;\
    default: { constexpr int LPR = 64; CALL; } break;\
  }

static int check_d(const char* fn, int d) {
  if (d <= 0 || d % 4 != 0 || d > 256) {
    set_error("%s: embedding size d=%d unsupported (need d %% 4 == 0 and d <= 256)", fn, d);
    return ARX_EUNSUPPORTED;
  }
  return ARX_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int arx_gather_onehot_fwd(const float* E, const float* bias, const int32_t* cat_map,
                          const int32_t* ids, int64_t B, int d, float scale, int accumulate,
                          float* out, int64_t ldo, float* bias_out, void* stream) {
  ARX_CHECK_ARG(E && ids && out, "arx_gather_onehot_fwd: null pointer");
  ARX_CHECK_ARG((bias != nullptr) == (bias_out != nullptr) || bias_out == nullptr,
                "arx_gather_onehot_fwd: bias_out requires bias");
  ARX_CHECK_ARG(!(bias_out && !bias), "arx_gather_onehot_fwd: bias_out requires bias");
  int rc = check_d("arx_gather_onehot_fwd", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldo % 4 == 0 && ldo >= d && aligned16(E) && aligned16(out),
                "arx_gather_onehot_fwd: ldo %% 4 and 16-byte alignment required");
  if (B <= 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(B, 64 / lpr);
  ARX_DISPATCH_LPR(lpr, (k_gather_onehot<LPR><<<grid_waves(nwaves), 256, 0, as_stream(stream)>>>(
                            E, bias, cat_map, ids, B, d, scale, accumulate, out, ldo,
                            bias_out, 1)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gather_onehot_packed_fwd(const float* E, const float* bias, const int32_t* cat_map,
                                 const int32_t* ids, int64_t B, int d, float scale, float* out,
                                 int64_t ldo, void* stream) {
  if (B <= 0) return ARX_OK;
  ARX_CHECK_ARG(E && bias && ids && out, "arx_gather_onehot_packed_fwd: null pointer");
  int rc = check_d("arx_gather_onehot_packed_fwd", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldo % 4 == 0 && ldo > d && aligned16(E) && aligned16(out),
                "arx_gather_onehot_packed_fwd: ldo %% 4, ldo > d and 16-byte alignment required");
  if (B <= 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(B, 64 / lpr);
  ARX_DISPATCH_LPR(lpr, (k_gather_onehot<LPR><<<grid_waves(nwaves), 256, 0, as_stream(stream)>>>(
                            E, bias, cat_map, ids, B, d, scale, 0, out, ldo, out + d, ldo)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gather_onehot_multi(int nsites, const float* const* E, const float* const* bias,
                            const int32_t* const* cat_map, const int32_t* const* ids,
                            const int64_t* n, int d, const float* scale, float* const* out,
                            const int64_t* ldo, float* const* bias_out, void* stream) {
  return arx_gather_onehot_multi_ld(nsites, E, bias, nullptr, cat_map, ids, n, d, scale, out, ldo, bias_out, nullptr,
                                    stream);
}

int arx_gather_onehot_multi_ld(int nsites, const float* const* E, const float* const* bias, const int64_t* ldbi,
                               const int32_t* const* cat_map, const int32_t* const* ids,
                               const int64_t* n, int d, const float* scale, float* const* out,
                               const int64_t* ldo, float* const* bias_out, const int64_t* ldb, void* stream) {
  ARX_CHECK_ARG(nsites >= 1 && nsites <= kMaxSites, "arx_gather_onehot_multi: 1..8 sites");
  ARX_CHECK_ARG(E && ids && n && out && ldo && scale, "arx_gather_onehot_multi: null pointer");
  int rc = check_d("arx_gather_onehot_multi", d);
  if (rc) return rc;
  GatherSites gs = {};
  int64_t tot = 0;
  for (int s = 0; s < nsites; ++s) {
    ARX_CHECK_ARG(E[s] && n[s] >= 0 && (n[s] == 0 || (ids[s] && out[s])), "arx_gather_onehot_multi: bad site");
    ARX_CHECK_ARG(ldo[s] % 4 == 0 && ldo[s] >= d && aligned16(E[s]) && aligned16(out[s]),
                  "arx_gather_onehot_multi: ldo %% 4 and 16-byte alignment required");
    ARX_CHECK_ARG(!(bias_out && bias_out[s] && !(bias && bias[s])), "arx_gather_onehot_multi: bias_out requires bias");
    gs.E[s] = E[s];
    gs.bias[s] = bias ? bias[s] : nullptr;
    gs.cat_map[s] = cat_map ? cat_map[s] : nullptr;
    gs.ids[s] = ids[s];
    gs.out[s] = out[s];
    gs.bias_out[s] = bias_out ? bias_out[s] : nullptr;
    gs.ldo[s] = ldo[s];
    gs.ldb[s] = ldb ? ldb[s] : 1;
    gs.ldbi[s] = ldbi ? ldbi[s] : 1;
    ARX_CHECK_ARG(gs.ldb[s] >= 1 && gs.ldbi[s] >= 1, "arx_gather_onehot_multi: bad bias stride");
    gs.scale[s] = scale[s];
    gs.n[s] = n[s];
    tot += n[s];
  }
  gs.count = nsites;
  if (tot == 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t rpb = 4 * (64 / lpr) * kGatherUnroll;       // rows per workgroup (one site each)
  int64_t nblk = 0;
  for (int s = 0; s < nsites; ++s) {
    nblk += ceil_div(n[s], rpb);
    ARX_CHECK_ARG(nblk < (int64_t)0x7fffffff, "arx_gather_onehot_multi: too many rows");
    gs.blk_end[s] = (int32_t)nblk;
  }
  ARX_DISPATCH_LPR(lpr, (k_gather_onehot_multi<LPR><<<(int)nblk, 256, 0, as_stream(stream)>>>(gs, d)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_lookup_multi(int nsites, const float* const* E_id, const float* const* bias_id,
                     const int32_t* const* cat_map, const float* const* E_tok,
                     const float* const* bias_tok, const int32_t* const* vals,
                     const int32_t* const* starts, const int32_t* const* lens,
                     const int32_t* const* ids, const int64_t* n, int d, const float* scale,
                     float* const* out, const int64_t* ldo, float* const* bias_out, void* stream) {
  ARX_CHECK_ARG(nsites >= 1 && nsites <= kMaxSites, "arx_lookup_multi: 1..8 sites");
  ARX_CHECK_ARG(E_id && E_tok && ids && n && out && ldo && scale, "arx_lookup_multi: null pointer");
  int rc = check_d("arx_lookup_multi", d);
  if (rc) return rc;
  LookupSites ls = {};
  int64_t tot = 0;
  for (int s = 0; s < nsites; ++s) {
    ARX_CHECK_ARG((E_id[s] || E_tok[s]) && n[s] >= 0 && (n[s] == 0 || (ids[s] && out[s])),
                  "arx_lookup_multi: bad site (a one-hot table, a multi-hot table or both)");
    ARX_CHECK_ARG(!E_tok[s] || (vals && starts && lens && vals[s] && starts[s] && lens[s]),
                  "arx_lookup_multi: a multi-hot feature needs vals / starts / lens");
    ARX_CHECK_ARG(ldo[s] % 4 == 0 && ldo[s] >= d && (!E_id[s] || aligned16(E_id[s])) &&
                      (!E_tok[s] || aligned16(E_tok[s])) && aligned16(out[s]),
                  "arx_lookup_multi: ldo %% 4 and 16-byte alignment required");
    const bool wb = bias_out && bias_out[s];
    ARX_CHECK_ARG(!wb || ((!E_id[s] || (bias_id && bias_id[s])) && (!E_tok[s] || (bias_tok && bias_tok[s]))),
                  "arx_lookup_multi: bias_out requires the bias of every feature of the site");
    ls.E_id[s] = E_id[s];
    ls.bias_id[s] = (wb && E_id[s]) ? bias_id[s] : nullptr;
    ls.cat_map[s] = cat_map ? cat_map[s] : nullptr;
    ls.E_tok[s] = E_tok[s];
    ls.bias_tok[s] = (wb && E_tok[s]) ? bias_tok[s] : nullptr;
    ls.vals[s] = E_tok[s] ? vals[s] : nullptr;
    ls.starts[s] = E_tok[s] ? starts[s] : nullptr;
    ls.lens[s] = E_tok[s] ? lens[s] : nullptr;
    ls.ids[s] = ids[s];
    ls.out[s] = out[s];
    ls.bias_out[s] = wb ? bias_out[s] : nullptr;
    ls.ldo[s] = ldo[s];
    ls.scale[s] = scale[s];
    ls.n[s] = n[s];
    tot += n[s];
  }
  ls.count = nsites;
  if (tot == 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t rpb = 4 * (64 / lpr) * kLookupRows;
  int64_t nblk = 0;
  for (int s = 0; s < nsites; ++s) {
    nblk += ceil_div(n[s], rpb);
    ARX_CHECK_ARG(nblk < (int64_t)0x7fffffff, "arx_lookup_multi: too many rows");
    ls.blk_end[s] = (int32_t)nblk;
  }
  ARX_DISPATCH_LPR(lpr, (k_lookup_multi<LPR><<<(int)nblk, 256, 0, as_stream(stream)>>>(ls, d)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gather_mulhot_mean_fwd(const float* E, const float* bias, const int32_t* vals,
                               const int32_t* starts, const int32_t* lens, const int32_t* ids,
                               int64_t B, int d, float scale, int accumulate, float* out,
                               int64_t ldo, float* bias_out, void* stream) {
  ARX_CHECK_ARG(E && vals && starts && lens && ids && out,
                "arx_gather_mulhot_mean_fwd: null pointer");
  ARX_CHECK_ARG(!(bias_out && !bias), "arx_gather_mulhot_mean_fwd: bias_out requires bias");
  int rc = check_d("arx_gather_mulhot_mean_fwd", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldo % 4 == 0 && ldo >= d && aligned16(E) && aligned16(out),
                "arx_gather_mulhot_mean_fwd: ldo %% 4 and 16-byte alignment required");
  if (B <= 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(B, 64 / lpr);
  const float* b = bias_out ? bias : nullptr;
  ARX_DISPATCH_LPR(lpr, (k_gather_mulhot<LPR><<<grid_waves(nwaves), 256, 0, as_stream(stream)>>>(
                            E, b, vals, starts, lens, ids, B, d, scale, accumulate, out, ldo,
                            bias_out, nullptr, nullptr, nullptr)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gather_id_plus_bag(const float* E_id, const float* bias_id, const int32_t* cat_map,
                           const float* E_tok, const float* bias_tok, const int32_t* vals,
                           const int32_t* starts, const int32_t* lens, const int32_t* ids, int64_t B,
                           int d, float scale, int accumulate, float* out, int64_t ldo,
                           float* bias_out, void* stream) {
  ARX_CHECK_ARG(E_id && E_tok && vals && starts && lens && ids && out, "arx_gather_id_plus_bag: null pointer");
  ARX_CHECK_ARG(!(bias_out && !(bias_id && bias_tok)), "arx_gather_id_plus_bag: bias_out requires both biases");
  int rc = check_d("arx_gather_id_plus_bag", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldo % 4 == 0 && ldo >= d && aligned16(E_id) && aligned16(E_tok) && aligned16(out),
                "arx_gather_id_plus_bag: ldo %% 4 and 16-byte alignment required");
  if (B <= 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(B, 64 / lpr);
  ARX_DISPATCH_LPR(lpr, (k_gather_mulhot<LPR><<<grid_waves(nwaves), 256, 0, as_stream(stream)>>>(
                            E_tok, bias_out ? bias_tok : nullptr, vals, starts, lens, ids, B, d, scale,
                            accumulate, out, ldo, bias_out, E_id, bias_out ? bias_id : nullptr, cat_map)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dot_score_fwd(const float* U, int64_t ldu, const float* T, int64_t ldt,
                      const float* tbias, int64_t B, int d, float* score, void* stream) {
  ARX_CHECK_ARG(U && T && score, "arx_dot_score_fwd: null pointer");
  int rc = check_d("arx_dot_score_fwd", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldu % 4 == 0 && ldt % 4 == 0 && aligned16(U) && aligned16(T),
                "arx_dot_score_fwd: leading dims %% 4 and 16-byte alignment required");
  if (B <= 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(B, 64 / lpr);
  ARX_DISPATCH_LPR(lpr, (k_dot_fwd<LPR><<<grid_waves(nwaves), 256, 0, as_stream(stream)>>>(
                            U, ldu, T, ldt, tbias, B, d, score)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dot_score_bwd(const float* U, int64_t ldu, const float* T, int64_t ldt,
                      const float* dscore, int64_t B, int d, float* dU, int64_t lddu,
                      int acc_dU, float* dT, int64_t lddt, void* stream) {
  ARX_CHECK_ARG(U && T && dscore && dU, "arx_dot_score_bwd: null pointer");
  int rc = check_d("arx_dot_score_bwd", d);
  if (rc) return rc;
  ARX_CHECK_ARG(ldu % 4 == 0 && ldt % 4 == 0 && lddu % 4 == 0 && (!dT || lddt % 4 == 0),
                "arx_dot_score_bwd: leading dims must be multiples of 4");
  if (B <= 0) return ARX_OK;
  int64_t total = B * (d / 4);
  int64_t g = ceil_div(total, 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_dot_bwd<<<(int)g, 256, 0, as_stream(stream)>>>(U, ldu, T, ldt, dscore, B, d / 4, dU, lddu,
                                                   acc_dU, dT, lddt);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_sparse_site_onehot(const int32_t* cat_map, const int32_t* ids, int64_t n,
                           int32_t row_base, float coef, int32_t* keys_out, int32_t* src_out,
                           float* coef_out, void* stream) {
  ARX_CHECK_ARG(ids && keys_out, "arx_sparse_site_onehot: null pointer");
  if (n <= 0) return ARX_OK;
  int64_t g = ceil_div(n, 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_site_onehot<<<(int)g, 256, 0, as_stream(stream)>>>(cat_map, ids, n, row_base, coef, keys_out,
                                                       src_out, coef_out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_pool_blocks(const int32_t* ids, int64_t S, int world, int rank, int32_t zero_row, int64_t cap,
                    int32_t* counts, int32_t* gidx, int32_t* my_slots, int32_t* pool_rows, void* stream) {
  ARX_CHECK_ARG(ids && counts && world > 0 && world <= 256 && rank >= 0 && rank < world, "arx_pool_blocks: bad argument (world <= 256)");
  ARX_CHECK_ARG(S > 0 && S < (1ll << 31), "arx_pool_blocks: 1 <= S < 2^31");
  ARX_CHECK_ARG(cap == 0 || (gidx && my_slots && pool_rows), "arx_pool_blocks: cap > 0 needs the three outputs");
  k_pool_blocks<<<1, 1024, 0, as_stream(stream)>>>(ids, (int)S, world, rank, zero_row, (int)cap, counts, gidx,
                                                    my_slots, pool_rows);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_shard_route(const int32_t* ids, int64_t n, int world, int rank, int32_t zero_row,
                    int32_t* rows_out, int32_t* keys_out, void* stream) {
  ARX_CHECK_ARG(ids && world > 0 && rank >= 0 && rank < world, "arx_shard_route: bad argument");
  if (n <= 0) return ARX_OK;
  int64_t g = ceil_div(n, 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_shard_route<<<(int)g, 256, 0, as_stream(stream)>>>(ids, n, world, rank, zero_row, rows_out, keys_out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_copy_2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows,
                int64_t cols, void* stream) {
  ARX_CHECK_ARG(src && dst, "arx_copy_2d: null pointer");
  ARX_CHECK_ARG(cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "arx_copy_2d: cols/ld must be multiples of 4");
  if (rows <= 0 || cols <= 0) return ARX_OK;
  int64_t g = ceil_div(rows * (cols / 4), 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_copy_2d<<<(int)g, 256, 0, as_stream(stream)>>>(src, lds, dst, ldd, rows, (int)(cols / 4));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_copy_strided_f32(const float* src, int64_t src_stride, float* dst, int64_t dst_stride,
                         int64_t n, void* stream) {
  ARX_CHECK_ARG(src && dst, "arx_copy_strided_f32: null pointer");
  ARX_CHECK_ARG(src_stride >= 1 && dst_stride >= 1, "arx_copy_strided_f32: strides must be >= 1");
  if (n <= 0) return ARX_OK;
  int64_t g = ceil_div(n, 256);
  const int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_copy_strided<<<(int)g, 256, 0, as_stream(stream)>>>(src, src_stride, dst, dst_stride, n);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

size_t arx_csr_expand_workspace_bytes(int64_t B) {
  int64_t nb = ceil_div(B > 0 ? B : 1, kExpBlock);
  return (size_t)(nb + 1) * sizeof(int32_t);
}

int arx_csr_expand(const int32_t* vals, const int32_t* starts, const int32_t* lens,
                   const int32_t* row_ids, int64_t B, int32_t* token_ids, int32_t* segids,
                   int64_t capacity, int32_t* offsets, int32_t* total_out, int32_t pad_token,
                   int32_t pad_seg, int32_t seg_base, float coef_scale, float* coef_out,
                   void* workspace, size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(vals && starts && lens && token_ids && offsets, "arx_csr_expand: null pointer");
  ARX_CHECK_ARG(B >= 0 && capacity >= 0, "arx_csr_expand: negative size");
  if (workspace_bytes < arx_csr_expand_workspace_bytes(B) || !workspace) {
    set_error("arx_csr_expand: workspace too small (%zu < %zu)", workspace_bytes,
              arx_csr_expand_workspace_bytes(B));
    return ARX_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  int32_t* block_tot = reinterpret_cast<int32_t*>(workspace);
  const int nb = (int)ceil_div(B > 0 ? B : 1, kExpBlock);
  k_expand_local<<<nb, kExpBlock, 0, s>>>(lens, row_ids, B, offsets, block_tot);
  ARX_CHECK_LAUNCH();
  k_expand_scan_blocks<<<1, kExpBlock, 0, s>>>(block_tot, nb, offsets + B, total_out);
  ARX_CHECK_LAUNCH();
  k_expand_fix<<<nb, kExpBlock, 0, s>>>(offsets, block_tot, B);
  ARX_CHECK_LAUNCH();
  {
    int64_t g = ceil_div(capacity > 0 ? capacity : 1, 256);
    int64_t cap = (int64_t)cu_count() * 16;
    if (g > cap) g = cap;
    k_expand_emit<<<(int)g, 256, 0, s>>>(vals, starts, lens, row_ids, B, offsets, token_ids, segids,
                                         capacity, pad_token, pad_seg, seg_base, coef_scale, coef_out);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

int arx_bag_expand_padded(const int32_t* vals, const int32_t* starts, const int32_t* lens,
                          const int32_t* row_ids, int64_t B, int max_len, int32_t pad_token,
                          int32_t seg_base, float coef_scale, int32_t* token_ids, int32_t* segids,
                          float* coef_out, void* stream) {
  ARX_CHECK_ARG(vals && starts && lens && token_ids && segids, "arx_bag_expand_padded: null pointer");
  ARX_CHECK_ARG(B >= 0 && max_len > 0, "arx_bag_expand_padded: bad size");
  if (B == 0) return ARX_OK;
  int64_t g = ceil_div(B * (int64_t)max_len, 256);
  const int64_t cap = (int64_t)cu_count() * 16;
  if (g > cap) g = cap;
  k_bag_expand_padded<<<(int)g, 256, 0, as_stream(stream)>>>(vals, starts, lens, row_ids, B, max_len,
                                                             pad_token, seg_base, coef_scale, token_ids,
                                                             segids, coef_out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
