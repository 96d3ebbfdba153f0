// radix_sort.hip -- device-wide stable LSD radix sort of (key, src, coef) triples for K7
// (sparse Adagrad) when the batch exceeds the single-launch LDS rank sort (n > 8192).
//
// Written for hipGraph replay: kernel launches only.  rocPRIM 4.2's onesweep resets its
// ordered-block-id counter on gfx942/gfx950 with a BLOCKING hipMemset
// (rocprim/device/detail/ordered_block_id.hpp, reset_from_host) -- not captured, so a
// replayed graph walks off the lookback array -- and its merge sort takes 17 launches
// (~90 us) at n = 100k.
//
// Layout: <= 256 blocks, each owning one contiguous slice of ipb items split into four
// wave-contiguous quarters; 2-3 passes of <= 10 bits; three launches per pass:
//   k_rs_hist     per-WAVE digit histogram -> hist[blk*4 + wave][bin]
//   k_rs_scan     exclusive scan down every histogram column (<= 1024 rows), in place
//   k_rs_scatter  wave start offset = bin base + its scanned row; lanes with equal digits
//                 find each other with one ballot per digit bit and the lowest such lane
//                 bumps the wave's running counter in LDS.  Stable by construction
//                 (block < wave < round < lane).
// (An inline scan in the scatter kernel was measured first: 16 dependent batches of loads
// per block at 256 blocks, ~16 us per pass -- the extra launch is cheaper.)
// (Round 5: a ONE-SWEEP form of the two-pass sort -- both digits' histograms from one read of the keys, pass 1 with
// the column prefix formed inside the scatter kernel, pass 2 with a decoupled look-back over published tile
// histograms: 3 launches instead of 6 -- was built, passed every K7 test bit for bit, and measured SLOWER in the step:
// C3 291 against 231 us.  Beside the scorer's streaming kernels a small kernel pays ~3 us per DEPENDENT global round
// trip in its CU's memory queue whatever the bytes; the all-to-all prefix over tile histograms is either a scan
// launch (2 round trips spread over bins / 4 workgroups) or, inlined, tiles x bins loads per workgroup = many
// round trips in a row (k_rs1_scat 68 us in the step), and few fat tiles turn a wave's four-round batch into a
// serial chain of them.  Removed again)
// The first pass reads the caller's raw int32 keys (out-of-range ones become the
// sentinel) and raw src/coef; the payload travels with the key (coalesced 12 B per item
// per pass instead of two random 4-byte gathers -- 64 B sectors each -- at the end).
//
// Wave priority: the kernels of the step's sort branch (this file, k_runs_extract) raise their waves' issue
// priority (s_setprio 3).  They are short, latency-bound launches next to the scorer's long MFMA kernels; with
// priority they get through and free their slots sooner (same box, three alternating runs: C3 227.4 -> 226.0,
// C2 163.8 -> 162.4, C3-MIX 247.7 -> 245.7 us).  The opposite -- priority for the scorer's kernels -- measured
// slower (C3 228.2 -> 229.9).
#include <hip/hip_runtime.h>

#include "common.h"

namespace arx {

namespace {

// 128 threads: the scatter kernel's LDS (per-wave cursors + bin bases) is 24 KB then -- it has to fit the
// 32 KB the scorer GEMMs (2 x 64 KB per CU) leave, or the sort waits for a whole GEMM (measured: 52 us)
constexpr int kRsThreads = 128;
constexpr int kRsMaxBlocks = 512;
constexpr int kRsMaxBits = 11;    // 2 passes cover 22 key bits (a 1M-row table + the sentinel bit)
constexpr int kRsPer = (1 << kRsMaxBits) / kRsThreads;   // bins per thread in the bin-base scan
constexpr int kRsMaxBins = 1 << kRsMaxBits;

__device__ __forceinline__ uint32_t norm_key(int32_t k, uint32_t sentinel) {
  return (k == ARX_KEY_NONE || k < 0 || (uint32_t)k >= sentinel) ? sentinel : (uint32_t)k;
}

// RAW: 0 = keys of an earlier pass, 1 = the caller's raw int32 keys, 2 = the lookups of a CatSites set: the first
// pass forms key / source row / coefficient of contribution i itself (what k_site_keys wrote to three arrays for it
// to read back: one launch and 3 x 4 n bytes each way less at the head of the step's sort branch)
__device__ __forceinline__ uint32_t site_key(const CatSites& st, int64_t i, uint32_t sentinel, int32_t* src, float* coef) {
  int s = 0;
#pragma unroll
  for (int q = 1; q < kMaxSites; ++q)
    if (q < st.nsites && i >= st.offs[q]) s = q;
  const int64_t j = i - st.offs[s];
  const int id = st.ids[s][j];
  // (id < 0: an empty pool slot -- a short device draw leaves -1 -- looks nothing up: no map read, no contribution)
  const int key = id < 0 ? -1 : (st.cat_map[s] ? st.cat_map[s][id] : id);
  const int tb = st.table[s];
  const int64_t rows = tb == 0 ? st.rows[0] : tb == 1 ? st.rows[1] : tb == 2 ? st.rows[2] : st.rows[3];
  if (src) {
    *src = st.row_base[s] + (int32_t)j;
    *coef = st.coef[s];
  }
  return (key < 0 || key >= rows) ? sentinel : (uint32_t)((tb << st.kb) | key);
}

struct NoSites {};
template <int RAW> struct SitesArg { typedef NoSites type; };
template <> struct SitesArg<2> { typedef CatSites type; };

template <int RAW>
__device__ __forceinline__ uint32_t load_key(const void* keys_in, int64_t i, uint32_t sentinel) {
  if constexpr (RAW == 1) return norm_key(reinterpret_cast<const int32_t*>(keys_in)[i], sentinel);
  else return reinterpret_cast<const uint32_t*>(keys_in)[i];
}

constexpr int kRsWaves = kRsThreads / 64;

// Items per block when the live count is only known on the device (the grid is sized for the
// host-side capacity): the same rule as rs_plan, applied to the device count, so that the live
// items spread over ALL blocks of the grid instead of filling the first few of them.
__device__ __forceinline__ int64_t rs_ipb_dev(int64_t n, int nblk) {
  const int64_t per = (n + nblk - 1) / nblk;
  return ((per + kRsThreads - 1) / kRsThreads) * kRsThreads;
}

// per-WAVE digit histograms: hist[(blk * 4 + wave)][bin] over the wave's contiguous quarter
template <int RAW>
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const void* __restrict__ keys_in, int64_t n_host,
                                                        const int32_t* __restrict__ n_dev,
                                                        uint32_t sentinel, int shift, int bits,
                                                        int64_t ipb_host, int32_t* __restrict__ hist,
                                                        int32_t* __restrict__ list_count,
                                                        const int32_t* __restrict__ n_in_dev,
                                                        typename SitesArg<RAW>::type st) {
  // n_dev: live entries after the first pass dropped the sentinels (device-side count; the grid
  // is sized for the host-side capacity: the live entries are dealt over ALL its blocks)
  // (first pass: n_in_dev, if given, is the number of input entries -- a compacted list whose length
  // is only known on the device)
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: see radix_sort.hip, "wave priority")
  const int32_t* cnt = RAW ? n_in_dev : n_dev;
  const int64_t n = cnt ? min((int64_t)*cnt, n_host) : n_host;
  const int64_t ipb = cnt ? rs_ipb_dev(n, gridDim.x) : ipb_host;
  __shared__ __attribute__((aligned(16))) int h[kRsWaves][kRsMaxBins];
  const int bins = 1 << bits;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
#pragma unroll 4
  for (int b = lane; b < bins; b += 64) h[w][b] = 0;
  // the caller's 2 KB counter block: [0], [1] run lists of the window apply, [2] live count (written by this
  // pass's scatter), [8 ..] counters of the run-centric apply, [32 ..] its look-back cells (k7.h)
  if (list_count && blockIdx.x == 0)
    for (int t = threadIdx.x; t < 512; t += kRsThreads) list_count[t] = 0;
  const int64_t base = blockIdx.x * ipb;
  const int64_t end = min(n, base + ipb);
  const int64_t ipw = ipb / kRsWaves;
  const int64_t wb = base + w * ipw;
  const int64_t we = min(end, wb + ipw);
  const uint32_t mask = (uint32_t)bins - 1u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (register budget: the kernels of this file run beside the scorer's k_sc_hinge in the step, two waves of which
  // leave 48 VGPRs of a SIMD's register file -- with 62, from a 16-fold unrolled row copy below, this kernel waited
  // for the whole hinge launch whenever the two coincided: 5 -> 23 us, r04 timelines)
  for (int64_t i0 = wb + lane; i0 < we; i0 += 8 * 64) {
    uint32_t k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // 8 loads in flight: the loop is a chain of L2 round trips otherwise
      const int64_t i = i0 + u * 64;
      if constexpr (RAW == 2) k[u] = (i < we) ? site_key(st, i, sentinel, nullptr, nullptr) : 0u;
      else k[u] = (i < we) ? load_key<RAW>(keys_in, i, sentinel) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 64 < we && !(RAW && k[u] >= sentinel))   // pads / dropped keys are not sorted at all
        atomicAdd(&h[w][(k[u] >> shift) & mask], 1);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int32_t* row = hist + ((int64_t)blockIdx.x * kRsWaves + w) * bins;
  if (bins >= 256) {               // (16-byte pieces, two in flight: register budget above)
#pragma unroll 2
    for (int b = lane * 4; b < bins; b += 256)
      *reinterpret_cast<int4*>(row + b) = *reinterpret_cast<const int4*>(&h[w][b]);
  } else {
#pragma unroll 1
    for (int b = lane; b < bins; b += 64) row[b] = h[w][b];
  }
}

// Exclusive scan over the rows (waves, in sort order) of each histogram column, in place;
// column totals to tot[].  One workgroup per 4 columns (int4) of AT MOST 256 threads: thread t owns the kRows =
// ceil(rows / threads) consecutive rows [t * kRows, ...) -- all its loads in flight together, a serial prefix over
// them, one workgroup scan over the thread totals.  (One thread per row made the token list's scan, ~590 rows, a
// 640-thread workgroup that waits for ten free wave slots on ONE CU beside the step's GEMMs: 7.6 us alone, 15-17 us
// in the step, on the chain that ends it.)
constexpr int kScanThreads = 256;
constexpr int kScanRowsMax = (kRsMaxBlocks * (kRsThreads / 64) + kScanThreads - 1) / kScanThreads;   // 4
__global__ __launch_bounds__(kScanThreads) void k_rs_scan(int32_t* __restrict__ hist, int nrows, int bins,
                                                         int32_t* __restrict__ tot) {
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: see radix_sort.hip, "wave priority")
  __shared__ int4 wsum[kScanThreads / 64];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int per = (nrows + (int)blockDim.x - 1) / (int)blockDim.x;        // <= kScanRowsMax
  const int r0 = t * per;
  int4 v[kScanRowsMax];
#pragma unroll
  for (int q = 0; q < kScanRowsMax; ++q) {
    const int r = r0 + q;
    v[q] = (q < per && r < nrows) ? *(reinterpret_cast<const int4*>(hist + (int64_t)r * bins) + blockIdx.x)
                                  : make_int4(0, 0, 0, 0);
  }
  int4 sum = make_int4(0, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < kScanRowsMax; ++q) { sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
  int4 incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int tx = __shfl_up(incl.x, o, 64), ty = __shfl_up(incl.y, o, 64);
    const int tz = __shfl_up(incl.z, o, 64), tw = __shfl_up(incl.w, o, 64);
    if (lane >= o) { incl.x += tx; incl.y += ty; incl.z += tz; incl.w += tw; }
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int4 off = make_int4(0, 0, 0, 0);
  for (int ww = 0; ww < w; ++ww) {
    off.x += wsum[ww].x; off.y += wsum[ww].y; off.z += wsum[ww].z; off.w += wsum[ww].w;
  }
  int4 run = make_int4(off.x + incl.x - sum.x, off.y + incl.y - sum.y, off.z + incl.z - sum.z,
                       off.w + incl.w - sum.w);                            // rows in front of this thread's
#pragma unroll
  for (int q = 0; q < kScanRowsMax; ++q) {
    const int r = r0 + q;
    if (q < per && r < nrows) {
      *(reinterpret_cast<int4*>(hist + (int64_t)r * bins) + blockIdx.x) = run;
      run.x += v[q].x; run.y += v[q].y; run.z += v[q].z; run.w += v[q].w;
      if (r == nrows - 1) reinterpret_cast<int4*>(tot)[blockIdx.x] = run;
    }
  }
}

template <int RAW>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(
    const void* __restrict__ keys_in, const int32_t* __restrict__ src_in,
    const float* __restrict__ coef_in, int64_t n_host, int32_t* __restrict__ n_live,
    uint32_t sentinel, int shift, int bits, int64_t ipb_host, const int32_t* __restrict__ hist,
    const int32_t* __restrict__ tot, uint32_t* __restrict__ keys_out, int32_t* __restrict__ src_out,
    float* __restrict__ coef_out, const int32_t* __restrict__ n_in_dev, typename SitesArg<RAW>::type st) {
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: see radix_sort.hip, "wave priority")
  const int32_t* cnt = RAW ? n_in_dev : n_live;
  const int64_t n = cnt ? min((int64_t)*cnt, n_host) : n_host;
  const int64_t ipb = cnt ? rs_ipb_dev(n, gridDim.x) : ipb_host;
  constexpr int NW = kRsWaves;
  __shared__ int wcnt[NW][kRsMaxBins];
  __shared__ int gbase[kRsMaxBins];
  __shared__ int wsum[NW];
  const int bins = 1 << bits;
  const uint32_t mask = (uint32_t)bins - 1u;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  // ---- bin bases: exclusive scan of the column totals (thread t owns a contiguous chunk);
  // this wave's starting offsets: bin base + rows before it (k_rs_scan)
  {
    const int per = (bins + kRsThreads - 1) / kRsThreads;   // 1..kRsPer
    const int b0 = threadIdx.x * per;
    int tv[kRsPer];
    int s = 0;
#pragma unroll
    for (int u = 0; u < kRsPer; ++u) {
      tv[u] = (u < per && b0 + u < bins) ? tot[b0 + u] : 0;
      s += tv[u];
    }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int ww = 0; ww < w; ++ww) woff += wsum[ww];
    int run = woff + incl - s;
#pragma unroll
    for (int u = 0; u < kRsPer; ++u)
      if (u < per && b0 + u < bins) {
        gbase[b0 + u] = run;
        run += tv[u];
      }
    // first pass: the grand total = entries that survive (sentinels dropped)
    if (RAW && n_live && blockIdx.x == 0 && threadIdx.x == kRsThreads - 1) *n_live = run;
  }
  __syncthreads();
  {
    const int32_t* row = hist + ((int64_t)blockIdx.x * NW + w) * bins;
    for (int b = lane; b < bins; b += 64) wcnt[w][b] = gbase[b] + row[b];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- stable rank + scatter over the wave's contiguous quarter, 4 rounds of loads in flight
  const int64_t base = blockIdx.x * ipb;
  const int64_t end = min(n, base + ipb);
  const int64_t ipw = ipb / NW;
  const int64_t wb = base + w * ipw;
  const int64_t we = min(end, wb + ipw);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int64_t i0 = wb; i0 < we; i0 += 4 * 64) {
    uint32_t k[4];
    int32_t sv[4];
    float cv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 64 + lane;
      const bool valid = i < we;
      if constexpr (RAW == 2) {
        sv[u] = 0;
        cv[u] = 0.f;
        k[u] = valid ? site_key(st, i, sentinel, &sv[u], &cv[u]) : 0u;
      } else {
        k[u] = valid ? load_key<RAW>(keys_in, i, sentinel) : 0u;
      }
      if constexpr (RAW == 2) {
      } else if constexpr (RAW == 1) {
        sv[u] = (valid && src_in) ? src_in[i] : (int32_t)i;
        cv[u] = (valid && coef_in) ? coef_in[i] : 1.f;
      } else {
        sv[u] = valid ? src_in[i] : 0;
        cv[u] = valid ? coef_in[i] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * 64 + lane;
      const bool valid = (i < we) && !(RAW && k[u] >= sentinel);
      if (i0 + u * 64 >= we) break;                 // wave-uniform
      const uint32_t dgt = (k[u] >> shift) & mask;
      unsigned long long peers = __ballot(valid);
      for (int bit = 0; bit < bits; ++bit) {
        const bool one = (dgt >> bit) & 1u;
        const unsigned long long bb = __ballot(one);
        peers &= one ? bb : ~bb;
      }
      if (valid) {
        const int rank_in = __popcll(peers & lt);
        const int old = wcnt[w][dgt];
        if (rank_in == 0) wcnt[w][dgt] = old + __popcll(peers);
        const int64_t dst = (int64_t)old + rank_in;
        keys_out[dst] = k[u];
        src_out[dst] = sv[u];
        coef_out[dst] = cv[u];
      }
    }
  }
}


struct RsPlan {
  int nblk, passes, bits[4], shift[4];
  int64_t ipb;
};

RsPlan rs_plan(int64_t n, int total_bits) {
  RsPlan p;
  if (total_bits < 2) total_bits = 2;
  p.passes = (total_bits + kRsMaxBits - 1) / kRsMaxBits;
  const int r = (total_bits + p.passes - 1) / p.passes;
  int sh = 0;
  for (int i = 0; i < p.passes; ++i) {
    p.bits[i] = (total_bits - sh) < r ? (total_bits - sh) : r;
    p.shift[i] = sh;
    sh += p.bits[i];
  }
  int64_t nblk = ceil_div(n, 4 * kRsThreads);   // one 4-round batch per wave at small n (latency-bound there)
  if (nblk > kRsMaxBlocks) nblk = kRsMaxBlocks;
  if (nblk < 1) nblk = 1;
  p.ipb = ceil_div(ceil_div(n, nblk), kRsThreads) * kRsThreads;
  p.nblk = (int)ceil_div(n, p.ipb);
  return p;
}

}  // namespace

size_t radix_sort_hist_bytes() {   // per-wave rows + the column totals
  return (size_t)(kRsMaxBlocks * kRsWaves + 1) * kRsMaxBins * sizeof(int32_t);
}



int launch_radix_sort(const int32_t* keys_raw, const int32_t* src_raw, const float* coef_raw, int64_t n,
                      uint32_t sentinel, int total_bits, uint32_t* keys_tmp, uint32_t* keys_out,
                      int32_t* src_tmp, int32_t* src_out, float* coef_tmp, float* coef_out,
                      int32_t* hist, int32_t* list_count, int32_t* n_live, hipStream_t s,
                      const int32_t* n_in_dev, const CatSites* sites) {
  const RsPlan p = rs_plan(n, total_bits);
  const void* in_k = keys_raw;
  const int32_t* in_s = src_raw;
  const float* in_c = coef_raw;
  for (int i = 0; i < p.passes; ++i) {
    // the last pass must write the *_out buffers: passes alternate backwards from there
    const bool to_out = ((p.passes - 1 - i) % 2) == 0;
    uint32_t* out_k = to_out ? keys_out : keys_tmp;
    int32_t* out_s = to_out ? src_out : src_tmp;
    float* out_c = to_out ? coef_out : coef_tmp;
    int32_t* tot = hist + (int64_t)kRsMaxBlocks * kRsWaves * kRsMaxBins;
    const int bins = 1 << p.bits[i];
    if (i == 0 && sites)
      k_rs_hist<2><<<p.nblk, kRsThreads, 0, s>>>(in_k, n, n_live, sentinel, p.shift[i], p.bits[i], p.ipb,
                                                 hist, list_count, n_in_dev, *sites);
    else if (i == 0)
      k_rs_hist<1><<<p.nblk, kRsThreads, 0, s>>>(in_k, n, n_live, sentinel, p.shift[i], p.bits[i], p.ipb,
                                                 hist, list_count, n_in_dev, NoSites{});
    else
      k_rs_hist<0><<<p.nblk, kRsThreads, 0, s>>>(in_k, n, n_live, sentinel, p.shift[i], p.bits[i],
                                                 p.ipb, hist, nullptr, nullptr, NoSites{});
    ARX_CHECK_LAUNCH();
    {
      // (at most 256 threads per workgroup: a 16-wave workgroup waits for a whole CU's worth of slots next to the
      // step's GEMMs -- measured 44 us for a 132-row scan)
      const int rows = p.nblk * kRsWaves;
      const int thr = rows < kScanThreads ? (rows + 63) / 64 * 64 : kScanThreads;
      k_rs_scan<<<bins / 4, thr, 0, s>>>(hist, rows, bins, tot);
    }
    ARX_CHECK_LAUNCH();
    if (i == 0 && sites)
      k_rs_scatter<2><<<p.nblk, kRsThreads, 0, s>>>(in_k, in_s, in_c, n, n_live, sentinel, p.shift[i], p.bits[i],
                                                    p.ipb, hist, tot, out_k, out_s, out_c, n_in_dev, *sites);
    else if (i == 0)
      k_rs_scatter<1><<<p.nblk, kRsThreads, 0, s>>>(in_k, in_s, in_c, n, n_live, sentinel, p.shift[i], p.bits[i],
                                                    p.ipb, hist, tot, out_k, out_s, out_c, n_in_dev, NoSites{});
    else
      k_rs_scatter<0><<<p.nblk, kRsThreads, 0, s>>>(in_k, in_s, in_c, n, n_live, sentinel, p.shift[i], p.bits[i],
                                                    p.ipb, hist, tot, out_k, out_s, out_c, nullptr, NoSites{});
    ARX_CHECK_LAUNCH();
    in_k = out_k;
    in_s = out_s;
    in_c = out_c;
  }
  return ARX_OK;
}

}  // namespace arx
