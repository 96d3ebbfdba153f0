// csc.hip -- K7, multi-hot tables: the token order of the bags is STATIC, so the step never sorts it.
//
// Reference semantics (embed_attribute.py:397-406 embedding_lookup of the bag tokens + unsorted_segment_sum /
// length; hmf_model.py:146-151 one Adagrad apply per variable): a token row's gradient is the sum, over the
// entities of the step whose bag holds the token, of the entity's merged, 1/len-scaled gradient row.
//
// Rounds 2-6 expanded the distinct entities' bags into (token, source) pairs every step and sorted them by
// token: expansion + 2 radix passes = 7 dependent launches of the step's sort branch (~58 us in the C3 step,
// profiles/r06_c3_step_timeline.txt).  But the bag index -- vals / starts / lens of attributes/attribute.py,
// the feature CSR of mulhot_index.py -- never changes after _init_attributes: the TOKEN-MAJOR order of all
// its (token, entity) pairs can be built once per table (arx/ops.py BagCSC: place q of every CSR position,
// {token, entity} of every place).  A step then only has to say WHICH pairs are live:
//   k_csc_mark     one 16-lane group per entity lookup: flags[qpos[starts[e] + j]] = 1 for its bag (+ one coarse
//                  byte per 16 places), and
//                  slot_of[e] = the lookup's index (any of the duplicates: the row of the merged gradients Gu
//                  the one-hot apply writes for e; plain stores, no atomics -- marking is idempotent);
//   k_csc_compact  one sweep over the flag bytes in place order (thread = contiguous chunk, block totals
//                  published, ordered prefix by look-back over the blocks in front): the live pairs come out
//                  as the token-sorted list (token, Gu row, 1.0) the run extraction and the run-centric apply
//                  of group.hip already take, and the flags are zero again.
// Inside a token the entries follow place order = CSR position order = (entity, position in bag): for a
// standard CSR the same order the stable sort produced, so the sums -- and the tables -- are bit-identical.
// Flag bytes, not bits: setting a bit needs a device-scope atomic (~0.65 ns each on this part, group.hip);
// a byte store needs none.
#include <hip/hip_runtime.h>

#include "common.h"
#include "k7.h"

namespace arx {
namespace {

constexpr int kMarkLanes = 16;
constexpr int kCscThreads = 256;

__global__ __launch_bounds__(256) void k_csc_mark(MarkSites ms, const int32_t* __restrict__ starts,
                                                  const int32_t* __restrict__ lens, int max_len,
                                                  const int32_t* __restrict__ qpos, uint8_t* __restrict__ flags,
                                                  uint8_t* __restrict__ cflags, int32_t* __restrict__ slot_of, int32_t* __restrict__ zero_i,
                                                  int n_zero_i, unsigned long long* __restrict__ zero_l,
                                                  int n_zero_l) {
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: radix_sort.hip, "wave priority")
  if (blockIdx.x == 0) {              // the token pass's counter block + the sweep's look-back cells
    for (int t = threadIdx.x; t < n_zero_i; t += blockDim.x) zero_i[t] = 0;
    for (int t = threadIdx.x; t < n_zero_l; t += blockDim.x) zero_l[t] = 0ull;
  }
  const int64_t n0 = ms.offs[ms.n];
  const int lig = threadIdx.x % kMarkLanes;
  const int64_t g0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kMarkLanes;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x / kMarkLanes;
  for (int64_t i = g0; i < n0; i += gs) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < kMaxSites; ++k)
      if (k < ms.n && i >= ms.offs[k]) s = k;
    const int id = ms.ids[s][i - ms.offs[s]];
    const int e = id < 0 ? -1 : (ms.cat_map[s] ? ms.cat_map[s][id] : id);
    if (e < 0 || e >= ms.rows) continue;
    const int len = min(lens[e], max_len);
    const int64_t st = starts[e];
    if (lig == 0) slot_of[e] = (int32_t)i;
    for (int j = lig; j < len; j += kMarkLanes) {
      const int q = qpos[st + j];
      if (q >= 0) {
        flags[q] = (uint8_t)1;
        cflags[q >> 4] = (uint8_t)1;
      }
    }
  }
}

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// flags: 16-byte words [0, nw), and one COARSE byte per word (cw[w] != 0: the word holds a live place): at C3 one
// word in eight is live, so the sweep reads nq / 16 coarse bytes + the live words instead of nq bytes -- twice
// (3.6 MB for 20).  Thread t of block b owns the wpt words [(b * 256 + t) * wpt, + wpt), wpt a multiple of 16:
// one 16-byte load of coarse bytes per 16 words.  The order of the list is (block, thread, word, byte) = place
// order.  Every global round trip is wave-wide: (1) coarse bytes, (2) the live words, counted; (3) the block totals
// of the blocks in front (published before anybody waits; all polled at once); (4) coarse bytes + live words again
// (L2), live places into an LDS list; (5) + (6) {token, entity} and the entity's Gu row for one list entry per
// thread.
// Measured, C3 step (profiles/r06_csc_ab.txt): first form -- flag bytes only, each thread walking its own live places
// with two dependent gathers each and polling the blocks in front one after the other -- 72 us in the step; wave-wide
// round trips 27 us, but its 40 MB of flag reads slowed the scorer's row kernel beside it from 24 to 46 us; with
// coarse bytes 33 us in the step, 25 us alone: cold dependent round trips of 3 - 4 us each (the static arrays are
// 240 MB) are what is left.  A form without step (4) -- every thread keeps its first places in LDS slots of its own,
// 8 words per thread, 588 blocks, 49 VGPRs -- measured SLOWER in the step (40 us; C3-MIX 239 against 228 us): twice
// the blocks to place beside the scorer's workgroups, and one allocation granule too many to sit next to two
// k_sc_hinge waves; the merged gradients addressed by entity row (no step (6), 512 MB of address space) bought nothing
// either (C3-MIX 238 - 245 against 236 - 237 us).  This form: 294 blocks at C3, 46 VGPRs.
constexpr int kCscWin = 4096;           // LDS list entries per sweep of a block (more: the block sweeps again)
constexpr int kCscLb = 4;               // look-back cells polled at once per thread
constexpr int kCscMaxBlocksLb = kCscThreads * kCscLb;

__device__ __forceinline__ uint32_t coarse_word(const uint4& c, int h) {     // coarse bytes of words h .. h + 3
  return h < 4 ? c.x : h < 8 ? c.y : h < 12 ? c.z : c.w;
}

__global__ __launch_bounds__(kCscThreads) void k_csc_compact(uint4* __restrict__ fw, uint4* __restrict__ cw,
                                                            int64_t nw, int wpt, const int2* __restrict__ qte,
                                                            const int32_t* __restrict__ slot_of,
                                                            uint32_t* __restrict__ sk, int32_t* __restrict__ ssrc,
                                                            float* __restrict__ scoef, int64_t cap,
                                                            int32_t* __restrict__ n_out,
                                                            unsigned long long* __restrict__ lookback) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ int wsum[kCscThreads / 64];
  __shared__ int psum[kCscThreads / 64];
  __shared__ int s_list[kCscWin];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t w0 = min(nw, ((int64_t)blockIdx.x * kCscThreads + tid) * wpt);      // (multiples of 16; nw too)
  const int64_t w1 = min(nw, w0 + wpt);
  // (1), (2): live pairs of the thread's words (a flag byte is 0 or 1: popcount of the word)
  int cnt = 0;
  for (int64_t w = w0; w < w1; w += 16) {
    const uint4 c = cw[w >> 4];
    if (!(c.x | c.y | c.z | c.w)) continue;
#pragma unroll 1
    for (int h = 0; h < 16; h += 4) {
      const uint32_t cx = coarse_word(c, h);
      if (!cx) continue;
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ((cx >> (8 * u)) & 0xffu) ? fw[w + h + u] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int u = 0; u < 4; ++u) cnt += __popc(v[u].x) + __popc(v[u].y) + __popc(v[u].z) + __popc(v[u].w);
    }
  }
  const int incl = wave_incl_scan_i(cnt, lane);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kCscThreads / 64; ++q) {
    if (q < wv) wbase += wsum[q];
    total += wsum[q];
  }
  // (3) ordered prefix over the blocks: every block publishes its total before it waits, and only waits for
  // blocks in front of it (in-order dispatch: group.hip, k_runs_extract)
  if (tid == 0)
    __hip_atomic_store(&lookback[blockIdx.x], (1ull << 63) | (unsigned long long)total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  int part = 0;
  {
    unsigned long long v[kCscLb];
#pragma unroll
    for (int u = 0; u < kCscLb; ++u) {
      const int q = tid + u * kCscThreads;
      v[u] = q < (int)blockIdx.x ? __hip_atomic_load(&lookback[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : (1ull << 63);
    }
#pragma unroll
    for (int u = 0; u < kCscLb; ++u) {
      const int q = tid + u * kCscThreads;
      while (!(v[u] >> 63)) {
        __builtin_amdgcn_s_sleep(2);
        v[u] = __hip_atomic_load(&lookback[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      part += (int)(v[u] & 0x7fffffffull);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) psum[wv] = part;
  __syncthreads();
  int prefix = 0;
#pragma unroll
  for (int q = 0; q < kCscThreads / 64; ++q) prefix += psum[q];
  if (tid == 0 && blockIdx.x == gridDim.x - 1) *n_out = (int32_t)min((int64_t)prefix + total, cap);
  if (total == 0) return;                             // (block-uniform)
  // (4) .. (6): sweeps of kCscWin list entries (one, unless a block holds more live pairs than that)
  const int my0 = wbase + incl - cnt;                 // the thread's first entry in the block's list
  for (int base = 0; base < total; base += kCscWin) {
    if (cnt > 0 && my0 < base + kCscWin && my0 + cnt > base) {
      const bool last = my0 + cnt <= base + kCscWin;  // the thread's last sweep: its flags are cleared
      int o = my0 - base;
      for (int64_t w = w0; w < w1; w += 16) {
        const uint4 c = cw[w >> 4];
        if (!(c.x | c.y | c.z | c.w)) continue;
        if (last) cw[w >> 4] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
        for (int h = 0; h < 16; h += 2) {              // (two words at a time: the kernel has to stay under 48 VGPRs)
          const uint32_t cx = (coarse_word(c, h) >> (8 * (h & 2))) & 0xffffu;
          if (!cx) continue;
          uint4 v[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) v[u] = ((cx >> (8 * u)) & 0xffu) ? fw[w + h + u] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (!(v[u].x | v[u].y | v[u].z | v[u].w)) continue;
            if (last) fw[w + h + u] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              uint32_t m = x[cc];
              while (m) {
                const int b = __builtin_ctz(m) >> 3;
                m &= ~(0xffu << (b * 8));
                if (o >= 0 && o < kCscWin) s_list[o] = (int)((w + h + u) * 16 + cc * 4 + b);
                ++o;
              }
            }
          }
        }
      }
    }
    __syncthreads();
    const int m = min(kCscWin, total - base);
    for (int k = tid; k < m; k += kCscThreads) {
      const int q = s_list[k];
      const int2 te = qte[q];
      const int sl = slot_of[te.y];
      const int64_t o = (int64_t)prefix + base + k;
      if (o < cap) {
        sk[o] = (uint32_t)te.x;
        ssrc[o] = sl;
        scoef[o] = 1.f;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_csc_mark(const MarkSites& ms, const BagCsc& csc, const int32_t* starts, const int32_t* lens,
                    int max_len, int32_t* zero_i, int n_zero_i, void* zero_l, int n_zero_l, hipStream_t s) {
  const int64_t n0 = ms.offs[ms.n];
  int64_t g = ceil_div((n0 > 0 ? n0 : 1) * kMarkLanes, 256);
  const int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_csc_mark<<<(int)g, 256, 0, s>>>(ms, starts, lens, max_len, csc.qpos, csc.flags, csc.cflags, csc.slot_of, zero_i,
                                    n_zero_i, reinterpret_cast<unsigned long long*>(zero_l), n_zero_l);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

// words per thread: 16 (256 places, one load of coarse bytes; every block reads the totals of all blocks in front of
// it -- blocks^2 / 2 cells through the fabric: 294 blocks at C3) unless that takes more blocks than one look-back
// round polls.  The word count is rounded up to 16 (flags: 256 places, coarse bytes: 16).
static void csc_plan(int64_t nq, int64_t* nw, int* wpt, int* blocks) {
  *nw = ceil_div(nq > 0 ? nq : 1, 256) * 16;
  int64_t w = 16;
  while (ceil_div(*nw, kCscThreads * w) > kCscMaxBlocksLb) w += 16;
  *wpt = (int)w;
  *blocks = (int)ceil_div(*nw, kCscThreads * w);
}

int csc_compact_blocks(int64_t nq) {
  int64_t nw;
  int wpt, g;
  csc_plan(nq, &nw, &wpt, &g);
  return g;
}

int launch_csc_compact(const BagCsc& csc, uint32_t* sk, int32_t* ssrc, float* scoef, int64_t cap,
                       int32_t* n_out, void* lookback, hipStream_t s) {
  int64_t nw;
  int wpt, g;
  csc_plan(csc.nq, &nw, &wpt, &g);
  k_csc_compact<<<g, kCscThreads, 0, s>>>(reinterpret_cast<uint4*>(csc.flags), reinterpret_cast<uint4*>(csc.cflags),
                                          nw, wpt, csc.qte, csc.slot_of, sk, ssrc, scoef, cap, n_out,
                                          reinterpret_cast<unsigned long long*>(lookback));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // namespace arx
