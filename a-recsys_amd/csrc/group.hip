// group.hip -- K7 apply, run-centric: one sub-group per touched row, no windows, no finish launch.
//
// Reference semantics (hmf_model.py:146-151, seqModel.py:173-182, embed_attribute.py:397-400;
// TF-1.0 AdagradOptimizer on IndexedSlices): duplicate rows are summed, every touched row gets
// ONE update  acc += g^2 ; w -= lr * g / sqrt(acc).
//
// Input: the stable radix sort of the pass (radix_sort.hip): keys ascending, (src, coef) in
// contribution order inside a run.  The window kernels of optim.hip cut that list into 64-position
// windows, so runs that cross a window leave partial rows and need a second (finish) launch, and a
// window with one long run has one busy sub-group.  Here
//   k_runs_extract (side branch, ids only)  every run becomes a record {key, offset, length, head position}
//                  + its first entry {src, coef}; runs longer than a sub-group are cut into 256-entry
//                  work items;
//   k_run_apply    blocks [0, glong): one workgroup per work item of a long run (partial rows travel as
//                  agent-scope atomics; the last item to arrive -- ticket -- adds them up in item
//                  order); the other blocks: one sub-group per short run -- round trip 1 the record,
//                  round trip 2 table row + slot row + bias cells + first gradient row + the other
//                  entries, round trip 3 the other gradient rows.
// The partition of a run into sub-sums depends on its length alone and every sub-sum runs in sorted
// (= contribution) order, so the result is bit-reproducible.
//
// (A direct-address grouping of the contributions -- integer atomics on per-row maps, or owner
// workgroups counting in LDS -- was built and measured in round 3 instead of the sort: per-entry
// global atomics cost ~0.65 ns each on this part, ~10 ns on one address, and the 1024-thread /
// 64 KB-LDS counting workgroups took the CUs from the scorer GEMMs they ran under (dU 46 -> 86 us).
// The radix sort's small launches stay)
#include <hip/hip_runtime.h>
#include <limits.h>
#include <cstdlib>

#include "common.h"
#include "k7.h"

namespace arx {
namespace {

// (round 6, alternating runs on one box: 256 -> 128 entries per item: C2 157.8 -> 151.6 us, C3 / C3-MIX / C4 within
// +-1 us; 64: C3 +6 us; 8 instead of 4 gradient rows in flight per sub-group: +-1 -- profiles/r06_run_item_ab.txt)
#ifndef ARX_RUN_ITEM
#define ARX_RUN_ITEM 128
#endif
constexpr int kItem = ARX_RUN_ITEM;   // entries per work item of a long run (one 256-thread workgroup: kItem / 8 per sub-group, <= its lanes)

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// ---- run records from the sorted list ------------------------------------------------------------
// A workgroup takes a tile of 8192 sorted positions, 8 consecutive ones per thread (two 16-byte key
// loads).  Heads are marked in an LDS bitmap; a head finds its run's end as the next marked bit (the
// bitmap also covers the 64 positions behind the tile) or, past 128 positions, by a 16-ary search in
// the sorted keys.  Per workgroup: one atomic per list counter (few workgroups: ~10 ns each on one
// address), and -- for the riding bag table -- the ORDERED prefix of the bag lengths of the entity
// heads (published totals of the workgroups in front: the token list they address stays deterministic).
constexpr int kExtThreads = 128;     // (small workgroups: they must find a slot next to the GEMMs of the step; 128: twice the
                                     // workgroups for the ~75 live tiles of the C3 token list -- the last kernel of the sort branch, 2 us)
constexpr int kExtPer = 8;
constexpr int kExtTile = kExtThreads * kExtPer;

struct HeadLens {                 // optional: compacted bag expansion riding on this list
  const int32_t* lens;            // bag length per entity (null: no such output)
  int ent_kb;                     // entity keys: (key >> ent_kb) == ent_tag
  uint32_t ent_tag;
  int max_len;
  int32_t* hoff;                  // [hoff_n] exclusive offsets of the heads' bags in the compact token list
  int64_t hoff_n;                 // (entity keys sort first: positions past their count carry no bag)
  int32_t* total;                 // its length
  unsigned long long* lookback;   // [blocks] published totals (zeroed by the sort's first launch)
};

__global__ __launch_bounds__(kExtThreads) void k_runs_extract(const uint32_t* __restrict__ sk,
                                                              const int32_t* __restrict__ ssrc,
                                                              const float* __restrict__ scoef, int64_t n_host,
                                                              const int32_t* __restrict__ n_dev, uint32_t sentinel,
                                                              RunLists rl, int kshort, HeadLens hl) {
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: see radix_sort.hip, "wave priority")
  __shared__ uint32_t bm[kExtTile / 32 + 2];             // head (or dead) flags of the tile + 64 positions
  __shared__ int wtot[2][kExtThreads / 64];
  __shared__ int s_cnt[4], s_base[5], s_cnt4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
  const int64_t t0 = (int64_t)blockIdx.x * kExtTile;
  const bool dead_tile = t0 >= n;
  if (dead_tile && !hl.lens) return;                     // (whole workgroup; with a look-back chain it still publishes)
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 4) s_cnt4 = 0;
  const int64_t p0 = t0 + (int64_t)tid * kExtPer;
  uint32_t k[kExtPer + 1];
  k[0] = (p0 > 0 && p0 <= n) ? sk[p0 - 1] : 0xffffffffu;
  if (p0 + kExtPer <= n) {
#pragma unroll
    for (int v = 0; v < kExtPer / 4; ++v) {
      const uint4 a = *reinterpret_cast<const uint4*>(sk + p0 + 4 * v);
      k[4 * v + 1] = a.x; k[4 * v + 2] = a.y; k[4 * v + 3] = a.z; k[4 * v + 4] = a.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < kExtPer; ++u) k[u + 1] = p0 + u < n ? sk[p0 + u] : 0xffffffffu;
  }
  // (src, coef) of the tile's positions, wanted by the records of its run heads: loaded HERE, with the keys, as
  // vector loads -- in the record loop below each pair was a load behind the previous record's stores (the
  // compiler cannot move it across them: the record arrays are plain pointers), eight dependent round trips
  int sv[kExtPer];
  float cv[kExtPer];
  if (rl.R != nullptr && p0 + kExtPer <= n) {
#pragma unroll
    for (int v = 0; v < kExtPer / 4; ++v) {
      const int4 a = *reinterpret_cast<const int4*>(ssrc + p0 + 4 * v);
      const float4 c4 = *reinterpret_cast<const float4*>(scoef + p0 + 4 * v);
      sv[4 * v] = a.x; sv[4 * v + 1] = a.y; sv[4 * v + 2] = a.z; sv[4 * v + 3] = a.w;
      cv[4 * v] = c4.x; cv[4 * v + 1] = c4.y; cv[4 * v + 2] = c4.z; cv[4 * v + 3] = c4.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < kExtPer; ++u) {
      const bool in = rl.R != nullptr && p0 + u < n;
      sv[u] = in ? ssrc[p0 + u] : 0;
      cv[u] = in ? scoef[p0 + u] : 0.f;
    }
  }
  uint32_t hm = 0, mk = 0;                               // heads; heads or dead positions
  int llen[kExtPer];
  int lsum = 0, nh0 = 0;                                 // bag lengths / number of the thread's ENTITY heads
#pragma unroll
  for (int u = 0; u < kExtPer; ++u) {
    const bool live = p0 + u < n && k[u + 1] < sentinel;
    const bool head = live && (p0 + u == 0 || k[u] != k[u + 1]);
    hm |= head ? (1u << u) : 0u;
    mk |= (head || !live) ? (1u << u) : 0u;
    llen[u] = 0;
    if (hl.lens && head && (k[u + 1] >> hl.ent_kb) == hl.ent_tag) {
      llen[u] = min(hl.lens[k[u + 1] & ((1u << hl.ent_kb) - 1u)], hl.max_len);
      ++nh0;
    }
  }
#pragma unroll
  for (int u = 0; u < kExtPer; ++u) lsum += llen[u];
  reinterpret_cast<unsigned char*>(bm)[tid] = (unsigned char)mk;
  if (tid < 64) {                                        // the 64 positions behind the tile
    const int64_t q = t0 + kExtTile + tid;
    // (n == 0 -- every lookup dropped -- : no probe at all, everything behind the tile is marked)
    const int64_t qq = n > 0 ? min(q, n - 1) : 0;
    const uint32_t kq = n > 0 ? sk[qq] : sentinel, kp = n > 1 ? sk[qq - (qq > 0 ? 1 : 0)] : kq;
    const bool mark = q >= n || kq >= sentinel || (qq > 0 && kq != kp);
    const unsigned long long m2 = __ballot(mark);
    if (tid == 0) {
      bm[kExtTile / 32] = (uint32_t)m2;
      bm[kExtTile / 32 + 1] = (uint32_t)(m2 >> 32);
    }
  }
  const int nh = __popc(hm);
  const int ih = wave_incl_scan(nh, lane), il = wave_incl_scan(lsum, lane);
  if (lane == 63) {
    wtot[0][wv] = ih;
    wtot[1][wv] = il;
  }
  __syncthreads();
  // run ends (records wanted: rl.R; a sweep for the bag offsets alone skips them)
  const bool rec = rl.R != nullptr;
  int cnt[kExtPer];
  int nlong = 0, nitem = 0, npart = 0;
#pragma unroll
  for (int u = 0; u < kExtPer; ++u) {
    cnt[u] = 0;
    if (rec && ((hm >> u) & 1u)) {
      const int i = tid * kExtPer + u;                   // bit of the head; next marked bit behind it
      const int i1 = i + 1;
      int w = i1 >> 5;
      uint32_t word = bm[w] & (0xffffffffu << (i1 & 31));
      int c = 0;
      // (the whole bitmap, not its next 128 positions: a run that ends inside the tile costs LDS reads, the search
      // below is two to four dependent rounds of global probes per long run)
      for (int step = 0; step < kExtTile / 32 + 2 && c == 0; ++step) {
        if (word) c = (w << 5) + __builtin_ctz(word) - i;
        else if (++w > kExtTile / 32 + 1) break;
        else word = bm[w];
      }
      if (c == 0) {                                      // it leaves the bitmap: 9-ary search in the sorted keys
        const uint32_t key = k[u + 1];
        int64_t lo = p0 + u + 1, hi = n;                   // sk[lo - 1] == key; answer in [lo, hi]
        while (lo < hi) {
          const int64_t span = hi - lo;
          constexpr int NP = 8;                            // probes in flight (rare path since the whole-tile bitmap
          bool same[NP];                                   // search: 16 of them were 40 of the kernel's 125 VGPRs)
          int64_t pos[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            // the whole rest when it is that short, else 1/(NP+1)-th steps
            pos[q] = span <= NP ? lo + q : lo + (span * (q + 1)) / (NP + 1);
          }
          uint32_t pv[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) pv[q] = sk[min(pos[q], hi - 1)];   // (unconditional: all loads fly together)
#pragma unroll
          for (int q = 0; q < NP; ++q) same[q] = pos[q] < hi && pv[q] == key;
          int64_t nlo = lo, nhi = hi;
          bool closed = false;
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            if (closed || pos[q] >= hi) continue;
            if (same[q]) nlo = pos[q] + 1;
            else { nhi = pos[q]; closed = true; }
          }
          lo = nlo;
          hi = (span <= NP && !closed) ? nlo : nhi;      // (all of a short rest equal: the run ends at hi)
        }
        c = (int)(lo - (p0 + u));
      }
      cnt[u] = c;
      if (c > kshort) {
        const int nch = (c + kItem - 1) / kItem;
        ++nlong;
        nitem += nch;
        if (nch > 1) npart += nch;
      }
    }
  }
  int li = 0, it = 0, pb = 0;
  if (nh0) atomicAdd(&s_cnt4, nh0);
  if (nlong) {
    li = atomicAdd(&s_cnt[1], nlong);
    it = atomicAdd(&s_cnt[2], nitem);
    if (npart) pb = atomicAdd(&s_cnt[3], npart);
  }
  __syncthreads();
  if (tid == 0) {
    int th = 0, tl = 0;
    for (int q = 0; q < kExtThreads / 64; ++q) {
      const int a = wtot[0][q], c = wtot[1][q];
      wtot[0][q] = th;
      wtot[1][q] = tl;
      th += a;
      tl += c;
    }
    s_cnt[0] = th;
    s_base[4] = hl.lens ? tl : 0;                        // (becomes the prefix below)
  }
  __syncthreads();
  if (tid < 4) {                                         // the four list counters: one atomic each, in flight together
    const int v = s_cnt[tid];
    s_base[tid] = (rec && v > 0) ? atomicAdd(&rl.ctr[tid == 0 ? kNRuns : tid == 1 ? kNLong : tid == 2 ? kNItems : kNPart], v) : 0;
  } else if (tid == 4 && rec && s_cnt4 > 0) {            // runs of the ENTITY table (they sort first: records [0, that))
    atomicAdd(&rl.ctr[kNRuns0], s_cnt4);
  }
  __syncthreads();
  if (hl.lens && wv == 0) {
    // ordered prefix of the bag lengths: every workgroup publishes its total BEFORE it waits for the ones in front,
    // and it only waits for lower block indices.  Progress therefore needs in-order dispatch (block b is never
    // started before every block < b has been) -- which the hardware dispatcher gives -- not full co-residency: the
    // lowest unfinished block never waits for anything that has not started.  The caller bounds the grid
    // (runs_extract_blocks(n) <= 224, optim.hip) and otherwise takes the one-workgroup k_head_len_scan.
    // (the word also carries the workgroup's number of runs: with it the run RECORDS are written in sorted order
    // -- the entity table's runs first -- instead of in the order the counter atomics were served)
    if (lane == 0)
      __hip_atomic_store(&hl.lookback[blockIdx.x],
                         (1ull << 63) | ((unsigned long long)s_cnt[0] << 31) | (unsigned long long)s_base[4],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int part = 0, rpart = 0;
    for (int q = lane; q < (int)blockIdx.x; q += 64) {
      unsigned long long v;
      while (!((v = __hip_atomic_load(&hl.lookback[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63))
        __builtin_amdgcn_s_sleep(2);
      part += (int)(v & 0x7fffffffull);
      rpart += (int)((v >> 31) & 0x7fffffffull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_xor(part, o, 64);
      rpart += __shfl_xor(rpart, o, 64);
    }
    if (lane == 0) {
      if (blockIdx.x == gridDim.x - 1) *hl.total = part + s_base[4];
      s_base[4] = part;
      s_base[0] = rpart;
    }
  }
  __syncthreads();
  if (dead_tile) return;
  int u_run = s_base[0] + wtot[0][wv] + ih - nh;
  int loff = s_base[4] + wtot[1][wv] + il - lsum;
  li += s_base[1];
  it += s_base[2];
  pb += s_base[3];
#pragma unroll
  for (int u = 0; u < kExtPer; ++u) {
    const int64_t p = p0 + u;
    if (hl.lens && p < n && p < hl.hoff_n) hl.hoff[p] = loff;
    loff += llen[u];
    if (!rec || !((hm >> u) & 1u)) continue;
    const int c = cnt[u];
    const int ur = u_run++;
    rl.R[ur] = make_int4((int)k[u + 1], (int)p, c, (int)p);
    rl.R2[ur] = make_int2(sv[u], __float_as_int(cv[u]));
    if (c > kshort) {
      const int nch = (c + kItem - 1) / kItem;
      const int pbu = nch > 1 ? pb : 0;
      if (li < rl.cap_long && it + nch <= rl.cap_items && pbu + nch <= rl.cap_part) {   // (bounds hold by construction)
        rl.LR[li] = make_int4(ur, pbu, nch, 0);
        for (int x = 0; x < nch; ++x) rl.items[it + x] = make_int2(li, x);
      }
      ++li;
      it += nch;
      if (nch > 1) pb += nch;
    }
  }
}

// ---- apply ------------------------------------------------------------------------------------
// Partial rows of a long run travel as agent-scope atomics (they bypass the per-XCD L2), so the
// ticket needs no cache-wide release / acquire fence: a __threadfence() per work item is an L2
// write-back scan each (measured: the long-run items alone 63 us with fences).
__device__ __forceinline__ void agent_store4(float* p, float4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 agent_load4(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
template <bool SGD>
__device__ __forceinline__ void finish_row(const TabRow& T, const MergeOut& mo, bool side, int d, int col,
                                           bool colok, int lig, float4 a, float gb, int first, float4 w4, float4 a4,
                                           float bv, float bav, float lr, float gs) {
  if (side) merge_row(mo, d, T.row, first, col, colok, lig, a, gb);
  if (!T.E) return;                                   // virtual table: the entity ids of a riding bag table
  if (colok) {
    const float4 gg = make_float4(a.x * gs, a.y * gs, a.z * gs, a.w * gs);
    if (!SGD) {
      a4.x += gg.x * gg.x; a4.y += gg.y * gg.y; a4.z += gg.z * gg.z; a4.w += gg.w * gg.w;
      w4.x -= adagrad_delta(lr, gg.x, a4.x);
      w4.y -= adagrad_delta(lr, gg.y, a4.y);
      w4.z -= adagrad_delta(lr, gg.z, a4.z);
      w4.w -= adagrad_delta(lr, gg.w, a4.w);
      row_store(T.acc + (int64_t)T.row * d + col, a4);
    } else {
      w4.x -= lr * gg.x; w4.y -= lr * gg.y; w4.z -= lr * gg.z; w4.w -= lr * gg.w;
    }
    row_store(T.E + (int64_t)T.row * d + col, w4);
  }
  if (T.bias && lig == 0) {
    const float gg = gb * gs;
    if (!SGD) {
      const float ba = bav + gg * gg;
      T.bias_acc[T.row] = ba;
      T.bias[T.row] = bv - adagrad_delta(lr, gg, ba);
    } else {
      T.bias[T.row] = bv - lr * gg;
    }
  }
}

#ifndef ARX_RUN_NB
#define ARX_RUN_NB 1
#endif
#ifndef ARX_RUN_RU
#define ARX_RUN_RU 4
#endif

// One run-centric apply inside a launch: a table set, the run lists of its sorted pass, the gradient rows.
// rec_lo / rec_hi: indices into sg.ctr of the record range [lo, hi) taken (-1: 0 / all) -- with the records in
// sorted order (k_runs_extract on the look-back path) the entity table's runs are records [0, ctr[kNRuns0]);
// tsel: which LONG runs (they live in their own unordered list): -1 all, t >= 0 those of table t, t <= -2 those of
// every table but -2 - t.
struct ApplyJob {
  TableSet ts;
  RunLists sg;
  const float* G;
  int64_t ldg;
  const float* Gb;
  MergeOut mo;
  int gshort, glong;
  int rec_lo, rec_hi, tsel;
};

__device__ __forceinline__ bool job_takes(const ApplyJob& jb, uint32_t key) {
  if (jb.tsel == -1) return true;
  const int t = (int)(key >> jb.ts.kb);
  return jb.tsel >= 0 ? t == jb.tsel : t != -2 - jb.tsel;
}

// blocks [0, glong) of the job: one workgroup per work item of a long run;
// blocks [glong, glong + gshort): one sub-group per short run, NB runs in flight per sub-group.
template <int LPR, bool MT, bool SGD>
__device__ __forceinline__ void run_apply_job(const ApplyJob& jb, int d, const int bid, const float lr, const float gs,
                                              int kshort) {
  const TableSet& ts = jb.ts;
  const RunLists& sg = jb.sg;
  const float* __restrict__ G = jb.G;
  const int64_t ldg = jb.ldg;
  const float* __restrict__ Gb = jb.Gb;
  const MergeOut& mo = jb.mo;
  const int gshort = jb.gshort;
  constexpr int NSG = 64 / LPR;        // sub-groups per wave
  constexpr int NSGB = 256 / LPR;      // sub-groups per workgroup
  constexpr int NB = ARX_RUN_NB;
  constexpr int RU = ARX_RUN_RU;
#ifndef ARX_RUN_LRU
#define ARX_RUN_LRU 4
#endif
  constexpr int LRU = ARX_RUN_LRU;     // rows in flight per sub-group of a long run's work item
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lig = lane % LPR;
  const int g = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // the long runs' work items take the FIRST blocks of the grid: they are the long poles
  const int glong = jb.glong;
  if (bid >= glong) {
    constexpr int KS = LPR < 32 ? LPR : 32;          // entries of a short run
    // entries 1.. of the sub-group's runs, in summation order (private to the sub-group: wave-synchronous)
    __shared__ int s_src[4 * NSG][NB][KS];
    __shared__ float s_coef[4 * NSG][NB][KS];
    __shared__ float s_gb[4 * NSG][NB][KS];
    const int sgl = wv * NSG + g;
    const int nruns = sg.ctr[jb.rec_hi >= 0 ? jb.rec_hi : (int)kNRuns];
    const int rlo = jb.rec_lo >= 0 ? sg.ctr[jb.rec_lo] : 0;
    const int64_t nsg_tot = (int64_t)gshort * 4 * NSG;
    const int64_t sgid = (((int64_t)bid - glong) * 4 + wv) * NSG + g;
    for (int64_t base = rlo; base < nruns; base += nsg_tot * NB) {
      // round trip 1: the run records -- key, segment, length AND the first entry
      int4 r[NB];
      int2 r2[NB];
      bool act[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int64_t u = base + j * nsg_tot + sgid;
        act[j] = u < nruns;
        r[j] = act[j] ? sg.R[u] : make_int4(0, 0, 0, 0);
        r2[j] = act[j] ? sg.R2[u] : make_int2(0, 0);
        act[j] = act[j] && r[j].z > 0 && r[j].z <= kshort && job_takes(jb, (uint32_t)r[j].x);
        if (!act[j]) r[j].z = 0;                            // (long runs: the work items below)
      }
      // round trip 2: table row, slot row, bias cells, the first gradient row, the other entries
      float bv[NB], bav[NB], gb0[NB];
      float4 wrow[NB], arow[NB], g0[NB];
      int maxcnt = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bv[j] = bav[j] = gb0[j] = 0.f;
        wrow[j] = arow[j] = g0[j] = z4;
        if (act[j]) {
          if (colok) g0[j] = *reinterpret_cast<const float4*>(G + (int64_t)r2[j].x * ldg + col);
          if (Gb) gb0[j] = Gb[r2[j].x];
          const TabRow T = tab_of<MT>(ts, (uint32_t)r[j].x);
          if (T.E) {
            if (colok) {
              wrow[j] = row_load(T.E + (int64_t)T.row * d + col);
              if (!SGD) arow[j] = row_load(T.acc + (int64_t)T.row * d + col);
            }
            if (T.bias && lig == 0) {
              bv[j] = T.bias[T.row];
              if (!SGD) bav[j] = T.bias_acc[T.row];
            }
          }
          if (lig >= 1 && lig < r[j].z) {
            const int es = sg.ssrc[r[j].y + lig];
            s_src[sgl][j][lig] = es;
            s_coef[sgl][j][lig] = sg.scoef[r[j].y + lig];
            if (Gb) s_gb[sgl][j][lig] = Gb[es];
          }
        }
        maxcnt = max(maxcnt, r[j].z);
      }
      float4 a[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] = act[j] ? f4_fma(__int_as_float(r2[j].y), g0[j], z4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (maxcnt > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the NB runs advance together, RU gradient rows of each in flight
        for (int k0 = 1; k0 < maxcnt; k0 += RU) {
          float4 v[NB][RU];
          float c[NB][RU];
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int u = 0; u < RU; ++u) {
              const bool ok = k0 + u < r[j].z;
              const int s = s_src[sgl][j][ok ? k0 + u : 1];
              c[j][u] = s_coef[sgl][j][ok ? k0 + u : 1];
              v[j][u] = (ok && colok) ? *reinterpret_cast<const float4*>(G + (int64_t)s * ldg + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int u = 0; u < RU; ++u)
              if (k0 + u < r[j].z) a[j] = f4_fma(c[j][u], v[j][u], a[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (!act[j]) continue;
        float gb = 0.f;
        if (Gb) {
          gb = fmaf(__int_as_float(r2[j].y), gb0[j], 0.f);
          for (int k = 1; k < r[j].z; ++k) gb = fmaf(s_coef[sgl][j][k], s_gb[sgl][j][k], gb);
        }
        const TabRow T = tab_of<MT>(ts, (uint32_t)r[j].x);
        finish_row<SGD>(T, mo, merge_side(mo, (uint32_t)r[j].x), d, col, colok, lig, a[j], gb, r[j].w, wrow[j],
                        arow[j], bv[j], bav[j], lr, gs);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // ---- long runs: one workgroup per 256-entry item ----
  __shared__ __attribute__((aligned(16))) float sh[NSGB][LPR * 4];
  __shared__ float shb[NSGB];
  __shared__ int s_last;
  const int sgb = threadIdx.x / LPR;
  const int nitems = min(sg.ctr[kNItems], (int)sg.cap_items);
  for (int it = bid; it < nitems; it += glong) {
    const int2 item = sg.items[it];
    const int4 lr4 = sg.LR[item.x];
    const int4 r = sg.R[lr4.x];
    if (!job_takes(jb, (uint32_t)r.x)) continue;        // (block-uniform)
    const int nch = lr4.z;
    const int64_t off = r.y;
    const int cnt = r.z;
    constexpr int EPS = kItem / NSGB < LPR ? kItem / NSGB : LPR;      // entries per sub-group
    static_assert(EPS * NSGB == kItem || LPR * NSGB < kItem, "a work item is dealt whole to the sub-groups");
    const int e0 = item.y * kItem + sgb * EPS;
    const int m = max(0, min(EPS, cnt - e0));
    int es = 0;
    float ec = 0.f, gbv = 0.f;
    if (lig < m) {
      es = sg.ssrc[off + e0 + lig];
      ec = sg.scoef[off + e0 + lig];
      if (Gb) gbv = Gb[es];
    }
    float4 a = z4;
    for (int t = 0; t < m; t += LRU) {
      float4 v[LRU];
      float c[LRU];
#pragma unroll
      for (int u = 0; u < LRU; ++u) {
        const bool ok = t + u < m;
        const int s = __shfl(es, ok ? t + u : 0, LPR);
        c[u] = ok ? __shfl(ec, ok ? t + u : 0, LPR) : 0.f;
        v[u] = (ok && colok) ? *reinterpret_cast<const float4*>(G + (int64_t)s * ldg + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < LRU; ++u)
        if (t + u < m) a = f4_fma(c[u], v[u], a);
    }
    float gb = 0.f;
    if (Gb)
      for (int t = 0; t < m; ++t) gb = fmaf(__shfl(ec, t, LPR), __shfl(gbv, t, LPR), gb);
    __syncthreads();                                  // (sh / shb free again)
    if (colok) *reinterpret_cast<float4*>(&sh[sgb][col]) = a;
    if (lig == 0) shb[sgb] = gb;
    __syncthreads();
    float4 tot = z4;
    float tb = 0.f;
    if (sgb == 0) {
      for (int q = 0; q < NSGB; ++q) {
        if (colok) tot = f4_add2(tot, *reinterpret_cast<const float4*>(&sh[q][col]));
        tb += shb[q];
      }
    }
    bool fin = nch == 1;
    if (nch > 1) {
      // several items: partial row out, the last item to arrive adds them up in item order
      const int64_t prow = (int64_t)lr4.y + item.y;
      if (sgb == 0) {
        if (colok) agent_store4(sg.part + prow * d + col, tot);
        if (lig == 0) __hip_atomic_store(sg.part_b + prow, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);                 // the partial is out before the ticket is taken
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
      }
      __syncthreads();
      if (threadIdx.x == 0)
        s_last = (__hip_atomic_fetch_add(&sg.LR[item.x].w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nch - 1) ? 1 : 0;
      __syncthreads();
      if (s_last) {
        const int per = (nch + NSGB - 1) / NSGB;
        const int q0 = sgb * per, q1 = min(nch, q0 + per);
        float4 p = z4;
        float pb = 0.f;
        for (int q = q0; q < q1; q += LRU) {
          float4 v[LRU];
          float vb[LRU];
#pragma unroll
          for (int u = 0; u < LRU; ++u) {
            const bool ok = q + u < q1;
            const int64_t pr = (int64_t)lr4.y + (ok ? q + u : q0);
            v[u] = colok ? agent_load4(sg.part + pr * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            vb[u] = __hip_atomic_load(sg.part_b + pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < LRU; ++u)
            if (q + u < q1) {
              p = f4_add2(p, v[u]);
              pb += vb[u];
            }
        }
        __syncthreads();
        if (colok) *reinterpret_cast<float4*>(&sh[sgb][col]) = p;
        if (lig == 0) shb[sgb] = pb;
        __syncthreads();
        if (sgb == 0) {
          tot = z4;
          tb = 0.f;
          for (int q = 0; q < NSGB; ++q) {
            if (colok) tot = f4_add2(tot, *reinterpret_cast<const float4*>(&sh[q][col]));
            tb += shb[q];
          }
        }
        fin = true;
        if (threadIdx.x == 0)                           // (the apply can be replayed on the same lists)
          __hip_atomic_store(&sg.LR[item.x].w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (fin && sgb == 0) {
      const TabRow T = tab_of<MT>(ts, (uint32_t)r.x);
      float4 w4 = z4, a4 = z4;
      float bv = 0.f, bav = 0.f;
      if (T.E) {
        if (colok) {
          w4 = row_load(T.E + (int64_t)T.row * d + col);
          if (!SGD) a4 = row_load(T.acc + (int64_t)T.row * d + col);
        }
        if (T.bias && lig == 0) {
          bv = T.bias[T.row];
          if (!SGD) bav = T.bias_acc[T.row];
        }
      }
      finish_row<SGD>(T, mo, merge_side(mo, (uint32_t)r.x), d, col, colok, lig, tot, tb, r.w, w4, a4, bv, bav, lr, gs);
    }
  }
}

template <int LPR, bool MT, bool SGD>
__global__ __launch_bounds__(256) void k_run_apply(ApplyJob jb, int d, const float* __restrict__ lr_dev,
                                                   const float* __restrict__ gscale_dev, int kshort) {
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  run_apply_job<LPR, MT, SGD>(jb, d, (int)blockIdx.x, lr, gs, kshort);
}

// Two jobs in one launch (the tail of a step with a riding bag table: the token runs over the merged rows + the
// one-hot runs of the tables that do not feed them).  Job B's blocks first: its long runs are the long poles.
template <int LPR, bool MTA, bool SGD>
__global__ __launch_bounds__(256) void k_run_apply2(ApplyJob ja, ApplyJob jbb, int d,
                                                    const float* __restrict__ lr_dev,
                                                    const float* __restrict__ gscale_dev, int kshort) {
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  const int nb = jbb.glong + jbb.gshort;
  if ((int)blockIdx.x < nb) run_apply_job<LPR, false, SGD>(jbb, d, (int)blockIdx.x, lr, gs, kshort);
  else run_apply_job<LPR, MTA, SGD>(ja, d, (int)blockIdx.x - nb, lr, gs, kshort);
}

#define ARX_GRP_LPR(lpr, CALL)                        \
  switch (lpr) {                                      \
    case 8: { constexpr int LPR = 8; CALL; } break;   \
    case 16: { constexpr int LPR = 16; CALL; } break; \
    case 32: { constexpr int LPR = 32; CALL; } break; \
    default: { constexpr int LPR = 64; CALL; } break; \
  }

static ApplyJob make_job(const TableSet& ts, int d, const RunLists& sg, int64_t n, const float* G, int64_t ldg,
                         const float* Gb, const MergeOut& mo, const ApplySel& sel) {
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  const int nsg = 64 / lpr;
  int64_t gshort = ceil_div(n, (int64_t)4 * nsg * ARX_RUN_NB);
#ifndef ARX_RUN_CAP_PER_CU
#define ARX_RUN_CAP_PER_CU 16       // (round 5, same box: 8 -> 12 / 24: C3 229.0 -> 227.1..227.9, C2 160 -> 157..159 us)
#endif
  const int64_t cap = (int64_t)cu_count() * ARX_RUN_CAP_PER_CU;
  if (gshort > cap) gshort = cap;
  if (gshort < 1) gshort = 1;
  int64_t glong = n / (kshort + 1) + 1;       // upper bound of the work items is larger; the loop strides
  const int64_t capl = (int64_t)cu_count() * 4;
  if (glong > capl) glong = capl;
  ApplyJob jb;
  jb.ts = ts;
  jb.sg = sg;
  jb.G = G;
  jb.ldg = ldg;
  jb.Gb = Gb;
  jb.mo = mo;
  jb.gshort = (int)gshort;
  jb.glong = (int)glong;
  jb.rec_lo = sel.rec_lo;
  jb.rec_hi = sel.rec_hi;
  jb.tsel = sel.tsel;
  return jb;
}

template <bool MT>
int launch_run_apply_t(const TableSet& ts, int d, const RunLists& sg, int64_t n, const float* G, int64_t ldg,
                       const float* Gb, const float* lr_dev, const float* gscale_dev, const MergeOut& mo, bool sgd,
                       hipStream_t s, const ApplySel& sel) {
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  const ApplyJob jb = make_job(ts, d, sg, n, G, ldg, Gb, mo, sel);
  const int grid = jb.gshort + jb.glong;
  if (sgd) {
    ARX_GRP_LPR(lpr, (k_run_apply<LPR, MT, true><<<grid, 256, 0, s>>>(jb, d, lr_dev, gscale_dev, kshort)));
  } else {
    ARX_GRP_LPR(lpr, (k_run_apply<LPR, MT, false><<<grid, 256, 0, s>>>(jb, d, lr_dev, gscale_dev, kshort)));
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

template <bool MTA>
int launch_run_apply_pair_t(const ApplyJob& ja, const ApplyJob& jb, int d, const float* lr_dev,
                            const float* gscale_dev, bool sgd, hipStream_t s) {
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  const int grid = ja.gshort + ja.glong + jb.gshort + jb.glong;
  if (sgd) {
    ARX_GRP_LPR(lpr, (k_run_apply2<LPR, MTA, true><<<grid, 256, 0, s>>>(ja, jb, d, lr_dev, gscale_dev, kshort)));
  } else {
    ARX_GRP_LPR(lpr, (k_run_apply2<LPR, MTA, false><<<grid, 256, 0, s>>>(ja, jb, d, lr_dev, gscale_dev, kshort)));
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

size_t al(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

bool runs_path(int d) {
  return lanes_per_row(d) >= 8;
}

// long runs have more than 8 entries (LPR >= 8)
static void run_caps(int64_t n, int64_t* cl, int64_t* ci, int64_t* cp) {
  *cl = n / 9 + 2;
  *ci = *cl + n / kItem + 2;
  *cp = n / (kItem / 2) + 4;
}

size_t run_lists_bytes(int64_t n, int d) {
  int64_t cl, ci, cp;
  run_caps(n, &cl, &ci, &cp);
  return al((size_t)n * 16) + al((size_t)n * 8) + al((size_t)cl * 16) + al((size_t)ci * 8) +
         al((size_t)cp * (size_t)d * 4) + al((size_t)cp * 4);
}

RunLists run_lists_of(char* base, int64_t n, int d, const int32_t* ssrc, const float* scoef, int32_t* ctr) {
  RunLists r;
  run_caps(n, &r.cap_long, &r.cap_items, &r.cap_part);
  size_t o = 0;
  r.R = reinterpret_cast<int4*>(base + o); o += al((size_t)n * 16);
  r.R2 = reinterpret_cast<int2*>(base + o); o += al((size_t)n * 8);
  r.LR = reinterpret_cast<int4*>(base + o); o += al((size_t)r.cap_long * 16);
  r.items = reinterpret_cast<int2*>(base + o); o += al((size_t)r.cap_items * 8);
  r.part = reinterpret_cast<float*>(base + o); o += al((size_t)r.cap_part * (size_t)d * 4);
  r.part_b = reinterpret_cast<float*>(base + o);
  r.ssrc = ssrc;
  r.scoef = scoef;
  r.ctr = ctr;
  return r;
}

int launch_runs_extract(const uint32_t* sk, int64_t n, const int32_t* n_dev, uint32_t sentinel, const RunLists& rl,
                        int d, hipStream_t s, const int32_t* ent_lens, int ent_kb, uint32_t ent_tag, int max_len,
                        int32_t* hoff, int64_t hoff_n, int32_t* total, void* lookback) {
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  HeadLens hl = {ent_lens, ent_kb, ent_tag, max_len, hoff, hoff_n, total,
                 reinterpret_cast<unsigned long long*>(lookback)};
  k_runs_extract<<<(int)ceil_div(n, kExtTile), kExtThreads, 0, s>>>(sk, rl.ssrc, rl.scoef, n, n_dev, sentinel, rl,
                                                                   kshort, hl);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int runs_extract_blocks(int64_t n) { return (int)ceil_div(n, kExtTile); }

int launch_run_apply(const TableSet& ts, bool multi, int d, const RunLists& rl, int64_t n, const float* G,
                     int64_t ldg, const float* Gb, const float* lr_dev, const float* gscale_dev, const MergeOut& mo,
                     bool sgd, hipStream_t s, const ApplySel& sel) {
  if (multi) return launch_run_apply_t<true>(ts, d, rl, n, G, ldg, Gb, lr_dev, gscale_dev, mo, sgd, s, sel);
  return launch_run_apply_t<false>(ts, d, rl, n, G, ldg, Gb, lr_dev, gscale_dev, mo, sgd, s, sel);
}

int launch_run_apply_pair(const TableSet& tsa, bool multi_a, const RunLists& rla, int64_t na, const float* Ga,
                          int64_t ldga, const float* Gba, const MergeOut& moa, const ApplySel& sela,
                          const TableSet& tsb, const RunLists& rlb, int64_t nb, const float* Gbm, int64_t ldgb,
                          const float* Gbb, int d, const float* lr_dev, const float* gscale_dev, bool sgd,
                          hipStream_t s) {
  const ApplyJob ja = make_job(tsa, d, rla, na, Ga, ldga, Gba, moa, sela);
  const MergeOut none = {nullptr, nullptr, nullptr, -1, 0};
  const ApplyJob jb = make_job(tsb, d, rlb, nb, Gbm, ldgb, Gbb, none, ApplySel{-1, -1, -1});
  if (multi_a) return launch_run_apply_pair_t<true>(ja, jb, d, lr_dev, gscale_dev, sgd, s);
  return launch_run_apply_pair_t<false>(ja, jb, d, lr_dev, gscale_dev, sgd, s);
}

}  // namespace arx

