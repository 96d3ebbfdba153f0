// group.hip -- K7 without a sort: direct-address grouping of the lookup contributions and a
// run-centric sparse-Adagrad apply.
//
// Reference semantics (hmf_model.py:146-151, seqModel.py:173-182, embed_attribute.py:397-400;
// TF-1.0 AdagradOptimizer on IndexedSlices): duplicate rows are summed, every touched row gets
// ONE update  acc += g^2 ; w -= lr * g / sqrt(acc).
//
// The radix path (radix_sort.hip + the window kernels of optim.hip) needs ~14 dependent launches
// for a C3 step and two launches more to finish the runs that cross its 64-position windows.
// Here every table of the pass owns a zeroed int32 map with one cell per row (the caller's
// aux_cnt arrays), and a pass is
//   A  count    one thread per contribution: c = atomicAdd(map[row], 1) -- the arrival rank; the
//               contribution that saw 0 appends the key to the run list (wave-aggregated append)
//   B  offsets  one thread per run: segment offset from a cursor (wave scan + one atomic per wave),
//               map[row] := offset; runs longer than one sub-group are listed as LONG and cut into
//               256-entry work items
//   C  place    one thread per contribution: (src, coef) -> segment[offset + arrival rank]
//   D  long     one workgroup per LONG run: its entries ordered by src (LDS bitmap + popcount
//               prefix: a counting sort over the gradient-row range); maps zeroed again
//   apply       one sub-group per SHORT run (<= LPR entries, ordered by src in registers), one
//               workgroup per work item of a LONG run (the last item to arrive -- ticket -- adds
//               the items' partial rows up in item order).  No partial-row windows, no finish launch.
// Integer atomics only decide WHERE an entry is stored; every run is summed in ascending src order
// and long runs in a partition that depends on their length alone, so the result is bit-reproducible.
// (key, src) pairs are unique within a pass: one-hot sites contribute each gradient row once, and the
// bag stage merges the duplicates of a token inside one bag (multiplicity as coefficient).
//
// A multi-hot table riding on one-hot table 0 (BagStage, arx_sparse_adagrad_cat_multi_bags): the
// distinct entities are the runs of table 0; their bags are counted / placed by the same kernels
// (launches B..E also carry the token stage's A..D), the merged, 1/len-scaled gradient row of an
// entity is a side output of the one-hot apply at Gu[first contribution of the entity among table
// 0's] -- an index that does not depend on arrival order -- and the token apply sums those rows.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "common.h"
#include "k7.h"

namespace arx {
namespace {

constexpr int kItem = 256;            // entries per work item of a long run (one 256-thread workgroup)
constexpr int kSlabWords = 8192;      // bitmap words per slab of the long-run ordering (262 144 gradient rows)
constexpr int kLongThreads = 1024;

// counters of one stage (zeroed by a memset node in front of launch A)
enum { kNRuns = 0, kCursor = 1, kNLong = 2, kNItems = 3, kNPart = 4, kNRuns0 = 5, kCtrInts = 16 };

struct Stage {
  int4* R;            // run records {key, offset, count, first}
  int2* R2;           // short runs: their first entry in summation order {src, coef bits}
  int4* LR;           // long runs {run, first partial row, items, ticket}
  int2* items;        // work items {long run, item}
  int32_t* asrc;      // segments in arrival order
  float* acoef;
  int32_t* ssrc;      // long runs: entries ordered by src
  float* scoef;
  int32_t* ctr;
  float* part;        // partial rows of the long runs with several items
  float* part_b;
  int64_t cap_long, cap_items, cap_part;
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// append to a list: one atomic per wave
__device__ __forceinline__ int wave_append(bool take, int32_t* counter, int lane) {
  const unsigned long long m = __ballot(take);
  int base = 0;
  if (m) {
    if (lane == 0) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, 0, 64);
  }
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// Block-private aggregation of equal keys in front of a global atomic.  Same-address atomics of a
// Zipf-hot row serialise at L2 (~10 ns each, measured: 12 k contributions of the hottest token =
// ~100 us for the whole launch); with the keys of a 1024-thread workgroup merged in an LDS hash
// table first, a hot row costs one global atomic per workgroup.
constexpr int kHashBits = 11;
constexpr int kHash = 1 << kHashBits;
constexpr int kAggThreads = 1024;

__device__ __forceinline__ int hash_slot(int* __restrict__ hkey, int key) {
  uint32_t h = ((uint32_t)key * 2654435761u) >> (32 - kHashBits);
  for (int p = 0; p < 24; ++p) {
    const int old = atomicCAS(&hkey[h], -1, key);
    if (old == -1 || old == key) return (int)h;
    h = (h + 1) & (kHash - 1);
  }
  return -1;                                          // crowded neighbourhood: the caller goes to memory directly
}

struct SiteExtra {
  int32_t t0off[kMaxSites];   // contributions of table 0 in front of the site (index among table 0's)
};

// ---- A: one-hot sites -> keys, arrival ranks, run list ---------------------------------------
__device__ __forceinline__ void body_count_sites(int64_t i, const CatSites& st, const TableSet& ts,
                                                 int32_t* __restrict__ keys, int32_t* __restrict__ src,
                                                 float* __restrict__ coef, int32_t* __restrict__ arr,
                                                 int4* __restrict__ R, int32_t* __restrict__ ctr,
                                                 int32_t* __restrict__ U0, int* __restrict__ hkey,
                                                 int* __restrict__ hcnt, int* __restrict__ hbase) {
  const int lane = threadIdx.x & 63;
  const int64_t n = st.offs[st.nsites];
  for (int h = threadIdx.x; h < kHash; h += kAggThreads) {
    hkey[h] = -1;
    hcnt[h] = 0;
  }
  __syncthreads();
  bool valid = false;
  int key = ARX_KEY_NONE, tb = 0, row = 0, slot = -1, lr = 0;
  if (i < n) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < kMaxSites; ++q)
      if (q < st.nsites && i >= st.offs[q]) s = q;
    const int64_t j = i - st.offs[s];
    const int id = st.ids[s][j];
    row = st.cat_map[s] ? st.cat_map[s][id] : id;
    tb = st.table[s];
    const int64_t rows = tb == 0 ? st.rows[0] : tb == 1 ? st.rows[1] : tb == 2 ? st.rows[2] : st.rows[3];
    valid = row >= 0 && row < rows;
    key = valid ? ((tb << st.kb) | row) : ARX_KEY_NONE;
    keys[i] = key;
    src[i] = st.row_base[s] + (int32_t)j;
    coef[i] = st.coef[s];
    if (valid) {
      slot = hash_slot(hkey, key);
      if (slot >= 0) lr = atomicAdd(&hcnt[slot], 1);
    }
  }
  __syncthreads();
  // one global atomic per distinct key of the workgroup (all of a thread's slots in flight together); the
  // one that finds the cell at 0 lists the run -- one append per wave and list, both in flight together
  {
    constexpr int SL = kHash / kAggThreads;
    int k[SL], base[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
      const int h = threadIdx.x + q * kAggThreads;
      k[q] = hkey[h];
      base[q] = -1;
      if (k[q] != -1)
        base[q] = atomicAdd(pick4(ts.cnt, (uint32_t)k[q] >> st.kb) + (k[q] & ((1 << st.kb) - 1)), hcnt[h]);
    }
    int nl = 0, nl0 = 0;
#pragma unroll
    for (int q = 0; q < SL; ++q) {
      if (k[q] != -1) hbase[threadIdx.x + q * kAggThreads] = base[q];
      nl += base[q] == 0 ? 1 : 0;
      nl0 += (base[q] == 0 && ((uint32_t)k[q] >> st.kb) == 0u) ? 1 : 0;
    }
    const int incl = wave_incl_scan(nl, lane);
    const int incl0 = U0 ? wave_incl_scan(nl0, lane) : 0;
    int wb = 0, wb0 = 0;
    if (lane == 63) {
      if (incl > 0) wb = atomicAdd(&ctr[kNRuns], incl);
      if (incl0 > 0) wb0 = atomicAdd(&ctr[kNRuns0], incl0);
    }
    int u = __shfl(wb, 63, 64) + incl - nl;
    int kk = __shfl(wb0, 63, 64) + incl0 - nl0;
#pragma unroll
    for (int q = 0; q < SL; ++q)
      if (base[q] == 0) {
        R[u++] = make_int4(k[q], 0, 0, INT_MAX);
        if (U0 && ((uint32_t)k[q] >> st.kb) == 0u) U0[kk++] = k[q] & ((1 << st.kb) - 1);
      }
  }
  __syncthreads();
  int c = -1;
  if (valid) c = slot >= 0 ? hbase[slot] + lr : atomicAdd(pick4(ts.cnt, (uint32_t)tb) + row, 1);
  if (i < n) arr[i] = c;
  const bool leader = valid && slot < 0 && c == 0;    // (only contributions that missed the hash table)
  const int u = wave_append(leader, &ctr[kNRuns], lane);
  if (leader) R[u] = make_int4(key, 0, 0, INT_MAX);
  if (U0) {
    const bool l0 = leader && tb == 0;
    const int kk = wave_append(l0, &ctr[kNRuns0], lane);
    if (l0) U0[kk] = row;
  }
}

// ---- A (token stage): bags of the distinct entities -> token keys, arrival ranks, run list ----
// W lanes per entity (W = 2^wshift <= 64, at most kBagNJ tokens per lane); duplicates of a token
// inside one bag are merged (the first occurrence keeps the entry, coefficient = multiplicity).  A
// workgroup takes 1024 / W entities and merges their tokens in its hash table before it touches the map.
constexpr int kBagNJ = 2;                               // bags of up to 128 tokens
__device__ __forceinline__ void body_count_bags(int bid, int wshift, const int32_t* __restrict__ U0,
                                                const int32_t* __restrict__ ctr0, const int32_t* __restrict__ vals,
                                                const int32_t* __restrict__ starts, const int32_t* __restrict__ lens,
                                                int max_len, int64_t bag_rows, int32_t* __restrict__ tcnt,
                                                int32_t* __restrict__ tkey, int32_t* __restrict__ tarr,
                                                float* __restrict__ tmul, int4* __restrict__ R,
                                                int32_t* __restrict__ ctr, int* __restrict__ hkey,
                                                int* __restrict__ hcnt, int* __restrict__ hbase) {
  const int lane = threadIdx.x & 63;
  const int W = 1 << wshift;
  const int lig = threadIdx.x & (W - 1);
  const int n0 = ctr0[kNRuns0];
  const int64_t sub = (int64_t)bid * (kAggThreads >> wshift) + (threadIdx.x >> wshift);
  if ((int64_t)bid * (kAggThreads >> wshift) >= n0) return;          // (whole workgroup)
  for (int h = threadIdx.x; h < kHash; h += kAggThreads) {
    hkey[h] = -1;
    hcnt[h] = 0;
  }
  __syncthreads();
  const bool ent = sub < n0;
  const int e = ent ? U0[sub] : 0;
  const int len = ent ? lens[e] : 0;
  const int64_t st = ent ? (int64_t)starts[e] : 0;
  int tok[kBagNJ], mult[kBagNJ], slot[kBagNJ], lr[kBagNJ];
  bool keep[kBagNJ];
#pragma unroll
  for (int q = 0; q < kBagNJ; ++q) {
    const int j = lig + q * W;
    tok[q] = ARX_KEY_NONE;
    mult[q] = 0;
    slot[q] = -1;
    lr[q] = 0;
    keep[q] = false;
    if (j < len) {
      tok[q] = vals[st + j];
      keep[q] = tok[q] >= 0 && tok[q] < bag_rows;
      for (int jj = 0; jj < len; jj += 4) {             // (the bag is a few cache lines; every lane reads the same words)
        int t2[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) t2[x] = vals[st + min(jj + x, len - 1)];
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (jj + x < len && t2[x] == tok[q]) {
            ++mult[q];
            if (jj + x < j) keep[q] = false;
          }
      }
      if (keep[q]) {
        slot[q] = hash_slot(hkey, tok[q]);
        if (slot[q] >= 0) lr[q] = atomicAdd(&hcnt[slot[q]], 1);
      }
    }
  }
  __syncthreads();
  {
    constexpr int SL = kHash / kAggThreads;
    int k[SL], base[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
      const int h = threadIdx.x + q * kAggThreads;
      k[q] = hkey[h];
      base[q] = -1;
      if (k[q] != -1) base[q] = atomicAdd(&tcnt[k[q]], hcnt[h]);
    }
    int nl = 0;
#pragma unroll
    for (int q = 0; q < SL; ++q) {
      if (k[q] != -1) hbase[threadIdx.x + q * kAggThreads] = base[q];
      nl += base[q] == 0 ? 1 : 0;
    }
    const int incl = wave_incl_scan(nl, lane);
    int wb = 0;
    if (lane == 63 && incl > 0) wb = atomicAdd(&ctr[kNRuns], incl);
    int u = __shfl(wb, 63, 64) + incl - nl;
#pragma unroll
    for (int q = 0; q < SL; ++q)
      if (base[q] == 0) R[u++] = make_int4(k[q], 0, 0, INT_MAX);
  }
  __syncthreads();
  // arrival rank = base of the workgroup's share of the row + rank within the workgroup
#pragma unroll
  for (int q = 0; q < kBagNJ; ++q) {
    const int j = lig + q * W;
    int c = 0;
    bool direct = false;
    if (keep[q]) {
      if (slot[q] >= 0) {
        c = hbase[slot[q]] + lr[q];
      } else {
        direct = true;
        c = atomicAdd(&tcnt[tok[q]], 1);
      }
    }
    if (ent && j < max_len) {
      const int64_t sl = sub * (int64_t)max_len + j;
      tkey[sl] = keep[q] ? tok[q] : ARX_KEY_NONE;
      tarr[sl] = c;
      tmul[sl] = (float)mult[q];
    }
    const bool leader = direct && c == 0;
    const int u = wave_append(leader, &ctr[kNRuns], lane);
    if (leader) R[u] = make_int4(tok[q], 0, 0, INT_MAX);
  }
}

// ---- B: one thread per run: segment offset, map := offset, long runs listed -------------------
template <bool MT>
__device__ __forceinline__ void body_offsets(int64_t u, const TableSet& ts, const Stage& sg, int kshort,
                                             int32_t* __restrict__ runfirst) {
  const int lane = threadIdx.x & 63;
  const int nruns = sg.ctr[kNRuns];
  const bool act = u < nruns;
  int key = 0, c = 0;
  int32_t* cell = nullptr;
  if (act) {
    key = sg.R[u].x;
    const TabRow T = tab_of<MT>(ts, (uint32_t)key);
    cell = T.cnt + T.row;
    c = *cell;
  }
  const int incl = wave_incl_scan(c, lane);
  int base = 0;
  if (lane == 63 && incl > 0) base = atomicAdd(&sg.ctr[kCursor], incl);
  base = __shfl(base, 63, 64);
  if (!act) return;
  const int off = base + incl - c;
  *cell = off;
  sg.R[u] = make_int4(key, off, c, INT_MAX);
  if (runfirst) runfirst[off] = INT_MAX;
  if (c > kshort) {
    const int nch = (c + kItem - 1) / kItem;
    const int li = atomicAdd(&sg.ctr[kNLong], 1);
    const int it = atomicAdd(&sg.ctr[kNItems], nch);
    const int pb = nch > 1 ? atomicAdd(&sg.ctr[kNPart], nch) : 0;
    if (li < sg.cap_long && it + nch <= sg.cap_items && pb + nch <= sg.cap_part) {   // (bounds hold by construction)
      sg.LR[li] = make_int4((int)u, pb, nch, 0);
      for (int k = 0; k < nch; ++k) sg.items[it + k] = make_int2(li, k);
    }
  }
}

// ---- C: one thread per contribution: (src, coef) -> its run's segment -------------------------
// runfirst (bag riding on table 0): the smallest table-0 contribution index of every entity run, merged
// per workgroup in the hash table (key = the run's offset) before the global atomicMin.
__device__ __forceinline__ void body_place_sites(int64_t i, const CatSites& st, const SiteExtra& sx,
                                                 const TableSet& ts, const int32_t* __restrict__ keys,
                                                 const int32_t* __restrict__ src, const float* __restrict__ coef,
                                                 const int32_t* __restrict__ arr, int32_t* __restrict__ asrc,
                                                 float* __restrict__ acoef, int32_t* __restrict__ runfirst,
                                                 int* __restrict__ hkey, int* __restrict__ hmin) {
  const int64_t n = st.offs[st.nsites];
  if (runfirst) {
    for (int h = threadIdx.x; h < kHash; h += kAggThreads) {
      hkey[h] = -1;
      hmin[h] = INT_MAX;
    }
    __syncthreads();
  }
  const int key = i < n ? keys[i] : ARX_KEY_NONE;
  if (key != ARX_KEY_NONE) {
    const uint32_t tb = (uint32_t)key >> st.kb;
    const int row = key & ((1 << st.kb) - 1);
    const int off = pick4(ts.cnt, tb)[row];
    const int pos = off + arr[i];
    asrc[pos] = src[i];
    acoef[pos] = coef[i];
    if (runfirst && tb == 0) {
      int s = 0;
#pragma unroll
      for (int q = 1; q < kMaxSites; ++q)
        if (q < st.nsites && i >= st.offs[q]) s = q;
      const int i0 = sx.t0off[s] + (int32_t)(i - st.offs[s]);
      const int slot = hash_slot(hkey, off);
      if (slot >= 0) atomicMin(&hmin[slot], i0);
      else atomicMin(&runfirst[off], i0);
    }
  }
  if (runfirst) {
    __syncthreads();
    for (int h = threadIdx.x; h < kHash; h += kAggThreads)
      if (hkey[h] != -1) atomicMin(&runfirst[hkey[h]], hmin[h]);
  }
}

__device__ __forceinline__ void body_place_bags(int64_t sub, int lig, int wshift, const int32_t* __restrict__ U0,
                                                const int32_t* __restrict__ ctr0, int max_len,
                                                const int32_t* __restrict__ map0 /* offset of the entity's run */,
                                                const int32_t* __restrict__ runfirst,
                                                const int32_t* __restrict__ tcnt, const int32_t* __restrict__ tkey,
                                                const int32_t* __restrict__ tarr, const float* __restrict__ tmul,
                                                int32_t* __restrict__ asrc, float* __restrict__ acoef) {
  const int W = 1 << wshift;
  if (sub >= ctr0[kNRuns0]) return;
  const int first = runfirst[map0[U0[sub]]];
  for (int j = lig; j < max_len; j += W) {
    const int64_t slot = sub * (int64_t)max_len + j;
    const int tok = tkey[slot];
    if (tok == ARX_KEY_NONE) continue;
    const int pos = tcnt[tok] + tarr[slot];
    asrc[pos] = first;
    acoef[pos] = tmul[slot];
  }
}

// ---- D: one workgroup per long run: entries ordered by src -----------------------------------
// Counting sort over the range of gradient rows: a bitmap of the run's src values in LDS, an
// exclusive popcount prefix per word; entry rank = prefix[word] + popcount of the lower bits.
// (key, src) pairs are unique within a pass, so every bit belongs to one entry.
__device__ __forceinline__ void body_long_sort(int bid, int nblk, const Stage& sg, int64_t src_range,
                                               uint32_t* __restrict__ bm, int* __restrict__ wp, int* __restrict__ wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nthr = blockDim.x, nwv = nthr >> 6;
  const int nlong = min(sg.ctr[kNLong], (int)sg.cap_long);
  for (int li = bid; li < nlong; li += nblk) {
    const int4 r = sg.R[sg.LR[li].x];
    const int64_t off = r.y;
    const int cnt = r.z;
    int base = 0;
    for (int64_t sb = 0; sb < src_range; sb += (int64_t)kSlabWords * 32) {
      const int words = (int)min((int64_t)kSlabWords, (src_range - sb + 31) / 32);
      for (int w = tid; w < words; w += nthr) bm[w] = 0u;
      __syncthreads();
      for (int m = tid; m < cnt; m += nthr) {
        const int64_t s = (int64_t)sg.asrc[off + m] - sb;
        if (s >= 0 && s < (int64_t)words * 32) atomicOr(&bm[s >> 5], 1u << (s & 31));
      }
      __syncthreads();
      const int per = (words + nthr - 1) / nthr;
      const int w0 = tid * per;
      int local = 0;
      for (int k = 0; k < per; ++k)
        if (w0 + k < words) local += __popc(bm[w0 + k]);
      const int incl = wave_incl_scan(local, lane);
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      int woff = 0, total = 0;
      for (int q = 0; q < nwv; ++q) {
        if (q < wv) woff += wsum[q];
        total += wsum[q];
      }
      int run = woff + incl - local;
      for (int k = 0; k < per; ++k)
        if (w0 + k < words) {
          wp[w0 + k] = run;
          run += __popc(bm[w0 + k]);
        }
      __syncthreads();
      for (int m = tid; m < cnt; m += nthr) {
        const int sv = sg.asrc[off + m];
        const int64_t s = (int64_t)sv - sb;
        if (s >= 0 && s < (int64_t)words * 32) {
          const int w = (int)(s >> 5);
          const int rank = base + wp[w] + __popc(bm[w] & ((1u << (s & 31)) - 1u));
          sg.ssrc[off + rank] = sv;
          sg.scoef[off + rank] = sg.acoef[off + m];
        }
      }
      base += total;
      __syncthreads();
    }
  }
}

// runs of 2 .. kshort entries: ordered by src by a sub-group of 32 lanes (rank = entries with a
// smaller src, read as LDS broadcasts), so that the apply sums every run straight from memory order
__device__ __forceinline__ void body_short_sort(int64_t u, int l32, int sgl, const Stage& sg, int kshort,
                                                int (*stash)[32]) {
  const int nruns = sg.ctr[kNRuns];
  int4 r = make_int4(0, 0, 0, 0);
  if (u < nruns) r = sg.R[u];
  const int cnt = (r.z >= 1 && r.z <= kshort) ? r.z : 0;
  int sv = 0;
  float cv = 0.f;
  if (l32 < cnt) {
    sv = sg.asrc[r.y + l32];
    cv = sg.acoef[r.y + l32];
  }
  stash[sgl][l32] = sv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (l32 < cnt) {
    int rank = 0;
    for (int m = 0; m < cnt; ++m) {
      const int sm = stash[sgl][m];
      rank += (sm < sv || (sm == sv && m < l32)) ? 1 : 0;
    }
    if (cnt > 1) {
      sg.ssrc[r.y + rank] = sv;
      sg.scoef[r.y + rank] = cv;
    }
    if (rank == 0) sg.R2[u] = make_int2(sv, __float_as_int(cv));   // the apply starts from the run record alone
  }
}

// maps back to zero, one thread per run
template <bool MT>
__device__ __forceinline__ void body_reset(int64_t u, const TableSet& ts, const Stage& sg) {
  if (u >= sg.ctr[kNRuns]) return;
  const TabRow T = tab_of<MT>(ts, (uint32_t)sg.R[u].x);
  T.cnt[T.row] = 0;
}

// first contribution of every run of table 0 -> its run record (the apply's side-output row)
__device__ __forceinline__ void body_fill_first(int64_t u, const Stage& sg, const int32_t* __restrict__ runfirst) {
  if (u >= sg.ctr[kNRuns]) return;
  int4 r = sg.R[u];
  r.w = runfirst[r.y];
  sg.R[u] = r;
}

// ---- the launches: every workgroup belongs to one job (block ranges) --------------------------
struct BagJob {
  const int32_t* U0;
  const int32_t* vals;
  const int32_t* starts;
  const int32_t* lens;
  int max_len, wshift;
  int64_t bag_rows;
  int32_t* tcnt;
  int32_t* tkey;
  int32_t* tarr;
  float* tmul;
  int32_t* runfirst;
};

__global__ __launch_bounds__(kAggThreads) void k_grp_a(CatSites st, TableSet ts, int32_t* keys, int32_t* src,
                                                       float* coef, int32_t* arr, Stage sl, int32_t* U0) {
  __shared__ int hkey[kHash], hcnt[kHash], hbase[kHash];
  body_count_sites(blockIdx.x * (int64_t)kAggThreads + threadIdx.x, st, ts, keys, src, coef, arr, sl.R, sl.ctr, U0,
                   hkey, hcnt, hbase);
}

// B: offsets of the one-hot runs | token counting of the distinct entities' bags
template <bool MT>
__global__ __launch_bounds__(kAggThreads) void k_grp_b(TableSet ts, Stage sl, int kshort, int nb_off, BagJob bj,
                                                       Stage st) {
  __shared__ int hkey[kHash], hcnt[kHash], hbase[kHash];
  if ((int)blockIdx.x < nb_off) {
    body_offsets<MT>(blockIdx.x * (int64_t)kAggThreads + threadIdx.x, ts, sl, kshort, bj.runfirst);
    return;
  }
  body_count_bags((int)blockIdx.x - nb_off, bj.wshift, bj.U0, sl.ctr, bj.vals, bj.starts, bj.lens, bj.max_len,
                  bj.bag_rows, bj.tcnt, bj.tkey, bj.tarr, bj.tmul, st.R, st.ctr, hkey, hcnt, hbase);
}

// C: placement of the one-hot contributions | offsets of the token runs
__global__ __launch_bounds__(kAggThreads) void k_grp_c(CatSites stt, SiteExtra sx, TableSet ts, const int32_t* keys,
                                                       const int32_t* src, const float* coef, const int32_t* arr,
                                                       Stage sl, int32_t* runfirst, int nb_place, TableSet tst,
                                                       Stage st, int kshort) {
  __shared__ int hkey[kHash], hmin[kHash];
  if ((int)blockIdx.x < nb_place) {
    body_place_sites(blockIdx.x * (int64_t)kAggThreads + threadIdx.x, stt, sx, ts, keys, src, coef, arr, sl.asrc,
                     sl.acoef, runfirst, hkey, hmin);
    return;
  }
  body_offsets<false>(((int64_t)blockIdx.x - nb_place) * kAggThreads + threadIdx.x, tst, st, kshort, nullptr);
}

// D: long one-hot runs ordered | short one-hot runs ordered | (no bag: maps reset) | (bag: first ->
// run records, token placement)
template <bool MT>
__global__ __launch_bounds__(kLongThreads) void k_grp_d(TableSet ts, Stage sl, int64_t src_range, int kshort,
                                                        int nb_long, int nb_runs, int nb_short, int has_bag,
                                                        BagJob bj, Stage st) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ int wsum[kLongThreads / 64];
  __shared__ int stash[kLongThreads / 32][32];
  int b = blockIdx.x;
  if (b < nb_long) {
    const int words = (int)min((int64_t)kSlabWords, (src_range + 31) / 32);
    body_long_sort(b, nb_long, sl, src_range, smem, reinterpret_cast<int*>(smem + words), wsum);
    return;
  }
  b -= nb_long;
  if (b < nb_runs) {
    const int64_t u = (int64_t)b * kLongThreads + threadIdx.x;
    if (has_bag) body_fill_first(u, sl, bj.runfirst);
    else body_reset<MT>(u, ts, sl);
    return;
  }
  b -= nb_runs;
  if (b < nb_short) {
    body_short_sort((int64_t)b * (kLongThreads / 32) + (threadIdx.x >> 5), threadIdx.x & 31, threadIdx.x >> 5, sl,
                    kshort, stash);
    return;
  }
  b -= nb_short;
  const int64_t t = (int64_t)b * kLongThreads + threadIdx.x;
  body_place_bags(t >> bj.wshift, (int)(t & ((1 << bj.wshift) - 1)), bj.wshift, bj.U0, sl.ctr, bj.max_len, ts.cnt[0],
                  bj.runfirst, bj.tcnt, bj.tkey, bj.tarr, bj.tmul, st.asrc, st.acoef);
}

// E (bag only): long token runs ordered | short token runs ordered | both map sets reset
template <bool MT>
__global__ __launch_bounds__(kLongThreads) void k_grp_e(TableSet ts, Stage sl, TableSet tst, Stage st,
                                                        int64_t src_range, int kshort, int nb_long, int nb_short,
                                                        int nb_runs_l) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ int wsum[kLongThreads / 64];
  __shared__ int stash[kLongThreads / 32][32];
  int b = blockIdx.x;
  if (b < nb_long) {
    const int words = (int)min((int64_t)kSlabWords, (src_range + 31) / 32);
    body_long_sort(b, nb_long, st, src_range, smem, reinterpret_cast<int*>(smem + words), wsum);
    return;
  }
  b -= nb_long;
  if (b < nb_short) {
    body_short_sort((int64_t)b * (kLongThreads / 32) + (threadIdx.x >> 5), threadIdx.x & 31, threadIdx.x >> 5, st,
                    kshort, stash);
    return;
  }
  b -= nb_short;
  if (b < nb_runs_l) {
    body_reset<MT>((int64_t)b * kLongThreads + threadIdx.x, ts, sl);
    return;
  }
  body_reset<false>(((int64_t)b - nb_runs_l) * kLongThreads + threadIdx.x, tst, st);
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
#ifndef ARX_ABL_NOSTORE
      row_store(T.acc + (int64_t)T.row * d + col, a4);
#endif
    } else {
      w4.x -= lr * gg.x; w4.y -= lr * gg.y; w4.z -= lr * gg.z; w4.w -= lr * gg.w;
    }
#ifndef ARX_ABL_NOSTORE
    row_store(T.E + (int64_t)T.row * d + col, w4);
#else
    if (w4.x == 1.2345e-30f) row_store(T.E + (int64_t)T.row * d + col, w4);
#endif
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

// blocks [0, gshort): one sub-group per short run, NB runs in flight per sub-group;
// blocks [gshort, ...): one workgroup per work item of a long run.
template <int LPR, bool MT, bool SGD>
__global__ __launch_bounds__(256) void k_run_apply(TableSet ts, int d, Stage sg, const float* __restrict__ G,
                                                   int64_t ldg, const float* __restrict__ Gb,
                                                   const float* __restrict__ lr_dev,
                                                   const float* __restrict__ gscale_dev, MergeOut mo, int gshort,
                                                   int kshort) {
  constexpr int NSG = 64 / LPR;        // sub-groups per wave
  constexpr int NSGB = 256 / LPR;      // sub-groups per workgroup
  constexpr int NB = ARX_RUN_NB;
  constexpr int RU = ARX_RUN_RU;
  constexpr int LRU = 4;               // rows in flight per sub-group of a long run's work item
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lig = lane % LPR;
  const int g = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // the long runs' work items take the FIRST blocks of the grid: they are the long poles
  const int glong = (int)gridDim.x - gshort;
  if ((int)blockIdx.x >= glong) {
#ifdef ARX_ABL_NOSHORT
    return;
#endif
    constexpr int KS = LPR < 32 ? LPR : 32;          // entries of a short run
    // entries 1.. of the sub-group's runs, in summation order (private to the sub-group: wave-synchronous)
    __shared__ int s_src[4 * NSG][NB][KS];
    __shared__ float s_coef[4 * NSG][NB][KS];
    __shared__ float s_gb[4 * NSG][NB][KS];
    const int sgl = wv * NSG + g;
    const int nruns = sg.ctr[kNRuns];
    const int64_t nsg_tot = (int64_t)gshort * 4 * NSG;
    const int64_t sgid = (((int64_t)blockIdx.x - glong) * 4 + wv) * NSG + g;
    for (int64_t base = 0; base < nruns; base += nsg_tot * NB) {
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
        act[j] = act[j] && r[j].z > 0 && r[j].z <= kshort;
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
#ifndef ARX_ABL_NOROW
            if (colok) {
              wrow[j] = row_load(T.E + (int64_t)T.row * d + col);
              if (!SGD) arow[j] = row_load(T.acc + (int64_t)T.row * d + col);
            }
#endif
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
      for (int j = 0; j < NB; ++j) a[j] = act[j] ? f4_fma(__int_as_float(r2[j].y), g0[j], z4) : z4;
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
              v[j][u] = (ok && colok) ? *reinterpret_cast<const float4*>(G + (int64_t)s * ldg + col) : z4;
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
#ifdef ARX_ABL_NOLONG
  return;
#endif
  __shared__ __attribute__((aligned(16))) float sh[NSGB][LPR * 4];
  __shared__ float shb[NSGB];
  __shared__ int s_last;
  const int sgb = threadIdx.x / LPR;
  const int nitems = min(sg.ctr[kNItems], (int)sg.cap_items);
  for (int it = (int)blockIdx.x; it < nitems; it += glong) {
    const int2 item = sg.items[it];
    const int4 lr4 = sg.LR[item.x];
    const int4 r = sg.R[lr4.x];
    const int nch = lr4.z;
    const int64_t off = r.y;
    const int cnt = r.z;
    const int e0 = item.y * kItem + sgb * LPR;
    const int m = max(0, min(LPR, cnt - e0));
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
        v[u] = (ok && colok) ? *reinterpret_cast<const float4*>(G + (int64_t)s * ldg + col) : z4;
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
            v[u] = colok ? agent_load4(sg.part + pr * d + col) : z4;
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

// ---- workspace layouts --------------------------------------------------------------------------
size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct StageWs {
  size_t ctr, R, R2, LR, items, asrc, acoef, ssrc, scoef, part, part_b, total;
  int64_t cap_long, cap_items, cap_part;
};
void stage_layout(int64_t n, int64_t nruns_cap, int d, size_t* o, StageWs* w) {
  // long runs have more than 8 entries (LPR >= 8)
  w->cap_long = n / 9 + 2;
  w->cap_items = w->cap_long + n / kItem + 2;
  w->cap_part = n / (kItem / 2) + 4;
  w->ctr = *o; *o += al(kCtrInts * 4);
  w->R = *o; *o += al((size_t)nruns_cap * 16);
  w->R2 = *o; *o += al((size_t)nruns_cap * 8);
  w->LR = *o; *o += al((size_t)w->cap_long * 16);
  w->items = *o; *o += al((size_t)w->cap_items * 8);
  w->asrc = *o; *o += al((size_t)n * 4);
  w->acoef = *o; *o += al((size_t)n * 4);
  w->ssrc = *o; *o += al((size_t)n * 4);
  w->scoef = *o; *o += al((size_t)n * 4);
  w->part = *o; *o += al((size_t)w->cap_part * (size_t)d * 4);
  w->part_b = *o; *o += al((size_t)w->cap_part * 4);
  w->total = *o;
}
Stage stage_of(char* base, const StageWs& w) {
  Stage s;
  s.ctr = reinterpret_cast<int32_t*>(base + w.ctr);
  s.R = reinterpret_cast<int4*>(base + w.R);
  s.R2 = reinterpret_cast<int2*>(base + w.R2);
  s.LR = reinterpret_cast<int4*>(base + w.LR);
  s.items = reinterpret_cast<int2*>(base + w.items);
  s.asrc = reinterpret_cast<int32_t*>(base + w.asrc);
  s.acoef = reinterpret_cast<float*>(base + w.acoef);
  s.ssrc = reinterpret_cast<int32_t*>(base + w.ssrc);
  s.scoef = reinterpret_cast<float*>(base + w.scoef);
  s.part = reinterpret_cast<float*>(base + w.part);
  s.part_b = reinterpret_cast<float*>(base + w.part_b);
  s.cap_long = w.cap_long;
  s.cap_items = w.cap_items;
  s.cap_part = w.cap_part;
  return s;
}

struct SiteWs {
  StageWs st;
  size_t arr, runfirst, u0, total;
};
void site_layout(int64_t n, int d, SiteWs* w) {
  size_t o = 0;
  stage_layout(n, n, d, &o, &w->st);
  w->arr = o; o += al((size_t)n * 4);
  w->runfirst = o; o += al((size_t)n * 4);
  w->u0 = o; o += al((size_t)n * 4);
  w->total = o;
}

struct BagWs2 {
  StageWs st;
  size_t tkey, tarr, tmul, gu, gub, total;
};
void bag_layout(int64_t n0, int max_len, int d, BagWs2* w) {
  const int64_t nt = n0 * (int64_t)max_len;
  size_t o = 0;
  stage_layout(nt, nt, d, &o, &w->st);
  w->tkey = o; o += al((size_t)nt * 4);
  w->tarr = o; o += al((size_t)nt * 4);
  w->tmul = o; o += al((size_t)nt * 4);
  w->gu = o; o += al((size_t)n0 * (size_t)d * 4);
  w->gub = o; o += al((size_t)n0 * 4);
  w->total = o;
}

int blocks_for(int64_t threads, int per_block) {
  int64_t b = (threads + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : b);
}

#define ARX_GRP_LPR(lpr, CALL)                        \
  switch (lpr) {                                      \
    case 8: { constexpr int LPR = 8; CALL; } break;   \
    case 16: { constexpr int LPR = 16; CALL; } break; \
    case 32: { constexpr int LPR = 32; CALL; } break; \
    default: { constexpr int LPR = 64; CALL; } break; \
  }

template <bool MT>
int launch_run_apply(const TableSet& ts, int d, const Stage& sg, int64_t nruns_cap, int64_t n, const float* G,
                     int64_t ldg, const float* Gb, const float* lr_dev, const float* gscale_dev, const MergeOut& mo,
                     bool sgd, hipStream_t s) {
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  const int nsg = 64 / lpr;
  int64_t gshort = ceil_div(nruns_cap, (int64_t)4 * nsg * ARX_RUN_NB);
  const int64_t cap = (int64_t)cu_count() * 8;
  if (gshort > cap) gshort = cap;
  if (gshort < 1) gshort = 1;
  int64_t glong = n / (kshort + 1) + 1;       // upper bound of the work items is larger; the loop strides
  const int64_t capl = (int64_t)cu_count() * 4;
  if (glong > capl) glong = capl;
  if (sgd) {
    ARX_GRP_LPR(lpr, (k_run_apply<LPR, MT, true><<<(int)(gshort + glong), 256, 0, s>>>(
                         ts, d, sg, G, ldg, Gb, lr_dev, gscale_dev, mo, (int)gshort, kshort)));
  } else {
    ARX_GRP_LPR(lpr, (k_run_apply<LPR, MT, false><<<(int)(gshort + glong), 256, 0, s>>>(
                         ts, d, sg, G, ldg, Gb, lr_dev, gscale_dev, mo, (int)gshort, kshort)));
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

size_t long_sort_lds(int64_t src_range) {
  int64_t words = (src_range + 31) / 32;
  if (words > kSlabWords) words = kSlabWords;
  if (words < 1) words = 1;
  return (size_t)words * 8;
}

}  // namespace

bool grouped_supported(const TableSet& ts, int ntables, int d, const CatSites& st, const BagStage* bag) {
  static const bool off = getenv("ARX_K7_SORTED") != nullptr;
  if (off || st.nextra != 0 || lanes_per_row(d) < 8) return false;
  for (int t = 0; t < ntables; ++t)
    if (!ts.cnt[t]) return false;
  if (bag && (!bag->aux_cnt || bag->max_len > 64 * kBagNJ)) return false;
  return true;
}

size_t grouped_ws_bytes(int64_t n, int d) {
  SiteWs w;
  site_layout(n > 0 ? n : 1, d, &w);
  return w.total;
}

size_t grouped_bag_ws_bytes(int64_t n0, int max_len, int64_t, int d) {
  BagWs2 w;
  bag_layout(n0 > 0 ? n0 : 1, max_len, d, &w);
  return w.total;
}

int sparse_adagrad_sites_grouped(const TableSet& ts, int ntables, int d, const CatSites& st, const float* G,
                                 int64_t ldg, int64_t, const float* Gb, const float* lr_dev,
                                 const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf, float* coef_buf,
                                 void* workspace, size_t workspace_bytes, hipStream_t s, int phase,
                                 const BagStage* bag) {
  const int64_t n = st.offs[st.nsites];
  if (n == 0) return ARX_OK;
  SiteWs w;
  site_layout(n, 256, &w);
  if (!workspace || workspace_bytes < w.total) {
    set_error("arx_sparse_adagrad_cat: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return ARX_EWORKSPACE;
  }
  int tbits = 0;
  while ((1 << tbits) < ntables) ++tbits;
  if (ts.kb + tbits > 30) {
    set_error("arx_sparse_adagrad_cat: %d key bits (rows + tables) exceed 30", ts.kb + tbits);
    return ARX_EUNSUPPORTED;
  }
  char* base = reinterpret_cast<char*>(workspace);
  const Stage sl = stage_of(base, w.st);
  int32_t* arr = reinterpret_cast<int32_t*>(base + w.arr);
  int32_t* runfirst = reinterpret_cast<int32_t*>(base + w.runfirst);
  int32_t* U0 = reinterpret_cast<int32_t*>(base + w.u0);
  const int lpr = lanes_per_row(d);
  const int kshort = lpr < 32 ? lpr : 32;
  const bool multi = ntables > 1;
  // gradient rows the sites read: the range of src values
  int64_t src_range = 1;
  SiteExtra sx;
  int64_t n0 = 0;
  for (int q = 0; q < kMaxSites; ++q) {
    sx.t0off[q] = (int32_t)n0;
    if (q < st.nsites) {
      const int64_t cnt = st.offs[q + 1] - st.offs[q];
      if (st.row_base[q] + cnt > src_range) src_range = st.row_base[q] + cnt;
      if (st.table[q] == 0) n0 += cnt;
    }
  }
  if (bag && n0 == 0) bag = nullptr;
  BagWs2 bw;
  char* bbase = nullptr;
  Stage stt = {};
  BagJob bj = {};
  TableSet tst = {};
  int64_t nt = 0;
  if (bag) {
    bag_layout(n0, bag->max_len, d, &bw);
    if (!bag->ws || bag->ws_bytes < bw.total) {
      set_error("arx_sparse_adagrad_cat_multi_bags: bag workspace too small (%zu < %zu)", bag->ws_bytes, bw.total);
      return ARX_EWORKSPACE;
    }
    bbase = reinterpret_cast<char*>(bag->ws);
    stt = stage_of(bbase, bw.st);
    nt = n0 * (int64_t)bag->max_len;
    int wshift = 0;
    while ((1 << wshift) < bag->max_len && wshift < 6) ++wshift;
    bj.U0 = U0;
    bj.vals = bag->vals;
    bj.starts = bag->starts;
    bj.lens = bag->lens;
    bj.max_len = bag->max_len;
    bj.wshift = wshift;
    bj.bag_rows = bag->rows;
    bj.tcnt = bag->aux_cnt;
    bj.tkey = reinterpret_cast<int32_t*>(bbase + bw.tkey);
    bj.tarr = reinterpret_cast<int32_t*>(bbase + bw.tarr);
    bj.tmul = reinterpret_cast<float*>(bbase + bw.tmul);
    bj.runfirst = runfirst;
    int kbt = 1;
    while ((1ll << kbt) < bag->rows && kbt < 30) ++kbt;
    tst.E[0] = bag->E;
    tst.acc[0] = bag->acc;
    tst.bias[0] = bag->bias;
    tst.bias_acc[0] = bag->bias_acc;
    tst.cnt[0] = bag->aux_cnt;
    tst.kb = kbt;
  }
  if (phase & 1) {
    ARX_CHECK_HIP(hipMemsetAsync(sl.ctr, 0, kCtrInts * 4, s));
    if (bag) ARX_CHECK_HIP(hipMemsetAsync(stt.ctr, 0, kCtrInts * 4, s));
    k_grp_a<<<blocks_for(n, kAggThreads), kAggThreads, 0, s>>>(st, ts, keys_buf, src_buf, coef_buf, arr, sl,
                                                               bag ? U0 : nullptr);
    ARX_CHECK_LAUNCH();
    const int nb_off = blocks_for(n, kAggThreads);
    const int nb_tcount = bag ? blocks_for(n0 << bj.wshift, kAggThreads) : 0;
    if (multi) k_grp_b<true><<<nb_off + nb_tcount, kAggThreads, 0, s>>>(ts, sl, kshort, nb_off, bj, stt);
    else k_grp_b<false><<<nb_off + nb_tcount, kAggThreads, 0, s>>>(ts, sl, kshort, nb_off, bj, stt);
    ARX_CHECK_LAUNCH();
    const int nb_place = blocks_for(n, kAggThreads);
    const int64_t truns = bag ? (nt < bag->rows ? nt : bag->rows) : 0;      // token runs: at most one per row
    const int nb_toff = bag ? blocks_for(truns, kAggThreads) : 0;
    k_grp_c<<<nb_place + nb_toff, kAggThreads, 0, s>>>(st, sx, ts, keys_buf, src_buf, coef_buf, arr, sl,
                                               bag ? runfirst : nullptr, nb_place, tst, stt, kshort);
    ARX_CHECK_LAUNCH();
    int nb_long = (int)(n / (kshort + 1) + 1);
    if (nb_long > cu_count()) nb_long = cu_count();
    const int nb_runs = blocks_for(n, kLongThreads);
    const int nb_short = blocks_for(n * 32, kLongThreads);
    const int nb_tplace = bag ? blocks_for(n0 << bj.wshift, kLongThreads) : 0;
    const size_t lds = long_sort_lds(src_range);
    if (multi) k_grp_d<true><<<nb_long + nb_runs + nb_short + nb_tplace, kLongThreads, lds, s>>>(
        ts, sl, src_range, kshort, nb_long, nb_runs, nb_short, bag ? 1 : 0, bj, stt);
    else k_grp_d<false><<<nb_long + nb_runs + nb_short + nb_tplace, kLongThreads, lds, s>>>(
        ts, sl, src_range, kshort, nb_long, nb_runs, nb_short, bag ? 1 : 0, bj, stt);
    ARX_CHECK_LAUNCH();
    if (bag) {
      int nb_tlong = (int)(nt / (kshort + 1) + 1);
      if (nb_tlong > cu_count()) nb_tlong = cu_count();
      const int nb_truns = blocks_for(truns, kLongThreads);
      const int nb_tshort = blocks_for(truns * 32, kLongThreads);
      const size_t ldst = long_sort_lds(n0);
      if (multi) k_grp_e<true><<<nb_tlong + nb_tshort + nb_runs + nb_truns, kLongThreads, ldst, s>>>(
          ts, sl, tst, stt, n0, kshort, nb_tlong, nb_tshort, nb_runs);
      else k_grp_e<false><<<nb_tlong + nb_tshort + nb_runs + nb_truns, kLongThreads, ldst, s>>>(
          ts, sl, tst, stt, n0, kshort, nb_tlong, nb_tshort, nb_runs);
      ARX_CHECK_LAUNCH();
    }
  }
  if (!(phase & 2)) return ARX_OK;
  bool any_bias = bag && bag->bias;
  for (int t = 0; t < ntables; ++t) any_bias = any_bias || ts.bias[t] != nullptr;
  const float* gb_in = any_bias ? Gb : nullptr;
  bool sgd = ts.acc[0] == nullptr;
  if (!ts.E[0] && ntables > 1) sgd = ts.acc[1] == nullptr;          // (table 0 virtual)
  if (!ts.E[0] && ntables == 1 && bag) sgd = bag->acc == nullptr;
  MergeOut side = {nullptr, nullptr, nullptr, -1, 0};
  if (bag)
    side = MergeOut{reinterpret_cast<float*>(bbase + bw.gu),
                    bag->bias ? reinterpret_cast<float*>(bbase + bw.gub) : nullptr, bag->lens, 0, ts.kb};
  int rc;
  if (multi) rc = launch_run_apply<true>(ts, d, sl, n, n, G, ldg, gb_in, lr_dev, gscale_dev, side, sgd, s);
  else rc = launch_run_apply<false>(ts, d, sl, n, n, G, ldg, gb_in, lr_dev, gscale_dev, side, sgd, s);
  if (rc || !bag) return rc;
  const MergeOut none = {nullptr, nullptr, nullptr, -1, 0};
  int64_t truns = nt < bag->rows ? nt : bag->rows;
  return launch_run_apply<false>(tst, d, stt, truns, nt, side.Gu, d, bag->bias ? side.Gub : nullptr, lr_dev,
                                 gscale_dev, none, bag->acc == nullptr, s);
}

}  // namespace arx
