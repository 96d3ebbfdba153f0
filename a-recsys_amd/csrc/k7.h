// k7.h -- device helpers shared by the sparse-Adagrad kernels (optim.hip: sorts, window apply;
// group.hip: run records + run-centric apply).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace arx {

// Table rows are touched ONCE per pass (one Adagrad update per row): streamed past L2 so that they
// do not evict the gradient rows the token stage re-reads (ARX_WIN_NT=0: plain loads / stores).
#ifndef ARX_WIN_NT
#define ARX_WIN_NT 1
#endif
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 row_load(const float* p) {
#if ARX_WIN_NT
  const v4f_nt q = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
  return make_float4(q.x, q.y, q.z, q.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void row_store(float* p, float4 v) {
#if ARX_WIN_NT
  v4f_nt q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<v4f_nt*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

__device__ __forceinline__ float4 f4_fma(float c, float4 v, float4 a) {
  return make_float4(fmaf(c, v.x, a.x), fmaf(c, v.y, a.y), fmaf(c, v.z, a.z), fmaf(c, v.w, a.w));
}
__device__ __forceinline__ float4 f4_add2(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

struct TabRow {
  float* E;
  float* acc;
  float* bias;
  float* bias_acc;
  int32_t* cnt;
  uint32_t row;
};
// key = (table << kb) | row  ->  that table's pointers (selects, not a runtime-indexed
// kernel-argument array: that would go through scratch)
// a[t] for t < 4 without indexing the kernel-argument struct.  The plain select chain
// (t == 0 ? a[0] : t == 1 ? ...) is folded back into an indexed load by the optimiser, which
// copies the whole TableSet to scratch and pays a scratch round trip per looked-up pointer;
// the empty asm between the selects keeps them selects (v_cndmask on kernel-argument SGPRs).
template <class P>
__device__ __forceinline__ P* pick4(P* const (&a)[4], uint32_t t) {
  uint64_t r = reinterpret_cast<uint64_t>(a[0]);
  r = (t == 1u) ? reinterpret_cast<uint64_t>(a[1]) : r;
  asm volatile("" : "+v"(r));
  r = (t == 2u) ? reinterpret_cast<uint64_t>(a[2]) : r;
  asm volatile("" : "+v"(r));
  r = (t == 3u) ? reinterpret_cast<uint64_t>(a[3]) : r;
  return reinterpret_cast<P*>(r);
}

template <bool MT>
__device__ __forceinline__ TabRow tab_of(const TableSet& ts, uint32_t key) {
  TabRow r;
  if (!MT) {   // single table: wave-uniform pointers straight from the kernel arguments
    r.row = key;
    r.E = ts.E[0];
    r.acc = ts.acc[0];
    r.bias = ts.bias[0];
    r.bias_acc = ts.bias_acc[0];
    r.cnt = ts.cnt[0];
    return r;
  }
  const uint32_t t = key >> ts.kb;
  r.row = key & ((1u << ts.kb) - 1u);
  r.E = pick4(ts.E, t);
  r.acc = pick4(ts.acc, t);
  r.bias = pick4(ts.bias, t);
  r.bias_acc = pick4(ts.bias_acc, t);
  r.cnt = pick4(ts.cnt, t);
  return r;
}

// lr * g / sqrt(a) as lr * g * rsq(a): v_rsq_f32 is good to 1 ulp, the IEEE sqrt + divide
// sequences cost ~25 VALU instructions per component and the apply kernels are VALU-issue bound
// (a wave64 instruction holds its SIMD for 4 cycles) -- profiles/README.md, K7.
// (v_rsq_f32 flushes denormal inputs to zero -> inf; the slot is clamped to FLT_MIN so a slot
// that was initialised to 0 and met a gradient below ~1e-19 yields a finite step, not inf.)
__device__ __forceinline__ float adagrad_delta(float lr, float g, float a) {
  return lr * g * __frsqrt_rn(a < 1.17549435e-38f ? 1.17549435e-38f : a);   // (NaN slots stay NaN)
}

// MERGE mode of the apply kernels (arx_sparse_adagrad_bags, stage 1): the sorted
// keys are ENTITY ids (items of the batch and of the pool), a run = all gradient rows of one
// entity; instead of an Adagrad update the run's sum, times 1/len(entity's bag), is written to
// Gu[slot of the run] -- the one gradient row the entity's bag tokens then
// share in stage 2 (duplicate entities cost one row, not one row per occurrence).
struct MergeOut {
  float* Gu;               // [n, d] merged rows (window path: indexed by the head's sorted position;
                           // grouped path: by the entity's first contribution among table 0's)
  float* Gub;              // [n] merged bias gradients (null: none)
  const int32_t* lens;     // bag length per entity
  // table < 0: MERGE mode proper (every key is an entity id).  table >= 0: SIDE output of an
  // ordinary Adagrad pass (arx_sparse_adagrad_cat_multi_bags): runs of that table -- an item's id
  // row, whose lookups are also the lookups of the item's bag -- are applied AND written here, so
  // the bag's token stage needs no entity sort and no merge pass of its own.
  int table;
  int kb;                  // row bits of the key (side output: key = (table << kb) | entity)
  // csc.hip: the row of Gu an entity's merged gradient goes to (null: the head's sorted position) -- the
  // static token order addresses Gu through this map, written by k_csc_mark from the step's lookups
  const int32_t* slot_of;
};

__device__ __forceinline__ bool merge_side(const MergeOut& mo, uint32_t key) {
  return mo.Gu != nullptr && mo.table >= 0 && (int)(key >> mo.kb) == mo.table;
}

__device__ __forceinline__ void merge_row(const MergeOut& mo, int d, uint32_t key, int64_t head_pos, int col,
                                          bool colok, int lig, float4 g, float gb) {
  const float inv = 1.f / (float)mo.lens[key];
  if (mo.slot_of) head_pos = mo.slot_of[key];
  if (colok)
    *reinterpret_cast<float4*>(mo.Gu + head_pos * (int64_t)d + col) =
        make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
  if (mo.Gub && lig == 0) mo.Gub[head_pos] = gb * inv;
}

// csc.hip: the static token-major order of a multi-hot table's bags (built once per table: arx/ops.py BagCSC)
struct BagCsc {
  const int32_t* qpos;     // [len(vals)] CSR position -> place in the token-major order (-1: token out of range)
  const int2* qte;         // [nq] {token row, entity} of every place
  uint8_t* flags;          // [nq rounded up to 256] live pairs of the step; zero between steps
  uint8_t* cflags;         // [flags bytes / 16] one coarse byte per 16 places (a 16-byte flag word); zero between steps
  int32_t* slot_of;        // [entities] row of the merged gradients Gu of an entity of the step
  int64_t nq;
};
struct MarkSites {         // the entity lookups of the step (the sites of table 0 of the one-hot pass)
  int n;
  const int32_t* ids[kMaxSites];
  const int32_t* cat_map[kMaxSites];
  int64_t offs[kMaxSites + 1];
  int64_t rows;            // entities (rows of table 0)
};
int launch_csc_mark(const MarkSites& ms, const BagCsc& csc, const int32_t* starts, const int32_t* lens,
                    int max_len, int32_t* zero_i, int n_zero_i, void* zero_l, int n_zero_l, hipStream_t s);
int csc_compact_blocks(int64_t nq);
int launch_csc_compact(const BagCsc& csc, uint32_t* sk, int32_t* ssrc, float* scoef, int64_t cap,
                       int32_t* n_out, void* lookback, hipStream_t s);

// group.hip: run records of a radix-sorted pass + the run-centric apply (d >= 32).
enum { kNRuns = 0, kNLong = 2, kNItems = 3, kNPart = 4, kNRuns0 = 5 };   // counters of a pass (ints, zeroed by the sort)
struct RunLists {
  int4* R;              // run records {key, offset, count, head position}
  int2* R2;             // first entry of every run {src, coef bits}
  int4* LR;             // long runs {run, first partial row, items, ticket}
  int2* items;          // work items {long run, item}
  const int32_t* ssrc;  // the pass's sorted gradient-source rows / coefficients
  const float* scoef;
  int32_t* ctr;
  float* part;          // partial rows of the long runs with several items
  float* part_b;
  int64_t cap_long, cap_items, cap_part;
};
size_t run_lists_bytes(int64_t n, int d);
RunLists run_lists_of(char* base, int64_t n, int d, const int32_t* ssrc, const float* scoef, int32_t* ctr);
bool runs_path(int d);
// ent_lens (optional): also the ordered exclusive offsets hoff[p] of the bags of the entity heads
// ((key >> ent_kb) == ent_tag) in a compact token list, and its length; lookback: 8 bytes per
// workgroup (runs_extract_blocks(n)), zero on entry.
int launch_runs_extract(const uint32_t* sk, int64_t n, const int32_t* n_dev, uint32_t sentinel, const RunLists& rl,
                        int d, hipStream_t s, const int32_t* ent_lens = nullptr, int ent_kb = 0,
                        uint32_t ent_tag = 0, int max_len = 0, int32_t* hoff = nullptr, int64_t hoff_n = 0,
                        int32_t* total = nullptr,
                        void* lookback = nullptr);
int runs_extract_blocks(int64_t n);
// which runs of the lists one apply takes: the record range [ctr[rec_lo], ctr[rec_hi]) (-1: from 0 / to the end;
// meaningful when the records are in sorted order: k_runs_extract on its look-back path) and, for the long runs
// (their own unordered list), the table: -1 all, t >= 0 table t, t <= -2 every table but -2 - t
struct ApplySel {
  int rec_lo, rec_hi, tsel;
};
int launch_run_apply(const TableSet& ts, bool multi, int d, const RunLists& rl, int64_t n, const float* G,
                     int64_t ldg, const float* Gb, const float* lr_dev, const float* gscale_dev, const MergeOut& mo,
                     bool sgd, hipStream_t s, const ApplySel& sel = ApplySel{-1, -1, -1});
// two applies in ONE launch: job A (table set a, selection sela, side output moa) and job B (single table b, all runs)
int launch_run_apply_pair(const TableSet& tsa, bool multi_a, const RunLists& rla, int64_t na, const float* Ga,
                          int64_t ldga, const float* Gba, const MergeOut& moa, const ApplySel& sela,
                          const TableSet& tsb, const RunLists& rlb, int64_t nb, const float* Gbm, int64_t ldgb,
                          const float* Gbb, int d, const float* lr_dev, const float* gscale_dev, bool sgd,
                          hipStream_t s);

}  // namespace arx
