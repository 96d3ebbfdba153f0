// optim.hip -- K7: embedding_lookup gradient scatter fused with sparse Adagrad,
//              K8: dense Adagrad, squared norms, clip coefficient.
//
// Reference semantics (hmf_model.py:146-151, seqModel.py:173-182; TF-1.0
// AdagradOptimizer): duplicate indices of an IndexedSlices gradient are summed
// first, then every touched row gets ONE update
//     acc += g^2 ;  w -= lr * g / sqrt(acc).
// Rows that receive no gradient are untouched, which is why the reference's
// dense update over the whole table can be restricted to the touched rows.
//
// Algorithm (deterministic -- no floating-point atomics):
//   1. contributions (key = table row, src = gradient row, coef) are sorted by key, stably:
//      n <= 8192: chip-wide LDS rank sort, one launch (k_rank_sort);
//      larger   : own LSD radix sort (radix_sort.hip), which also drops padded / invalid keys.
//   2. apply: runs of equal keys are summed in sorted (= original) order and each touched row
//      gets ONE Adagrad update.
//      n <= 8192: one sub-group (LPR lanes x float4) per sorted position; multi-piece runs are
//                 finished by their last-arriving piece (per-row ticket counter) -- k_sparse_onepass;
//      larger   : one wave (or 8) per window of 64 sorted positions -- k_sparse_win -- plus
//                 k_sparse_finish for runs that cross windows (short: one sub-group each,
//                 Zipf-hot: one 1024-thread workgroup each, fixed-order LDS combine).
//   Several tables can share steps 1-2: the key carries the table index above the row bits.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include "common.h"
#include "k7.h"

namespace arx {

constexpr int kPiece = 64;          // positions per piece
constexpr int kRankSortMax = 8192;
// entity stage of arx_sparse_adagrad_bags.  (Measured: ONE rank-sort launch at n = 17.4 k -- 70 KB of
// keys staged per workgroup, O(n^2) compares -- costs ~115 us against ~45 us for the 7 launches of
// the two-pass radix sort; the rank sort stays below 8192.)
static const int kTokenWpw = 4;   // waves per window of the token stage (merged rows: 342 us/step with 1, 336 with 4, 344 with 8 at C3 B=16384)
static const int kRankSortEntities = kRankSortMax;
constexpr int kPassBBlocks = 128;   // persistent grid of pass B

// Stable rank sort for n <= 16384 keys, chip-wide in ONE launch: every workgroup keeps
// the whole (normalised) key list in LDS (<= 64 KB); TPE lanes share one element and
// count, each over a slice of the list, how many (key, index) pairs precede it; the
// count IS the element's sorted position.  O(n^2 / CUs) integer compares with
// ds_read_b128 -- a few microseconds at n = 5k -- instead of several dependent radix
// passes by a single workgroup (15-27 us measured).
template <int TPE>
__global__ __launch_bounds__(256) void k_rank_sort(const int32_t* __restrict__ keys, int64_t n,
                                                   uint32_t sentinel, uint32_t* __restrict__ sk,
                                                   uint32_t* __restrict__ spos,
                                                   int32_t* __restrict__ list_count,
                                                   const int32_t* __restrict__ src_in,
                                                   const float* __restrict__ coef_in,
                                                   int32_t* __restrict__ ssrc,
                                                   float* __restrict__ scoef) {
  extern __shared__ __attribute__((aligned(16))) uint32_t skeys[];
  const int n4 = (int)((n + 3) & ~(int64_t)3);
  // stage the key list: 8 independent loads in flight per thread (the loop is otherwise a
  // chain of ~1 us L2 round trips: 22 us measured at n = 5k with one load per iteration)
  for (int base = 0; base < n4; base += 256 * 8) {
    int32_t r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      r[u] = (i < n) ? keys[i] : -2;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      if (i < n4) {
        uint32_t kk = 0xffffffffu;          // padding: greater than every real key
        if (i < n)
          kk = (r[u] == ARX_KEY_NONE || r[u] < 0 || (uint32_t)r[u] >= sentinel) ? sentinel
                                                                                : (uint32_t)r[u];
        skeys[i] = kk;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { list_count[0] = 0; list_count[1] = 0; }
  __syncthreads();
  constexpr int EPB = 256 / TPE;
  const int e = blockIdx.x * EPB + threadIdx.x / TPE;
  const int part = threadIdx.x % TPE;
  const uint32_t ki = (e < n) ? skeys[e] : 0u;
  int chunk = ((n4 / 4 + TPE - 1) / TPE) * 4;
  if ((chunk & 63) == 0) chunk += 4;      // slices 256 B apart would all hit one LDS bank row
  const int jb = min(n4, part * chunk);
  const int je = min(n4, jb + chunk);
  int cnt = 0;
  for (int j = jb; j < je; j += 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(&skeys[j]);
    cnt += (v.x < ki) || (v.x == ki && j < e);
    cnt += (v.y < ki) || (v.y == ki && j + 1 < e);
    cnt += (v.z < ki) || (v.z == ki && j + 2 < e);
    cnt += (v.w < ki) || (v.w == ki && j + 3 < e);
  }
#pragma unroll
  for (int o = TPE / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, TPE);
  if (part == 0 && e < n) {
    sk[cnt] = ki;
    spos[cnt] = (uint32_t)e;
    if (ssrc) {   // gradient-source row and coefficient in sorted order: one hop less in the apply pass
      ssrc[cnt] = src_in ? src_in[e] : e;
      scoef[cnt] = coef_in ? coef_in[e] : 1.f;
    }
  }
}

static inline int launch_rank_sort(const int32_t* keys, int64_t n, uint32_t sentinel, uint32_t* sk,
                                   uint32_t* spos, int32_t* count, hipStream_t s,
                                   const int32_t* src_in = nullptr, const float* coef_in = nullptr,
                                   int32_t* ssrc = nullptr, float* scoef = nullptr) {
  const int tpe = 16;      // lanes sharing one element (32 / 8 / 4 measured slower at n = 5 k .. 16 k)
  const size_t lds = (size_t)((n + 3) & ~(int64_t)3) * sizeof(uint32_t);
  if (lds > 64 * 1024) {       // past the default dynamic-LDS limit (n > 16384): opt in once
    static bool raised = false;
    if (!raised) {
      const int cap = 160 * 1024;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_sort<32>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_sort<16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_sort<8>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_sort<4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      raised = true;
    }
  }
  if (tpe == 32)
    k_rank_sort<32><<<(int)ceil_div(n, 8), 256, lds, s>>>(keys, n, sentinel, sk, spos, count, src_in,
                                                              coef_in, ssrc, scoef);
  else if (tpe == 16)
    k_rank_sort<16><<<(int)ceil_div(n, 16), 256, lds, s>>>(keys, n, sentinel, sk, spos, count, src_in,
                                                              coef_in, ssrc, scoef);
  else if (tpe == 4)
    k_rank_sort<4><<<(int)ceil_div(n, 64), 256, lds, s>>>(keys, n, sentinel, sk, spos, count, src_in,
                                                              coef_in, ssrc, scoef);
  else
    k_rank_sort<8><<<(int)ceil_div(n, 32), 256, lds, s>>>(keys, n, sentinel, sk, spos, count, src_in,
                                                              coef_in, ssrc, scoef);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

// multi-site key generation (wide launch; the gathers are latency-bound)
__global__ __launch_bounds__(256) void k_site_keys(CatSites st, int32_t* __restrict__ keys,
                                                   int32_t* __restrict__ src,
                                                   float* __restrict__ coef) {
  const int64_t n = st.offs[st.nsites];
  const int64_t nall = n + st.xoffs[st.nextra];
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nall; i += stride) {
    if (i >= n) {   // pre-expanded contribution: validate + add the table prefix
      const int64_t j = i - n;
      int e = 0;
#pragma unroll
      for (int q = 1; q < kMaxSites; ++q)
        if (q < st.nextra && j >= st.xoffs[q]) e = q;
      const int tb = st.xtable[e];
      const int64_t rows = tb == 0 ? st.rows[0] : tb == 1 ? st.rows[1] : tb == 2 ? st.rows[2] : st.rows[3];
      const int key = keys[i];
      keys[i] = (key == ARX_KEY_NONE || key < 0 || key >= rows) ? ARX_KEY_NONE : ((tb << st.kb) | key);
      continue;
    }
    int s = 0;
#pragma unroll
    for (int q = 1; q < kMaxSites; ++q)
      if (q < st.nsites && i >= st.offs[q]) s = q;
    const int64_t j = i - st.offs[s];
    const int id = st.ids[s][j];
    const int key = id < 0 ? -1 : (st.cat_map[s] ? st.cat_map[s][id] : id);   // (id < 0: an empty pool slot)
    const int tb = st.table[s];
    const int64_t rows = tb == 0 ? st.rows[0] : tb == 1 ? st.rows[1] : tb == 2 ? st.rows[2] : st.rows[3];
    keys[i] = (key < 0 || key >= rows) ? ARX_KEY_NONE : ((tb << st.kb) | key);
    src[i] = st.row_base[s] + (int32_t)j;
    coef[i] = st.coef[s];
  }
}

// Sum coef*G[src] over sorted positions [q, pend) that carry `key`, stopping at
// the first different key.  Returns the number of positions consumed.
template <int LPR>
__device__ __forceinline__ int walk_piece(const uint32_t* __restrict__ sk,
                                          const uint32_t* __restrict__ spos,
                                          const int32_t* __restrict__ src,
                                          const float* __restrict__ coef,
                                          const float* __restrict__ G, int64_t ldg,
                                          const float* __restrict__ Gb, uint32_t key, int64_t q,
                                          int64_t pend, int col, bool colok, int lig, int gid,
                                          float4& acc_out, float& gb_out) {
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  float gb = 0.f;
  int consumed = 0;
  for (int64_t q0 = q; q0 < pend; q0 += LPR) {
    const int64_t p = q0 + lig;
    const bool same = (p < pend) && (sk[p] == key);
    unsigned long long bal = __ballot(same);
    unsigned long long bits = bal;
    if constexpr (LPR < 64) bits = (bal >> (gid * LPR)) & ((1ull << LPR) - 1ull);
    // length of the leading run of 'same' lanes
    int cnt = (~bits == 0ull) ? 64 : (int)__builtin_ctzll(~bits);
    if (cnt > LPR) cnt = LPR;
    int mysrc = 0;
    float mycoef = 0.f;
    if (lig < cnt) {
      const uint32_t i = spos ? spos[p] : (uint32_t)p;   // null: src/coef already in sorted order
      mysrc = src ? src[i] : (int32_t)i;
      mycoef = coef ? coef[i] : 1.f;
      if (Gb) gb = fmaf(mycoef, Gb[mysrc], gb);
    }
    int t = 0;
    for (; t + 2 <= cnt; t += 2) {
      const int s0 = __shfl(mysrc, t, LPR), s1 = __shfl(mysrc, t + 1, LPR);
      const float c0 = __shfl(mycoef, t, LPR), c1 = __shfl(mycoef, t + 1, LPR);
      if (colok) {
        const float4 v0 = *reinterpret_cast<const float4*>(G + (int64_t)s0 * ldg + col);
        const float4 v1 = *reinterpret_cast<const float4*>(G + (int64_t)s1 * ldg + col);
        a0 = f4_fma(c0, v0, a0);
        a1 = f4_fma(c1, v1, a1);
      }
    }
    if (t < cnt) {
      const int s0 = __shfl(mysrc, t, LPR);
      const float c0 = __shfl(mycoef, t, LPR);
      if (colok) a0 = f4_fma(c0, *reinterpret_cast<const float4*>(G + (int64_t)s0 * ldg + col), a0);
    }
    consumed += cnt;
    if (cnt < LPR) break;
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) gb += __shfl_xor(gb, o, LPR);
  acc_out = f4_add2(a0, a1);
  gb_out = gb;
  return consumed;
}

__device__ __forceinline__ void adagrad_row(float* __restrict__ E, float* __restrict__ acc,
                                            float* __restrict__ bias, float* __restrict__ bias_acc,
                                            int d, uint32_t row, int col, bool colok, int lig,
                                            float4 g, float gb, float lr, float gs) {
  // acc == nullptr: plain gradient descent (tf.train.GradientDescentOptimizer, seqModel.py:176)
  if (colok) {
    float4* wp = reinterpret_cast<float4*>(E + (int64_t)row * d + col);
    float4 w = *wp;
    g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs;
    if (acc) {
      float4* ap = reinterpret_cast<float4*>(acc + (int64_t)row * d + col);
      float4 a = *ap;
      a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
      w.x -= adagrad_delta(lr, g.x, a.x);
      w.y -= adagrad_delta(lr, g.y, a.y);
      w.z -= adagrad_delta(lr, g.z, a.z);
      w.w -= adagrad_delta(lr, g.w, a.w);
      *ap = a;
    } else {
      w.x -= lr * g.x; w.y -= lr * g.y; w.z -= lr * g.z; w.w -= lr * g.w;
    }
    *wp = w;
  }
  if (bias && lig == 0) {
    const float gg = gb * gs;
    if (bias_acc) {
      const float a = bias_acc[row] + gg * gg;
      bias_acc[row] = a;
      bias[row] -= adagrad_delta(lr, gg, a);
    } else {
      bias[row] -= lr * gg;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Window apply (replaces the position-per-sub-group passes): one WAVE owns 64 consecutive
// sorted positions (= one piece).  Keys, gradient-source rows and coefficients of the
// window are staged with three coalesced loads; run boundaries come from one ballot, so
// the per-run walk needs no further key loads and issues its gradient rows 8 at a time.
// The wave's 64/LPR sub-groups take the window's runs round-robin.
//   complete run (starts and ends inside the window)      -> Adagrad at once
//   open-left piece (run started in an earlier window)    -> partial row to scratch[slot]
//   head piece whose run continues past the window        -> partial row to scratch_h[slot]
// Multi-piece runs are finished
//   * with `cnt` (per-table-row counter, zero on entry and exit) and <= 17 pieces: by the
//     LAST arriving piece (ticket: every piece adds 1, the head adds 1 + kBig - T);
//   * otherwise by k_sparse_long (second launch, one workgroup per listed run): the
//     Zipf-hot rows with hundreds of pieces, summed by 32 sub-groups in parallel.
// Partials are always summed head first, then in ascending piece order => bit-reproducible.
#ifndef ARX_WIN_NB1
#define ARX_WIN_NB1 2
#endif
#ifndef ARX_WIN_RU1
#define ARX_WIN_RU1 8
#endif
constexpr int kBig = 1 << 30;
constexpr int kShortMaxAligned = 16;    // runs with more aligned pieces get a whole workgroup

template <int LPR, int WPW, bool MT, bool SGD, bool MERGE = false>
__global__ __launch_bounds__((WPW > 4 ? WPW : 4) * 64) void k_sparse_win(
    TableSet ts, int d, const uint32_t* __restrict__ sk,
    const uint32_t* __restrict__ spos, const int32_t* __restrict__ ssrc,
    const float* __restrict__ scoef, int64_t n_host, const int32_t* __restrict__ n_dev,
    uint32_t sentinel, const float* __restrict__ G,
    int64_t ldg, const float* __restrict__ Gb, const float* __restrict__ lr_dev,
    const float* __restrict__ gscale_dev, float* __restrict__ scratch,
    float* __restrict__ scratch_b, float* __restrict__ scratch_h, float* __restrict__ scratch_hb,
    int32_t* __restrict__ list_long, int32_t* __restrict__ list_short,
    int32_t* __restrict__ list_count, MergeOut mo) {
  constexpr int NSG = 64 / LPR;
  constexpr int NWV = (WPW > 4) ? WPW : 4;    // waves per workgroup
  // n_dev: entries that survived the radix sort's first pass (pads dropped); the grid is sized
  // for the padded capacity and surplus windows leave at once
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  __shared__ int s_src[NWV][64];
  __shared__ float s_coef[NWV][64];
  __shared__ uint32_t s_key[NWV][64];
  __shared__ int s_lead[NWV][66];              // window positions of the run leaders, then nvalid
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lig = lane % LPR;
  const int g = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  // WPW == 1: every wave owns a window (long runs: few leaders per window).
  // WPW  > 1: the workgroup's WPW waves share ONE window and split its leaders (one-hot
  //           ids: up to 64 single-row runs per window, a lone wave would walk them serially).
  const int64_t w0 = (WPW == 1) ? ((int64_t)blockIdx.x * 4 + wv) * 64 : (int64_t)blockIdx.x * 64;
  const int sgi = (WPW == 1) ? g : wv * NSG + g;          // this sub-group's first leader
  constexpr int kStride = (WPW == 1) ? NSG : WPW * NSG;   // leaders taken per round
  if (w0 >= n) return;                       // whole wave (no workgroup barrier below)
  const int64_t p = w0 + lane;
  const bool inb = p < n;
  const uint32_t key = inb ? sk[p] : 0xffffffffu;
  const uint32_t prev = (inb && p > 0) ? sk[p - 1] : 0xffffffffu;
  const uint32_t knext = (w0 + 64 < n) ? sk[w0 + 64] : 0xffffffffu;
  const bool valid = inb && key < sentinel;
  const bool head = valid && (p == 0 || prev != key);
  {
    int sv = 0;
    float cv = 0.f;
    if (inb) {          // (not `valid`: these loads then fly with the key loads)
      const uint32_t i = spos ? spos[p] : (uint32_t)p;
      sv = ssrc ? ssrc[i] : (int32_t)i;
      cv = scoef ? scoef[i] : 1.f;
    }
    s_src[wv][lane] = sv;
    s_coef[wv][lane] = cv;
    s_key[wv][lane] = key;
  }
  const unsigned long long V = __ballot(valid);
  const unsigned long long H = __ballot(head);
  const unsigned long long L = H | (V & 1ull);            // leaders: heads + the window's first position
  const int nvalid = __popcll(V);                          // valid positions are a prefix
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  // leader list: the j-th leader's window position at s_lead[j], its run ends at s_lead[j + 1]
  const int nlead = __popcll(L);
  if ((L >> lane) & 1ull) {
    const int r = __builtin_amdgcn_mbcnt_hi((uint32_t)(L >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)L, 0u));
    s_lead[wv][r] = lane;
  }
  if (lane == 0) s_lead[wv][nlead] = nvalid;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // Leaders are taken NB at a time: everything a run needs from HBM that does not depend on its
  // gradient sum -- table row, slot row, bias cells, its first gradient row and the bias
  // gradients -- is requested for all NB runs before the first one is summed, so a sub-group
  // pays one HBM round trip per NB runs instead of three per run (rows -> bias gradient ->
  // bias cells used to be dependent trips).
  constexpr int NB = (WPW > 1) ? 4 : ARX_WIN_NB1;   // one-hot windows: all 4 leaders of a sub-group at once
  constexpr int RU = (WPW > 1) ? 4 : ARX_WIN_RU1;   // further gradient rows of a run in flight
  for (int j0 = sgi; j0 < nlead; j0 += kStride * NB) {
    int li[NB], lrows[NB];
    bool lhead[NB], lcomp[NB];
    float4 wrow[NB], arow[NB], g0[NB];
    float gbv[NB], bv[NB], bav[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int jl = j0 + j * kStride;
      const bool act = jl < nlead;
      const int i = act ? s_lead[wv][jl] : 0;
      const int e = act ? s_lead[wv][jl + 1] : 1;
      const uint32_t rkey = s_key[wv][i];
      const bool is_head = (H >> i) & 1ull;
      const bool continues = (e == 64) && (knext == rkey);
      li[j] = act ? i : -1;
      lrows[j] = e - i;
      lhead[j] = is_head;
      lcomp[j] = act && is_head && !continues;
      wrow[j] = arow[j] = g0[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      gbv[j] = bv[j] = bav[j] = 0.f;
      if (act) {
        const TabRow T = tab_of<MT>(ts, rkey);
        if (colok) {
          g0[j] = *reinterpret_cast<const float4*>(G + (int64_t)s_src[wv][i] * ldg + col);
          if (!MERGE && lcomp[j] && T.E) {      // (T.E null: a virtual table -- entity ids of a riding bag table)
            wrow[j] = row_load(T.E + (int64_t)T.row * d + col);
            if (!SGD) arow[j] = row_load(T.acc + (int64_t)T.row * d + col);
          }
        }
        if (Gb && lig < lrows[j]) gbv[j] = Gb[s_src[wv][i + lig]];
        if (!MERGE && lcomp[j] && T.bias && lig == 0) {
          bv[j] = T.bias[T.row];
          if (!SGD) bav[j] = T.bias_acc[T.row];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (li[j] < 0) continue;
      const int i = li[j];
      const int rows = lrows[j];
      const uint32_t rkey = s_key[wv][i];
      const TabRow T = tab_of<MT>(ts, rkey);
      const uint32_t rrow = T.row;
      const bool is_head = lhead[j];
      // same order as a plain walk over the run: row 0 first, then 8 rows in flight at a time
      float4 a = f4_fma(s_coef[wv][i], g0[j], make_float4(0.f, 0.f, 0.f, 0.f));
      for (int t = 1; t < rows; t += RU) {
        float4 v[RU];
        float c[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const bool ok = t + u < rows;
          const int idx = i + (ok ? t + u : 0);
          const int s = s_src[wv][idx];
          c[u] = ok ? s_coef[wv][idx] : 0.f;
          v[u] = (ok && colok) ? *reinterpret_cast<const float4*>(G + (int64_t)s * ldg + col)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) a = f4_fma(c[u], v[u], a);
      }
      float gb = 0.f;
      if (Gb) {
        if (lig < rows) gb = fmaf(s_coef[wv][i + lig], gbv[j], 0.f);
        for (int t = lig + LPR; t < rows; t += LPR) gb = fmaf(s_coef[wv][i + t], Gb[s_src[wv][i + t]], gb);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) gb += __shfl_xor(gb, o, LPR);
      }
      if (MERGE && lcomp[j]) {
        merge_row(mo, d, rkey, w0 + i, col, colok, lig, a, gb);
      } else if (lcomp[j]) {
        if (!MERGE && merge_side(mo, rkey)) merge_row(mo, d, rrow, w0 + i, col, colok, lig, a, gb);
        if (colok && T.E) {
          float4 gg = make_float4(a.x * gs, a.y * gs, a.z * gs, a.w * gs);
          float4 w4 = wrow[j];
          if (!SGD) {
            float4 a4 = arow[j];
            a4.x += gg.x * gg.x; a4.y += gg.y * gg.y; a4.z += gg.z * gg.z; a4.w += gg.w * gg.w;
            w4.x -= adagrad_delta(lr, gg.x, a4.x);
            w4.y -= adagrad_delta(lr, gg.y, a4.y);
            w4.z -= adagrad_delta(lr, gg.z, a4.z);
            w4.w -= adagrad_delta(lr, gg.w, a4.w);
            row_store(T.acc + (int64_t)rrow * d + col, a4);
          } else {                                   // gradient descent
            w4.x -= lr * gg.x; w4.y -= lr * gg.y; w4.z -= lr * gg.z; w4.w -= lr * gg.w;
          }
          row_store(T.E + (int64_t)rrow * d + col, w4);
        }
        if (T.bias && lig == 0) {
          const float gg = gb * gs;
          if (!SGD) {
            const float ba = bav[j] + gg * gg;
            T.bias_acc[rrow] = ba;
            T.bias[rrow] = bv[j] - adagrad_delta(lr, gg, ba);
          } else {
            T.bias[rrow] = bv[j] - lr * gg;
          }
        }
      } else {
        // ---- piece of a multi-piece run ----
        const int64_t slot = w0 / 64;
        const int64_t q = w0 + i;
        float* prow = (is_head ? scratch_h : scratch) + slot * (int64_t)d;
        if (colok) *reinterpret_cast<float4*>(prow + col) = a;
        if (lig == 0) (is_head ? scratch_hb : scratch_b)[slot] = gb;
        if (is_head) {
          // aligned pieces that follow: probe 16 window starts per round trip; runs with at
          // most kShortMaxAligned of them go to the short list (one sub-group each in
          // k_sparse_finish), Zipf-hot ones to the long list (one workgroup each)
          const int64_t first = w0 + 64;
          int na = 0;
          bool stop = false;
          while (!stop) {
            int mc = 0;
            bool all = true;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
              const int64_t aa = first + (int64_t)(na + jj) * 64;
              const bool same = (aa < n) && (sk[aa] == rkey);
              if (all && same) ++mc; else all = false;
            }
            na += mc;
            stop = (mc < 16) || (na > kShortMaxAligned);
          }
          if (lig == 0) {
            if (na > kShortMaxAligned) {
              const int sl = atomicAdd(&list_count[0], 1);
              list_long[sl] = (int32_t)q;
            } else {
              const int sl = atomicAdd(&list_count[1], 1);
              list_short[2 * sl] = (int32_t)q;
              list_short[2 * sl + 1] = na;
            }
          }
        }
      }
    }
  }
}

// Second launch of the window apply: finishes the multi-piece runs.
//   blocks [0, nlong_blocks): one workgroup (1024 threads = 1024/LPR sub-groups) per LONG
//     run: the sub-groups sum the aligned partials in parallel (4 rows in flight each) and
//     combine through LDS in a fixed order -- a Zipf-hot token with 2000 pieces costs ~16
//     dependent round trips instead of 2000;
//   remaining blocks: one SUB-GROUP per SHORT run (head partial + <= 16 aligned partials,
//     two round trips), grid-stride over the short list.
template <int LPR, bool MT, bool MERGE = false>
__global__ __launch_bounds__(1024) void k_sparse_finish(
    TableSet ts, int d, const uint32_t* __restrict__ sk, int64_t n_host,
    const int32_t* __restrict__ n_dev, const float* __restrict__ lr_dev, const float* __restrict__ gscale_dev,
    const float* __restrict__ scratch, const float* __restrict__ scratch_b,
    const float* __restrict__ scratch_h, const float* __restrict__ scratch_hb,
    const int32_t* __restrict__ list_long, const int32_t* __restrict__ list_short,
    const int32_t* __restrict__ list_count, int nlong_blocks, MergeOut mo) {
  constexpr int NSG = 1024 / LPR;
  __shared__ __attribute__((aligned(16))) float sh[NSG][LPR * 4];
  __shared__ float shb[NSG];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  const int lig = threadIdx.x % LPR;
  const int sg = threadIdx.x / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  if ((int)blockIdx.x >= nlong_blocks) {
    // ---- short runs: one sub-group each ----
    const int count = list_count[1];
    const int64_t nsg_total = (int64_t)(gridDim.x - nlong_blocks) * NSG;
    for (int64_t it = (int64_t)(blockIdx.x - nlong_blocks) * NSG + sg; it < count; it += nsg_total) {
      const int64_t h = list_short[2 * it];
      const int na = list_short[2 * it + 1];
      const uint32_t key = sk[h];
      const int64_t hslot = h / 64;
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      if (colok) tot = *reinterpret_cast<const float4*>(scratch_h + hslot * (int64_t)d + col);
      float tb = scratch_hb[hslot];
      for (int j0 = 0; j0 < na; j0 += 8) {
        float4 v[8];
        float vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t sl = hslot + 1 + ((j0 + u < na) ? j0 + u : 0);
          v[u] = colok ? *reinterpret_cast<const float4*>(scratch + sl * (int64_t)d + col)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
          vb[u] = scratch_b[sl];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < na) {
            tot = f4_add2(tot, v[u]);
            tb += vb[u];
          }
      }
      if (MERGE) {
        merge_row(mo, d, key, h, col, colok, lig, tot, tb);
      } else {
        const TabRow T = tab_of<MT>(ts, key);
        if (merge_side(mo, key)) merge_row(mo, d, T.row, h, col, colok, lig, tot, tb);
        if (T.E) adagrad_row(T.E, T.acc, T.bias, T.bias_acc, d, T.row, col, colok, lig, tot, tb, lr, gs);
      }
    }
    return;
  }
  // ---- long runs: one workgroup each ----
  const int count = list_count[0];
  for (int it = blockIdx.x; it < count; it += nlong_blocks) {
    const int64_t h = list_long[it];
    const uint32_t key = sk[h];
    const int64_t hslot = h / 64;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    float tb = 0.f;
    bool more = true;
    for (int64_t j0 = sg; more; j0 += (int64_t)NSG * 4) {
      float4 v[4];
      float vb[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t aa = (hslot + 1 + j0 + (int64_t)u * NSG) * 64;
        const bool in = aa < n;
        ok[u] = in && sk[aa] == key;
        const int64_t sl = in ? aa / 64 : hslot;
        v[u] = colok ? *reinterpret_cast<const float4*>(scratch + sl * (int64_t)d + col)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
        vb[u] = scratch_b[sl];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          tot = f4_add2(tot, v[u]);
          tb += vb[u];
        } else {
          more = false;
        }
      }
    }
    __syncthreads();
    if (colok) *reinterpret_cast<float4*>(&sh[sg][col]) = tot;
    if (lig == 0) shb[sg] = tb;
    __syncthreads();
    if (sg == 0) {
      float4 t2 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (colok) t2 = *reinterpret_cast<const float4*>(scratch_h + hslot * (int64_t)d + col);
      float t2b = scratch_hb[hslot];
      for (int k = 0; k < NSG; ++k) {
        if (colok) t2 = f4_add2(t2, *reinterpret_cast<const float4*>(&sh[k][col]));
        t2b += shb[k];
      }
      if (MERGE) {
        merge_row(mo, d, key, h, col, colok, lig, t2, t2b);
      } else {
        const TabRow T = tab_of<MT>(ts, key);
        if (merge_side(mo, key)) merge_row(mo, d, T.row, h, col, colok, lig, t2, t2b);
        if (T.E) adagrad_row(T.E, T.acc, T.bias, T.bias_acc, d, T.row, col, colok, lig, t2, t2b, lr, gs);
      }
    }
  }
}

// Small-n variant without the second launch: multi-piece runs are finished by their LAST
// arriving piece (ticket on a per-table-row counter).  Every piece of such a run writes
// its partial row, releases (agent scope), and adds 1 to cnt[key]; the head adds
// 1 + (kBig - T) with T = pieces of the run (binary search over the aligned positions), so
// the arrival that makes the counter reach kBig knows all T partials are published: it
// acquires, sums them in piece order (fixed order => deterministic) and applies Adagrad.

template <int LPR, bool MT>
__global__ __launch_bounds__(256) void k_sparse_onepass(
    TableSet ts, int d, const uint32_t* __restrict__ sk,
    const uint32_t* __restrict__ spos /* null: ssrc/scoef already sorted */,
    const int32_t* __restrict__ ssrc, const float* __restrict__ scoef, int64_t n, uint32_t sentinel,
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ Gb,
    const float* __restrict__ lr_dev, const float* __restrict__ gscale_dev,
    float* __restrict__ scratch, float* __restrict__ scratch_b, float* __restrict__ scratch_h,
    float* __restrict__ scratch_hb) {
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t q = wave * GPW + gid;
  if (q >= n) return;
  const uint32_t key = sk[q];
  if (key >= sentinel) return;
  const bool head = (q == 0) || (sk[q - 1] != key);
  const bool aligned = (q % kPiece) == 0;
  if (!head && !aligned) return;
  int64_t pend = (q / kPiece + 1) * kPiece;
  if (pend > n) pend = n;
  float4 g;
  float gb;
  const int consumed = walk_piece<LPR>(sk, spos, ssrc, scoef, G, ldg, Gb, key, q, pend, col, colok,
                                       lig, gid, g, gb);
  const int64_t e = q + consumed;
  const bool continues = (e == pend) && (e < n) && (sk[e] == key);
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  const TabRow T = tab_of<MT>(ts, key);
  int32_t* __restrict__ cnt = T.cnt;
  if (head && !continues) {
    adagrad_row(T.E, T.acc, T.bias, T.bias_acc, d, T.row, col, colok, lig, g, gb, lr, gs);
    return;
  }
  // ---- piece of a multi-piece run ----
  const int64_t slot = q / kPiece;
  float* prow = (head ? scratch_h : scratch) + slot * (int64_t)d;
  if (colok) *reinterpret_cast<float4*>(prow + col) = g;
  if (lig == 0) (head ? scratch_hb : scratch_b)[slot] = gb;
  __threadfence();                                   // release the partial (agent scope)
  int inc = 1;
  if (head) {
    const int64_t first = (q / kPiece + 1) * kPiece;
    int64_t lo = 0, hi = (n - 1 - first) / kPiece;   // sk[first + lo*kPiece] == key (continues)
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (sk[first + mid * kPiece] == key) lo = mid; else hi = mid - 1;
    }
    inc = 1 + (kBig - (int)(lo + 2));                // T = head piece + (lo + 1) aligned pieces
  }
  int old = 0;
  if (lig == 0) old = atomicAdd(&cnt[T.row], inc);
  old = __shfl(old, 0, LPR);
  if (old + inc != kBig) return;
  __threadfence();                                   // acquire: drop stale L1 lines
  int64_t h = q;
  if (!head) {                                       // lower bound of key in sk[0..q]
    int64_t lo = 0, hi = q;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sk[mid] < key) lo = mid + 1; else hi = mid;
    }
    h = lo;
  }
  const int64_t ch = h / kPiece;
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
  if (colok) tot = *reinterpret_cast<const float4*>(scratch_h + ch * (int64_t)d + col);
  float tb = scratch_hb[ch];
  // 8 partials per round trip (a serial walk is one ~1.5 us L2 round trip per piece: a hot
  // row with hundreds of pieces stalled the whole launch); rows past the run's end are
  // loaded speculatively (in-bounds scratch) and dropped.  Fixed order => deterministic.
  bool more = true;
  for (int64_t a = (ch + 1) * kPiece; more && a < n; a += 8 * kPiece) {
    float4 v[8];
    float vb[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t aa = a + (int64_t)u * kPiece;
      const bool in = aa < n;
      ok[u] = in && sk[aa] == key;
      const int64_t sl = in ? aa / kPiece : ch;
      v[u] = colok ? *reinterpret_cast<const float4*>(scratch + sl * (int64_t)d + col)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      vb[u] = scratch_b[sl];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (ok[u]) {
        tot = f4_add2(tot, v[u]);
        tb += vb[u];
      } else {
        more = false;
      }
    }
  }
  adagrad_row(T.E, T.acc, T.bias, T.bias_acc, d, T.row, col, colok, lig, tot, tb, lr, gs);
  if (lig == 0) cnt[T.row] = 0;
}

__global__ void k_adagrad_dense(float* __restrict__ w, float* __restrict__ acc,
                                const float* __restrict__ g, int64_t n,
                                const float* __restrict__ lr_dev,
                                const float* __restrict__ gscale_dev) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  for (; i < n; i += stride) {
    const float gg = g[i] * gs;
    if (acc) {
      const float a = acc[i] + gg * gg;
      acc[i] = a;
      w[i] -= lr * gg / sqrtf(a);
    } else {
      w[i] -= lr * gg;                             // gradient descent
    }
  }
}

// Adagrad over the rows of a table whose dense gradient rows are mostly zero (arx.dist.ShardedHMFRepTokens: the merged,
// all-reduced gradient of a replicated token table -- ~30 % of 100 k rows touched per step): a sub-group reads its
// gradient row, skips the row when every element is zero (what the dense step would do to it: acc += 0, w -= 0), else
// applies arx_adagrad_dense's arithmetic and ZEROES the gradient row it consumed -- the table is left all zero for the
// next step's accumulation, no fill pass.  bias / bias_acc / Gb (nullable): the same per row.
template <int LPR>
__global__ __launch_bounds__(256) void k_adagrad_rows_nonzero(float* __restrict__ W, float* __restrict__ acc,
                                                              float* __restrict__ bias, float* __restrict__ bias_acc,
                                                              float* __restrict__ G, float* __restrict__ Gb,
                                                              int64_t rows, int d, const float* __restrict__ lr_dev) {
  constexpr int NSG = 256 / LPR;
  const int lig = threadIdx.x % LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const float lr = *lr_dev;
  const int64_t stride = (int64_t)gridDim.x * NSG;
  for (int64_t r0 = (int64_t)blockIdx.x * NSG + threadIdx.x / LPR; r0 < rows; r0 += 4 * stride) {
    float4 g[4];
    float gb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {          // four rows in flight per sub-group
      const int64_t r = r0 + u * stride;
      g[u] = (r < rows && colok) ? *reinterpret_cast<const float4*>(G + r * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      gb[u] = (r < rows && Gb && lig == 0) ? Gb[r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + u * stride;
      // (x != 0 is true for NaN: a poisoned gradient row is applied, not skipped)
      bool nz = (g[u].x != 0.f) | (g[u].y != 0.f) | (g[u].z != 0.f) | (g[u].w != 0.f) | (gb[u] != 0.f);
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) nz = nz | (bool)__shfl_xor((int)nz, o, LPR);
      if (r >= rows || !nz) continue;
      if (colok) {
        float4 w = *reinterpret_cast<const float4*>(W + r * d + col);
        if (acc) {
          float4 a = *reinterpret_cast<const float4*>(acc + r * d + col);
          a.x += g[u].x * g[u].x; a.y += g[u].y * g[u].y; a.z += g[u].z * g[u].z; a.w += g[u].w * g[u].w;
          *reinterpret_cast<float4*>(acc + r * d + col) = a;
          w.x -= lr * g[u].x / sqrtf(a.x); w.y -= lr * g[u].y / sqrtf(a.y);
          w.z -= lr * g[u].z / sqrtf(a.z); w.w -= lr * g[u].w / sqrtf(a.w);
        } else {
          w.x -= lr * g[u].x; w.y -= lr * g[u].y; w.z -= lr * g[u].z; w.w -= lr * g[u].w;
        }
        *reinterpret_cast<float4*>(W + r * d + col) = w;
        *reinterpret_cast<float4*>(G + r * d + col) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (bias && lig == 0) {
        if (bias_acc) {
          const float a = bias_acc[r] + gb[u] * gb[u];
          bias_acc[r] = a;
          bias[r] -= lr * gb[u] / sqrtf(a);
        } else {
          bias[r] -= lr * gb[u];
        }
        Gb[r] = 0.f;
      }
    }
  }
}

// Up to 8 dense parameters in one launch (the LSTM weights / biases / input projections of a step:
// 2-6 tensors of 256 B - 128 KB, each a launch of its own otherwise).  Blocks [blk_end[t-1], blk_end[t])
// own tensor t.
struct DenseSet {
  float* w[8];
  float* acc[8];
  const float* g[8];
  int64_t n[8];
  int blk_end[8];
  int count;
};
__global__ __launch_bounds__(256) void k_adagrad_dense_multi(DenseSet ds, const float* __restrict__ lr_dev,
                                                             const float* __restrict__ gscale_dev) {
  int t = 0;
  while ((int)blockIdx.x >= ds.blk_end[t]) ++t;
  const int b0 = t ? ds.blk_end[t - 1] : 0;
  float* __restrict__ w = ds.w[t];
  float* __restrict__ acc = ds.acc[t];
  const float* __restrict__ g = ds.g[t];
  const int64_t n = ds.n[t];
  const int64_t stride = (int64_t)(ds.blk_end[t] - b0) * blockDim.x;
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  for (int64_t i = ((int)blockIdx.x - b0) * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = g[i] * gs;
    if (acc) {
      const float a = acc[i] + gg * gg;
      acc[i] = a;
      w[i] -= lr * gg / sqrtf(a);
    } else {
      w[i] -= lr * gg;
    }
  }
}

// Deterministic squared norm in ONE launch: per-block partials in fixed slots, then the block
// that arrives last (ticket) adds them up in slot order.  float4 loads, 4 in flight per thread.
static_assert(64 + 512 * 4 <= kReduceScratchBytes, "reduce scratch too small");
constexpr int kNormBlocks = 512;     // fixed slots of block partials (combined in slot order: deterministic)
// (round 4: block partials and the arrival ticket live in the CALLER's reduce scratch -- common.h reduce_scratch_* --
// so that two launches on different streams cannot meet in library-owned memory)

template <bool VEC>
__global__ __launch_bounds__(1024) void k_sq_norm(const float* __restrict__ x, int64_t n, int d,
                                                 const float* __restrict__ row_scale,
                                                 float* __restrict__ part, unsigned int* ticket,
                                                 float* __restrict__ out) {
  __shared__ float sh[16];
  __shared__ bool s_last;
  float s = 0.f;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {              // n % 4 == 0, d % 4 == 0, 16-byte aligned, n < 2^31
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const uint32_t n4 = (uint32_t)(n >> 2), d4 = (uint32_t)(d >> 2);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    uint32_t i = (uint32_t)tid;
    const uint32_t st = (uint32_t)stride;
    for (; i + 3 * st < n4; i += 4 * st) {
      const float4 v0 = x4[i], v1 = x4[i + st], v2 = x4[i + 2 * st], v3 = x4[i + 3 * st];
      const float w0 = row_scale ? row_scale[i / d4] : 1.f;
      const float w1 = row_scale ? row_scale[(i + st) / d4] : 1.f;
      const float w2 = row_scale ? row_scale[(i + 2 * st) / d4] : 1.f;
      const float w3 = row_scale ? row_scale[(i + 3 * st) / d4] : 1.f;
      a0 += w0 * (v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w);
      a1 += w1 * (v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w);
      a2 += w2 * (v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w);
      a3 += w3 * (v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w);
    }
    for (; i < n4; i += st) {
      const float4 v = x4[i];
      a0 += (row_scale ? row_scale[i / d4] : 1.f) * (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    }
    s = (a0 + a1) + (a2 + a3);
  } else {
    for (int64_t i = tid; i < n; i += stride) {
      const float v = x[i];
      s += (row_scale ? row_scale[i / d] : 1.f) * v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  // The partials travel as agent-scope atomics (they bypass the per-XCD L2), so no cache-wide
  // release/acquire fence is needed: a __threadfence() per block costs ~0.2 us of L2 write-back
  // scan each, 13 us over 512 blocks (measured).  The store is complete (vmcnt 0) before the
  // ticket is taken.
  if (threadIdx.x == 0) {
    float bs = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) bs += sh[k];
    __hip_atomic_store(&part[blockIdx.x], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64)
      t += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x == 0) {
      *out += t;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The same reduction over up to 8 tensors in one launch (the clip norm of an LSTM step sums 7
// gradient tensors: one launch instead of seven).  Blocks [blk_end[t-1], blk_end[t]) own tensor t.
struct NormSet {
  const float* x[8];
  const float* rs[8];
  int64_t n[8];
  int d[8];
  int vec[8];
  int blk_end[8];
  int count;
};

// init: out = sum instead of out += sum; coef (nullable): the last block also forms the
// clip_by_global_norm coefficient max_norm / max(||g||, max_norm) and ||g|| from the final sum
// (arx_sq_norm_clip_multi: the fill / norm / coefficient chain of a step in one launch).
__global__ __launch_bounds__(1024) void k_sq_norm_multi(NormSet ns, float* __restrict__ part,
                                                        unsigned int* ticket, float* __restrict__ out,
                                                        int init, float max_norm, float* __restrict__ coef,
                                                        float* __restrict__ gnorm) {
  __shared__ float sh[16];
  __shared__ bool s_last;
  int t = 0;
  while ((int)blockIdx.x >= ns.blk_end[t]) ++t;
  const int b0 = t ? ns.blk_end[t - 1] : 0;
  const int64_t lb = (int)blockIdx.x - b0, nb = ns.blk_end[t] - b0;
  const float* __restrict__ x = ns.x[t];
  const float* __restrict__ row_scale = ns.rs[t];
  const int64_t n = ns.n[t];
  const int d = ns.d[t];
  float s = 0.f;
  const int64_t tid = lb * blockDim.x + threadIdx.x, stride = nb * blockDim.x;
  if (ns.vec[t]) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const uint32_t n4 = (uint32_t)(n >> 2), d4 = (uint32_t)(d >> 2), st = (uint32_t)stride;
    float a0 = 0.f, a1 = 0.f;
    uint32_t i = (uint32_t)tid;
    for (; i + st < n4; i += 2 * st) {
      const float4 v0 = x4[i], v1 = x4[i + st];
      const float w0 = row_scale ? row_scale[i / d4] : 1.f, w1 = row_scale ? row_scale[(i + st) / d4] : 1.f;
      a0 += w0 * (v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w);
      a1 += w1 * (v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w);
    }
    for (; i < n4; i += st) {
      const float4 v = x4[i];
      a0 += (row_scale ? row_scale[i / d4] : 1.f) * (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    }
    s = a0 + a1;
  } else {
    for (int64_t i = tid; i < n; i += stride) {
      const float v = x[i];
      s += (row_scale ? row_scale[i / d] : 1.f) * v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float bs = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) bs += sh[k];
    __hip_atomic_store(&part[blockIdx.x], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < 64) {
    float tt = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64)
      tt += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o, 64);
    if (threadIdx.x == 0) {
      const float total = init ? tt : *out + tt;
      *out = total;
      if (coef) {
        const float nrm = sqrtf(total);
        if (gnorm) *gnorm = nrm;
        *coef = max_norm / fmaxf(nrm, max_norm);
      }
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void k_clip_coef(const float* __restrict__ sq, float max_norm, float* __restrict__ coef,
                            float* __restrict__ gnorm) {
  const float nrm = sqrtf(*sq);
  if (gnorm) *gnorm = nrm;
  *coef = max_norm / fmaxf(nrm, max_norm);
}


static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Squared norm of a gradient AFTER merging contributions by table row:
//   out += sum_t sum_rows || sum_{contributions c of the row} coef_c * X_t[src_c, :] ||^2
// over sorted contributions (sk ascending; entries >= sentinel are pads).  This is the norm
// tf.clip_by_global_norm sees for the dense gradient a matmul'd attribute table receives in
// each unrolled step (seqModel.py:180; embed_attribute.py:171,188: innerp = E . u^T is taken
// over the WHOLE table and gathered afterwards, so pool items that share a row are summed
// before the norm).  One sub-group of GS lanes per run head; Xb is the d = 1 bias analogue.

template <int GS>
__global__ __launch_bounds__(256) void k_merged_sq_norm(
    const uint32_t* __restrict__ sk, const int32_t* __restrict__ ssrc, const float* __restrict__ scoef,
    int64_t n_host, const int32_t* __restrict__ n_dev, uint32_t sentinel, const float* __restrict__ X,
    int64_t ldx, int d, int L, int64_t step_stride, const float* __restrict__ Xb, int Lb,
    int64_t stepb_stride, float* __restrict__ bpart, unsigned int* ticket, float* __restrict__ out) {
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
  const int lane = threadIdx.x % GS;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) / GS;
  float tot = 0.f;
  if (p < n) {
    const uint32_t key = sk[p];
    if (key < sentinel && (p == 0 || sk[p - 1] != key)) {
      int64_t q1 = p + 1;
      while (q1 < n && sk[q1] == key) ++q1;
      for (int t = 0; t < L; ++t) {
        const float* Xt = X + (int64_t)t * step_stride;
        for (int e = lane; e < d; e += GS) {
          float a = 0.f;
          for (int64_t q = p; q < q1; ++q) a = fmaf(scoef[q], Xt[(int64_t)ssrc[q] * ldx + e], a);
          tot = fmaf(a, a, tot);
        }
      }
      for (int t = lane; t < Lb; t += GS) {
        const float* Xt = Xb + (int64_t)t * stepb_stride;
        float a = 0.f;
        for (int64_t q = p; q < q1; ++q) a = fmaf(scoef[q], Xt[ssrc[q]], a);
        tot = fmaf(a, a, tot);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  __shared__ float part[4];
  __shared__ bool s_last;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = tot;
  __syncthreads();
  // fixed-slot block partials, summed in slot order by the block that arrives last (same scheme
  // as k_sq_norm): no float atomics, so the clip norm is bit-reproducible
  if (threadIdx.x == 0) {
    const float v = (part[0] + part[1]) + (part[2] + part[3]);
    __hip_atomic_store(&bpart[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64)
      t += __hip_atomic_load(&bpart[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x == 0) {
      *out += t;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Stage 1b of arx_sparse_adagrad_bags: the bag of every DISTINCT entity of the step, in padded
// slots: sorted entity position p owns out[p * max_len .. + max_len); a run head writes its bag's
// tokens (key = table row, src = p = row of the merged gradient Gu), every other slot holds
// ARX_KEY_NONE, which the token sort's first pass drops.
__global__ __launch_bounds__(256) void k_bag_expand_heads(
    const uint32_t* __restrict__ sk, int64_t n_host, const int32_t* __restrict__ n_dev,
    uint32_t sentinel, int ent_kb, uint32_t ent_tag, const int32_t* __restrict__ vals,
    const int32_t* __restrict__ starts, const int32_t* __restrict__ lens, int max_len,
    int64_t table_rows, int32_t* __restrict__ tkeys, int32_t* __restrict__ tsrc) {
  // sk may be the sorted key list of a pass over several tables: entity keys carry ent_tag above
  // their ent_kb row bits, every other key is skipped
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
  const int64_t total = n_host * (int64_t)max_len;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = q / max_len;
    const int j = (int)(q - p * max_len);
    int32_t key = ARX_KEY_NONE;
    if (p < n) {
      const uint32_t ke = sk[p];
      const uint32_t e = ke & ((1u << ent_kb) - 1u);
      if (ke < sentinel && (ke >> ent_kb) == ent_tag && (p == 0 || sk[p - 1] != ke) && j < lens[e]) {
        const int32_t t = vals[(int64_t)starts[e] + j];
        key = (t < 0 || t >= table_rows) ? ARX_KEY_NONE : t;
      }
    }
    tkeys[q] = key;
    tsrc[q] = (int32_t)p;
  }
}

// Compacted bag expansion (round 3): the padded slot layout above made the token sort's first pass read
// lookups x max_len slots at ~13 % live (1.1 M at C3).  k_head_len_scan: exclusive scan of the bag
// lengths of the DISTINCT entities over the sorted entity list (one workgroup, thread = contiguous
// chunk; positions follow the sorted order, so the token list -- and with it the summation order of
// every token run -- stays deterministic); k_bag_expand_compact writes the tokens densely.
__global__ __launch_bounds__(1024) void k_head_len_scan(
    const uint32_t* __restrict__ sk, int64_t n_host, const int32_t* __restrict__ n_dev, uint32_t sentinel,
    int ent_kb, uint32_t ent_tag, const int32_t* __restrict__ lens, int max_len, int32_t* __restrict__ hoff,
    int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  constexpr int BT = 32;                                 // positions per batch: their loads fly together
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
  const int64_t per = (n + 1023) / 1024;
  const int64_t p0 = tid * per, p1 = min(n, p0 + per);
  // pass 1: bag length of every run head of the entity table (0 elsewhere) -> hoff, chunk sum
  int local = 0;
  for (int64_t c0 = p0; c0 < p1; c0 += BT) {
    uint32_t k[BT + 1];
    k[0] = c0 > 0 ? sk[c0 - 1] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < BT; ++u) k[u + 1] = c0 + u < p1 ? sk[c0 + u] : 0xffffffffu;
    int l[BT];
#pragma unroll
    for (int u = 0; u < BT; ++u) {
      const uint32_t ke = k[u + 1];
      const bool h = c0 + u < p1 && ke < sentinel && (ke >> ent_kb) == ent_tag && (c0 + u == 0 || k[u] != ke);
      l[u] = h ? lens[ke & ((1u << ent_kb) - 1u)] : 0;
    }
#pragma unroll
    for (int u = 0; u < BT; ++u)
      if (c0 + u < p1) {
        const int v = min(l[u], max_len);
        hoff[c0 + u] = v;
        local += v;
      }
  }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int run = incl - local;
  for (int q = 0; q < wv; ++q) run += wsum[q];
  if (tid == 1023) *total = run + local;
  // pass 2: lengths -> exclusive offsets (the thread re-reads its own stores)
  for (int64_t c0 = p0; c0 < p1; c0 += BT) {
    int l[BT];
#pragma unroll
    for (int u = 0; u < BT; ++u) l[u] = c0 + u < p1 ? hoff[c0 + u] : 0;
#pragma unroll
    for (int u = 0; u < BT; ++u)
      if (c0 + u < p1) {
        hoff[c0 + u] = run;
        run += l[u];
      }
  }
}

__global__ __launch_bounds__(256) void k_bag_expand_compact(
    const uint32_t* __restrict__ sk, int64_t n_host, const int32_t* __restrict__ n_dev, uint32_t sentinel,
    int ent_kb, uint32_t ent_tag, const int32_t* __restrict__ vals, const int32_t* __restrict__ starts,
    const int32_t* __restrict__ lens, int max_len, int64_t table_rows, const int32_t* __restrict__ hoff,
    int32_t* __restrict__ tkeys, int32_t* __restrict__ tsrc) {
  __builtin_amdgcn_s_setprio(3);      // (sort-branch kernel: radix_sort.hip, "wave priority")
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
  const int64_t total = n * (int64_t)max_len;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = q / max_len;
    const int j = (int)(q - p * max_len);
    const uint32_t ke = sk[p];
    const uint32_t e = ke & ((1u << ent_kb) - 1u);
    if (ke < sentinel && (ke >> ent_kb) == ent_tag && (p == 0 || sk[p - 1] != ke) && j < lens[e]) {
      const int32_t t = vals[(int64_t)starts[e] + j];
      const int64_t slot = (int64_t)hoff[p] + j;
      tkeys[slot] = (t < 0 || t >= table_rows) ? ARX_KEY_NONE : t;
      tsrc[slot] = (int32_t)p;
    }
  }
}

struct SparseWs {
  size_t off_keys_tmp, off_keys_out, off_pos_in, off_pos_out, off_list, off_count, off_scratch,
      off_scratch_b, off_scratch_h, off_scratch_hb, off_ssrc, off_scoef, off_hist, off_runs, total;
};

static int sparse_ws_layout(int64_t n, int d, SparseWs* w) {
  const size_t ni = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
  const size_t pieces = (size_t)(n / kPiece + 2);
  size_t o = 0;
  w->off_keys_tmp = o; o += ni;
  w->off_keys_out = o; o += ni;
  w->off_pos_in = o; o += ni;
  w->off_pos_out = o; o += ni;
  w->off_list = o; o += align_up(pieces * 4 * 3, 256);   // long list + (position, pieces) pairs of the short list
  w->off_count = o; o += 2048;
  w->off_scratch = o; o += align_up(pieces * (size_t)d * 4, 256);
  w->off_scratch_b = o; o += align_up(pieces * 4, 256);
  w->off_scratch_h = o; o += align_up(pieces * (size_t)d * 4, 256);
  w->off_scratch_hb = o; o += align_up(pieces * 4, 256);
  w->off_ssrc = o; o += ni;
  w->off_scoef = o; o += ni;
  w->off_hist = o; o += align_up(radix_sort_hist_bytes(), 256);
  w->off_runs = o; o += run_lists_bytes(n > 0 ? n : 1, d);      // run records of the run-centric apply (group.hip)
  w->total = o;
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

#define ARX_DISPATCH_LPR(lpr, CALL)                   \
  switch (lpr) {                                      \
    case 1: { constexpr int LPR = 1; CALL; } break;   \
    case 2: { constexpr int LPR = 2; CALL; } break;   \
    case 4: { constexpr int LPR = 4; CALL; } break;   \
    case 8: { constexpr int LPR = 8; CALL; } break;   \
    case 16: { constexpr int LPR = 16; CALL; } break; \
    case 32: { constexpr int LPR = 32; CALL; } break; \
    default: { constexpr int LPR = 64; CALL; } break; \
  }


namespace arx {

// window apply (+ the long-run launch when runs can exceed the ticket's reach)
static int launch_apply(const TableSet& ts, int d, const uint32_t* sk, const uint32_t* spos, const int32_t* ssrc,
                        const float* scoef, int64_t n, uint32_t sentinel, const float* G, int64_t ldg,
                        const float* gb_in, const float* lr_dev, const float* gscale_dev,
                        float* scratch, float* scratch_b, float* scratch_h, float* scratch_hb,
                        int32_t* list, int32_t* count, int wpw, bool multi,
                        const int32_t* n_dev, hipStream_t s, const MergeOut* merge = nullptr,
                        const RunLists* runs = nullptr) {
  const int32_t* cnt = ts.cnt[0];
  const int lpr = lanes_per_row(d);
  const MergeOut mo = merge ? *merge : MergeOut{nullptr, nullptr, nullptr, -1, 0};
  const bool pure_merge = merge && merge->table < 0;      // (table >= 0: side output of an Adagrad pass)
  if (runs && n_dev && runs_path(d)) {
    // radix-sorted pass: the run records were extracted behind the sort (phase 1) -- one launch, no
    // windows, no finish launch (group.hip)
    bool sgd2 = ts.acc[0] == nullptr;
    if (!ts.E[0] && ts.E[1]) sgd2 = ts.acc[1] == nullptr;
    MergeOut m2 = mo;
    if (pure_merge) {           // every key is an entity id: a side output of "table 0", which has no rows
      m2.table = 0;
      m2.kb = 30;
    }
    return launch_run_apply(ts, multi, d, *runs, n, G, ldg, gb_in, lr_dev, gscale_dev, m2, sgd2, s);
  }
  if (!merge && cnt != nullptr && n <= kRankSortMax && n_dev == nullptr) {   // (radix-sorted input has a live count: window path)
    // small batches (single-launch LDS rank sort regime): one sub-group per sorted position --
    // mostly-unique one-hot ids need the parallelism (80 windows would leave the chip idle);
    // every multi-piece run is finished by its last arriver, no second launch.
    const int64_t nwaves = ceil_div(n, 64 / lpr);
    const int grid_a = (int)ceil_div(nwaves, 4);
    if (multi) {
      ARX_DISPATCH_LPR(lpr, (k_sparse_onepass<LPR, true><<<grid_a, 256, 0, s>>>(
                                ts, d, sk, spos, ssrc, scoef, n, sentinel, G, ldg,
                                gb_in, lr_dev, gscale_dev, scratch, scratch_b, scratch_h, scratch_hb)));
    } else {
      ARX_DISPATCH_LPR(lpr, (k_sparse_onepass<LPR, false><<<grid_a, 256, 0, s>>>(
                                ts, d, sk, spos, ssrc, scoef, n, sentinel, G, ldg,
                                gb_in, lr_dev, gscale_dev, scratch, scratch_b, scratch_h, scratch_hb)));
    }
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  const int grid = (int)ceil_div(ceil_div(n, 64), 4);
  int32_t* list_long = list;                       // <= n/64/17 entries
  int32_t* list_short = list + (n / 64 / (kShortMaxAligned + 1) + 2);   // pairs, <= n/64 entries
  const int grid8 = (int)ceil_div(n, 64);
  bool sgd = ts.acc[0] == nullptr;                 // gradient descent: no slots (all tables alike)
  if (!ts.E[0] && ts.E[1]) sgd = ts.acc[1] == nullptr;   // (table 0 virtual: the entity ids of a riding bag table)
#define ARX_WIN_GO2(WPW_, MT_, SGD_, GRID_, THREADS_)                                                \
  ARX_DISPATCH_LPR(lpr, (k_sparse_win<LPR, WPW_, MT_, SGD_><<<GRID_, THREADS_, 0, s>>>(             \
                            ts, d, sk, spos, ssrc, scoef, n, n_dev, sentinel, G, ldg, gb_in, lr_dev, \
                            gscale_dev, scratch, scratch_b, scratch_h, scratch_hb, list_long,       \
                            list_short, count, mo)))
#define ARX_WIN_GO(WPW_, MT_, GRID_, THREADS_)                                              \
  do {                                                                                      \
    if (sgd) { ARX_WIN_GO2(WPW_, MT_, true, GRID_, THREADS_); }                             \
    else { ARX_WIN_GO2(WPW_, MT_, false, GRID_, THREADS_); }                                \
  } while (0)
  // wpw = waves sharing one 64-position window.  8: one-hot ids, alone or fused in -- up to 64
  // single-row runs per window, a lone wave would walk them serially; 1: pre-expanded multi-hot
  // segments -- few long runs per window (C3 B=16384, 350 k live tokens: 77 us with 8, 54 with 4,
  // 52 with 1; the id pass 24 / 27 / 54); 4: callers that cannot tell.
  if (pure_merge) {   // entity-id runs: the one-hot window shape (many short runs per window)
    ARX_DISPATCH_LPR(lpr, (k_sparse_win<LPR, 8, false, false, true><<<grid8, 512, 0, s>>>(
                              ts, d, sk, spos, ssrc, scoef, n, n_dev, sentinel, G, ldg, gb_in, lr_dev,
                              gscale_dev, scratch, scratch_b, scratch_h, scratch_hb, list_long,
                              list_short, count, mo)));
  } else if (wpw >= 8) {
    if (multi) { ARX_WIN_GO(8, true, grid8, 512); } else { ARX_WIN_GO(8, false, grid8, 512); }
  } else if (wpw >= 4) {
    if (multi) { ARX_WIN_GO(4, true, grid8, 256); } else { ARX_WIN_GO(4, false, grid8, 256); }
  } else {
    if (multi) { ARX_WIN_GO(1, true, grid, 256); } else { ARX_WIN_GO(1, false, grid, 256); }
  }
#undef ARX_WIN_GO2
#undef ARX_WIN_GO
  ARX_CHECK_LAUNCH();
  {
    int64_t nlong = ceil_div(n, 64 * (int64_t)(kShortMaxAligned + 2));      // upper bound of long runs
    if (nlong > kPassBBlocks) nlong = kPassBBlocks;
    const int nsg = 1024 / lpr;
    int64_t nshort = ceil_div(ceil_div(n, 64), nsg);                         // upper bound of short runs
    const int64_t cap = (int64_t)cu_count() * 2;
    if (nshort > cap) nshort = cap;
    if (pure_merge) {
      ARX_DISPATCH_LPR(lpr, (k_sparse_finish<LPR, false, true><<<(int)(nlong + nshort), 1024, 0, s>>>(
                                ts, d, sk, n, n_dev, lr_dev, gscale_dev, scratch, scratch_b,
                                scratch_h, scratch_hb, list_long, list_short, count, (int)nlong, mo)));
    } else if (multi) {
      ARX_DISPATCH_LPR(lpr, (k_sparse_finish<LPR, true><<<(int)(nlong + nshort), 1024, 0, s>>>(
                                ts, d, sk, n, n_dev, lr_dev, gscale_dev, scratch, scratch_b,
                                scratch_h, scratch_hb, list_long, list_short, count, (int)nlong, mo)));
    } else {
      ARX_DISPATCH_LPR(lpr, (k_sparse_finish<LPR, false><<<(int)(nlong + nshort), 1024, 0, s>>>(
                                ts, d, sk, n, n_dev, lr_dev, gscale_dev, scratch, scratch_b,
                                scratch_h, scratch_hb, list_long, list_short, count, (int)nlong, mo)));
    }
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

}  // namespace arx

// ---- multi-hot lookups: merge per entity first, then per token (see arx.h) ----
namespace {
struct BagWs {
  SparseWs wi, wt;                        // sort / apply workspaces of the two stages
  size_t off_wi, off_wt, off_ikeys, off_isrc, off_icoef, off_tkeys, off_tsrc, off_gu, off_gub, off_hoff, off_csc_lb, total;
};
int bag_ws_layout(int64_t n_i, int max_len, int d, BagWs* w) {
  const int64_t n_t = n_i * (int64_t)max_len;
  if (sparse_ws_layout(n_i, 256, &w->wi) != ARX_OK || sparse_ws_layout(n_t, 256, &w->wt) != ARX_OK)
    return ARX_EINVAL;
  size_t o = 0;
  w->off_wi = o; o += align_up(w->wi.total, 256);
  w->off_wt = o; o += align_up(w->wt.total, 256);
  w->off_ikeys = o; o += align_up((size_t)n_i * 4, 256);
  w->off_isrc = o; o += align_up((size_t)n_i * 4, 256);
  w->off_icoef = o; o += align_up((size_t)n_i * 4, 256);
  w->off_tkeys = o; o += align_up((size_t)n_t * 4, 256);
  w->off_tsrc = o; o += align_up((size_t)n_t * 4, 256);
  w->off_gu = o; o += align_up((size_t)n_i * (size_t)d * 4, 256);
  w->off_gub = o; o += align_up((size_t)n_i * 4, 256);
  w->off_hoff = o; o += align_up((size_t)(n_i + 64) * 4, 256);       // head offsets of the compacted expansion + its total
  w->off_csc_lb = o; o += 1024 * 8;                                  // look-back cells of the static-order sweep (csc.hip)
  w->total = o;
  return ARX_OK;
}
}  // namespace

namespace arx {
namespace {

static bool bag_compact(int64_t n_i, int max_len) {
  return n_i * (int64_t)max_len > kRankSortMax && n_i <= (1 << 18);
}

// Stage 1b: the bags of the DISTINCT entities of a sorted entity-key list, sorted by token.
int bag_token_sort(const BagWs& w, char* base, const uint32_t* sk_ent, int64_t n_i, const int32_t* ndev_i,
                   uint32_t sent_i, int ent_kb, uint32_t ent_tag, const int32_t* vals, const int32_t* starts,
                   const int32_t* lens, int max_len, int64_t table_rows, int d, hipStream_t s, bool have_hoff) {
  const int64_t n_t = n_i * (int64_t)max_len;
  char* bt = base + w.off_wt;
  int32_t* tkeys = reinterpret_cast<int32_t*>(base + w.off_tkeys);
  int32_t* tsrc = reinterpret_cast<int32_t*>(base + w.off_tsrc);
  uint32_t* sk_t = reinterpret_cast<uint32_t*>(bt + w.wt.off_keys_out);
  int32_t* ssrc_t = reinterpret_cast<int32_t*>(bt + w.wt.off_ssrc);
  float* scoef_t = reinterpret_cast<float*>(bt + w.wt.off_scoef);
  int32_t* count_t = reinterpret_cast<int32_t*>(bt + w.wt.off_count);
  int kbt = 1;
  while ((1ll << kbt) < table_rows && kbt < 30) ++kbt;
  const uint32_t sent_t = 1u << kbt;
  const bool compact = bag_compact(n_i, max_len);
  int32_t* hoff = reinterpret_cast<int32_t*>(base + w.off_hoff);
  int32_t* ttotal = hoff + n_i;
  {
    int64_t g = ceil_div(n_t, 256);
    const int64_t cap = (int64_t)cu_count() * 16;
    if (g > cap) g = cap;
    if (compact) {
      if (!have_hoff) {        // (normally a by-product of the entity list's run extraction, group.hip)
        k_head_len_scan<<<1, 1024, 0, s>>>(sk_ent, n_i, ndev_i, sent_i, ent_kb, ent_tag, lens, max_len, hoff, ttotal);
        ARX_CHECK_LAUNCH();
      }
      k_bag_expand_compact<<<(int)g, 256, 0, s>>>(sk_ent, n_i, ndev_i, sent_i, ent_kb, ent_tag, vals, starts, lens,
                                                  max_len, table_rows, hoff, tkeys, tsrc);
    } else {
      k_bag_expand_heads<<<(int)g, 256, 0, s>>>(sk_ent, n_i, ndev_i, sent_i, ent_kb, ent_tag, vals, starts, lens,
                                                max_len, table_rows, tkeys, tsrc);
    }
    ARX_CHECK_LAUNCH();
  }
  if (n_t <= kRankSortMax)
    return launch_rank_sort(tkeys, n_t, sent_t, sk_t, reinterpret_cast<uint32_t*>(bt + w.wt.off_pos_out),
                            count_t, s, tsrc, nullptr, ssrc_t, scoef_t);
  const int rc = launch_radix_sort(tkeys, tsrc, nullptr, n_t, sent_t, kbt,
                                   reinterpret_cast<uint32_t*>(bt + w.wt.off_keys_tmp), sk_t,
                                   reinterpret_cast<int32_t*>(bt + w.wt.off_pos_in), ssrc_t,
                                   reinterpret_cast<float*>(bt + w.wt.off_pos_out), scoef_t,
                                   reinterpret_cast<int32_t*>(bt + w.wt.off_hist), count_t, count_t + 2, s,
                                   compact ? ttotal : nullptr);
  if (rc || !runs_path(d)) return rc;
  const RunLists rl = run_lists_of(bt + w.wt.off_runs, n_t, 256, ssrc_t, scoef_t, count_t + 8);
  return launch_runs_extract(sk_t, n_t, count_t + 2, sent_t, rl, d, s);
}

// Stage 1b with the table's STATIC token order (csc.hip): no expansion, no sort -- mark the live pairs from the
// step's entity lookups, sweep them out in place order, extract the run records.  Depends on the lookup ids only
// (not on the one-hot sort of the pass).
bool csc_path(const BagStage* bag, int64_t n_i, int d) {
  return bag && bag->csc && runs_path(d) && n_i * (int64_t)bag->max_len > kRankSortMax;
}
// part 0: the marks (phase A of the pass: the one-hot apply reads slot_of, and it only waits for phase A);
// part 1: the sweep + the run records (phase B)
int csc_token_sort(const BagWs& w, char* base, const CatSites& st, int64_t n_i, const BagStage& bag, int d,
                   hipStream_t s, int part) {
  const int64_t n_t = n_i * (int64_t)bag.max_len;
  char* bt = base + w.off_wt;
  uint32_t* sk_t = reinterpret_cast<uint32_t*>(bt + w.wt.off_keys_out);
  int32_t* ssrc_t = reinterpret_cast<int32_t*>(bt + w.wt.off_ssrc);
  float* scoef_t = reinterpret_cast<float*>(bt + w.wt.off_scoef);
  int32_t* count_t = reinterpret_cast<int32_t*>(bt + w.wt.off_count);
  void* lb = base + w.off_csc_lb;
  // timing-only ablation (WRONG results): ARX_ABL_CSC_SKIP=1 -- after the first call only the marks go out (the token
  // apply walks the first batch's list): what the step would cost if the apply read the flags itself
  static const bool abl_skip = getenv("ARX_ABL_CSC_SKIP") != nullptr;
  static int abl_calls = 0;
  if (part == 0) {
    MarkSites ms = {};
    ms.rows = st.rows[0];
    ms.offs[0] = 0;
    for (int q = 0; q < st.nsites; ++q) {
      if (st.table[q] != 0) continue;
      ms.ids[ms.n] = st.ids[q];
      ms.cat_map[ms.n] = st.cat_map[q];
      ms.offs[ms.n + 1] = ms.offs[ms.n] + (st.offs[q + 1] - st.offs[q]);
      ++ms.n;
    }
    for (int q = ms.n; q < kMaxSites; ++q) ms.offs[q + 1] = ms.offs[ms.n];
    const bool skip = abl_skip && abl_calls++ >= 1;
    return launch_csc_mark(ms, *bag.csc, bag.starts, bag.lens, bag.max_len, count_t, skip ? 0 : 512, lb,
                           skip ? 0 : 1024, s);
  }
  if (abl_skip && abl_calls > 1) return ARX_OK;
  int kbt = 1;
  while ((1ll << kbt) < bag.rows && kbt < 30) ++kbt;
  int rc = launch_csc_compact(*bag.csc, sk_t, ssrc_t, scoef_t, n_t, count_t + 2, lb, s);
  if (rc) return rc;
  const RunLists rl = run_lists_of(bt + w.wt.off_runs, n_t, 256, ssrc_t, scoef_t, count_t + 8);
  return launch_runs_extract(sk_t, n_t, count_t + 2, 1u << kbt, rl, d, s);
}

// Stage 2b: token runs over the merged rows Gu (every coefficient is 1) -> Adagrad.
int bag_token_apply(const BagWs& w, char* base, int64_t n_i, int max_len, float* E, float* acc, float* bias,
                    float* bias_acc, int64_t table_rows, int32_t* aux_cnt, int d, const float* lr_dev,
                    const float* gscale_dev, hipStream_t s) {
  const int64_t n_t = n_i * (int64_t)max_len;
  char* bt = base + w.off_wt;
  const bool rank_t = n_t <= kRankSortMax;
  int32_t* count_t = reinterpret_cast<int32_t*>(bt + w.wt.off_count);
  int kbt = 1;
  while ((1ll << kbt) < table_rows && kbt < 30) ++kbt;
  TableSet ts = {};
  ts.E[0] = E;
  ts.acc[0] = acc;
  ts.bias[0] = bias;
  ts.bias_acc[0] = bias_acc;
  ts.cnt[0] = rank_t ? aux_cnt : nullptr;
  ts.kb = kbt;
  const RunLists rl = run_lists_of(bt + w.wt.off_runs, n_t, 256, reinterpret_cast<int32_t*>(bt + w.wt.off_ssrc),
                                   reinterpret_cast<float*>(bt + w.wt.off_scoef), count_t + 8);
  return launch_apply(ts, d, reinterpret_cast<uint32_t*>(bt + w.wt.off_keys_out), nullptr,
                      reinterpret_cast<int32_t*>(bt + w.wt.off_ssrc), nullptr, n_t, 1u << kbt,
                      reinterpret_cast<float*>(base + w.off_gu), d,
                      bias ? reinterpret_cast<float*>(base + w.off_gub) : nullptr, lr_dev, gscale_dev,
                      reinterpret_cast<float*>(bt + w.wt.off_scratch),
                      reinterpret_cast<float*>(bt + w.wt.off_scratch_b),
                      reinterpret_cast<float*>(bt + w.wt.off_scratch_h),
                      reinterpret_cast<float*>(bt + w.wt.off_scratch_hb),
                      reinterpret_cast<int32_t*>(bt + w.wt.off_list), count_t, kTokenWpw, false,
                      rank_t ? nullptr : count_t + 2, s, nullptr, &rl);
}

}  // namespace
}  // namespace arx

namespace arx {

#ifndef ARX_K7_RIDER_DEFAULT
#define ARX_K7_RIDER_DEFAULT kRiderWin
#endif
static bool rider_hoff_ok(int64_t n, int64_t n0, int max_len, int d) {
  return runs_path(d) && bag_compact(n0, max_len) && runs_extract_blocks(n) <= 224;
}

// Round 5: the rider's one-hot pass on run records too (they have to be in sorted order: the extraction sweep's
// look-back path) -- ARX_K7_RIDER = "win" (the default: window + finish launches over the one-hot list, then the token
// apply: rounds 3-4) or "split" (phase 7 = the entity table's runs, phase 8 = one launch with the token runs and the
// other tables' runs).  Measured at C3, same box, alternating runs: win 230.5 / 228.9 us, split 231.6 / 231.4 us --
// the records lengthen the one-hot list's extraction sweep by ~7 us on the sort branch and the two run-centric
// launches (23 + 34 us in the step) are no shorter than window + finish + token apply (24 + 8 + 29).  A third form,
// the WHOLE apply in one launch with the token runs waiting inside it for the entity runs (arrival counters,
// write-through merged rows), passed every test and measured 297 us: the one-hot runs alone fill the chip's
// resident workgroups, so the token workgroups behind them cannot start early and only pay for the coupling.
// Removed again.
enum { kRiderWin = 0, kRiderSplit = 1 };
static int rider_mode() {
  static const int mode = [] {
    const char* e = getenv("ARX_K7_RIDER");
    if (e && !strcmp(e, "split")) return (int)kRiderSplit;
    if (e && !strcmp(e, "win")) return (int)kRiderWin;
    return (int)ARX_K7_RIDER_DEFAULT;
  }();
  return mode;
}
// (bag->rider >= 0: the caller's choice for this pass -- bit 0x100 of the entry point's phase argument)
static bool rider_records(bool hoff_ok, const BagStage* bag) {
  const int mode = (bag && bag->rider >= 0) ? bag->rider : rider_mode();
  return hoff_ok && mode != kRiderWin;
}

int sparse_adagrad_sites_sorted(const TableSet& ts, int ntables, int d, const CatSites& st,
                                const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                                const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                float* coef_buf, void* workspace, size_t workspace_bytes,
                                hipStream_t s, int phase, const BagStage* bag) {
  // phase 1 (keys + sort) depends on the lookup ids only, not on any gradient: the caller may run
  // it on a side stream under the forward / backward GEMMs and join before phase 2 (apply).  Both
  // phases must see the same workspace, untouched in between.  With a riding bag table each half
  // splits again (phases 5 .. 8): A one-hot keys + sort (+ bag offsets), B the token chain, C the
  // one-hot apply (its side output: the merged rows), D the token apply -- the caller joins the
  // branch after A for C and after B for D, so the tail of the token chain hides under C.
  const int mask = phase == 1 ? 3 : phase == 2 ? 12 : phase == 3 ? 15 : phase == 5 ? 1 : phase == 6 ? 2
                   : phase == 7 ? 4 : phase == 8 ? 8 : 0;
  const int64_t n = st.offs[st.nsites] + st.xoffs[st.nextra];   // one-hot + pre-expanded
  if (n == 0) return ARX_OK;
  SparseWs w;
  int rc = sparse_ws_layout(n, 256, &w);
  if (rc) { set_error("arx_sparse_adagrad_cat: workspace layout failed"); return rc; }
  if (!workspace || workspace_bytes < w.total) {
    set_error("arx_sparse_adagrad_cat: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return ARX_EWORKSPACE;
  }
  int tbits = 0;                              // key = (table << kb) | row, sentinel above both
  while ((1 << tbits) < ntables) ++tbits;
  const int key_bits = ts.kb + tbits;
  if (key_bits > 30) {
    set_error("arx_sparse_adagrad_cat: %d key bits (rows + tables) exceed 30", key_bits);
    return ARX_EUNSUPPORTED;
  }
  const uint32_t sentinel = 1u << key_bits;
  char* base = reinterpret_cast<char*>(workspace);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + w.off_keys_out);
  uint32_t* pos_out = reinterpret_cast<uint32_t*>(base + w.off_pos_out);
  int32_t* list = reinterpret_cast<int32_t*>(base + w.off_list);
  int32_t* count = reinterpret_cast<int32_t*>(base + w.off_count);
  float* scratch = reinterpret_cast<float*>(base + w.off_scratch);
  float* scratch_b = reinterpret_cast<float*>(base + w.off_scratch_b);
  // radix-sorted pass without pre-expanded segments: the sort's first pass forms the keys itself (radix_sort.hip),
  // no k_site_keys launch, keys_buf / src_buf / coef_buf stay untouched
  const bool fused_keys = n > kRankSortMax && st.nextra == 0;
  if ((mask & 1) && !fused_keys) {
    int64_t g = ceil_div(n, 256);
    k_site_keys<<<(int)g, 256, 0, s>>>(st, keys_buf, src_buf, coef_buf);
    ARX_CHECK_LAUNCH();
  }
  int32_t* ssrc = reinterpret_cast<int32_t*>(base + w.off_ssrc);
  float* scoef = reinterpret_cast<float*>(base + w.off_scoef);
  float* scratch_h = reinterpret_cast<float*>(base + w.off_scratch_h);
  float* scratch_hb = reinterpret_cast<float*>(base + w.off_scratch_hb);
  const uint32_t* spos_arg = nullptr;
  const int32_t* src_arg = ssrc;
  const float* coef_arg = scoef;
  const int32_t* n_dev = nullptr;      // live-entry count of the radix sort (pads dropped)
  bool sorted_runs = false;
  if (n <= kRankSortMax && st.nextra == 0) {     // (pre-expanded multi-hot segments: radix path only)
    if (mask & 1) {
      rc = launch_rank_sort(keys_buf, n, sentinel, keys_out, pos_out, count, s, src_buf, coef_buf,
                            ssrc, scoef);
      if (rc) return rc;
    }
  } else {   // own LSD radix sort: src/coef come out in sorted order too
    if (mask & 1) {
      rc = launch_radix_sort(keys_buf, src_buf, coef_buf, n, sentinel, key_bits /* survivors of the first pass are < sentinel */,
                             reinterpret_cast<uint32_t*>(base + w.off_keys_tmp), keys_out,
                             reinterpret_cast<int32_t*>(base + w.off_pos_in), ssrc,
                             reinterpret_cast<float*>(base + w.off_pos_out), scoef,
                             reinterpret_cast<int32_t*>(base + w.off_hist), count, count + 2, s, nullptr,
                             fused_keys ? &st : nullptr);
      if (rc) return rc;
      sorted_runs = runs_path(d);                // run records: extracted below, once the riding bag table is known
    }
    n_dev = count + 2;
  }
  // ---- a multi-hot table whose entity lookups are the lookups of table 0 (BagStage): its
  // distinct entities are the run heads of table 0 in the sorted list just made -- no entity sort
  // of its own; the merged gradient row of each comes out of the apply below as a side output
  BagWs bw;
  char* bbase = nullptr;
  bool have_hoff = false;
  int64_t n0 = 0;                              // lookups of table 0 (host count: its keys sort first)
  if (bag) {
    if (st.nextra != 0) { set_error("arx_sparse_adagrad_cat_multi_bags: no pre-expanded segments"); return ARX_EINVAL; }
    for (int q = 0; q < st.nsites; ++q)
      if (st.table[q] == 0) n0 += st.offs[q + 1] - st.offs[q];
    if (n0 == 0) bag = nullptr;
  }
  if (bag) {
    if (bag_ws_layout(n0, bag->max_len, d, &bw) != ARX_OK || !bag->ws || bag->ws_bytes < bw.total) {
      set_error("arx_sparse_adagrad_cat_multi_bags: bag workspace too small (%zu < %zu)", bag->ws_bytes, bw.total);
      return ARX_EWORKSPACE;
    }
    bbase = reinterpret_cast<char*>(bag->ws);
  }
  const bool csc = n_dev && csc_path(bag, n0, d);      // static token order: no bag offsets, no token sort
  if (csc && (mask & 1)) {        // the marks: part of the first quarter (the one-hot apply reads slot_of)
    rc = csc_token_sort(bw, bbase, st, n0, *bag, d, s, 0);
    if (rc) return rc;
  }
  const bool hoff_ok = bag && n_dev && rider_hoff_ok(n, n0, bag->max_len, d);
  if (sorted_runs) {
    // run records of the one-hot list -- unless a bag table rides on it: then the sort branch of the step has
    // no room for them (measured at C3: the branch ends 40 us behind the backward GEMMs), the one-hot pass
    // keeps the window apply (same time for mostly single-entry runs) and the sweep only produces the ordered
    // offsets of the distinct entities' bags in the compact token list
    const bool hoff = hoff_ok;
    int32_t* hp = hoff ? reinterpret_cast<int32_t*>(bbase + bw.off_hoff) : nullptr;
    RunLists rl = run_lists_of(base + w.off_runs, n, 256, ssrc, scoef, count + 8);
    if (bag && !rider_records(hoff_ok, bag)) rl.R = nullptr;
    if (!bag || (hoff && (!csc || rl.R))) {      // (static token order: the sweep is only wanted for the records)
      rc = launch_runs_extract(keys_out, n, count + 2, sentinel, rl, d, s, hoff ? bag->lens : nullptr, ts.kb, 0u,
                               bag ? bag->max_len : 0, hp, n0, hoff ? hp + n0 : nullptr, count + 32);
      if (rc) return rc;
    }
  }
  have_hoff = hoff_ok;
  if (bag && (mask & 2)) {
    rc = csc ? csc_token_sort(bw, bbase, st, n0, *bag, d, s, 1)
             : bag_token_sort(bw, bbase, keys_out, n0, n_dev, sentinel, ts.kb, 0u, bag->vals, bag->starts,
                              bag->lens, bag->max_len, bag->rows, d, s, have_hoff);
    if (rc) return rc;
  }
  if (!(mask & 12)) return ARX_OK;
  bool any_bias = bag && bag->bias;
  for (int t = 0; t < ntables; ++t) any_bias = any_bias || ts.bias[t] != nullptr;
  const float* gb_in = any_bias ? Gb : nullptr;
  MergeOut side = {nullptr, nullptr, nullptr, -1, 0};
  float* gu = bag ? reinterpret_cast<float*>(bbase + bw.off_gu) : nullptr;
  float* gub = bag ? reinterpret_cast<float*>(bbase + bw.off_gub) : nullptr;
  if (bag)      // (static token order: the rows of Gu are addressed through slot_of, not by sorted head position)
    side = MergeOut{gu, bag->bias ? gub : nullptr, bag->lens, 0, ts.kb, csc ? bag->csc->slot_of : nullptr};
  const RunLists rl_sites = run_lists_of(base + w.off_runs, n, 256, ssrc, scoef, count + 8);
  rc = ARX_OK;
  bool sgd_sites = ts.acc[0] == nullptr;
  if (!ts.E[0] && ts.E[1]) sgd_sites = ts.acc[1] == nullptr;
  if (bag && n_dev && runs_path(d) && rider_records(hoff_ok, bag) && n0 * (int64_t)bag->max_len > kRankSortMax &&
      sgd_sites == (bag->acc == nullptr)) {
    // Round 5: the rider's one-hot pass on run records too, cut where the data flow allows --
    //   C  the ENTITY table's runs (records [0, ctr[kNRuns0]): the records are in sorted order, k_runs_extract's
    //      look-back path): Adagrad on the id rows + the merged rows Gu the token stage reads;
    //   D  ONE launch: the token runs over Gu + the runs of the other one-hot tables (users), which nothing waits for.
    // (before: window + finish launches over the whole one-hot list, then the token apply: three launches in a row)
    if (mask & 4)
      rc = launch_run_apply(ts, ntables > 1, d, rl_sites, n0, G, ldg, gb_in, lr_dev, gscale_dev, side, sgd_sites, s,
                            ApplySel{-1, kNRuns0, 0});
    if (rc || !(mask & 8)) return rc;
    char* bt = bbase + bw.off_wt;
    const int64_t n_t = n0 * (int64_t)bag->max_len;
    int32_t* count_t = reinterpret_cast<int32_t*>(bt + bw.wt.off_count);
    int kbt = 1;
    while ((1ll << kbt) < bag->rows && kbt < 30) ++kbt;
    TableSet tt = {};
    tt.E[0] = bag->E;
    tt.acc[0] = bag->acc;
    tt.bias[0] = bag->bias;
    tt.bias_acc[0] = bag->bias_acc;
    tt.kb = kbt;
    const RunLists rl_t = run_lists_of(bt + bw.wt.off_runs, n_t, 256, reinterpret_cast<int32_t*>(bt + bw.wt.off_ssrc),
                                       reinterpret_cast<float*>(bt + bw.wt.off_scoef), count_t + 8);
    const MergeOut none = {nullptr, nullptr, nullptr, -1, 0};
    return launch_run_apply_pair(ts, ntables > 1, rl_sites, n - n0 > 0 ? n - n0 : 1, G, ldg, gb_in, none,
                                 ApplySel{kNRuns0, -1, -2}, tt, rl_t, n_t, gu,
                                 d, bag->bias ? gub : nullptr, d, lr_dev,
                                 gscale_dev, sgd_sites, s);
  }
  if (mask & 4)
    rc = launch_apply(ts, d, keys_out, spos_arg, src_arg, coef_arg, n, sentinel, G, ldg, gb_in, lr_dev,
                    gscale_dev, scratch, scratch_b, scratch_h, scratch_hb, list, count,
                    /*wpw=*/(ntables > 1 || st.nextra == 0) ? 8 : 1, /*multi=*/ntables > 1, n_dev, s,
                    bag ? &side : nullptr, bag ? nullptr : &rl_sites);
  if (rc || !bag || !(mask & 8)) return rc;
  return bag_token_apply(bw, bbase, n0, bag->max_len, bag->E, bag->acc, bag->bias, bag->bias_acc, bag->rows,
                         bag->aux_cnt, d, lr_dev, gscale_dev, s);
}

}  // namespace arx

extern "C" {

size_t arx_sparse_adagrad_workspace_bytes(int64_t n) {
  SparseWs w;
  if (sparse_ws_layout(n, 256, &w) != ARX_OK) return 0;  // d <= 256 (largest supported row)
  return w.total;
}

int arx_sparse_adagrad(float* E, float* acc, float* bias, float* bias_acc, int d,
                       const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                       const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                       const float* gscale_dev, int key_bits, void* workspace,
                       size_t workspace_bytes, void* stream) {
  return arx_sparse_adagrad_ticket(E, acc, bias, bias_acc, d, keys, src, coef, n, G, ldg, Gb, lr_dev,
                                   gscale_dev, key_bits, nullptr, workspace, workspace_bytes, stream);
}

int arx_sparse_adagrad_ticket(float* E, float* acc, float* bias, float* bias_acc, int d,
                              const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                              const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                              const float* gscale_dev, int key_bits, int32_t* aux_cnt,
                              void* workspace, size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(E && keys && G && lr_dev, "arx_sparse_adagrad: null pointer");
  ARX_CHECK_ARG(acc ? (bias == nullptr) == (bias_acc == nullptr) : bias_acc == nullptr,
                "arx_sparse_adagrad: bias and bias_acc go together (acc == NULL: gradient descent, no slots)");
  ARX_CHECK_ARG(!(bias && !Gb), "arx_sparse_adagrad: bias table given without Gb");
  if (d <= 0 || d % 4 != 0 || d > 256) {
    set_error("arx_sparse_adagrad: d=%d unsupported (d %% 4 == 0, d <= 256)", d);
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_ARG(ldg % 4 == 0 && ldg >= d, "arx_sparse_adagrad: ldg must be a multiple of 4 and >= d");
  ARX_CHECK_ARG(n >= 0 && n < (int64_t)0x7fffffff, "arx_sparse_adagrad: bad n");
  if (n == 0) return ARX_OK;
  SparseWs w;
  int rc = sparse_ws_layout(n, 256, &w);
  if (rc) { set_error("arx_sparse_adagrad: workspace layout failed"); return rc; }
  if (!workspace || workspace_bytes < w.total) {
    set_error("arx_sparse_adagrad: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return ARX_EWORKSPACE;
  }
  if (key_bits <= 0 || key_bits > 30) key_bits = 30;
  const uint32_t sentinel = 1u << key_bits;
  hipStream_t s = as_stream(stream);
  char* base = reinterpret_cast<char*>(workspace);
  uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(base + w.off_keys_tmp);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + w.off_keys_out);
  uint32_t* pos_in = reinterpret_cast<uint32_t*>(base + w.off_pos_in);
  uint32_t* pos_out = reinterpret_cast<uint32_t*>(base + w.off_pos_out);
  int32_t* list = reinterpret_cast<int32_t*>(base + w.off_list);
  int32_t* count = reinterpret_cast<int32_t*>(base + w.off_count);
  float* scratch = reinterpret_cast<float*>(base + w.off_scratch);
  float* scratch_b = reinterpret_cast<float*>(base + w.off_scratch_b);
  const uint32_t* spos_arg = pos_out;
  const int32_t* n_dev = nullptr;      // live-entry count of the radix sort (pads dropped)
  if (n <= kRankSortMax) {
    // id-only batches (B + S keys): one workgroup sorts everything in LDS -- a
    // single launch instead of the 5-6 of the device-wide radix sort.
    rc = launch_rank_sort(keys, n, sentinel, keys_out, pos_out, count, s);
    if (rc) return rc;
  } else {
    // own LSD radix sort; src/coef are emitted in sorted order (one hop less per contribution)
    int32_t* ssrc = reinterpret_cast<int32_t*>(base + w.off_ssrc);
    float* scoef = reinterpret_cast<float*>(base + w.off_scoef);
    rc = launch_radix_sort(keys, src, coef, n, sentinel, key_bits /* survivors of the first pass are < sentinel */, keys_tmp, keys_out,
                           reinterpret_cast<int32_t*>(pos_in), ssrc,
                           reinterpret_cast<float*>(pos_out), scoef,
                           reinterpret_cast<int32_t*>(base + w.off_hist), count, count + 2, s);
    if (rc) return rc;
    n_dev = count + 2;
    spos_arg = nullptr;
    src = ssrc;
    coef = scoef;
    if (runs_path(d)) {
      rc = launch_runs_extract(keys_out, n, count + 2, sentinel,
                               run_lists_of(base + w.off_runs, n, 256, ssrc, scoef, count + 8), d, s);
      if (rc) return rc;
    }
  }
  const float* gb_in = bias ? Gb : nullptr;
  TableSet ts = {};
  ts.E[0] = E;
  ts.acc[0] = acc;
  ts.bias[0] = bias;
  ts.bias_acc[0] = bias_acc;
  ts.cnt[0] = aux_cnt;
  ts.kb = key_bits;                            // single table: key >> kb == 0 for every real key
  const RunLists rl_one = run_lists_of(base + w.off_runs, n, 256, reinterpret_cast<int32_t*>(base + w.off_ssrc),
                                       reinterpret_cast<float*>(base + w.off_scoef), count + 8);
  return launch_apply(ts, d, keys_out, spos_arg, src, coef, n, sentinel, G, ldg, gb_in, lr_dev,
                      gscale_dev, scratch, scratch_b,
                      reinterpret_cast<float*>(base + w.off_scratch_h),
                      reinterpret_cast<float*>(base + w.off_scratch_hb), list, count,
                      /*wpw=*/n <= (1 << 21) ? 4 : 1   /* explicit (key, src, coef) triples: one-hot or multi-hot, not known here; n is the padded capacity (~3x the live count for multi-hot sites) */, /*multi=*/false, n_dev, s,
                      nullptr, &rl_one);
}

int arx_merged_sq_norm(const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                       int key_bits, const float* X, int64_t ldx, int d, int L, int64_t step_stride,
                       const float* Xb, int Lb, int64_t stepb_stride, float* out, void* workspace,
                       size_t workspace_bytes, void* scratch, void* stream) {
  ARX_CHECK_ARG(keys && src && coef && out, "arx_merged_sq_norm: null pointer");
  ARX_CHECK_ARG(X || Xb, "arx_merged_sq_norm: neither X nor Xb given");
  ARX_CHECK_ARG(!X || (d > 0 && ldx >= d && L > 0), "arx_merged_sq_norm: bad X shape");
  ARX_CHECK_ARG(!Xb || Lb > 0, "arx_merged_sq_norm: bad Xb shape");
  ARX_CHECK_ARG(n >= 0 && n < (int64_t)0x7fffffff, "arx_merged_sq_norm: bad n");
  if (n == 0) return ARX_OK;
  SparseWs w;
  int rc = sparse_ws_layout(n, 256, &w);
  if (rc) { set_error("arx_merged_sq_norm: workspace layout failed"); return rc; }
  if (!workspace || workspace_bytes < w.total) {
    set_error("arx_merged_sq_norm: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return ARX_EWORKSPACE;
  }
  if (key_bits <= 0 || key_bits > 30) key_bits = 30;
  const uint32_t sentinel = 1u << key_bits;
  hipStream_t s = as_stream(stream);
  char* base = reinterpret_cast<char*>(workspace);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + w.off_keys_out);
  int32_t* count = reinterpret_cast<int32_t*>(base + w.off_count);
  int32_t* ssrc = reinterpret_cast<int32_t*>(base + w.off_ssrc);
  float* scoef = reinterpret_cast<float*>(base + w.off_scoef);
  const int32_t* n_dev = nullptr;
  if (n <= kRankSortMax) {
    rc = launch_rank_sort(keys, n, sentinel, keys_out, reinterpret_cast<uint32_t*>(base + w.off_pos_out),
                          count, s, src, coef, ssrc, scoef);
  } else {
    rc = launch_radix_sort(keys, src, coef, n, sentinel, key_bits /* survivors of the first pass are < sentinel */,
                           reinterpret_cast<uint32_t*>(base + w.off_keys_tmp), keys_out,
                           reinterpret_cast<int32_t*>(base + w.off_pos_in), ssrc,
                           reinterpret_cast<float*>(base + w.off_pos_out), scoef,
                           reinterpret_cast<int32_t*>(base + w.off_hist), count, count + 2, s);
    n_dev = count + 2;
  }
  if (rc) return rc;
  constexpr int GS = 16;
  const int64_t blocks = ceil_div(n * GS, (int64_t)256);
  ARX_CHECK_ARG(blocks < (int64_t)0x7fffffff, "arx_merged_sq_norm: n too large");
  // block partials: the apply pass's scratch rows are free here (>= 4 * n bytes >= 4 * blocks)
  float* bpart = reinterpret_cast<float*>(base + w.off_scratch);
  ARX_CHECK_ARG(scratch, "arx_merged_sq_norm: reduce scratch is NULL (arx_reduce_scratch_bytes)");
  unsigned int* ticket = reduce_scratch_ticket(scratch);
  k_merged_sq_norm<GS><<<(int)blocks, 256, 0, s>>>(keys_out, ssrc, scoef, n, n_dev, sentinel, X, ldx,
                                                    X ? d : 0, X ? L : 0, step_stride, Xb, Xb ? Lb : 0,
                                                    stepb_stride, bpart, ticket, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}


size_t arx_sparse_adagrad_bags_workspace_bytes(int64_t n_lookups, int max_len, int d) {
  BagWs w;
  if (n_lookups <= 0 || max_len <= 0 || d <= 0) return 0;
  if (bag_ws_layout(n_lookups, max_len, d, &w) != ARX_OK) return 0;
  return w.total;
}

int arx_sparse_adagrad_bags(int phase, float* E, float* acc, float* bias, float* bias_acc,
                            int64_t table_rows, int d, const int32_t* vals, const int32_t* starts,
                            const int32_t* lens, int64_t n_entities, int max_len, int nsites,
                            const int32_t* const* site_ids, const int64_t* site_n,
                            const int32_t* site_row_base, const float* site_coef, const float* G,
                            int64_t ldg, const float* Gb, const float* lr_dev, const float* gscale_dev,
                            int32_t* aux_cnt, void* workspace, size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(phase == 1 || phase == 2 || phase == 3, "arx_sparse_adagrad_bags: phase 1, 2 or 3");
  ARX_CHECK_ARG(E && vals && starts && lens && site_ids && site_n && site_row_base && site_coef && G &&
                    lr_dev, "arx_sparse_adagrad_bags: null pointer");
  ARX_CHECK_ARG(nsites > 0 && nsites <= kMaxSites, "arx_sparse_adagrad_bags: 1..8 lookup sites");
  ARX_CHECK_ARG(acc ? (bias == nullptr) == (bias_acc == nullptr) : bias_acc == nullptr,
                "arx_sparse_adagrad_bags: bias and bias_acc go together (acc == NULL: gradient descent)");
  ARX_CHECK_ARG(!(bias && !Gb), "arx_sparse_adagrad_bags: bias table given without Gb");
  ARX_CHECK_ARG(table_rows > 0 && n_entities > 0 && max_len > 0, "arx_sparse_adagrad_bags: bad sizes");
  if (d <= 0 || d % 4 != 0 || d > 256) {
    set_error("arx_sparse_adagrad_bags: d=%d unsupported (d %% 4 == 0, d <= 256)", d);
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_ARG(ldg % 4 == 0 && ldg >= d, "arx_sparse_adagrad_bags: ldg must be a multiple of 4 and >= d");
  CatSites st;
  st.nsites = nsites;
  st.offs[0] = 0;
  for (int q = 0; q < kMaxSites; ++q) {
    const bool live = q < nsites;
    st.cat_map[q] = nullptr;                      // key = the entity id itself
    st.ids[q] = live ? site_ids[q] : nullptr;
    st.row_base[q] = live ? site_row_base[q] : 0;
    st.coef[q] = live ? site_coef[q] : 0.f;
    st.table[q] = 0;
    st.offs[q + 1] = st.offs[q] + (live ? site_n[q] : 0);
    if (live) ARX_CHECK_ARG(site_n[q] >= 0 && (site_ids[q] || site_n[q] == 0), "arx_sparse_adagrad_bags: bad site");
  }
  for (int t = 0; t < kMaxTables; ++t) st.rows[t] = n_entities;
  int kbi = 1;
  while ((1ll << kbi) < n_entities && kbi < 30) ++kbi;
  st.kb = kbi;
  st.nextra = 0;
  st.xoffs[0] = 0;
  const int64_t n_i = st.offs[nsites];
  if (n_i == 0) return ARX_OK;
  const int64_t n_t = n_i * (int64_t)max_len;
  ARX_CHECK_ARG(n_t < (int64_t)0x7fffffff, "arx_sparse_adagrad_bags: too many bag slots");
  BagWs w;
  if (bag_ws_layout(n_i, max_len, d, &w) != ARX_OK) {
    set_error("arx_sparse_adagrad_bags: workspace layout failed");
    return ARX_EINVAL;
  }
  if (!workspace || workspace_bytes < w.total) {
    set_error("arx_sparse_adagrad_bags: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return ARX_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  char* base = reinterpret_cast<char*>(workspace);
  char* bi = base + w.off_wi;
  int32_t* ikeys = reinterpret_cast<int32_t*>(base + w.off_ikeys);
  int32_t* isrc = reinterpret_cast<int32_t*>(base + w.off_isrc);
  float* icoef = reinterpret_cast<float*>(base + w.off_icoef);
  float* Gu = reinterpret_cast<float*>(base + w.off_gu);
  float* Gub = reinterpret_cast<float*>(base + w.off_gub);
  const uint32_t sent_i = 1u << kbi;
  uint32_t* sk_i = reinterpret_cast<uint32_t*>(bi + w.wi.off_keys_out);
  int32_t* ssrc_i = reinterpret_cast<int32_t*>(bi + w.wi.off_ssrc);
  float* scoef_i = reinterpret_cast<float*>(bi + w.wi.off_scoef);
  int32_t* count_i = reinterpret_cast<int32_t*>(bi + w.wi.off_count);
  const bool rank_i = n_i <= kRankSortEntities;
  const int32_t* ndev_i = rank_i ? nullptr : count_i + 2;
  int rc;
  if (phase & 1) {
    // ---- stage 1a: (entity, gradient row, coef) of every lookup, sorted by entity ----
    k_site_keys<<<(int)ceil_div(n_i, 256), 256, 0, s>>>(st, ikeys, isrc, icoef);
    ARX_CHECK_LAUNCH();
    if (rank_i)
      rc = launch_rank_sort(ikeys, n_i, sent_i, sk_i, reinterpret_cast<uint32_t*>(bi + w.wi.off_pos_out),
                            count_i, s, isrc, icoef, ssrc_i, scoef_i);
    else
      rc = launch_radix_sort(ikeys, isrc, icoef, n_i, sent_i, kbi,
                             reinterpret_cast<uint32_t*>(bi + w.wi.off_keys_tmp), sk_i,
                             reinterpret_cast<int32_t*>(bi + w.wi.off_pos_in), ssrc_i,
                             reinterpret_cast<float*>(bi + w.wi.off_pos_out), scoef_i,
                             reinterpret_cast<int32_t*>(bi + w.wi.off_hist), count_i, count_i + 2, s);
    if (rc) return rc;
    bool have_hoff = false;
    if (!rank_i && runs_path(d)) {
      have_hoff = bag_compact(n_i, max_len) && runs_extract_blocks(n_i) <= 224;
      int32_t* hp = reinterpret_cast<int32_t*>(base + w.off_hoff);
      rc = launch_runs_extract(sk_i, n_i, count_i + 2, sent_i,
                               run_lists_of(bi + w.wi.off_runs, n_i, 256, ssrc_i, scoef_i, count_i + 8), d, s,
                               have_hoff ? lens : nullptr, kbi, 0u, max_len, hp, n_i, have_hoff ? hp + n_i : nullptr,
                               count_i + 32);
      if (rc) return rc;
    }
    // ---- stage 1b: bags of the DISTINCT entities, sorted by token ----
    rc = bag_token_sort(w, base, sk_i, n_i, ndev_i, sent_i, kbi, 0u, vals, starts, lens, max_len, table_rows, d, s, have_hoff);
    if (rc) return rc;
  }
  if (!(phase & 2)) return ARX_OK;
  // ---- stage 2a: one merged, 1/len-scaled gradient row per distinct entity ----
  {
    TableSet none = {};
    none.kb = kbi;
    MergeOut mo = {Gu, bias ? Gub : nullptr, lens, -1, 0};
    const RunLists rl_ent = run_lists_of(bi + w.wi.off_runs, n_i, 256, ssrc_i, scoef_i, count_i + 8);
    rc = launch_apply(none, d, sk_i, nullptr, ssrc_i, scoef_i, n_i, sent_i, G, ldg, bias ? Gb : nullptr,
                      lr_dev, gscale_dev, reinterpret_cast<float*>(bi + w.wi.off_scratch),
                      reinterpret_cast<float*>(bi + w.wi.off_scratch_b),
                      reinterpret_cast<float*>(bi + w.wi.off_scratch_h),
                      reinterpret_cast<float*>(bi + w.wi.off_scratch_hb),
                      reinterpret_cast<int32_t*>(bi + w.wi.off_list), count_i, 8, false, ndev_i, s, &mo, &rl_ent);
    if (rc) return rc;
  }
  // ---- stage 2b: token runs over the merged rows (every coefficient is 1) -> Adagrad ----
  return bag_token_apply(w, base, n_i, max_len, E, acc, bias, bias_acc, table_rows, aux_cnt, d, lr_dev,
                         gscale_dev, s);
}

int arx_adagrad_dense(float* w, float* acc, const float* g, int64_t n, const float* lr_dev,
                      const float* gscale_dev, void* stream) {
  ARX_CHECK_ARG(w && g && lr_dev, "arx_adagrad_dense: null pointer");
  if (n <= 0) return ARX_OK;
  int64_t gr = ceil_div(n, 256);
  int64_t cap = (int64_t)cu_count() * 8;
  if (gr > cap) gr = cap;
  k_adagrad_dense<<<(int)gr, 256, 0, as_stream(stream)>>>(w, acc, g, n, lr_dev, gscale_dev);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_adagrad_rows_nonzero(float* W, float* acc, float* bias, float* bias_acc, float* G, float* Gb, int64_t rows,
                             int d, const float* lr_dev, void* stream) {
  ARX_CHECK_ARG(W && G && lr_dev && rows >= 0, "arx_adagrad_rows_nonzero: null pointer");
  ARX_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 256, "arx_adagrad_rows_nonzero: d %% 4 == 0, d <= 256");
  ARX_CHECK_ARG(!(bias && !Gb) && !(bias_acc && !bias), "arx_adagrad_rows_nonzero: bias needs Gb, bias_acc needs bias");
  if (rows == 0) return ARX_OK;
  const int lpr = lanes_per_row(d);
  int64_t g = ceil_div(rows, (int64_t)(256 / lpr) * 4);
  const int64_t cap = (int64_t)cu_count() * 16;
  if (g > cap) g = cap;
  ARX_DISPATCH_LPR(lpr, (k_adagrad_rows_nonzero<LPR><<<(int)g, 256, 0, as_stream(stream)>>>(W, acc, bias, bias_acc, G, Gb,
                                                                                              rows, d, lr_dev)));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_adagrad_dense_multi(int count, float* const* w, float* const* acc, const float* const* g,
                            const int64_t* n, const float* lr_dev, const float* gscale_dev, void* stream) {
  ARX_CHECK_ARG(count >= 1 && count <= 8, "arx_adagrad_dense_multi: 1..8 tensors");
  ARX_CHECK_ARG(w && acc && g && n && lr_dev, "arx_adagrad_dense_multi: null pointer");
  DenseSet ds = {};
  int blocks = 0;
  const int64_t cap = (int64_t)cu_count() * 8 / count;
  for (int t = 0; t < count; ++t) {
    ARX_CHECK_ARG(w[t] && g[t] && n[t] >= 0, "arx_adagrad_dense_multi: bad tensor");
    ds.w[t] = w[t];
    ds.acc[t] = acc[t];
    ds.g[t] = g[t];
    ds.n[t] = n[t];
    int64_t nb = ceil_div(n[t] > 0 ? n[t] : 1, 256);
    if (nb > cap) nb = cap;
    blocks += (int)nb;
    ds.blk_end[t] = blocks;
  }
  for (int t = count; t < 8; ++t) ds.blk_end[t] = blocks;
  ds.count = count;
  k_adagrad_dense_multi<<<blocks, 256, 0, as_stream(stream)>>>(ds, lr_dev, gscale_dev);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_sq_norm_accum(const float* x, int64_t n, int d, const float* row_scale, float* out_accum,
                      void* scratch, void* stream) {
  ARX_CHECK_ARG(x && out_accum && d > 0, "arx_sq_norm_accum: bad argument");
  ARX_CHECK_ARG(scratch, "arx_sq_norm_accum: reduce scratch is NULL (arx_reduce_scratch_bytes)");
  if (n <= 0) return ARX_OK;
  float* part = reduce_scratch_f32(scratch);
  unsigned int* ticket = reduce_scratch_ticket(scratch);
  const bool vec = (n % 4 == 0) && (d % 4 == 0) && n < ((int64_t)1 << 31) &&
                   (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  // few, fat blocks: the ticket atomics serialise at ~16 ns each (1024 blocks: 16 us, measured)
  const int threads = (vec ? n / 4 : n) >= 64 * 1024 ? 1024 : 256;
  int nb = (int)ceil_div(vec ? n / 4 : n, (int64_t)threads * 4);
  if (nb < 1) nb = 1;
  const int cap = 128;
  if (nb > cap) nb = cap;
  if (nb > kNormBlocks) nb = kNormBlocks;
  if (vec)
    k_sq_norm<true><<<nb, threads, 0, as_stream(stream)>>>(x, n, d, row_scale, part, ticket, out_accum);
  else
    k_sq_norm<false><<<nb, threads, 0, as_stream(stream)>>>(x, n, d, row_scale, part, ticket, out_accum);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

static int sq_norm_multi_impl(const char* who, int count, const float* const* x, const int64_t* n, const int* d,
                             const float* const* row_scale, float* out_accum, int init, float max_norm,
                             float* coef_out, float* gnorm_out, void* scratch, void* stream) {
  (void)who;
  ARX_CHECK_ARG(count >= 1 && count <= 8, "arx_sq_norm_accum_multi / arx_sq_norm_clip_multi: 1..8 tensors");
  ARX_CHECK_ARG(x && n && d && out_accum, "arx_sq_norm_accum_multi / arx_sq_norm_clip_multi: null pointer");
  ARX_CHECK_ARG(scratch, "arx_sq_norm_accum_multi / arx_sq_norm_clip_multi: reduce scratch is NULL (arx_reduce_scratch_bytes)");
  float* part = reduce_scratch_f32(scratch);
  unsigned int* ticket = reduce_scratch_ticket(scratch);
  NormSet ns = {};
  int blocks = 0;
  // blocks per tensor: one per 8 K float4 per thread-row (32 KB), at most kNormBlocks / count -- the
  // step's big tensors (13 MB each at C4) were latency-bound with 16 blocks: 20.7 us for 39 MB
  // (more is not better: every block ends in a ticket atomic that serialises at ~16 ns)
  const int per_cap = 64;
  int per = kNormBlocks / (count < 1 ? 1 : count);
  if (per > per_cap) per = per_cap;
  for (int t = 0; t < count; ++t) {
    ARX_CHECK_ARG(x[t] && n[t] >= 0 && d[t] > 0, "arx_sq_norm_accum_multi: bad tensor");
    ns.x[t] = x[t];
    ns.rs[t] = row_scale ? row_scale[t] : nullptr;
    ns.n[t] = n[t];
    ns.d[t] = d[t];
    ns.vec[t] = (n[t] % 4 == 0) && (d[t] % 4 == 0) && n[t] < ((int64_t)1 << 31) &&
                (reinterpret_cast<uintptr_t>(x[t]) & 15) == 0;
    int nb = (int)ceil_div(ns.vec[t] ? n[t] / 4 : n[t], (int64_t)1024 * 8);
    if (nb < 1) nb = 1;
    if (nb > per) nb = per;
    blocks += nb;
    ns.blk_end[t] = blocks;
  }
  for (int t = count; t < 8; ++t) ns.blk_end[t] = blocks;
  ns.count = count;
  k_sq_norm_multi<<<blocks, 1024, 0, as_stream(stream)>>>(ns, part, ticket, out_accum, init, max_norm, coef_out,
                                                          gnorm_out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_sq_norm_accum_multi(int count, const float* const* x, const int64_t* n, const int* d,
                            const float* const* row_scale, float* out_accum, void* scratch, void* stream) {
  return sq_norm_multi_impl("arx_sq_norm_accum_multi", count, x, n, d, row_scale, out_accum, 0, 0.f, nullptr,
                            nullptr, scratch, stream);
}

int arx_sq_norm_clip_multi(int count, const float* const* x, const int64_t* n, const int* d,
                           const float* const* row_scale, int init, float* sqnorm_out, float max_norm,
                           float* coef_out, float* gnorm_out, void* scratch, void* stream) {
  ARX_CHECK_ARG(coef_out, "arx_sq_norm_clip_multi: null pointer");
  return sq_norm_multi_impl("arx_sq_norm_clip_multi", count, x, n, d, row_scale, sqnorm_out, init ? 1 : 0,
                            max_norm, coef_out, gnorm_out, scratch, stream);
}

int arx_clip_coef(const float* sqnorm_dev, float max_norm, float* coef_out, float* gnorm_out,
                  void* stream) {
  ARX_CHECK_ARG(sqnorm_dev && coef_out, "arx_clip_coef: null pointer");
  k_clip_coef<<<1, 1, 0, as_stream(stream)>>>(sqnorm_dev, max_norm, coef_out, gnorm_out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
