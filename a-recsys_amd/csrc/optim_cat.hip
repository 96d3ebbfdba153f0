// optim_cat.hip -- K7 fast path for one-hot lookups with few contributions
// (id-only batches: B user rows, or S pool rows + B targets).
//
// Same semantics as arx_sparse_adagrad (duplicates summed in contribution order,
// ONE Adagrad application per touched row, hmf_model.py:146-151) but no sort:
//   kernel A  (one thread per contribution): key = cat_map[id]; first[key] =
//             atomicMin(i); cnt[key] += 1      -- integer atomics only
//   kernel B  (one sub-group per contribution): the contribution with the smallest
//             index of a key is its leader; it sums coef*G[src] of itself and -- only
//             when cnt > 1 -- of its later duplicates, found by scanning the (L2
//             resident, <= 64 k) key list in index order; then applies Adagrad and
//             restores first/cnt for the next step.
// Float sums stay in a fixed order => bit-deterministic.  The per-site key
// generation (cat_map gather, gradient-source row, coefficient) is folded into
// kernel A, so a step needs 2 launches per table instead of sort + 2 passes + one
// launch per lookup site.
#include <limits.h>

#include "common.h"
#include "k7.h"

namespace arx {

constexpr int kHot = 16;          // rows with more duplicates than this go to the hot kernel
constexpr int kHotBlocks = 64;
constexpr int kHotCap = 2048;     // indices buffered in LDS per flush

__global__ __launch_bounds__(256) void k_cat_keys(CatSites st, int32_t* __restrict__ keys,
                                                  int32_t* __restrict__ src,
                                                  float* __restrict__ coef,
                                                  int32_t* __restrict__ first,
                                                  int32_t* __restrict__ cnt, int64_t table_rows,
                                                  int32_t* __restrict__ hot /*[0]=count,[1]=done,[2..]=keys*/,
                                                  int hot_cap) {
  const int64_t n = st.offs[st.nsites];
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < kMaxSites; ++k)
      if (k < st.nsites && i >= st.offs[k]) s = k;
    const int64_t j = i - st.offs[s];
    const int id = st.ids[s][j];
    int key = id < 0 ? -1 : (st.cat_map[s] ? st.cat_map[s][id] : id);     // (id < 0: an empty pool slot, no update)
    if (key < 0 || key >= table_rows) key = ARX_KEY_NONE;
    keys[i] = key;
    src[i] = st.row_base[s] + (int32_t)j;
    coef[i] = st.coef[s];
    if (key != ARX_KEY_NONE) {
      atomicMin(&first[key], (int32_t)i);
      if (atomicAdd(&cnt[key], 1) == kHot) {          // exactly one contribution sees this
        const int slot = atomicAdd(&hot[0], 1);
        if (slot < hot_cap) hot[2 + slot] = key;
      }
    }
  }
}

__device__ __forceinline__ float4 fma4(float c, float4 v, float4 a) {
  return make_float4(fmaf(c, v.x, a.x), fmaf(c, v.y, a.y), fmaf(c, v.z, a.z), fmaf(c, v.w, a.w));
}

template <int LPR>
__global__ __launch_bounds__(256) void k_cat_apply(
    float* __restrict__ E, float* __restrict__ acc, float* __restrict__ bias,
    float* __restrict__ bias_acc, int d, const int32_t* __restrict__ keys,
    const int32_t* __restrict__ src, const float* __restrict__ coef, int64_t n,
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ Gb,
    const float* __restrict__ lr_dev, const float* __restrict__ gscale_dev,
    int32_t* __restrict__ first, int32_t* __restrict__ cnt) {
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int lig = lane % LPR;
  const int gid = lane / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t i = wave * GPW + gid;
  if (i >= n) return;
  const int key = keys[i];
  if (key == ARX_KEY_NONE) return;
  if (first[key] != (int32_t)i) return;
  // leader: start the row loads early (they depend on the key only)
  float4 w = make_float4(0, 0, 0, 0), a = w;
  float4* wp = reinterpret_cast<float4*>(E + (int64_t)key * d + col);
  float4* ap = reinterpret_cast<float4*>(acc + (int64_t)key * d + col);
  if (colok) {
    w = *wp;
    a = *ap;
  }
  const int dup = cnt[key];
  if (dup > kHot) return;            // hot row: k_cat_hot sums it with a whole workgroup
  const float c0 = coef[i];
  const int s0 = src[i];
  float4 g = make_float4(0, 0, 0, 0);
  if (colok) g = fma4(c0, *reinterpret_cast<const float4*>(G + (int64_t)s0 * ldg + col), g);
  float gb = Gb ? c0 * Gb[s0] : 0.f;
  if (dup > 1) {
    int found = 1;
    for (int64_t j0 = i + 1; j0 < n && found < dup; j0 += LPR) {
      const int64_t j = j0 + lig;
      const bool hit = (j < n) && (keys[j] == key);
      unsigned long long bal = __ballot(hit);
      unsigned long long bits = bal;
      if constexpr (LPR < 64) bits = (bal >> (gid * LPR)) & ((1ull << LPR) - 1ull);
      while (bits) {
        const int t = __builtin_ctzll(bits);
        bits &= bits - 1;
        const int64_t jj = j0 + t;
        const float cj = coef[jj];
        const int sj = src[jj];
        if (colok) g = fma4(cj, *reinterpret_cast<const float4*>(G + (int64_t)sj * ldg + col), g);
        if (Gb) gb = fmaf(cj, Gb[sj], gb);
        ++found;
      }
    }
  }
  const float lr = *lr_dev;
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  if (colok) {
    g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs;
    a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
    w.x -= lr * g.x / sqrtf(a.x);
    w.y -= lr * g.y / sqrtf(a.y);
    w.z -= lr * g.z / sqrtf(a.z);
    w.w -= lr * g.w / sqrtf(a.w);
    *ap = a;
    *wp = w;
  }
  if (lig == 0) {
    if (bias) {
      const float gg = gb * gs;
      const float ba = bias_acc[key] + gg * gg;
      bias_acc[key] = ba;
      bias[key] -= lr * gg / sqrtf(ba);
    }
    first[key] = INT_MAX;   // leave the aux arrays clean for the next step
    cnt[key] = 0;
  }
}

// One workgroup per hot row: all 256 threads scan the key list (ordered compaction of
// the matching indices into LDS), the sub-groups sum the rows strided, fixed-order
// combine through LDS, one Adagrad application.  The last workgroup to finish clears
// the hot list for the next step.
template <int LPR>
__global__ __launch_bounds__(256) void k_cat_hot(
    float* __restrict__ E, float* __restrict__ acc, float* __restrict__ bias,
    float* __restrict__ bias_acc, int d, const int32_t* __restrict__ keys,
    const int32_t* __restrict__ src, const float* __restrict__ coef, int64_t n,
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ Gb,
    const float* __restrict__ lr_dev, const float* __restrict__ gscale_dev,
    int32_t* __restrict__ first, int32_t* __restrict__ cnt, int32_t* __restrict__ hot,
    int hot_cap) {
  constexpr int NSG = 256 / LPR;
  __shared__ int s_idx[kHotCap];
  __shared__ int s_wcnt[4];
  __shared__ int s_fill;
  __shared__ __attribute__((aligned(16))) float s_g[NSG][LPR * 4];
  __shared__ float s_gb[NSG];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lig = threadIdx.x % LPR, sg = threadIdx.x / LPR;
  const int col = lig * 4;
  const bool colok = col < d;
  int count = hot[0];
  if (count > hot_cap) count = hot_cap;
  for (int h = blockIdx.x; h < count; h += gridDim.x) {
    const int key = hot[2 + h];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gb = 0.f;
    int64_t base = 0;
    int taken = 0;                       // entries consumed so far (keeps the k % NSG split stable)
    while (base < n) {
      if (threadIdx.x == 0) s_fill = 0;
      __syncthreads();
      // fill the LDS list with matching indices, in index order
      while (base < n) {
        const int fill0 = s_fill;
        if (fill0 + 256 > kHotCap) break;
        const int64_t j = base + threadIdx.x;
        const bool hit = (j < n) && (keys[j] == key);
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = fill0;
        for (int w = 0; w < wv; ++w) off += s_wcnt[w];
        if (hit) s_idx[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)j;
        __syncthreads();
        if (threadIdx.x == 0) s_fill = fill0 + s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        base += 256;
        __syncthreads();
      }
      const int m = s_fill;
      // entry k (global order taken + k) belongs to sub-group (taken + k) % NSG
      for (int k = ((sg - taken) % NSG + NSG) % NSG; k < m; k += NSG) {
        const int jj = s_idx[k];
        const float cj = coef[jj];
        const int sj = src[jj];
        if (colok) g = fma4(cj, *reinterpret_cast<const float4*>(G + (int64_t)sj * ldg + col), g);
        if (Gb) gb = fmaf(cj, Gb[sj], gb);
      }
      taken += m;
      __syncthreads();
    }
    if (colok) *reinterpret_cast<float4*>(&s_g[sg][col]) = g;
    if (lig == 0) s_gb[sg] = gb;
    __syncthreads();
    if (sg == 0) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      float tb = 0.f;
      for (int k = 0; k < NSG; ++k) {
        if (colok) {
          const float4 v = *reinterpret_cast<const float4*>(&s_g[k][col]);
          tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
        }
        tb += s_gb[k];
      }
      const float lr = *lr_dev;
      const float gs = gscale_dev ? *gscale_dev : 1.f;
      if (colok) {
        float4* wp = reinterpret_cast<float4*>(E + (int64_t)key * d + col);
        float4* ap = reinterpret_cast<float4*>(acc + (int64_t)key * d + col);
        float4 w = *wp, a = *ap;
        tot.x *= gs; tot.y *= gs; tot.z *= gs; tot.w *= gs;
        a.x += tot.x * tot.x; a.y += tot.y * tot.y; a.z += tot.z * tot.z; a.w += tot.w * tot.w;
        w.x -= lr * tot.x / sqrtf(a.x);
        w.y -= lr * tot.y / sqrtf(a.y);
        w.z -= lr * tot.z / sqrtf(a.z);
        w.w -= lr * tot.w / sqrtf(a.w);
        *ap = a;
        *wp = w;
      }
      if (lig == 0) {
        if (bias) {
          const float gg = tb * gs;
          const float ba = bias_acc[key] + gg * gg;
          bias_acc[key] = ba;
          bias[key] -= lr * gg / sqrtf(ba);
        }
        first[key] = INT_MAX;
        cnt[key] = 0;
      }
    }
    __syncthreads();
  }
  // last workgroup out clears the list (every workgroup has read hot[0] by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&hot[1], 1) == (int)gridDim.x - 1) {
      hot[0] = 0;
      hot[1] = 0;
    }
  }
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

extern "C" {

int arx_sparse_adagrad_cat(float* E, float* acc, float* bias, float* bias_acc, int64_t table_rows,
                           int d, int nsites, const int32_t* const* site_cat_map,
                           const int32_t* const* site_ids, const int64_t* site_n,
                           const int32_t* site_row_base, const float* site_coef,
                           const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                           const float* gscale_dev, int32_t* aux_first, int32_t* aux_cnt,
                           int32_t* aux_hot, int64_t aux_hot_len, int32_t* keys_buf,
                           int32_t* src_buf, float* coef_buf, int mode, void* workspace,
                           size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(E && G && lr_dev && aux_first && aux_cnt && aux_hot && keys_buf && src_buf && coef_buf,
                "arx_sparse_adagrad_cat: null pointer");
  ARX_CHECK_ARG(acc || (mode == 0 || mode == 0x10 || mode == 0x20),
                "arx_sparse_adagrad_cat: acc == NULL (gradient descent) needs the sorted pass (mode 0)");
  ARX_CHECK_ARG(aux_hot_len >= 3, "arx_sparse_adagrad_cat: aux_hot too short");
  ARX_CHECK_ARG(nsites > 0 && nsites <= kMaxSites, "arx_sparse_adagrad_cat: 1..8 lookup sites");
  ARX_CHECK_ARG(acc ? (bias == nullptr) == (bias_acc == nullptr) : bias_acc == nullptr,
                "arx_sparse_adagrad_cat: bias/bias_acc");
  ARX_CHECK_ARG(!(bias && !Gb), "arx_sparse_adagrad_cat: bias table given without Gb");
  if (d <= 0 || d % 4 != 0 || d > 256) {
    set_error("arx_sparse_adagrad_cat: d=%d unsupported (d %% 4 == 0, d <= 256)", d);
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_ARG(ldg % 4 == 0 && ldg >= d, "arx_sparse_adagrad_cat: bad ldg");
  CatSites st;
  st.nsites = nsites;
  st.offs[0] = 0;
  for (int s = 0; s < kMaxSites; ++s) {
    const bool live = s < nsites;
    st.cat_map[s] = live ? site_cat_map[s] : nullptr;
    st.ids[s] = live ? site_ids[s] : nullptr;
    st.row_base[s] = live ? site_row_base[s] : 0;
    st.coef[s] = live ? site_coef[s] : 0.f;
    st.offs[s + 1] = st.offs[s] + (live ? site_n[s] : 0);
    if (live) ARX_CHECK_ARG(site_ids[s] && site_n[s] >= 0, "arx_sparse_adagrad_cat: bad site");
  }
  const int64_t n = st.offs[nsites];
  if (n == 0) return ARX_OK;
  ARX_CHECK_ARG(n < (int64_t)INT_MAX, "arx_sparse_adagrad_cat: too many contributions");
  hipStream_t s = as_stream(stream);
  if (mode == 0 || mode == 0x10 || mode == 0x20) {   // default: key generation + sort (LDS rank sort
    // or device radix) + apply; 0x10 / 0x20: only the first / the second half (the sort depends on
    // the ids alone and may run ahead on another stream; same workspace for both halves)
    const int phase = mode == 0 ? 3 : (mode == 0x10 ? 1 : 2);
    TableSet ts = {};
    ts.E[0] = E;
    ts.acc[0] = acc;
    ts.bias[0] = bias;
    ts.bias_acc[0] = bias_acc;
    ts.cnt[0] = aux_cnt;
    int kb = 1;
    while ((1ll << kb) < table_rows && kb < 30) ++kb;
    ts.kb = kb;
    st.kb = kb;
    st.nextra = 0;
    st.xoffs[0] = 0;
    for (int q = 0; q < kMaxSites; ++q) st.table[q] = 0;
    for (int q = 0; q < kMaxTables; ++q) st.rows[q] = table_rows;
    return sparse_adagrad_sites_sorted(ts, 1, d, st, G, ldg, Gb, lr_dev, gscale_dev, keys_buf, src_buf,
                                       coef_buf, workspace, workspace_bytes, s, phase);
  }
  {
    int64_t g = ceil_div(n, 256);
    int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    k_cat_keys<<<(int)g, 256, 0, s>>>(st, keys_buf, src_buf, coef_buf, aux_first, aux_cnt, table_rows,
                                      aux_hot, (int)(aux_hot_len - 2));
    ARX_CHECK_LAUNCH();
  }
  const int lpr = lanes_per_row(d);
  const int64_t nwaves = ceil_div(n, 64 / lpr);
  const int grid = (int)ceil_div(nwaves, 4);
  const float* gb_in = bias ? Gb : nullptr;
  ARX_DISPATCH_LPR(lpr, (k_cat_apply<LPR><<<grid, 256, 0, s>>>(
                            E, acc, bias, bias_acc, d, keys_buf, src_buf, coef_buf, n, G, ldg, gb_in,
                            lr_dev, gscale_dev, aux_first, aux_cnt)));
  ARX_CHECK_LAUNCH();
  ARX_DISPATCH_LPR(lpr, (k_cat_hot<LPR><<<kHotBlocks, 256, 0, s>>>(
                            E, acc, bias, bias_acc, d, keys_buf, src_buf, coef_buf, n, G, ldg, gb_in,
                            lr_dev, gscale_dev, aux_first, aux_cnt, aux_hot, (int)(aux_hot_len - 2))));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}


static int cat_multi_impl(int phase, int ntables, float* const* E, float* const* acc, float* const* bias,
                                 float* const* bias_acc, const int64_t* table_rows,
                                 int32_t* const* aux_cnt, int d, int nsites,
                                 const int32_t* site_table, const int32_t* const* site_cat_map,
                                 const int32_t* const* site_ids, const int64_t* site_n,
                                 const int32_t* site_row_base, const float* site_coef, const float* G,
                                 int64_t ldg, const float* Gb, const float* lr_dev,
                                 const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                 float* coef_buf, int nextra, const int64_t* extra_n,
                                 const int32_t* extra_table, void* workspace,
                                 size_t workspace_bytes, void* stream, const BagStage* bag = nullptr) {
  ARX_CHECK_ARG(ntables >= 1 && ntables <= kMaxTables, "arx_sparse_adagrad_cat_multi: 1..4 tables");
  ARX_CHECK_ARG(nsites >= 0 && nsites <= kMaxSites && nextra >= 0 && nextra <= kMaxSites &&
                    nsites + nextra > 0,
                "arx_sparse_adagrad_cat_multi: 0..8 one-hot sites, 0..8 pre-expanded segments");
  ARX_CHECK_ARG(nextra == 0 || (extra_n && extra_table), "arx_sparse_adagrad_cat_multi: extra arrays");
  ARX_CHECK_ARG(E && acc && bias && bias_acc && table_rows && aux_cnt && site_table && site_cat_map &&
                    site_ids && site_n && site_row_base && site_coef && G && lr_dev && keys_buf &&
                    src_buf && coef_buf,
                "arx_sparse_adagrad_cat_multi: null pointer");
  if (d <= 0 || d % 4 != 0 || d > 256) {
    set_error("arx_sparse_adagrad_cat_multi: d=%d unsupported (d %% 4 == 0, d <= 256)", d);
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_ARG(ldg % 4 == 0 && ldg >= d, "arx_sparse_adagrad_cat_multi: bad ldg");
  TableSet ts = {};
  CatSites st;
  int kb = 1;
  bool all_cnt = true, any_cnt = false;
  for (int t = 0; t < kMaxTables; ++t) {
    const bool live = t < ntables;
    if (live && t == 0 && bag && !E[0]) {
      // virtual table: the entity ids of the riding bag table, no rows of its own to update
      ARX_CHECK_ARG(table_rows[0] > 0 && !acc[0] && !bias[0] && !bias_acc[0] && ntables >= 2,
                    "arx_sparse_adagrad_cat_multi_bags: a virtual table 0 has no acc / bias and needs a real table beside it");
      while ((1ll << kb) < table_rows[0] && kb < 30) ++kb;
    } else if (live) {
      ARX_CHECK_ARG(E[t] && table_rows[t] > 0, "arx_sparse_adagrad_cat_multi: bad table");
      ARX_CHECK_ARG(acc[t] ? (bias[t] == nullptr) == (bias_acc[t] == nullptr) : bias_acc[t] == nullptr,
                    "arx_sparse_adagrad_cat_multi: bias/bias_acc");
      ARX_CHECK_ARG(!(bias[t] && !Gb), "arx_sparse_adagrad_cat_multi: bias table given without Gb");
      while ((1ll << kb) < table_rows[t] && kb < 30) ++kb;
      all_cnt = all_cnt && aux_cnt[t] != nullptr;
      any_cnt = any_cnt || aux_cnt[t] != nullptr;
    }
    ts.E[t] = live ? E[t] : nullptr;
    ts.acc[t] = live ? acc[t] : nullptr;
    ts.bias[t] = live ? bias[t] : nullptr;
    ts.bias_acc[t] = live ? bias_acc[t] : nullptr;
    ts.cnt[t] = live ? aux_cnt[t] : nullptr;
    st.rows[t] = live ? table_rows[t] : 0;
  }
  ARX_CHECK_ARG(all_cnt || !any_cnt, "arx_sparse_adagrad_cat_multi: aux_cnt for all tables or none");
  ts.kb = kb;
  st.kb = kb;
  st.nsites = nsites;
  st.offs[0] = 0;
  for (int q = 0; q < kMaxSites; ++q) {
    const bool live = q < nsites;
    st.cat_map[q] = live ? site_cat_map[q] : nullptr;
    st.ids[q] = live ? site_ids[q] : nullptr;
    st.row_base[q] = live ? site_row_base[q] : 0;
    st.coef[q] = live ? site_coef[q] : 0.f;
    st.table[q] = live ? site_table[q] : 0;
    st.offs[q + 1] = st.offs[q] + (live ? site_n[q] : 0);
    if (live)
      ARX_CHECK_ARG((site_ids[q] || site_n[q] == 0) && site_n[q] >= 0 && site_table[q] >= 0 &&
                        site_table[q] < ntables,
                    "arx_sparse_adagrad_cat_multi: bad site");
  }
  st.nextra = nextra;
  st.xoffs[0] = 0;
  for (int q = 0; q < kMaxSites; ++q) {
    const bool live = q < nextra;
    st.xtable[q] = live ? extra_table[q] : 0;
    st.xoffs[q + 1] = st.xoffs[q] + (live ? extra_n[q] : 0);
    if (live)
      ARX_CHECK_ARG(extra_n[q] >= 0 && extra_table[q] >= 0 && extra_table[q] < ntables,
                    "arx_sparse_adagrad_cat_multi: bad pre-expanded segment");
  }
  const int64_t n = st.offs[nsites] + st.xoffs[nextra];
  if (n == 0) return ARX_OK;
  ARX_CHECK_ARG(n < (int64_t)INT_MAX, "arx_sparse_adagrad_cat_multi: too many contributions");
  return sparse_adagrad_sites_sorted(ts, ntables, d, st, G, ldg, Gb, lr_dev, gscale_dev, keys_buf,
                                     src_buf, coef_buf, workspace, workspace_bytes, as_stream(stream), phase, bag);
}

int arx_sparse_adagrad_cat_multi(int ntables, float* const* E, float* const* acc, float* const* bias,
                                 float* const* bias_acc, const int64_t* table_rows,
                                 int32_t* const* aux_cnt, int d, int nsites,
                                 const int32_t* site_table, const int32_t* const* site_cat_map,
                                 const int32_t* const* site_ids, const int64_t* site_n,
                                 const int32_t* site_row_base, const float* site_coef, const float* G,
                                 int64_t ldg, const float* Gb, const float* lr_dev,
                                 const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                 float* coef_buf, int nextra, const int64_t* extra_n,
                                 const int32_t* extra_table, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return cat_multi_impl(3, ntables, E, acc, bias, bias_acc, table_rows, aux_cnt, d, nsites, site_table,
                        site_cat_map, site_ids, site_n, site_row_base, site_coef, G, ldg, Gb, lr_dev,
                        gscale_dev, keys_buf, src_buf, coef_buf, nextra, extra_n, extra_table, workspace,
                        workspace_bytes, stream);
}

int arx_sparse_adagrad_cat_multi_phase(int phase, int ntables, float* const* E, float* const* acc,
                                       float* const* bias, float* const* bias_acc,
                                       const int64_t* table_rows, int32_t* const* aux_cnt, int d,
                                       int nsites, const int32_t* site_table,
                                       const int32_t* const* site_cat_map,
                                       const int32_t* const* site_ids, const int64_t* site_n,
                                       const int32_t* site_row_base, const float* site_coef,
                                       const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                                       const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                       float* coef_buf, int nextra, const int64_t* extra_n,
                                       const int32_t* extra_table, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(phase == 1 || phase == 2 || phase == 3, "arx_sparse_adagrad_cat_multi_phase: phase 1, 2 or 3");
  return cat_multi_impl(phase, ntables, E, acc, bias, bias_acc, table_rows, aux_cnt, d, nsites, site_table,
                        site_cat_map, site_ids, site_n, site_row_base, site_coef, G, ldg, Gb, lr_dev,
                        gscale_dev, keys_buf, src_buf, coef_buf, nextra, extra_n, extra_table, workspace,
                        workspace_bytes, stream);
}

int arx_sparse_adagrad_cat_multi_bags_csc(int phase, int ntables, float* const* E, float* const* acc,
                                      float* const* bias, float* const* bias_acc,
                                      const int64_t* table_rows, int32_t* const* aux_cnt, int d,
                                      int nsites, const int32_t* site_table,
                                      const int32_t* const* site_cat_map,
                                      const int32_t* const* site_ids, const int64_t* site_n,
                                      const int32_t* site_row_base, const float* site_coef,
                                      const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                                      const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                      float* coef_buf, void* workspace, size_t workspace_bytes,
                                      float* bag_E, float* bag_acc, float* bag_bias, float* bag_bias_acc,
                                      int64_t bag_rows, const int32_t* vals, const int32_t* starts,
                                      const int32_t* lens, int max_len, int32_t* bag_aux_cnt,
                                      void* bag_workspace, size_t bag_workspace_bytes,
                                          const int32_t* csc_qpos, const int32_t* csc_qte, uint8_t* csc_flags,
                                          int32_t* csc_slot_of, int64_t csc_nq, void* stream) {
  const int rider = (phase & 0x100) ? 1 : (phase & 0x200) ? 0 : -1;     // this pass's apply form (arx.h)
  phase &= 0xff;
  ARX_CHECK_ARG((phase >= 1 && phase <= 3) || (phase >= 5 && phase <= 8),
                "arx_sparse_adagrad_cat_multi_bags_csc: phase 1, 2, 3 or the quarter phases 5 .. 8");
  ARX_CHECK_ARG(bag_E && vals && starts && lens && bag_rows > 0 && max_len > 0 && bag_workspace,
                "arx_sparse_adagrad_cat_multi_bags_csc: null pointer / bad sizes");
  ARX_CHECK_ARG(bag_acc ? (bag_bias == nullptr) == (bag_bias_acc == nullptr) : bag_bias_acc == nullptr,
                "arx_sparse_adagrad_cat_multi_bags_csc: bag bias and bias_acc go together");
  ARX_CHECK_ARG(!(bag_bias && !Gb), "arx_sparse_adagrad_cat_multi_bags_csc: bag bias given without Gb");
  ARX_CHECK_ARG(!csc_qpos || (csc_qte && csc_flags && csc_slot_of && csc_nq >= 0 && csc_nq < (1ll << 31)),
                "arx_sparse_adagrad_cat_multi_bags_csc: incomplete static token order");
  const int64_t csc_fb = (csc_nq + 255) / 256 * 256;     // flag bytes, then their coarse bytes (one per 16)
  BagCsc csc = {csc_qpos, reinterpret_cast<const int2*>(csc_qte), csc_flags, csc_flags ? csc_flags + csc_fb : nullptr,
                csc_slot_of, csc_nq};
  BagStage bag = {bag_E, bag_acc, bag_bias, bag_bias_acc, bag_rows, vals, starts, lens, max_len, bag_aux_cnt,
                  bag_workspace, bag_workspace_bytes, csc_qpos ? &csc : nullptr, rider};
  return cat_multi_impl(phase, ntables, E, acc, bias, bias_acc, table_rows, aux_cnt, d, nsites, site_table,
                        site_cat_map, site_ids, site_n, site_row_base, site_coef, G, ldg, Gb, lr_dev,
                        gscale_dev, keys_buf, src_buf, coef_buf, 0, nullptr, nullptr, workspace,
                        workspace_bytes, stream, &bag);
}


int arx_sparse_adagrad_cat_multi_bags(int phase, int ntables, float* const* E, float* const* acc,
                                      float* const* bias, float* const* bias_acc,
                                      const int64_t* table_rows, int32_t* const* aux_cnt, int d,
                                      int nsites, const int32_t* site_table,
                                      const int32_t* const* site_cat_map,
                                      const int32_t* const* site_ids, const int64_t* site_n,
                                      const int32_t* site_row_base, const float* site_coef,
                                      const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                                      const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                      float* coef_buf, void* workspace, size_t workspace_bytes,
                                      float* bag_E, float* bag_acc, float* bag_bias, float* bag_bias_acc,
                                      int64_t bag_rows, const int32_t* vals, const int32_t* starts,
                                      const int32_t* lens, int max_len, int32_t* bag_aux_cnt,
                                      void* bag_workspace, size_t bag_workspace_bytes, void* stream) {
  return arx_sparse_adagrad_cat_multi_bags_csc(phase, ntables, E, acc, bias, bias_acc, table_rows, aux_cnt, d, nsites,
                                               site_table, site_cat_map, site_ids, site_n, site_row_base, site_coef,
                                               G, ldg, Gb, lr_dev, gscale_dev, keys_buf, src_buf, coef_buf, workspace,
                                               workspace_bytes, bag_E, bag_acc, bag_bias, bag_bias_acc, bag_rows, vals,
                                               starts, lens, max_len, bag_aux_cnt, bag_workspace, bag_workspace_bytes,
                                               nullptr, nullptr, nullptr, nullptr, 0, stream);
}

}  // extern "C"
