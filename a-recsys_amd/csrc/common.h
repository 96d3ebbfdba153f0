// common.h -- shared helpers for libarx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "arx.h"

namespace arx {

void set_error(const char* fmt, ...);

constexpr int kWave = 64;  // CDNA wavefront

#define ARX_CHECK_ARG(cond, ...)                  \
  do {                                            \
    if (!(cond)) {                                \
      ::arx::set_error(__VA_ARGS__);              \
      return ARX_EINVAL;                          \
    }                                             \
  } while (0)

#define ARX_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::arx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                       __FILE__, __LINE__);                                        \
      return ARX_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define ARX_CHECK_LAUNCH()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      ::arx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                        \
      return ARX_EHIP;                                                             \
    }                                                                              \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// lanes of a wave that cooperate on one table row of d floats (float4 per lane)
static inline int lanes_per_row(int d) {
  int q = d / 4;
  int l = 1;
  while (l < q && l < 64) l <<= 1;
  return l;
}

// number of CUs of the current device (cached)
int cu_count();

// One-hot lookup sites of a table (arx_sparse_adagrad_cat): site s contributes, for
// j < offs[s+1]-offs[s], key = cat_map[s] ? cat_map[s][ids[s][j]] : ids[s][j], gradient
// row row_base[s] + j, factor coef[s].
constexpr int kMaxSites = 8;
constexpr int kMaxTables = 4;
struct CatSites {
  int nsites;
  int64_t offs[kMaxSites + 1];
  const int32_t* cat_map[kMaxSites];
  const int32_t* ids[kMaxSites];
  int32_t row_base[kMaxSites];
  float coef[kMaxSites];
  int32_t table[kMaxSites];        // which table of the TableSet the site updates
  int64_t rows[kMaxTables];        // rows of each table (key validation)
  int kb;                          // sort key = (table << kb) | row
  // pre-expanded contributions behind the one-hot ones (multi-hot sites: arx_csr_expand wrote
  // table-local keys, source rows and coefficients at keys_buf[offs[nsites] + xoffs[e] ...));
  // k_site_keys only adds the table prefix
  int nextra;
  int64_t xoffs[kMaxSites + 1];
  int32_t xtable[kMaxSites];
};

// Build-defined sampled softmax 'mce' (arx.h): loss_r = log(1 + sum_s m_rs e_rs), backward weight w_rs = g_r e_rs /
// (1 + s_r), with e_rs = exp(min(x_rs - t_r, kMceSat)).  Below the cap this is the plain sampled softmax; the cap keeps
// every path finite whatever the logits (e^64 * 65 536 columns < FLT_MAX) -- round-5 advisor: the fused family anchors
// its exponent at t_r and overflowed at x - t > 88.7, poisoning the tables with NaN.  (NaN inputs stay NaN.)
constexpr float kEmptySlotBias = -1e30f;     // bias of an EMPTY (-1) pool slot in a lookup's bias output (gather.hip)
constexpr float kMceSat = 64.f;
__device__ __forceinline__ float mce_sat(float x, float t) {
  const float hi = t + kMceSat;
  return x > hi ? hi : x;
}


// Tables updated by ONE sparse-Adagrad pass.  Several tables of equal width share a sort and
// an apply launch (the per-table kernel chains are launch-bound: 12 launches of ~5 us per
// table at B=16384): the sort key carries the table index above the row bits.
struct TableSet {
  float* E[kMaxTables];
  float* acc[kMaxTables];
  float* bias[kMaxTables];         // null: table without bias
  float* bias_acc[kMaxTables];
  int32_t* cnt[kMaxTables];        // per-row arrival counters (ticket apply), may be null
  int kb;                          // row bits of the key
};

// optim.hip: key generation + single-workgroup sort + the two Adagrad passes
// (n <= 16384 contributions); returns ARX_* codes.
int sparse_adagrad_sites_sorted(const TableSet& ts, int ntables, int d, const CatSites& st,
                                const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                                const float* gscale_dev, int32_t* keys_buf, int32_t* src_buf,
                                float* coef_buf, void* workspace, size_t workspace_bytes,
                                hipStream_t s, int phase = 3,    // 1: keys + sort, 2: apply, 3: both
                                const struct BagStage* bag = nullptr);

// A multi-hot table riding on a one-hot pass (arx_sparse_adagrad_cat_multi_bags): the lookups of
// one-hot table 0 of the pass are also the entity lookups of this table's bags.
struct BagCsc;
struct BagStage {
  float* E;
  float* acc;
  float* bias;
  float* bias_acc;
  int64_t rows;                    // token rows of the multi-hot table
  const int32_t* vals;             // CSR of the bags: tokens of entity e at vals[starts[e] .. + lens[e])
  const int32_t* starts;
  const int32_t* lens;
  int max_len;
  int32_t* aux_cnt;                // per-token-row ticket counters (may be null)
  void* ws;                        // arx_sparse_adagrad_bags_workspace_bytes(lookups of table 0, max_len, d)
  size_t ws_bytes;
  const BagCsc* csc;               // static token order of the bags (csc.hip), or null: sort the tokens every step
  int rider;                       // form of the apply: -1 the process default (ARX_K7_RIDER), 0 win, 1 split
};

// gemm_nt.hip: logits GEMM with the A operand register-resident (K in {32,64,128});
// ARX_EUNSUPPORTED for any other shape / alignment.
int gemm_nt_smallk(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                   const float* B, int64_t ldb, float* C, int64_t ldc, const float* col_bias,
                   hipStream_t s);

// Reduce scratch (arx_reduce_scratch_bytes): the caller's memory for the deterministic one-launch reductions
// (norms, running arg-max) -- [0] arrival ticket (zero between calls), [64 ..) block partials.
constexpr size_t kReduceScratchBytes = 64 + 512 * 16;
inline unsigned int* reduce_scratch_ticket(void* p) { return reinterpret_cast<unsigned int*>(p); }
inline float* reduce_scratch_f32(void* p) { return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + 64); }
bool bx6_enabled();      // true unless ARX_SCORER_F32 is set (read once)
// gemm_dma.hip: NN / TN GEMMs with N <= 128 (dU, dI): LDS-DMA streamed operands, dL read once.
bool gemm_dma_supported(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                        int64_t lda, const float* B, int64_t ldb);
void gemm_dma_plan(int64_t M, int64_t N, int64_t K, int* bm, int* splits, int64_t* kchunk);
int gemm_dma_launch(int transA, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                    int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc,
                    const float* col_bias, float* partial, int bm, int splits, int64_t kchunk,
                    float* a_rowsum, float* rowsum_partial, hipStream_t s);
int gemm_dma_launch_tn_pair(int64_t M, int64_t N, int n1, int64_t K, const float* A, int64_t lda,
                            const float* B1, int64_t ldb1, const float* B2, int64_t ldb2,
                            int64_t shift, float* partial, int bm, int splits, int64_t kchunk,
                            float* a_rowsum, float* rowsum_partial, hipStream_t s);

// topk.hip: radix-select top-k of every row (k <= 1024); indices are offset by idx_base.
int topk_select_launch(const float* logits, int64_t ld, int64_t B, int64_t V, int k, int32_t idx_base,
                       float* values, int32_t* indices, hipStream_t s);

// radix_sort.hip: graph-safe stable LSD sort of (key, src, coef) triples, n > 8192.
// keys_raw: caller keys (ARX_KEY_NONE / out-of-range -> sentinel); src_raw/coef_raw may be
// null (identity / 1.0).  *_tmp: ping-pong buffers of n entries; hist: radix_sort_hist_bytes().
// list_count (optional): 512 ints (2 KB) zeroed by the first launch.
// n_live (optional, device int32): the first pass drops sentinel entries (pads, invalid keys)
// and writes the number of survivors here; later passes -- and the caller's apply kernels --
// work on that many entries (the grids stay sized for n).
size_t radix_sort_hist_bytes();
int launch_radix_sort(const int32_t* keys_raw, const int32_t* src_raw, const float* coef_raw, int64_t n,
                      uint32_t sentinel, int total_bits, uint32_t* keys_tmp, uint32_t* keys_out,
                      int32_t* src_tmp, int32_t* src_out, float* coef_tmp, float* coef_out,
                      int32_t* hist, int32_t* list_count, int32_t* n_live, hipStream_t s,
                      const int32_t* n_in_dev = nullptr /* first pass: device-side number of input entries */,
                      const CatSites* sites = nullptr /* first pass: form (key, src, coef) from these lookups (keys_raw,
                                                         src_raw, coef_raw unused; no pre-expanded segments) */);

}  // namespace arx
