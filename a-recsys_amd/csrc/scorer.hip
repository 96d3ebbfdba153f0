// scorer.hip -- the sampled-loss scorer of a training step on the bf16 matrix pipe, f32-exact (round 4).
//
//   logits = U . P^T + b (embed_attribute.py:171,188-193), WMRB 'mw' (:641-649), target score (:208-220) and the two
//   backward products dU = dL . P, dP = dL^T . U (tf.gradients of the same lines) -- without [B, S] logits or
//   dlogits in HBM.  An f32 value is EXACTLY three bf16 pieces (8 + 8 + 8 mantissa bits), a bf16 x bf16 product
//   is exact in f32, and of the nine piece products the three smallest are below 2^-26 of the product: six
//   v_mfma_f32_32x32x16_bf16 per k-chunk with f32 accumulation (small terms in their own accumulator) carry
//   every bit an f32 multiply-add chain carries (three where one operand is the 0/1 hinge-activity matrix).
//
// What round 3's kernels (gemm_bx6.hip) got wrong, measured with tools/probe/mfma_issue.hip on the part
// (profiles/r04_mfma_issue_probe.txt): (1) a wave can issue <= 6 plain VALU or one ds_read_b128 per MFMA gap for
// free, the accumulator pattern does not matter, packed-f32 VALU costs +18 cycles each -- the matrix pipe was
// never the limit; (2) every workgroup streamed ALL planes of the small operand (768 KB) through its CU: 196 MB of
// L2 -> LDS traffic per launch, and the loaders prefetched ONE stage ahead of a ~2 600-cycle L2 round trip.
// This file: 128 x 512 (forward) and 256 x 32 (backward) workgroup tiles -- 2x / 4x less plane traffic --, LDS-DMA
// loader waves (global_load_lds_dwordx4: no staging registers) three stages ahead in a four-slot ring, the hinge
// epilogue of tile j - 1 interleaved with the MFMAs of tile j, the target score formed by the half waves of k_sc_prep
// that list a row's positives (round 4: in the forward kernel's prologue), and every operand plane produced by the
// kernel that has the data in registers anyway (k_sc_prep: pool rows in both layouts; k_sc_rows: g U transposed) --
// no split launches.
//
// Round 5, k_sc_hinge 28.2 -> 24.4 us (0.38 -> 0.42 of 2500 / 6; the C4 shape 56.4 -> 49.9): its start is bound by the
// bytes 256 workgroups ask for at once.  The T rows and the dot moved to k_sc_prep (+1.7 us there at the C3 shape, 0
// at C4's), the pieces of U are made chunk by chunk between the MFMAs of tile 0 instead of in front of them, loads are
// issued in cache-line order, and the two column splits of a row block run on ONE XCD (blockIdx mapping).  Measured
// on the way, with UNTRACED builds timed by rocprofv3 (tools/r05_sc_ab.sh) -- the cycle stamps of tools/sc_trace.py
// slow the stamped kernel by 8 us and mis-attribute it (a traced build made the bias re-reads look like a third of the
// loop): no MFMA 9.6 us, no LDS-DMA -2.3, no barriers -0.8, no re-initialising writes +-0, the bias through
// v_readlane instead of LDS +0.6, U pre-split by k_sc_prep 24.3 but +3 us in k_sc_prep.  The part runs this kernel at
// 1.85 - 1.95 GHz (s_memtime against s_memrealtime), not the 2.4 the guide's peak assumes.  s_setprio 3 in the compute
// waves: +-0.  k_sc_prep by role (blocks of the others returning at once): pool planes 5.0 us, hit lists + target scores
// 8.7 (a chain of four dependent loads per row), together 10.1.
//
// Launches of a step:  k_sc_prep -> k_sc_hinge -> k_sc_rows          (arx_mw_scorer_fwd)
//                      k_sc_bits (dU += g (act . P))                 (arx_mw_scorer_bwd_du)
//                      k_sc_bits (slices of act^T . (g U)) -> k_sc_tn_reduce   (arx_mw_scorer_bwd_di)
#include <stdlib.h>

#include "common.h"
#include "posmask.h"

namespace arx {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Ablation builds (tools/build_variant.sh <name> -DSC_ABL_...; WRONG results, the clock is what counts):
//   SC_ABL_NODMA   loaders issue no LDS-DMA (compute reads whatever is in LDS)
//   SC_ABL_NOEPI   the forward kernel's hinge epilogue is compiled out
//   SC_ABL_NOBAR   no per-stage barriers (loaders and compute waves run free)
//   SC_ABL_NOMFMA  the MFMAs are compiled out (reads, epilogue and barriers stay)
//   SC_TRACE       cycle stamps (s_memtime) of one compute wave of workgroup 5: arx_sc_trace_read (tools/sc_trace.py)
//   MC_ABL_NODMA / MC_ABL_NOEXP / MC_ABL_NOBAR   the same for k_mc_flow (SC_ABL_NOMFMA applies to it too); MC_SLOTS=3: a
//                  three-slot LDS ring (measured +-0)
#ifndef SC_TRACE_CHUNKS
#define SC_TRACE_CHUNKS 0            // 1: a stamp after every chunk (perturbs: each stamp drains the LDS counter)
#endif
#if defined(SC_TRACE) || defined(SC_TRACE_ROWS)
__device__ unsigned long long g_sc_trace[4096];
#define SC_T(ev_)                                                                                        \
  if (blockIdx.x == (gridDim.x > 5 ? 5 : 0) && threadIdx.x == 0 && tcount < 4096)                        \
    g_sc_trace[tcount++] = ((unsigned long long)(ev_) << 56) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFull);
#define SC_TDECL int tcount = 0;
#define SC_TREAL(ev_)                 /* the constant 100 MHz counter: shader clock = cycle stamps / these */ \
  if (blockIdx.x == (gridDim.x > 5 ? 5 : 0) && threadIdx.x == 0 && tcount < 4096)                        \
    g_sc_trace[tcount++] = ((unsigned long long)(ev_) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFFFFFull);
#else
#define SC_TREAL(ev_)
#define SC_T(ev_)
#define SC_TDECL
#endif
#ifdef SC_TRACE_ROWS                  // the same stamps in k_sc_rows (phases of one wave of workgroup 5)
#define SC_TR(ev_) SC_T(ev_)
#else
#define SC_TR(ev_)
#endif
constexpr int kScHits = 8;            // pool slots of a row's positives kept in the hit list (more: the row kernel walks)
constexpr int kScSlots = 4;           // LDS ring: the stage being read + three in flight
constexpr int kScAhead = 3;
constexpr int kScStageBytes = 3 * 32 * 256;      // one stage = 3 planes x 32 operand rows x 256 bytes (24 KB)

// (x, y) -> their three bf16 pieces, packed {lo: x, hi: y} per piece (the layout of two consecutive k of an MFMA
// operand): v_cvt_pk_bf16_f32 rounds to nearest even, both residual subtractions are exact in f32.  7.5 VALU per
// value pair and piece instead of ~20 for the integer rounding sequence (the forward kernel's prologue splits 64 values
// per lane: 2.8 us of its 27)
typedef __bf16 sc_bf2 __attribute__((ext_vector_type(2)));
typedef float sc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x2(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((sc_f2){x, y}, sc_bf2));
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xFFFF0000u);
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((sc_f2){rx, ry}, sc_bf2));
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xFFFF0000u);
  p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector((sc_f2){sx, sy}, sc_bf2));
}

#ifdef SC_ABL_NOBAR
__device__ __forceinline__ void sc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define SC_RAW_BARRIER() asm volatile("s_nop 0" ::: "memory")
#else
__device__ __forceinline__ void sc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define SC_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")
#endif
#ifdef SC_ABL_NOMFMA
#define SC_MFMA(a_, b_, c_) (c_)
#else
#define SC_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#endif

// 16-byte chunk position of chunk q of operand row r inside an LDS stage image (rows of ROWB bytes, unpadded: the
// LDS-DMA writes lane-linear): XOR swizzle such that the 16 lanes a ds_read_b128 services per cycle hit 16 slots
template <int ROWB>
__device__ __forceinline__ int sc_swz(int r) {
  return ROWB == 256 ? (r & 15) : ((r >> 1) & 7);
}

// ------------------------------------------------------------------------------------------------------------
// Loader waves: one stage = 24 pieces of 1 KiB (64 lanes x 16 B, LDS-DMA), six per loader wave.  The image of a
// stage: [tile][plane][32 rows][ROWB bytes]; lane -> (row of its piece, chunk position), fetching the chunk the
// swizzle puts there.  `src` is this lane's byte pointer for piece i of stage 0 without the stage advance.
// ------------------------------------------------------------------------------------------------------------
template <int ROWB>
struct ScLoader {
  const char* src[6];                // per piece: plane / tile / row / chunk resolved
  int64_t stage_step;                // bytes between consecutive stages in the source
  uint32_t dst[6];                   // LDS byte offset of the piece inside a stage image (wave-uniform)
  const char* srcx = nullptr;        // a seventh piece per stage (k_sc_bits: one 256-word row piece of the act bits)
  int64_t stepx = 0;

  // planes: [3][rows_total][row_bytes]; a stage = TPS tiles of 32 consecutive rows (TPS * ROWB == 256 ... see callers)
  __device__ __forceinline__ void init(const char* planes, int64_t plane_bytes, int64_t row_bytes, int64_t row0,
                                       int64_t col_byte0, int lw, int lane, int64_t rows_total) {
    constexpr int RP = 1024 / ROWB;            // rows per piece
    constexpr int PPT = 32 / RP;               // pieces per (tile, plane)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int p = lw * 6 + i;
      const int pt = p / PPT, rg = p % PPT;
      const int t = pt / 3, pl = pt % 3;
      const int row = rg * RP + lane / (ROWB / 16), pos = lane % (ROWB / 16);
      const int q = pos ^ sc_swz<ROWB>(row);
      int64_t r = row0 + t * 32 + row;
      if (r > rows_total - 1) r = rows_total - 1;          // (clamped: such rows are never used)
      src[i] = planes + pl * plane_bytes + r * row_bytes + col_byte0 + q * 16;
      dst[i] = (uint32_t)(p * 1024);
    }
  }
  template <bool EXTRA = false>
  __device__ __forceinline__ void issue(char* lds_stage, int64_t st, char* lds_x = nullptr) const {
#ifdef SC_ABL_NODMA
    return;
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + st * stage_step),
                                       (__attribute__((address_space(3))) void*)(lds_stage + dst[i]), 16, 0, 0);
    if (EXTRA)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + st * stepx),
                                       (__attribute__((address_space(3))) void*)lds_x, 16, 0, 0);
  }
};

// The loader waves' loop.  Ring of kScSlots stage images; protocol with the compute waves (one s_barrier per stage):
//   B0     stage 0 has landed (and whatever the workgroup staged in LDS itself)
//   E(st)  the compute waves arrive in the MIDDLE of stage st: they are done with stage st - 1 (its slot may be
//          overwritten: stage st + 3 goes there) and may read stage st + 1 from now on (the loaders waited for it) --
//          so the first fragments of stage st + 1 are requested under the last MFMAs of stage st, never behind a
//          barrier.
// EXTRA: a seventh piece per stage and loader wave lands at ldsx + slot * 4096 + lw * 1024.
// (Round 5, measured and not kept: k_sc_hinge's loaders requesting only stage 0 in front of B0 and stages 1 / 2 behind
// it -- the start of that kernel is bound by what all 256 workgroups ask of the memory system at once, ~7 TB/s -- 24.3
// against 24.4 us; the loaders forming the target scores from whole-row loads of their own under the first stages:
// the second copy of U's bytes cost more than the compute waves' strided T loads had.)
template <int ROWB, bool EXTRA = false>
__device__ __forceinline__ void sc_loader_loop(const ScLoader<ROWB>& ld, char* lds, int nstage, char* ldsx = nullptr,
                                               int lw = 0) {
#pragma unroll
  for (int a = 0; a < kScAhead; ++a)
    ld.template issue<EXTRA>(lds + a * kScStageBytes, a < nstage ? a : nstage - 1, ldsx + a * 4096 + lw * 1024);
  if (EXTRA) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // stage 0 has landed
  sc_barrier();                                               // B0
  for (int st = 0; st < nstage; ++st) {
    if (EXTRA) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // stage st + 1 has landed (st + 2 may be in flight)
    SC_RAW_BARRIER();                                         // E(st)
    const int nx = st + kScAhead;
    ld.template issue<EXTRA>(lds + (nx % kScSlots) * kScStageBytes, nx < nstage ? nx : nstage - 1,
                             ldsx + (nx % kScSlots) * 4096 + lw * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// f32 tile [R rows][ld floats] in LDS -> TRANSPOSED bf16 planes dst[pl][n][k0 .. k0 + R): thread -> (column n, 8 rows)
__device__ __forceinline__ void sc_emit_planes_t(const float* tile, int ld, int R, int N, uint16_t* dst,
                                                 int64_t plane_elems, int64_t ldk, int64_t k0, int tid, int nthreads) {
  const int groups = R / 8;
  for (int it = tid; it < N * groups; it += nthreads) {
    const int n = it % N, q = it / N;
    uint32_t w[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      split3x2(tile[(8 * q + 2 * e) * ld + n], tile[(8 * q + 2 * e + 1) * ld + n], w[0][e], w[1][e], w[2][e]);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(dst + p * plane_elems + (int64_t)n * ldk + k0 + 8 * q) =
          make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
  }
}

__device__ __forceinline__ float sc_wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct ScTScore {                     // target scores t_r = U_r . T_r + tb_r, formed by the hit-list half waves of k_sc_prep
  const float* U; int64_t ldu;
  const float* T; int64_t ldt;
  const float* tb; int64_t tb_stride;
  float* tscore; float* tscore2;      // [B] (tscore2 nullable)
};

// ------------------------------------------------------------------------------------------------------------
// k_sc_prep: blocks [0, S / 32): 32 pool rows -> planes Pp [3][S][d] (forward operand, k contiguous) and
// PT [3][d][S] (dU operand, pool index contiguous); the other blocks: one wave per batch row walks the user's
// positives (user -> pos_ptr -> pos_items -> item2slot, embed_attribute.py:729-741) and lists the ones that are
// pool slots: hits[r][0 .. nhit) (more than kScHits: nhit = -1, the row kernel walks the chain itself).
// ------------------------------------------------------------------------------------------------------------
// MASKS (the 'mce' family, k_mc_flow): the walk leaves the masked pairs as bits instead of hit lists, maskR[(s >> 5) *
// ldr + u] bit (s & 31) for EVERY positive of user row u = r % mask_rows that sits in the pool (no stop at kScHits) --
// a walk of its own (k_mc_mask) took 18 us in the C4 step, as long as this whole kernel.
template <bool MASKS>
__global__ __launch_bounds__(256) void k_sc_prep(const float* __restrict__ P, int64_t ldp, int64_t S, int d,
                                                 const float* __restrict__ pbias, uint16_t* __restrict__ Pp,
                                                 uint16_t* __restrict__ PT, int64_t ldpt, float* __restrict__ pool_bad,
                                                 PosMask pm,
                                                 int64_t mask_rows, int64_t B, int32_t* __restrict__ hits,
                                                 int32_t* __restrict__ nhit, const float* __restrict__ seq_w,
                                                 int64_t seq_rows, float* __restrict__ row_w_out, ScTScore ts,
                                                 uint32_t* __restrict__ maskR, int64_t ldr) {
  __shared__ float tile[32 * 129];
  __shared__ float sbad[4];
  const int tid = threadIdx.x;
  const int64_t pblocks = S / 32;
  const int64_t hblocks = (B + 7) / 8;
  if ((int64_t)blockIdx.x >= pblocks + hblocks) {
    // the sequence model's example weights (seqModel.py:561-567): row_w[t * seq_rows + b] = w_t[b] / (sum_t w_t[b] +
    // 1e-12) for the B = L * seq_rows time-major rows -- 32 sequences x 8 time lanes per workgroup, combined in
    // lane order (the arithmetic of arx_seq_weights, which was a launch of its own in front of this kernel)
    float* part = tile;                                      // [8][33]
    const int cx = tid & 31, ty = tid >> 5;
    const int64_t b = ((int64_t)blockIdx.x - pblocks - hblocks) * 32 + cx;
    const int64_t L = B / seq_rows;
    float sum = 0.f;
    if (b < seq_rows)
      for (int64_t t = ty; t < L; t += 8) sum += seq_w[t * seq_rows + b];
    part[ty * 33 + cx] = sum;
    __syncthreads();
    float tot = 1e-12f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += part[k * 33 + cx];
    if (b < seq_rows)
      for (int64_t t = ty; t < L; t += 8) row_w_out[t * seq_rows + b] = seq_w[t * seq_rows + b] / tot;
    return;
  }
  if ((int64_t)blockIdx.x < pblocks) {
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int ld = d + 1;
    const int c4n = d / 4;
    // 0 * x is NaN exactly for a non-finite x: pool_bad[block] poisons every row's loss (what an f32 chain would do)
    float bad = (pbias && tid < 32) ? pbias[r0 + tid] * 0.f : 0.f;
    for (int it = tid; it < 32 * c4n; it += 256) {
      const int r = it / c4n, c = (it % c4n) * 4;
      const float4 v = *reinterpret_cast<const float4*>(P + (r0 + r) * ldp + c);
      bad += (v.x + v.y + v.z + v.w) * 0.f;
      tile[r * ld + c] = v.x; tile[r * ld + c + 1] = v.y; tile[r * ld + c + 2] = v.z; tile[r * ld + c + 3] = v.w;
      uint32_t a[3], b[3];
      split3x2(v.x, v.y, a[0], a[1], a[2]);
      split3x2(v.z, v.w, b[0], b[1], b[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(Pp + ((int64_t)p * S + r0 + r) * d + c) = make_uint2(a[p], b[p]);
    }
    bad = sc_wsum(bad);
    if ((tid & 63) == 0) sbad[tid >> 6] = bad;
    __syncthreads();
    if (tid == 0) pool_bad[blockIdx.x] = (sbad[0] + sbad[1]) + (sbad[2] + sbad[3]);
    sc_emit_planes_t(tile, ld, 32, d, PT, (int64_t)d * ldpt, ldpt, r0, tid, 256);
    return;
  }
  // HALF a wave per batch row (round 5: one wave per row was 16 384 waves of a four-deep load chain, two rounds of the
  // chip's wave slots, for lists of ~20 entries).  The half wave also forms the row's target score: 32 lanes x 16 bytes
  // are a whole row of U and of T at d = 128 (at d = 64 the upper 16 lanes idle) -- two coalesced loads that ride under
  // the chain's latency.  k_sc_hinge formed t_r itself in round 4, per COLUMN split, with 16 strided loads per lane.
  const int lane = tid & 63, hl = lane & 31;
  const int64_t r = (((int64_t)blockIdx.x - pblocks) * 256 + tid) >> 5;
  const bool valid = r < B;
  float4 u4 = make_float4(0.f, 0.f, 0.f, 0.f), t4 = u4;
  if (valid && 4 * hl < d) {
    u4 = *reinterpret_cast<const float4*>(ts.U + r * ts.ldu + 4 * hl);
    t4 = *reinterpret_cast<const float4*>(ts.T + r * ts.ldt + 4 * hl);
  }
  const float tbv = (valid && ts.tb) ? ts.tb[r * ts.tb_stride] : 0.f;
  // MASKS: the masked pairs of row r are those of its user, r % mask_rows -- the sequence model's L * mask_rows
  // time-major rows repeat every user L times: only the rows r < mask_rows walk (1 024 walks instead of 51 200 at the
  // C4 shape) and leave the user's word row maskR[0 .. S / 32)[r], WRITTEN whole (lane hl keeps words hl and hl + 32 in
  // registers: no zeroing launch, no atomics); the flow kernels index the table with r % mask_rows.
  // (hit lists, !MASKS: the same once-per-user walk since round 5 -- the walker leaves nhit / hits for every row of
  // its user, r + t * mask_rows; k_sc_prep took 16 us of the C4 'mw' step for 51 200 walks of 1 024 distinct lists)
  const bool walker = valid && r < mask_rows;
  const int64_t reps = walker ? (B - r + mask_rows - 1) / mask_rows : 0;
  const int usr = walker ? pm.user_ids[r % mask_rows] : 0;
  const int beg = walker ? pm.pos_ptr[usr] : 0, end = walker ? pm.pos_ptr[usr + 1] : 0;
  int n = 0;
  uint32_t mw0 = 0u, mw1 = 0u;
  for (int p0 = beg;; p0 += 32) {
    const bool active = p0 < end && (MASKS || n >= 0);
    if (!__any(active)) break;
    const int p = p0 + hl;
    int j = -1;
    if (active && p < end) {
      j = pos_slot(pm, pm.pos_items[p]);
      if (j < 0 || j >= S) j = -1;
    }
    const unsigned long long hm64 = __ballot(j >= 0);
    uint32_t hm = (lane >> 5) ? (uint32_t)(hm64 >> 32) : (uint32_t)hm64;
    if (MASKS) {
      // every slot found by a lane of the half wave goes to the lane that keeps its word (wave-uniform loop: the
      // two half waves of a wave take their own lists in step)
      while (__any(hm != 0u)) {
        const bool some = hm != 0u;
        const int srcl = some ? __builtin_ctz(hm) : 0;
        hm &= hm - 1u;
        const int jj = __shfl(j, (lane & 32) + srcl, 64);
        if (some && hl == ((jj >> 5) & 31)) {
          if (jj < 1024) mw0 |= 1u << (jj & 31);
          else mw1 |= 1u << (jj & 31);
        }
      }
    } else {
      const int k = __popc(hm);
      if (active && n >= 0) {
        if (n + k > kScHits) n = -1;
        else {
          if (j >= 0) {
            const int at = n + __popc(hm & ((1u << hl) - 1u));
            for (int64_t t = 0; t < reps; ++t) hits[(r + t * mask_rows) * kScHits + at] = j;
          }
          n += k;
        }
      }
    }
  }
  if (MASKS && walker) {
    const int nwords = (int)(S >> 5);
    if (hl < nwords) maskR[(int64_t)hl * ldr + r] = mw0;
    if (hl + 32 < nwords) maskR[(int64_t)(hl + 32) * ldr + r] = mw1;
  }
  float dot = u4.x * t4.x + u4.y * t4.y + u4.z * t4.z + u4.w * t4.w;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (!MASKS && hl == 0)
    for (int64_t t = 0; t < reps; ++t) nhit[r + t * mask_rows] = n;
  if (hl == 0 && valid) {
    const float t = dot + tbv;
    ts.tscore[r] = t;
    if (ts.tscore2) ts.tscore2[r] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_sc_hinge<KD>: workgroup = 128 batch rows x CW pool columns (CW = S / nsplit).  Waves 0..3 compute: wave w
// keeps rows [32 w, 32 w + 32) of U as bf16 pieces in registers for the whole kernel (lane = row l % 32, k = 8 (l / 32)
// .. + 8 of each 16-chunk; t_r = U_r . T_r + tb_r comes from k_sc_prep); per 32-column tile 6 KD / 16 MFMAs out of
// the pool planes in LDS (pool tile = FIRST operand: D[pool column][row], a lane's 16 accumulator values are columns
// 8 g + 4 (l / 32) + e of its row).  The hinge of tile j - 1 (v = x - t + 1: one act bit per logit, running
// sum act * v and count) is written between the MFMAs of tile j (two accumulator sets).  Waves 4..7: LDS-DMA loaders.
// Outputs: bits[tile * ldbits + row] (bit c = column 32 tile + c), rs_part / cnt_part [nsplit][B].
// ------------------------------------------------------------------------------------------------------------
template <int KD>
__global__ __launch_bounds__(512) void k_sc_hinge(int64_t B, int64_t S, int CW, const float* __restrict__ U, int64_t ldu,
                                                  const float* __restrict__ tscore,
                                                  const uint16_t* __restrict__ Pp, const float* __restrict__ bias,
                                                  uint32_t* __restrict__ bits, int64_t ldbits, float* __restrict__ rs_part,
                                                  float* __restrict__ cnt_part) {
  constexpr int NCH = KD / 16;
  constexpr int ROWB = KD * 2;                     // bytes per operand row
  constexpr int TPS = 256 / ROWB;                  // tiles per stage (1 at K = 128, 2 at K = 64)
  constexpr int TILEB = 3 * 32 * ROWB;             // bytes per tile (three planes)
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [kScSlots][24 KB], bias [CW] f32, words [CW / 32][128]
  float* sbias = reinterpret_cast<float*>(lds + kScSlots * kScStageBytes);
  // the act words of the workgroup's rows wait in LDS and leave after the loop: a compute wave issues NO global
  // memory instruction inside its loop (with the loaders' LDS-DMA in flight a single global_store_dword per tile
  // stalled the wave's in-order issue for ~400 cycles: measured, tools/sc_trace.py)
  uint32_t* swords = reinterpret_cast<uint32_t*>(sbias + CW);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int nsplit = (int)(S / CW);
  // workgroup -> (row block, column split).  Workgroups are dealt to the 8 XCDs round-robin: the nsplit workgroups
  // that read the SAME 128 rows of U / T are 8 apart, i.e. on one XCD (its L2 then serves the second reader); with
  // plain division they sat on neighbouring XCDs and every row came over the fabric nsplit times
  int64_t rb;
  int ch;
  {
    const int64_t bid = blockIdx.x, grp = (int64_t)8 * nsplit, full = (int64_t)gridDim.x / grp * grp;
    if (bid < full) {
      rb = bid / grp * 8 + bid % 8;
      ch = (int)(bid % grp / 8);
    } else {
      rb = full / nsplit + (bid - full) / nsplit;
      ch = (int)((bid - full) % nsplit);
    }
  }
  const int64_t col0 = (int64_t)ch * CW;
  const int ntile = CW / 32;
  const int nstage = ntile / TPS;

  if (wv >= 4) {
    // ================================== loaders ==================================
    ScLoader<ROWB> ld;
    ld.init(reinterpret_cast<const char*>(Pp), S * (int64_t)ROWB, ROWB, col0, 0, wv - 4, lane, S);
    ld.stage_step = (int64_t)TPS * 32 * ROWB;
    sc_loader_loop<ROWB>(ld, lds, nstage);
    return;
  }

  // ================================== compute waves ==================================
  const int lr = lane & 31, kg = lane >> 5;
  const int64_t row = rb * 128 + wv * 32 + lr;
  const bool ok = row < B;
  SC_TDECL
  SC_TREAL(120)
  SC_T(99)
  // Prologue (round 5).  Round 4 staged the bias (a global round trip in front of everything), then loaded the rows of
  // U and T (32 strided 16-byte loads per lane), formed t_r and split ALL of U into bf16 pieces (~1 000 VALU per lane)
  // before the first MFMA: 12 000 of the kernel's 54 000 cycles with the matrix pipe idle (tools/sc_trace.py), and most
  // of that the loads -- 256 workgroups x (128 KB of rows + 72 KB of pool stages) at the ~7 TB/s the memory system
  // gives a cold start.  Now: t_r comes from k_sc_prep (its hit-list waves have the lanes to spare), every load is
  // issued at once (the bias first: its LDS stores need only that one back), chunk 0 is split as soon as ITS lines are
  // there -- and the pieces of chunk c + 1 are made between the MFMAs of chunk c of tile 0, whose gaps have no hinge to
  // hold yet.  The raw rows occupy the registers the pieces grow into (8 raw / 12 piece registers per chunk: never
  // more than the 96 of the steady state).
  bf16x8 a1[NCH], a2[NCH], a3[NCH];
  float4 ru[NCH][2];
  float tm1;
  {
    float bv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) bv[k] = (bias && tid + 256 * k < CW) ? bias[col0 + tid + 256 * k] : 0.f;
    tm1 = (ok ? tscore[row] : 0.f) - 1.f;                    // v = x - (t - 1); t_r from k_sc_prep
    // rows past B read row 0 (their results are never stored).  Load order = line order: the four 16-byte loads that
    // touch one 128-byte line of a row (chunks 2 p, 2 p + 1) are adjacent, U before T -- issued chunk by chunk for both
    // arrays, the 4 waves' working set (2 x 32 lines each) overflowed the L1 and every line came from L2 four times
    const float* up = U + (ok ? row : 0) * ldu + 8 * kg;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      ru[c][0] = *reinterpret_cast<const float4*>(up + 16 * c);
      ru[c][1] = *reinterpret_cast<const float4*>(up + 16 * c + 4);
      if (c & 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 256 * k < CW) sbias[tid + 256 * k] = bv[k];
  }
#define SC_SPLIT(c_)                                                                                     \
  {                                                                                                      \
    uint32_t p1[4], p2[4], p3[4];                                                                        \
    split3x2(ru[c_][0].x, ru[c_][0].y, p1[0], p2[0], p3[0]);                                             \
    split3x2(ru[c_][0].z, ru[c_][0].w, p1[1], p2[1], p3[1]);                                             \
    split3x2(ru[c_][1].x, ru[c_][1].y, p1[2], p2[2], p3[2]);                                             \
    split3x2(ru[c_][1].z, ru[c_][1].w, p1[3], p2[3], p3[3]);                                             \
    a1[c_] = __builtin_bit_cast(bf16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));                         \
    a2[c_] = __builtin_bit_cast(bf16x8, make_uint4(p2[0], p2[1], p2[2], p2[3]));                         \
    a3[c_] = __builtin_bit_cast(bf16x8, make_uint4(p3[0], p3[1], p3[2], p3[3]));                         \
  }
  SC_SPLIT(0)
  float rs = 0.f;
  int cnt = 0;
  // LDS byte address of this lane's fragment of chunk c in the CURRENT tile's image (plane pl: + pl * 32 * ROWB).
  // Advanced IN PLACE to the next tile's image right after its last use (a scalar step): the reads of a tile then
  // depend on nothing computed inside that tile's scheduling region, so the scheduler can hoist them.
  // K = 128 (ROWB = 256): ONE register -- chunk c's address is fa[0] ^ (c << 5): the chunk index only enters through
  // the XOR swizzle, (2 c + kg) ^ sw = ((c ^ (sw >> 1)) << 1) | (kg ^ (sw & 1)), and bits 5..7 of everything else
  // are zero.  Seven registers less: 239 -> 232, which is what lets a 40-VGPR wave of another kernel -- the radix
  // scatter of the K7 sort branch -- share a SIMD with two waves of this one (2 x 240 left 32: the branch stood still
  // for the 30 us of this kernel, r04 timelines).
  constexpr bool FA1 = ROWB == 256;
  uint32_t fa[FA1 ? 1 : NCH];
#pragma unroll
  for (int c = 0; c < (FA1 ? 1 : NCH); ++c) fa[c] = (uint32_t)(lr * ROWB + 16 * ((2 * c + kg) ^ sc_swz<ROWB>(lr)));
#define SC_FA(c_) (FA1 ? (fa[0] ^ (uint32_t)((c_) << 5)) : fa[FA1 ? 0 : (c_)])

  f32x16 hiA, loA, hiB, loB;
  // The instruction stream of a tile is laid out BY HAND, chunk by chunk (left alone the scheduler emits ds_read /
  // wait / MFMA triples and the matrix pipe idles for an LDS round trip per fragment; sched_group_barrier alone
  // does not hoist the reads either):
  //   [3 fragment reads of the NEXT chunk -- in the last chunk: of the next tile's chunk 0, plus the 4 bias reads of
  //    the tile after next]                                                          | sched_barrier |
  //   [6 MFMAs of chunk c, each followed by its share of the PREVIOUS tile's hinge: VPC values per chunk -- sign bit of
  //    -(v) into h (v_alignbit), rs += relu(v), then the value's accumulator pair is re-initialised for the tile after
  //    next: hi = bias - (t - 1), lo = 0; chunk 0 also assembles and stores the word of the tile before that]
  //                                                                                   | sched_barrier |
  // and the stage barrier E sits in the MIDDLE of a tile (sc_loader_loop): nothing waits behind it.
  // tools/probe/mfma_issue.hip: <= 6 plain VALU or one ds_read_b128 per MFMA gap are free.
  constexpr int VPC = 16 / NCH;                               // hinge values per chunk (2 at K = 128, 4 at K = 64)
  constexpr int VPM = KD == 128 ? 2 : 4;                      // VALU per MFMA gap inside a chunk (6 VPC + 1 over 6 MFMAs)
  // this lane's word of tile j: swords[j * 128 + 32 wv + lr] (both halves of the wave hold the full word after the
  // swap and write the same value)
  uint32_t* wptr = swords + wv * 32 + lr;
  // fragment ring: chunk c of a tile consumes set c % 4 and requests set (c + SC_PF) % 4 (SC_PF chunks ahead, into
  // the next tile's image at the end of a tile)
  uint4 fr[4][3];
  float4 bvn[4];
  uint32_t h = 0u;
  // A zero the compiler cannot fold: the `lo` accumulators are re-initialised by a real v_mov and stay LIVE between a
  // tile's hinge and the next tile's MFMAs.  With a literal 0 hipcc starts the next tile's first MFMA from an inline
  // constant and uses the 16 dead registers for the fragments of the running tile -- then every re-initialising VALU
  // write waits for the in-flight MFMAs that still read those registers, the wave's in-order issue stalls behind it,
  // and the hinge costs 1 450 cycles per tile instead of hiding in the MFMA gaps (measured: 30 vs 20 us per launch).
  float zero_v;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
#ifndef SC_PF
#define SC_PF 1
#endif
  static_assert(SC_PF == 1 || !FA1, "the single address register steps to the next image one chunk ahead");
#define SC_LD3(set_, c_)                                                                                 \
  {                                                                                                      \
    const uint32_t fa_ = SC_FA(c_);                                                                      \
    fr[set_][0] = *reinterpret_cast<const uint4*>(lds + fa_);                                            \
    fr[set_][1] = *reinterpret_cast<const uint4*>(lds + fa_ + 32 * ROWB);                                \
    fr[set_][2] = *reinterpret_cast<const uint4*>(lds + fa_ + 64 * ROWB);                                \
  }
#define SC_MFMA6(hi_, lo_, c_)                                                                           \
  lo_ = SC_MFMA(b3, a1[c_], lo_);                                                                        \
  hi_ = SC_MFMA(b1, a1[c_], hi_);                                                                        \
  lo_ = SC_MFMA(b1, a3[c_], lo_);                                                                        \
  lo_ = SC_MFMA(b2, a2[c_], lo_);                                                                        \
  lo_ = SC_MFMA(b2, a1[c_], lo_);                                                                        \
  lo_ = SC_MFMA(b1, a2[c_], lo_);
  // the finished word (hw_: bit i = value i = column 8 (i / 4) + 4 kg + i % 4 of the tile, upper 16 bits garbage):
  // the two halves of the wave meet through v_permlane32_swap
#define SC_WORD(hw_, j_)                                                                                 \
  {                                                                                                      \
    const uint32_t hh = (hw_) & 0xFFFFu;                                                                 \
    cnt += __popc(hh);                                                                                   \
    const uint32_t w = ((hh & 0xFu) | ((hh & 0xF0u) << 4) | ((hh & 0xF00u) << 8) | ((hh & 0xF000u) << 12)) << (4 * kg); \
    const auto sw = __builtin_amdgcn_permlane32_swap(w, w, false, false);                                \
    wptr[(j_) * 128] = sw[0] | sw[1];                                                                    \
  }
#ifdef SC_ABL_NOEPI
#define SC_EPI_ON(x_) false
#else
#define SC_EPI_ON(x_) (x_)
#endif
  // the hinge of the previous tile starts in chunk SC_EPI_C0 of a tile and takes SC_EPI_V values per chunk
#ifndef SC_EPI_C0
#define SC_EPI_C0 0
#endif
#ifndef SC_EPI_V
#define SC_EPI_V VPC
#endif
  // SC_ABL_EPI_NOREAD: the hinge arithmetic runs on a register that is NOT an accumulator; SC_ABL_EPI_NOWRITE: the
  // accumulators are not re-initialised (both: timing only)
#ifdef SC_ABL_EPI_NOREAD
#define SC_EPI_READ(x_) (bvn[i >> 2].x + tm1)
#else
#define SC_EPI_READ(x_) (x_)
#endif
#ifdef SC_ABL_EPI_NOWRITE
#define SC_EPI_WRITE(x_)
#else
#define SC_EPI_WRITE(x_) x_
#endif
  // where the VALU of a chunk go between its six MFMAs: evenly (default) or in two bursts (SC_SCHED_BURST: after the
  // first and the fourth MFMA, the layout of tools/probe's hand-written tile)
#ifdef SC_SCHED_BURST
#define SC_CHUNK_SCHED(c_)                                                                               \
  {                                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    if ((c_) == 0) __builtin_amdgcn_sched_group_barrier(0x002, 3 * VPM + 9, 0);                          \
    else __builtin_amdgcn_sched_group_barrier(0x002, 3 * VPM, 0);                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                   \
    if ((c_) == 0) __builtin_amdgcn_sched_group_barrier(0x002, 3 * VPM + 9, 0);                          \
    else __builtin_amdgcn_sched_group_barrier(0x002, 3 * VPM + 1, 0);                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                   \
  }
#else
#define SC_CHUNK_SCHED(c_)                                                                               \
  _Pragma("unroll") for (int m = 0; m < 6; ++m) {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    if ((c_) == 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM + 3, 0);                              \
    else __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                                            \
  }
#endif
  // tile 0: the split of the next chunk's rows (~70 VALU) spread over the chunk's six MFMA gaps
#define SC_CHUNK_SCHED_PRO                                                                               \
  _Pragma("unroll") for (int m = 0; m < 6; ++m) {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                                   \
  }
  // one tile: MFMAs into (hi_, lo_).  EPI_: the previous tile's accumulators (hy_, ly_) are turned into hinge bits /
  // sums and re-initialised from bvn; jw_ >= 0: the word of tile jw_ (finished one tile ago) is assembled and stored in
  // chunk 0; jb0_ / jb_: the tiles whose bias quad 0 (chunk 0) / quads 1..3 (later chunks) are re-read -- see below;
  // dl_: byte step to the next tile's image; BAR_: the stage barrier E in the middle of this tile
#define SC_TILE(hi_, lo_, hy_, ly_, EPI_, PRO_, jw_, jb0_, jb_, dl_, BAR_)                               \
  {                                                                                                      \
    const uint32_t dl = (uint32_t)(dl_);                                                                 \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                    \
      if ((BAR_) && c == NCH / 2) sc_barrier();                                                          \
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, fr[c % 4][0]), b2 = __builtin_bit_cast(bf16x8, fr[c % 4][1]), \
                   b3 = __builtin_bit_cast(bf16x8, fr[c % 4][2]);                                        \
      if (FA1 && c == NCH - 1) fa[0] += dl;        /* (one address register: step to the next image before its chunk 0) */ \
      SC_LD3((c + SC_PF) % 4, (c + SC_PF) % NCH)   /* (past the tile's end: fa[] already points into the next image) */ \
      /* the bias quads are re-read one by one, each in the chunk after its last use (quad g serves the values */ \
      /* 4 g .. 4 g + 3), with the bias of the tile after next (jb_); a quad used up to the tile's last chunk is */ \
      /* re-read in the NEXT tile's chunk 0 (then as jb0_ = that tile's next tile) */                   \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
        const int lastc = SC_EPI_C0 + (15 - 4 * g) / SC_EPI_V;          /* chunk of the quad's last use */ \
        if (lastc + 1 < NCH ? c == lastc + 1 : c == 0)                                                   \
          bvn[g] = *reinterpret_cast<const float4*>(sbias + (lastc + 1 < NCH ? (jb_) : (jb0_)) * 32 + 4 * kg + 8 * g); \
      }                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      SC_MFMA6(hi_, lo_, c)                                                                              \
      if (!FA1) fa[FA1 ? 0 : c] += dl;                                                                   \
      if (c == 0 && (jw_) >= 0) {                                                                        \
        const uint32_t hw = h;                                                                           \
        SC_WORD(hw, jw_)                                                                                 \
      }                                                                                                  \
      if (SC_EPI_ON(EPI_) && c >= SC_EPI_C0 && c < SC_EPI_C0 + 16 / SC_EPI_V) {                          \
        _Pragma("unroll") for (int q = 0; q < SC_EPI_V; ++q) {                                           \
          const int i = 15 - ((c - SC_EPI_C0) * SC_EPI_V + q);                                           \
          const float nv = SC_EPI_READ(-hy_[i] - ly_[i]);                                                \
          h = __builtin_amdgcn_alignbit(h, __float_as_uint(nv), 31);                                     \
          rs += fmaxf(-nv, 0.f);                                                                         \
          const float4 bq = bvn[i >> 2];                                                                 \
          SC_EPI_WRITE(hy_[i] = ((i & 3) == 0 ? bq.x : (i & 3) == 1 ? bq.y : (i & 3) == 2 ? bq.z : bq.w) - tm1;) \
          SC_EPI_WRITE({ float z_; asm volatile("v_mov_b32 %0, 0" : "=v"(z_)); ly_[i] = z_; })          \
        }                                                                                                \
      }                                                                                                  \
      if ((PRO_) && c + 1 < NCH) { SC_SPLIT(c + 1) }                                                     \
      if (PRO_) { SC_CHUNK_SCHED_PRO } else { SC_CHUNK_SCHED(c) }                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (SC_TRACE_CHUNKS || c == NCH - 1) { SC_T(1 + c) }                                               \
    }                                                                                                    \
  }
#define SC_INIT(hi_, lo_, j_)                                                                            \
  {                                                                                                      \
    const float* bt = sbias + (j_) * 32 + 4 * kg;                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                      \
      const float4 bv = *reinterpret_cast<const float4*>(bt + 8 * g);                                    \
      hi_[4 * g] = bv.x - tm1; hi_[4 * g + 1] = bv.y - tm1; hi_[4 * g + 2] = bv.z - tm1; hi_[4 * g + 3] = bv.w - tm1; \
    }                                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) lo_[e] = zero_v;                                      \
  }
  // byte offset of tile j's image in the ring, and the step from tile j to tile j + 1 (wave-uniform)
  auto img = [&](int j) -> int { return ((j / TPS) % kScSlots) * kScStageBytes + (j % TPS) * TILEB; };
  auto step = [&](int j) -> int { return img(j + 1) - img(j); };
  SC_T(100)
  sc_barrier();                                               // B0: stage 0 and the bias are in LDS
  SC_T(101)
  SC_INIT(hiA, loA, 0)
  SC_INIT(hiB, loB, 1)
#pragma unroll
  for (int c = 0; c < SC_PF; ++c) SC_LD3(c, c)
  // tile 0 -> A (no hinge to do yet); the stage barrier: in every tile at K = 128, in the odd ones at K = 64
  auto tcl = [&](int j) -> int { return j < ntile ? j : ntile - 1; };
  SC_TILE(hiA, loA, hiB, loB, false, true, -1, tcl(1), tcl(2), step(0), TPS == 1)
  // tile 1 -> B with the hinge of tile 0 (A), A re-initialised for tile 2
  SC_TILE(hiB, loB, hiA, loA, true, false, -1, tcl(2), tcl(3), step(1), true)
  for (int j = 2; j + 1 < ntile; j += 2) {
    // tile j -> A with the hinge of tile j - 1 (B) and the word of tile j - 2; B re-initialised for tile j + 1
    SC_TILE(hiA, loA, hiB, loB, true, false, j - 2, tcl(j + 1), tcl(j + 2), step(j), TPS == 1)
    // tile j + 1 -> B with the hinge of tile j (A) and the word of tile j - 1; A re-initialised for tile j + 2
    SC_TILE(hiB, loB, hiA, loA, true, false, j - 1, tcl(j + 2), tcl(j + 3), step(j + 1), true)
  }
  // tail: the word of tile ntile - 2 (h holds it), then the hinge and the word of the last tile (B)
  SC_WORD(h, ntile - 2)
#pragma unroll
  for (int i = 15; i >= 0; --i) {
    const float nv = -hiB[i] - loB[i];
    h = __builtin_amdgcn_alignbit(h, __float_as_uint(nv), 31);
    rs += fmaxf(-nv, 0.f);
  }
  SC_WORD(h, ntile - 1)
  SC_T(110)
  // the wave's words leave: lane (lr, kg) stores the tiles of parity kg (its own LDS writes: no barrier needed)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (ok) {
    for (int j = kg; j < ntile; j += 2) bits[(col0 / 32 + j) * ldbits + row] = swords[j * 128 + wv * 32 + lr];
  }
  SC_T(111)
  SC_TREAL(121)
#undef SC_LD3
#undef SC_SPLIT
#undef SC_MFMA6
#undef SC_TILE
#undef SC_WORD
#undef SC_INIT
  rs += __shfl_xor(rs, 32, 64);
  cnt += __shfl_xor(cnt, 32, 64);
  if (kg == 0 && ok) {
    rs_part[(int64_t)ch * B + row] = rs;
    cnt_part[(int64_t)ch * B + row] = (float)cnt;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_sc_rows: workgroup = 32 batch rows (8 waves x 4 rows).  Per row: the column splits' partial sums in fixed
// order, the user's positives that sit in the pool taken out again (their logit recomputed from the row's latent
// and the pool row, the act bit cleared), loss = log(1 + s), g = gscale * w / (1 + s), dt = -g * cnt and the
// rank-one terms dT = dt U, dU = dt T.  Per workgroup: g U as TRANSPOSED bf16 planes UgT [3][d][Bp] (the dI
// operand), the act bits TRANSPOSED (bitsT[blk][s]: bit r = row 32 blk + r) and the bias-gradient partials
// dbp[blk][s] = sum_r act[r][s] g[r].
// ------------------------------------------------------------------------------------------------------------
struct ScRows {
  const float* rs_part;      // [nsplit][B]
  const float* cnt_part;
  int nsplit;
  uint32_t* bits;            // word-major [S / 32][ldbits]
  int64_t ldbits;
  uint32_t* bitsT; int64_t ldbt;   // [Bp / 32][ldbt]
  const float* tscore;       // [B]
  const float* U; int64_t ldu;
  const float* T; int64_t ldt;
  const float* P; int64_t ldp;     // pool rows [S, d]
  const float* pb;                 // pool bias [S] (nullable)
  int d;
  float gscale;
  const float* row_w;              // nullable
  float* batch_loss;               // [B] nullable
  float* g_out;                    // [Bp]
  float* dtscore; int64_t dts_stride;
  float* dU; int64_t lddu;         // dt * T   (nullable)
  float* dT; int64_t lddt;         // dt * U   (nullable)
  uint16_t* UgT; int64_t ldug;     // [3][d][ldug]
  float* dbp;                      // [Bp / 32][S]
  const int32_t* hits;             // [B][kScHits]
  const int32_t* nhit;             // [B]  (-1: walk the positives)
  const float* pool_bad;           // [S / 32]: NaN where a block of pool rows / biases holds a non-finite value
};

__global__ __launch_bounds__(512) void k_sc_rows(ScRows a, PosMask pm, int64_t mask_rows, int64_t B, int64_t S) {
  __shared__ float tile[32 * 129];              // g U of the block's rows, f32 [32][d + 1]
  __shared__ uint32_t sbits[32 * 64];           // final act words of the rows [32][S / 32]
  __shared__ float sg[32];
#ifdef SC_TRACE_ROWS
  SC_TDECL
  SC_TREAL(120)
#endif
  SC_TR(10)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int nwords = (int)(S >> 5);
  const int ld = a.d + 1;
  const bool colok = lane * 4 < a.d;
  // non-finite inputs poison the loss like an f32 chain would (the hinge itself drops a NaN logit: v > 0 is false)
  const float pbad = sc_wsum(lane < (int)(S >> 5) ? a.pool_bad[lane] : 0.f);
  constexpr int RPW = 4;                          // rows per wave, all loads of the four rows issued up front
  float4 u[RPW], tr[RPW];
  float s[RPW], c[RPW], t[RPW];
  int nh[RPW];
  uint32_t w0[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int64_t r = r0 + wv * RPW + q;
    const bool live = r < B;
    u[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    tr[q] = u[q];
    s[q] = c[q] = t[q] = 0.f;
    nh[q] = 0;
    w0[q] = 0u;
    if (live) {
      for (int p = 0; p < a.nsplit; ++p) {                    // fixed order: bit-reproducible
        s[q] += a.rs_part[(int64_t)p * B + r];
        c[q] += a.cnt_part[(int64_t)p * B + r];
      }
      t[q] = a.tscore[r];
      if (colok) {
        u[q] = *reinterpret_cast<const float4*>(a.U + r * a.ldu + lane * 4);
        tr[q] = *reinterpret_cast<const float4*>(a.T + r * a.ldt + lane * 4);
      }
      nh[q] = a.nhit[r];
      if (lane < nwords) w0[q] = a.bits[(int64_t)lane * a.ldbits + r];
    }
  }
  SC_TR(16)
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int rl = wv * RPW + q;
    const int64_t r = r0 + rl;
    const bool live = r < B;
    uint32_t myw = w0[q];
    float sq = s[q], cq = c[q];
    const float4 uq = u[q], tq = tr[q];
    if (live && nh[q] != 0) {
      // the row's bit words live in the lanes (word w in lane w; S <= 2048): a slot the positives name twice is
      // taken out once, and the write-back is one store per changed word
      auto take_out = [&](int jj) {                   // jj wave-uniform
        const uint32_t w = __shfl(myw, jj >> 5, 64);
        if (!((w >> (jj & 31)) & 1u)) return;         // hinge not active there (or already taken out)
        float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colok) pr = *reinterpret_cast<const float4*>(a.P + (int64_t)jj * a.ldp + lane * 4);
        const float x = sc_wsum(uq.x * pr.x + uq.y * pr.y + uq.z * pr.z + uq.w * pr.w) + (a.pb ? a.pb[jj] : 0.f);
        const float v = x - t[q] + 1.f;
        sq -= v > 0.f ? v : 0.f;
        cq -= 1.f;
        if (lane == (jj >> 5)) myw &= ~(1u << (jj & 31));
      };
      if (nh[q] > 0) {
        const int myhit = lane < nh[q] ? a.hits[r * kScHits + lane] : 0;
        for (int k = 0; k < nh[q]; ++k) take_out(__shfl(myhit, k, 64));
      } else {                                        // long list: walk the positives here
        const int usr = pm.user_ids[r % mask_rows];
        const int beg = pm.pos_ptr[usr], end = pm.pos_ptr[usr + 1];
        for (int p0 = beg; p0 < end; p0 += 64) {
          const int p = p0 + lane;
          int j = -1;
          if (p < end) {
            j = pos_slot(pm, pm.pos_items[p]);
            if (j < 0 || j >= S) j = -1;
          }
          unsigned long long hm = __ballot(j >= 0);
          while (hm) {
            const int src = __builtin_ctzll(hm);
            hm &= hm - 1;
            take_out(__shfl(j, src, 64));
          }
        }
      }
      if (lane < nwords && myw != w0[q]) a.bits[(int64_t)lane * a.ldbits + r] = myw;
    }
    sq = fmaxf(sq, 0.f) + (pbad + sc_wsum(((uq.x + uq.y) + (uq.z + uq.w) + (tq.x + tq.y) + (tq.z + tq.w)) * 0.f) + t[q] * 0.f);
    const float g = live ? a.gscale * (a.row_w ? a.row_w[r] : 1.f) / (1.f + sq) : 0.f;
    const float dt = -g * cq;
    if (lane == 0) {
      sg[rl] = g;
      if (live) {
        if (a.batch_loss) a.batch_loss[r] = logf(1.f + sq);
        if (a.dtscore) a.dtscore[r * a.dts_stride] = dt;
      }
      a.g_out[r] = g;                                 // (rows past B inside the padded length: 0)
    }
    if (lane < nwords) sbits[rl * 64 + lane] = myw;
    if (colok) {
      if (live) {
        if (a.dT) *reinterpret_cast<float4*>(a.dT + r * a.lddt + lane * 4) = make_float4(dt * uq.x, dt * uq.y, dt * uq.z, dt * uq.w);
        if (a.dU) *reinterpret_cast<float4*>(a.dU + r * a.lddu + lane * 4) = make_float4(dt * tq.x, dt * tq.y, dt * tq.z, dt * tq.w);
      }
      float* tp = tile + rl * ld + lane * 4;
      tp[0] = g * uq.x; tp[1] = g * uq.y; tp[2] = g * uq.z; tp[3] = g * uq.w;
    }
  }
  SC_TR(11)
  __syncthreads();
  SC_TR(15)
  // g U -> transposed planes, 32 consecutive k (= batch rows) per column n
  sc_emit_planes_t(tile, ld, 32, a.d, a.UgT, (int64_t)a.d * a.ldug, a.ldug, r0, tid, 512);
  SC_TR(12)
  // act bits transposed: a 32 x 32 bit block is 32 ballots -- lane l of a half-wave holds row l's word of one word
  // column, ballot k collects bit k of the 32 rows = the transposed word of column 32 c + k (the low half of the
  // ballot for the word column of lanes 0..31, the high half for the one of lanes 32..63).  Then the block's
  // bias-gradient partial per column from that word (the per-column loop over 32 LDS words it replaces took 9.7 k
  // cycles of the workgroup's 28 k)
  uint32_t* tw = reinterpret_cast<uint32_t*>(tile);            // (the f32 tile is dead: planes emitted above)
  __syncthreads();
  for (int c0 = wv * 2; c0 < nwords; c0 += 16) {
    const int c = c0 + (lane >> 5);
    const uint32_t w = c < nwords ? sbits[(lane & 31) * 64 + c] : 0u;
    uint32_t mine = 0u;                                        // lane 32 h + k ends up with ballot k's half h
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const unsigned long long bal = __ballot((w >> k) & 1u);
      // (s_nop: a v_writelane that reads an SGPR a VALU compare has just written needs wait states the compiler
      // cannot insert inside inline asm -- without them the lanes got garbage; probed on gfx950)
      asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                   : "+v"(mine) : "s"((uint32_t)bal), "s"((uint32_t)(bal >> 32)), "n"(k), "n"(32 + k));
    }
    if (c < nwords) tw[c * 32 + (lane & 31)] = mine;
  }
  SC_TR(14)
  __syncthreads();
  // (g of the 32 rows read from LDS as broadcasts, four at a time: kept in 32 registers the kernel went from 88 to
  // 115 VGPRs and the radix scatter that runs beside it in the step no longer fitted on the SIMDs -- 13 -> 32 us)
  for (int col = tid; col < S; col += 512) {
    const uint32_t word = tw[col];
    float db = 0.f;
#pragma unroll 1
    for (int r4 = 0; r4 < 32; r4 += 4) {
      const float4 g4 = *reinterpret_cast<const float4*>(&sg[r4]);
      db += ((word >> r4) & 1u) ? g4.x : 0.f;
      db += ((word >> (r4 + 1)) & 1u) ? g4.y : 0.f;
      db += ((word >> (r4 + 2)) & 1u) ? g4.z : 0.f;
      db += ((word >> (r4 + 3)) & 1u) ? g4.w : 0.f;
    }
    a.bitsT[(int64_t)blockIdx.x * a.ldbt + col] = word;
    a.dbp[(int64_t)blockIdx.x * S + col] = db;
  }
  SC_TR(13)
#ifdef SC_TRACE_ROWS
  SC_TREAL(121)
#endif
}

// ------------------------------------------------------------------------------------------------------------
// k_sc_rows_g (round 6): k_sc_rows with the row phase laid out for the width the scorer family has -- d = 64 / 128 is
// LPR = 16 / 32 lanes of 16 bytes, so a wave serves 64 / LPR rows AT ONCE (one sub-group per row) instead of four
// rows one after the other with 48 / 32 idle lanes: one load instruction per operand for all of the wave's rows, 16-
// / 32-lane reductions (the upper steps of the 64-lane xor tree added exact zeros: same bits), the row's S / 32
// activity words spread over the sub-group's lanes (word w in lane w % LPR, slot w / LPR).  Measured on the C4 shape
// (51 200 rows, d = 64; cycle stamps of one wave, tools/sc_trace.py): the four-rows-in-a-row phase was 6 300 of the
// workgroup's 23 200 cycles behind an 8 900-cycle first load; k_sc_rows was the largest kernel of the C4 step (71.8 us).
// Results bit-identical to k_sc_rows (ARX_SC_ROWS_OLD=1 selects it: the A/B of DESIGN.md section 6).
// ------------------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(512) void k_sc_rows_g(ScRows a, PosMask pm, int64_t mask_rows, int64_t B, int64_t S) {
  constexpr int G = 64 / LPR;                    // rows a wave serves at once
  constexpr int WPL = 64 / LPR;                  // activity words per lane (S <= 2048: 64 words per row)
  constexpr int ROUNDS = 4 / G;                  // 32 rows per workgroup of 8 waves = 4 rows per wave
  static_assert(G == 2 || G == 4, "d = 128 or d = 64");
  __shared__ float tile[32 * 129];               // g U of the block's rows, f32 [32][d + 1]
  __shared__ uint32_t sbits[32 * 64];            // final act words of the rows [32][S / 32]
  __shared__ float sg[32];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int j = lane % LPR, grp = lane / LPR, gbase = grp * LPR;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int nwords = (int)(S >> 5);
  const int ld = a.d + 1;
  const float pbad = sc_wsum(lane < nwords ? a.pool_bad[lane] : 0.f);
  auto gsum = [](float v) {                      // sum over the sub-group's LPR lanes (xor tree, offsets LPR / 2 .. 1)
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll
  for (int it = 0; it < ROUNDS; ++it) {
    const int rl = wv * 4 + it * G + grp;
    const int64_t r = r0 + rl;
    const bool live = r < B;
    float4 uq = make_float4(0.f, 0.f, 0.f, 0.f), tq = uq;
    float sq = 0.f, cq = 0.f, t = 0.f;
    int nh = 0;
    uint32_t w[WPL], w0[WPL];
#pragma unroll
    for (int k = 0; k < WPL; ++k) w[k] = 0u;
    if (live) {
      for (int p = 0; p < a.nsplit; ++p) {                    // fixed order: bit-reproducible
        sq += a.rs_part[(int64_t)p * B + r];
        cq += a.cnt_part[(int64_t)p * B + r];
      }
      t = a.tscore[r];
      uq = *reinterpret_cast<const float4*>(a.U + r * a.ldu + j * 4);
      tq = *reinterpret_cast<const float4*>(a.T + r * a.ldt + j * 4);
      nh = a.nhit[r];
#pragma unroll
      for (int k = 0; k < WPL; ++k)
        if (k * LPR + j < nwords) w[k] = a.bits[(int64_t)(k * LPR + j) * a.ldbits + r];
    }
#pragma unroll
    for (int k = 0; k < WPL; ++k) w0[k] = w[k];
    if (__any(live && nh != 0)) {
      // a slot the positives name twice is taken out once; jj / act are uniform over the sub-group
      auto take_out = [&](int jj, bool act) {
        const int wi = jj >> 5, own = wi % LPR, slot = wi / LPR;
        uint32_t mine = w[0];
#pragma unroll
        for (int k = 1; k < WPL; ++k) mine = slot == k ? w[k] : mine;
        const uint32_t word = __shfl(mine, gbase + own, 64);
        const bool on = act && ((word >> (jj & 31)) & 1u);          // hinge active there (and not yet taken out)
        float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) pr = *reinterpret_cast<const float4*>(a.P + (int64_t)jj * a.ldp + j * 4);
        const float x = gsum(uq.x * pr.x + uq.y * pr.y + uq.z * pr.z + uq.w * pr.w) + ((on && a.pb) ? a.pb[jj] : 0.f);
        const float v = x - t + 1.f;
        if (on) {
          sq -= v > 0.f ? v : 0.f;
          cq -= 1.f;
          if (j == own) {
#pragma unroll
            for (int k = 0; k < WPL; ++k)
              if (slot == k) w[k] &= ~(1u << (jj & 31));
          }
        }
      };
      const bool listed = live && nh > 0;
      const int myhit = (listed && j < nh) ? a.hits[r * kScHits + j] : 0;
      for (int k = 0; k < kScHits; ++k) {
        const bool act = listed && k < nh;
        if (!__any(act)) break;
        take_out(__shfl(myhit, gbase + k, 64), act);
      }
      const bool walk = live && nh < 0;                     // long list: walk the positives here
      if (__any(walk)) {
        const int usr = walk ? pm.user_ids[r % mask_rows] : 0;
        const int beg = walk ? pm.pos_ptr[usr] : 0, end = walk ? pm.pos_ptr[usr + 1] : 0;
        for (int p0 = beg;; p0 += LPR) {
          const bool more = walk && p0 < end;
          if (!__any(more)) break;
          const int p = p0 + j;
          int js = -1;
          if (more && p < end) {
            js = pos_slot(pm, pm.pos_items[p]);
            if (js < 0 || js >= S) js = -1;
          }
          const unsigned long long bal = __ballot(js >= 0);
          uint32_t hm = (uint32_t)((bal >> gbase) & ((LPR == 32) ? 0xffffffffull : 0xffffull));
          while (__any(hm != 0u)) {                         // sub-groups drain their hit masks side by side
            const bool act = hm != 0u;
            const int src = act ? __builtin_ctz(hm) : 0;
            hm &= hm - 1u;
            take_out(__shfl(js, gbase + src, 64), act);
          }
        }
      }
      if (live && nh != 0) {
#pragma unroll
        for (int k = 0; k < WPL; ++k)
          if (k * LPR + j < nwords && w[k] != w0[k]) a.bits[(int64_t)(k * LPR + j) * a.ldbits + r] = w[k];
      }
    }
    // non-finite inputs poison the loss like an f32 chain would (the hinge itself drops a NaN logit: v > 0 is false)
    sq = fmaxf(sq, 0.f) + (pbad + gsum(((uq.x + uq.y) + (uq.z + uq.w) + (tq.x + tq.y) + (tq.z + tq.w)) * 0.f) + t * 0.f);
    const float g = live ? a.gscale * (a.row_w ? a.row_w[r] : 1.f) / (1.f + sq) : 0.f;
    const float dt = -g * cq;
    if (j == 0) {
      sg[rl] = g;
      if (live) {
        if (a.batch_loss) a.batch_loss[r] = logf(1.f + sq);
        if (a.dtscore) a.dtscore[r * a.dts_stride] = dt;
      }
      a.g_out[r] = g;                                 // (rows past B inside the padded length: 0)
    }
#pragma unroll
    for (int k = 0; k < WPL; ++k)
      if (k * LPR + j < nwords) sbits[rl * 64 + k * LPR + j] = w[k];
    if (live) {
      if (a.dT) *reinterpret_cast<float4*>(a.dT + r * a.lddt + j * 4) = make_float4(dt * uq.x, dt * uq.y, dt * uq.z, dt * uq.w);
      if (a.dU) *reinterpret_cast<float4*>(a.dU + r * a.lddu + j * 4) = make_float4(dt * tq.x, dt * tq.y, dt * tq.z, dt * tq.w);
    }
    float* tp = tile + rl * ld + j * 4;
    tp[0] = g * uq.x; tp[1] = g * uq.y; tp[2] = g * uq.z; tp[3] = g * uq.w;
  }
  __syncthreads();
  // g U -> transposed planes, 32 consecutive k (= batch rows) per column n
  sc_emit_planes_t(tile, ld, 32, a.d, a.UgT, (int64_t)a.d * a.ldug, a.ldug, r0, tid, 512);
  // act bits transposed + the block's bias-gradient partials: as in k_sc_rows
  uint32_t* tw = reinterpret_cast<uint32_t*>(tile);            // (the f32 tile is dead: planes emitted above)
  __syncthreads();
  for (int c0 = wv * 2; c0 < nwords; c0 += 16) {
    const int c = c0 + (lane >> 5);
    const uint32_t w = c < nwords ? sbits[(lane & 31) * 64 + c] : 0u;
    uint32_t mine = 0u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const unsigned long long bal = __ballot((w >> k) & 1u);
      asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                   : "+v"(mine) : "s"((uint32_t)bal), "s"((uint32_t)(bal >> 32)), "n"(k), "n"(32 + k));
    }
    if (c < nwords) tw[c * 32 + (lane & 31)] = mine;
  }
  __syncthreads();
  for (int col = tid; col < S; col += 512) {
    const uint32_t word = tw[col];
    float db = 0.f;
#pragma unroll 1
    for (int r4 = 0; r4 < 32; r4 += 4) {
      const float4 g4 = *reinterpret_cast<const float4*>(&sg[r4]);
      db += ((word >> r4) & 1u) ? g4.x : 0.f;
      db += ((word >> (r4 + 1)) & 1u) ? g4.y : 0.f;
      db += ((word >> (r4 + 2)) & 1u) ? g4.z : 0.f;
      db += ((word >> (r4 + 3)) & 1u) ? g4.w : 0.f;
    }
    a.bitsT[(int64_t)blockIdx.x * a.ldbt + col] = word;
    a.dbp[(int64_t)blockIdx.x * S + col] = db;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_sc_bits: C[m, n] (+)= sum_k act[m][k] X[k][n] with act a 0/1 matrix given as bits, word-major:
// act[m][k] = bit (k & 31) of bits[(k >> 5) * ldw + m], and X as TRANSPOSED bf16 planes XT [3][N][ldx].
// Workgroup = 256 m x 32 n x one K slice; compute wave = 64 m (two m-tiles) x the n-tile: per 16-chunk three plane
// fragments from LDS (FIRST operand, rows = n), two act fragments expanded from the lane's own bit words in ten
// VALU instructions each (v_bfe, v_lshl_or, 4 x (v_and, v_mul_u32_u24): exact 0.0 / 1.0 bf16), six MFMAs.
//   SLICED == false (dU):  C[m, :] = beta C[m, :] + g[m] * acc        (one slice = all of K)
//   SLICED == true  (dI):  part[slice][m, :] = acc                    (k_sc_tn_reduce adds the slices up)
// Stages of 128 k; loaders as in k_sc_hinge.  K (= Kp) is a multiple of 128, bits / planes zero-padded up to it.
// ------------------------------------------------------------------------------------------------------------
template <bool SLICED>
__global__ __launch_bounds__(512) void k_sc_bits(int64_t M, int N, int64_t kslice, int64_t Kp,
                                                 const uint32_t* __restrict__ bits, int64_t ldw,
                                                 const uint16_t* __restrict__ XT, int64_t ldx, float beta,
                                                 float* __restrict__ C, int64_t ldc, const float* __restrict__ gvec) {
  constexpr int ROWB = 256;
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [kScSlots][24 KB] planes, [kScSlots][4][256] bit words
  char* ldsw = lds + kScSlots * kScStageBytes;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int ntn = N / 32;
  const int64_t mblocks = (M + 255) / 256;
  int64_t bid = blockIdx.x;
  const int nt = (int)(bid % ntn);
  bid /= ntn;
  const int64_t mb = bid % mblocks;
  const int64_t sl = bid / mblocks;
  const int64_t k0 = sl * kslice;
  const int64_t kend = k0 + kslice < Kp ? k0 + kslice : Kp;
  const int nstage = (int)((kend - k0) / 128);

  if (wv >= 4) {
    ScLoader<ROWB> ld;
    ld.init(reinterpret_cast<const char*>(XT), (int64_t)N * ldx * 2, ldx * 2, (int64_t)nt * 32, k0 * 2, wv - 4, lane, N);
    ld.stage_step = 256;
    // ... and the stage's act words: loader wave q fetches word row (k0 / 32 + 4 st + q), rows mb * 256 .. + 256 (one
    // 1 KiB piece: the compute waves issue no global memory instruction in their loop)
    ld.srcx = reinterpret_cast<const char*>(bits + ((k0 >> 5) + (wv - 4)) * ldw + mb * 256 + lane * 4);
    ld.stepx = 4 * ldw * 4;
    sc_loader_loop<ROWB, true>(ld, lds, nstage, ldsw, wv - 4);
    return;
  }

  const int lr = lane & 31, kg = lane >> 5;
  const int64_t m_a = mb * 256 + wv * 64 + lr, m_b = m_a + 32;
  const bool ok_a = m_a < M, ok_b = m_b < M;
  // (the stream of a stage is laid out by hand, chunk by chunk -- see k_sc_hinge)
  uint32_t fa[8];                                              // LDS byte address of the lane's fragment of chunk c
#pragma unroll
  for (int c = 0; c < 8; ++c) fa[c] = (uint32_t)(lr * ROWB + 16 * ((2 * c + kg) ^ sc_swz<ROWB>(lr)));
  f32x16 hi0 = {0}, lo0 = {0}, hi1 = {0}, lo1 = {0};
  // 8 bits -> four dwords of two bf16 (0.0 / 1.0) each: y = b | b << 15 puts bit 2 j at position 2 j and bit 2 j + 1
  // at position 2 j + 16; (y & (0x10001 << 2 j)) * (0x3F80 >> 2 j) is the pair, no carries: ten VALU per fragment
#define SC_EXPAND(dst_, x_, odd_)                                                                       \
  {                                                                                                      \
    const uint32_t b = __builtin_amdgcn_ubfe((x_), (odd_) ? 16 : 0, 8);                                  \
    const uint32_t y = b | (b << 15);                                                                    \
    dst_ = make_uint4(__umul24(y & 0x00010001u, 0x3F80u), __umul24(y & 0x00040004u, 0x0FE0u),            \
                      __umul24(y & 0x00100010u, 0x03F8u), __umul24(y & 0x00400040u, 0x00FEu));           \
  }
#define SC_LD3(c_)                                                                                       \
  {                                                                                                      \
    f1 = *reinterpret_cast<const uint4*>(lds + fa[c_]);                                                  \
    f2 = *reinterpret_cast<const uint4*>(lds + fa[c_] + 32 * ROWB);                                      \
    f3 = *reinterpret_cast<const uint4*>(lds + fa[c_] + 64 * ROWB);                                      \
  }
  // the bit words of the stage being computed (pre-shifted by 8 kg: the lane's bytes are 0 and 2 of each word) and
  // of the next one (raw), both m-tiles, from the loaders' LDS images [slot][4 word rows][256 rows]; rows past M masked
  const uint32_t mk_a = ok_a ? 0xFFFFFFFFu : 0u, mk_b = ok_b ? 0xFFFFFFFFu : 0u;
  const uint32_t* wl = reinterpret_cast<const uint32_t*>(ldsw) + wv * 64 + lr;
  uint32_t wa[4], wb[4], na[4], nb[4];
  uint4 f1, f2, f3, ea, eb;
  sc_barrier();                                               // B0
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    wa[q] = (wl[q * 256] & mk_a) >> (8 * kg);
    wb[q] = (wl[q * 256 + 32] & mk_b) >> (8 * kg);
  }
  SC_EXPAND(ea, wa[0], 0)
  SC_EXPAND(eb, wb[0], 0)
  SC_LD3(0)
  for (int st = 0; st < nstage; ++st) {
    const uint32_t dl = (uint32_t)((((st + 1) % kScSlots) - (st % kScSlots)) * kScStageBytes);
    const uint32_t* wn = wl + ((st + 1) % kScSlots) * 1024;   // next stage's words (its image is there after E(st))
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c == 4) {
        sc_barrier();                                         // E(st), in the middle of the stage (sc_loader_loop)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          na[q] = wn[q * 256];
          nb[q] = wn[q * 256 + 32];
        }
      }
      const bf16x8 p1 = __builtin_bit_cast(bf16x8, f1), p2 = __builtin_bit_cast(bf16x8, f2),
                   p3 = __builtin_bit_cast(bf16x8, f3);
      const bf16x8 aa = __builtin_bit_cast(bf16x8, ea), ab = __builtin_bit_cast(bf16x8, eb);
      if (c + 1 < 8) SC_LD3(c + 1)
      else SC_LD3(0)                                          // (fa[0] already points into the next stage's image)
      __builtin_amdgcn_sched_barrier(0);
      hi0 = SC_MFMA(p1, aa, hi0);
      hi1 = SC_MFMA(p1, ab, hi1);
      lo0 = SC_MFMA(p2, aa, lo0);
      lo1 = SC_MFMA(p2, ab, lo1);
      lo0 = SC_MFMA(p3, aa, lo0);
      lo1 = SC_MFMA(p3, ab, lo1);
      fa[c] += dl;
      if (c + 1 < 8) {                                        // the act fragments of chunk c + 1
        SC_EXPAND(ea, wa[(c + 1) >> 1], (c + 1) & 1)
        SC_EXPAND(eb, wb[(c + 1) >> 1], (c + 1) & 1)
      } else {                                                // ... of the next stage's chunk 0; its words move in
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wa[q] = (na[q] & mk_a) >> (8 * kg);
          wb[q] = (nb[q] & mk_b) >> (8 * kg);
        }
        SC_EXPAND(ea, wa[0], 0)
        SC_EXPAND(eb, wb[0], 0)
      }
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (c + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        else __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef SC_EXPAND
#undef SC_LD3
  // epilogue: lane = output row m (one per m-tile), columns nt * 32 + 8 g + 4 kg + 0..3
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int64_t m = t ? m_b : m_a;
    if (!(t ? ok_b : ok_a)) continue;
    const f32x16& hi = t ? hi1 : hi0;
    const f32x16& lo = t ? lo1 : lo0;
    if (!SLICED) {
      const float gm = gvec ? gvec[m] : 1.f;
      float* crow = C + m * ldc + nt * 32 + 4 * kg;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4* cp4 = reinterpret_cast<float4*>(crow + 8 * g);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (beta != 0.f) {
          o = *cp4;
          o.x *= beta; o.y *= beta; o.z *= beta; o.w *= beta;
        }
        o.x += gm * (hi[4 * g] + lo[4 * g]);
        o.y += gm * (hi[4 * g + 1] + lo[4 * g + 1]);
        o.z += gm * (hi[4 * g + 2] + lo[4 * g + 2]);
        o.w += gm * (hi[4 * g + 3] + lo[4 * g + 3]);
        *cp4 = o;
      }
    } else {
      float* prow = C + (sl * M + m) * (int64_t)N + nt * 32 + 4 * kg;             // C = partials [slices][M][N]
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(prow + 8 * g) =
            make_float4(hi[4 * g] + lo[4 * g], hi[4 * g + 1] + lo[4 * g + 1], hi[4 * g + 2] + lo[4 * g + 2],
                        hi[4 * g + 3] + lo[4 * g + 3]);
    }
  }
}

// db_steps[t][s] = sum of the bps block partials of time step t (fixed order): a role of k_sc_tn_reduce's launch (blocks
// of 256 outputs; a launch of its own, 5 us on the sequence model's serial chain, before)
__device__ __forceinline__ void sc_db_steps_block(const float* __restrict__ dbp, int64_t bps, int64_t S, int64_t L,
                                                  float* __restrict__ db_steps, int64_t blk) {
  const int64_t q = blk * 256 + threadIdx.x;
  if (q >= L * S) return;
  const int64_t t = q / S, s = q % S;
  // (eight loads in flight, added in block order: one load per iteration was a chain of L2 round trips -- 12 us for
  // 6.5 MB at the C4 shape)
  const float* p = dbp + t * bps * S + s;
  float acc = 0.f;
  int64_t b = 0;
  for (; b + 8 <= bps; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(b + u) * S];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; b < bps; ++b) acc += p[b * S];
  db_steps[q] = acc;
}

// dI[m, :] = beta dI[m, :] + sum_slices part[sl][m, :]   (fixed order); the LAST blocks of the grid (dbblocks of them, 64
// columns each): db[m] = sum_blocks dbp[blk][m] -- four waves take a quarter of the partial rows each (coalesced
// 256-byte rows, independent loads), combined in wave order
__global__ __launch_bounds__(256) void k_sc_tn_reduce(const float* __restrict__ part, int nsl, int64_t M, int N,
                                                      float beta, float* __restrict__ C, int64_t ldc,
                                                      const float* __restrict__ dbp, int64_t nblk,
                                                      float* __restrict__ db, int dbblocks,
                                                      const float* __restrict__ lrow, const float* __restrict__ lw,
                                                      int64_t lrows, float lscale, float* __restrict__ loss_out,
                                                      int stepblocks, int64_t bps, int64_t Lsteps,
                                                      float* __restrict__ db_steps) {
  const int nmain = (int)gridDim.x - dbblocks - stepblocks - (loss_out ? 1 : 0);
  if (stepblocks && (int)blockIdx.x >= nmain + dbblocks && (int)blockIdx.x < nmain + dbblocks + stepblocks) {
    sc_db_steps_block(dbp, bps, M, Lsteps, db_steps, (int)blockIdx.x - nmain - dbblocks);
    return;
  }
  if (loss_out && blockIdx.x == gridDim.x - 1) {
    // the step's scalar loss, gscale * sum_r row_w[r] * loss[r] over the row kernel's batch_loss, in a fixed order
    // (thread t takes float4s t, t + 256, ..., eight pairs of loads in flight; the 256 sub-sums meet in a fixed
    // tree) -- was a launch of its own (mean / weighted dot) between the forward and the backward kernels.
    // (Block sums out of the row kernel would be less to read, but ANY code added to k_sc_rows -- one store --
    // cost a register, 88 -> 89 = one allocation granule, and ~6 us of the C3 step: measured.)
    __shared__ float sls[256];
    const int64_t n4 = lrows >> 2;
    float t = 0.f;
    for (int64_t b0 = threadIdx.x; b0 < n4; b0 += 8 * 256) {
      float4 v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t q = b0 + u * 256;
        v[u] = q < n4 ? reinterpret_cast<const float4*>(lrow)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        w[u] = (q < n4 && lw) ? reinterpret_cast<const float4*>(lw)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (v[u].x * w[u].x + v[u].y * w[u].y) + (v[u].z * w[u].z + v[u].w * w[u].w);
    }
    if (threadIdx.x == 0)
      for (int64_t r = n4 * 4; r < lrows; ++r) t += lrow[r] * (lw ? lw[r] : 1.f);
    sls[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) sls[threadIdx.x] += sls[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = lscale * sls[0];
    return;
  }
  if ((int)blockIdx.x >= nmain) {
    // 64 columns per workgroup: thread = (row group rg of 16, four columns cg); a thread adds up the partial rows
    // rg, rg + 16, ... with sixteen 16-byte loads in flight (four waves walking 128 rows each, four loads at a time,
    // were a chain of 32 L2 round trips: 12 us -- longer than the product this kernel finishes)
    __shared__ float4 sdb[16][16];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)((int)blockIdx.x - nmain) * 64 + cg * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 < M) {
      for (int64_t b = rg; b < nblk; b += 16 * 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int64_t bb = b + 16 * u;
          v[u] = bb < nblk ? *reinterpret_cast<const float4*>(dbp + bb * M + m0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
      }
    }
    sdb[rg][cg] = a;
    __syncthreads();
    if (rg == 0 && m0 < M) {
      float4 t = sdb[0][cg];
      for (int r = 1; r < 16; ++r) { const float4 o = sdb[r][cg]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      *reinterpret_cast<float4*>(db + m0) = t;
    }
    return;
  }
  // 64 float4 outputs per workgroup; wave w adds up its quarter of the slices (loads in batches of four, added in
  // slice order), the four sub-sums meet in LDS in wave order: every load of a thread is in flight at once (one
  // thread per output walked the slices as a chain of L2 round trips: 11-16 us for 16 slices of 0.5 MB)
  __shared__ float4 sq[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 64 + lane;
  const int n4 = N / 4;
  const bool live = q < M * n4;
  const int64_t m = live ? q / n4 : 0;
  const int c = live ? (int)(q % n4) * 4 : 0;
  const int per = (nsl + 3) / 4, s0 = wv * per, s1 = min(nsl, s0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* p0 = part + m * N + c;
    const int64_t sl = M * (int64_t)N;
    int sI = s0;
    for (; sI + 3 < s1; sI += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(p0 + (int64_t)sI * sl);
      const float4 v1 = *reinterpret_cast<const float4*>(p0 + (int64_t)(sI + 1) * sl);
      const float4 v2 = *reinterpret_cast<const float4*>(p0 + (int64_t)(sI + 2) * sl);
      const float4 v3 = *reinterpret_cast<const float4*>(p0 + (int64_t)(sI + 3) * sl);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
      acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; sI < s1; ++sI) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + (int64_t)sI * sl);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  sq[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && live) {
    const float4 a = sq[0][lane], b = sq[1][lane], e = sq[2][lane], f = sq[3][lane];
    acc.x = (a.x + b.x) + (e.x + f.x); acc.y = (a.y + b.y) + (e.y + f.y);
    acc.z = (a.z + b.z) + (e.z + f.z); acc.w = (a.w + b.w) + (e.w + f.w);
    float4* cp = reinterpret_cast<float4*>(C + m * ldc + c);
    if (beta != 0.f) {
      const float4 o = *cp;
      acc.x += beta * o.x; acc.y += beta * o.y; acc.z += beta * o.z; acc.w += beta * o.w;
    }
    *cp = acc;
  }
}

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

// ============================================================================================================
// The 'mce' family (round 5): sampled softmax in the shape of 'mw' (build-defined, arx.h), loss_r = log(1 + sum_s
// m_rs exp(x_rs - t_r)), WITHOUT [B, S] logits or weights in HBM.  The backward weight of a logit is dense,
// w_rs = coef_r e_rs with e_rs = m_rs exp(x_rs - t_r) and coef_r = g_r / (1 + s_r), not a bit -- so the weight tile is
// recomputed where it is used and consumed out of the accumulator registers:
//
//   k_mc_flow: a wave keeps 32 STATIONARY rows as bf16 pieces in registers and streams 32-index tiles of the other
//   side through LDS (LDS-DMA, double buffered): six-term x tile = D[stream index][stationary] as in k_sc_hinge, e =
//   exp(x + c_stream + c_stat) in the accumulator layout -- which IS the second-operand layout of the next product
//   if the k slots of a 32x32x16 MFMA are dealt as  slot (kg, i) <-> stream index 16 q + 8 (i / 4) + 4 kg + i % 4:
//   the lane's values 8 q .. 8 q + 7, split into three exact bf16 pieces, are operand fragments as they stand, and
//   the first operand (the streamed side's TRANSPOSED planes) is read from LDS as two 8-byte halves in the same
//   order.  O[stationary][:] += sum_stream e . X[stream][:] with six terms again; rsum[stationary] += sum e (. cw).
//     role dU (DI = false): stationary = batch rows (U split in the prologue), stream = pool columns (Pp / PT of
//       k_sc_prep), c_stat = -t_r, c_stream = bias: O_r = sum_s e_rs P_s, s_r = sum_s e_rs -- the loss and BOTH
//       the forward and the latent-side backward product in one pass (the row factor coef_r commutes with it).
//     role dI (DI = true): stationary = pool columns (pieces from Pp), stream = batch rows (Up: U as planes,
//       UgT: coef_r U_r transposed, both by k_mc_rows), c_stream = -t_r (-inf in the padding), cw = coef:
//       part[slice][s][:] = sum_r e_rs coef_r U_r, dbpart[slice][s] = sum_r e_rs coef_r; slices = time steps for
//       the sequence model (its per-step clip norm), summed by k_sc_tn_reduce.
//   The positives of a row that sit in the pool are MASKED (m_rs = 0) from one bit table, maskR[s / 32][user row],
//   WRITTEN by k_sc_prep<true>'s walk of the positives chain, one walk per user row (the sequence model's time-major
//   rows repeat theirs L times): the dU role's lane reads its row's word of the tile, the dI role reads the same
//   words across (a tile's 32 rows x the wave's 32 columns are the 32 words of one word row) -- a word is non-zero
//   once in thousands of tiles, one wave-wide test per tile.
//   k_mc_rows (between the two): s_r, loss_r = log1p(s_r), coef_r, dt_r = -coef_r s_r, dU_r = coef_r O_r + dt_r T_r,
//   dT_r = dt_r U_r, the planes Up / UgT, -t_r.
// Two workgroups of four waves per CU (two MFMA-issuing waves per SIMD); d = 64 (x accumulators 32 + O accumulators
// 64 + stationary pieces 48 registers; d = 128 would need 352).
// Measured at the C4 shape (51 200 rows x 1 024 columns, tools/mcebench.py; ablation builds give WRONG results, the
// clock is what counts): a flow launch 96 - 100 us = 0.33 of the bf16 peak for its 12 terms (k_sc_hinge<64>: 0.28);
// no MFMA 30 us, no product MFMAs 66, no exp -2, no split -8, no barrier -4, one workgroup per CU 122 (the second
// wave per SIMD gives 1.22x), three LDS slots +-0, s_setprio on every other workgroup +-0, the q = 1 split between
// the MFMAs of the q = 0 products (kept) +-0, fragments requested one group ahead -3; "no LDS-DMA" -23 is the
// matrix pipe multiplying zeros at a higher clock, not the DMA.  The walk of the positives in a launch of its own
// took 18 us in the step (now k_sc_prep<true>), slices on one XCD -8 us for the dI role.
// ============================================================================================================
#ifndef MC_SLOTS
#define MC_SLOTS 2
#endif
constexpr int kMcSlots = MC_SLOTS;           // LDS ring: the stage being read + kMcSlots - 1 in flight
// a stage: K part (3 planes x 32 rows x KD k) + T part (3 planes x KD rows x 32 indices) + 1 KB: the tile's 32 + 32
// constants -- 12 + 12 + 1 KB at KD = 64, 24 + 24 + 1 KB at KD = 128
constexpr int mc_part(int KD) { return 192 * KD; }
constexpr int mc_stage(int KD) { return 2 * mc_part(KD) + 1024; }

struct McFlow {
  int64_t nstat;                             // stationary extent
  int64_t span;                              // streamed indices per slice (multiple of 32)
  int64_t nstream;                           // streamed extent (multiple of 32; planes hold that many rows)
  const float* statF; int64_t ldstat;        // dU role: U [nstat][ldstat] f32
  const uint16_t* statP; int64_t statP_plane;   // dI role: planes [3][nstat][64]
  const uint16_t* XK; int64_t xk_plane;      // streamed planes, k contiguous [3][nstream][64] (elements per plane)
  const uint16_t* XT; int64_t ldx;           // streamed planes, transposed [3][64][ldx]
  const float* cstat; float sgn_stat;        // exponent constant of a stationary index (nullable), times sgn
  const float* cstream;                      // ... of a streamed index (nullable: 0)
  const float* cw;                           // dI role: weight of a streamed index in rsum
  const uint32_t* mask; int64_t ldmask;      // maskR [S / 32][ldmask]: bit (s & 31) of word [(s >> 5) * ldmask + r] <-> pair (r, s) masked
  const float* zeros;                        // >= 16 bytes of zeros (what an absent constant array reads)
  int64_t mask_mod;                          // the table's columns are USER rows: batch row r reads column r % mask_mod
  float* O; int64_t o_rows;                  // [slices][o_rows][64]
  float* rsum; int64_t rs_rows;              // [slices][rs_rows]
};

template <bool DI, int KD>
__global__ __launch_bounds__(256, KD == 64 ? 2 : 1) void k_mc_flow(McFlow a) {
  constexpr int NCH = KD / 16;                 // k-chunks of the x tile
  constexpr int NB = KD / 32;                  // 32-wide blocks of the product's second index
  constexpr int PART = mc_part(KD), kMcStage = mc_stage(KD);
  constexpr int NPC = PART / 2048;             // LDS-DMA pieces (1 KB) per loader wave and stage: 6 / 12
  constexpr int ROWB = 2 * KD;                 // bytes of a K-part row
  constexpr int PK = 32 * ROWB;                // plane stride of the K part
  constexpr int PTS = KD * 64;                 // plane stride of the T part
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // [kMcSlots][kMcStage]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 31, kg = lane >> 5;
  const int64_t nsb = (a.nstat + 127) / 128;
  // workgroup -> (stationary block sb, slice sl).  Workgroups are dealt to the 8 XCDs round-robin; the dI role's nsb
  // workgroups of a slice stream the SAME rows of the batch: they sit 8 apart, i.e. on one XCD, whose L2 then fetches
  // the slice once (with plain division every slice came over the fabric nsb times)
  int64_t sb, sl;
  if (DI) {
    const int64_t bid = blockIdx.x, grp = 8 * nsb, full = (int64_t)gridDim.x / grp * grp;
    if (bid < full) {
      sl = bid / grp * 8 + bid % 8;
      sb = bid % grp / 8;
    } else {
      sl = full / nsb + (bid - full) / nsb;
      sb = (bid - full) % nsb;
    }
  } else {
    sb = (int64_t)blockIdx.x % nsb;
    sl = (int64_t)blockIdx.x / nsb;
  }
  const int64_t n_begin = sl * a.span;
  const int64_t n_end = n_begin + a.span < a.nstream ? n_begin + a.span : a.nstream;
  const int ntile = (int)((n_end - n_begin) / 32);
  const int64_t stat = sb * 128 + wv * 32 + lr;
  const bool ok = stat < a.nstat;
  const int64_t stat_c = ok ? stat : 0;

  // ---- the LDS-DMA pieces of a stage: waves 0 / 1 the K part (pieces 0 .. 11), waves 2 / 3 the T part ----
  // Issued through inline asm: the compiler orders every LDS read behind an outstanding __builtin_amdgcn_global_load_lds
  // (s_waitcnt vmcnt(0) at the top of the loop: the prefetch of tile t + 1 was waited for in front of tile t, 107 us
  // per launch at the C4 shape) -- k_sc_hinge avoids that with loader waves of their own, here the four compute waves
  // are all a CU's register file holds.  The waits are the explicit ones in front of the stage barriers.
  const char* src[NPC];
  int64_t adv;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int p = (wv & 1) * NPC + i;
    if (wv < 2) {      // K part: pieces of 1024 / ROWB rows, a row = ROWB / 16 chunks, chunk position swizzled by the row
      constexpr int RPP = 1024 / ROWB, CPR = ROWB / 16, PPP = 32 / RPP;
      const int pl = p / PPP, rg = p % PPP;
      const int row = rg * RPP + lane / CPR, pos = lane % CPR, q = pos ^ sc_swz<ROWB>(row);
      src[i] = reinterpret_cast<const char*>(a.XK) + ((int64_t)pl * a.xk_plane + (n_begin + row) * KD) * 2 + q * 16;
    } else {           // T part: pieces of 16 rows of 64 bytes
      constexpr int PPP = KD / 16;
      const int pl = p / PPP, rg = p % PPP;
      const int row = rg * 16 + (lane >> 2), pos = lane & 3, q = pos ^ ((row >> 2) & 3);
      src[i] = reinterpret_cast<const char*>(a.XT) + ((int64_t)pl * KD * a.ldx + (int64_t)row * a.ldx + n_begin) * 2 + q * 16;
    }
  }
  adv = wv < 2 ? 32 * ROWB : 64;
  // the tile's constants and mask words (wave 0's seventh piece): lanes 0..7 c_stream[32 t ..], lanes 8..15 cw[32 t ..],
  // lanes 16..47 the mask words of the workgroup's 128 stationary indices, the others re-fetch `zeros`
  const char* srcx = reinterpret_cast<const char*>(a.zeros);
  int64_t advx = 0;
  if (lane < 8) {
    if (a.cstream) { srcx = reinterpret_cast<const char*>(a.cstream + n_begin + 4 * lane); advx = 128; }
  } else if (lane < 16) {
    if (DI) { srcx = reinterpret_cast<const char*>(a.cw + n_begin + 4 * (lane - 8)); advx = 128; }
  } else if (lane < 48) {
    if (!DI) {       // word (tile, row) of the workgroup's 128 rows: bit i <-> column 32 tile + i
      srcx = reinterpret_cast<const char*>(a.mask + (n_begin / 32) * a.ldmask + (sb * 128) % a.mask_mod + 4 * (lane - 16));
      advx = a.ldmask * 4;
    } else {         // the SAME table read across: wave w's 32 columns are word row 4 sb + w, the tile's 32 rows its words
      srcx = reinterpret_cast<const char*>(a.mask + (sb * 4 + ((lane - 16) >> 3)) * a.ldmask + n_begin % a.mask_mod +
                                           4 * ((lane - 16) & 7));
      advx = 128;
    }
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  auto dma = [&](uint32_t dst, const char* p) {
#ifdef MC_ABL_NODMA
    return;
#endif
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                 :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(p) : "memory");
  };
  auto issue = [&](int slot) {                     // (tiles are issued in order: the sources step along)
    const uint32_t stage = lds0 + (uint32_t)(slot * kMcStage + wv * NPC * 1024);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      dma(stage + i * 1024, src[i]);
      src[i] += adv;
    }
    if (wv == 0) {
      dma(lds0 + (uint32_t)(slot * kMcStage + 2 * PART), srcx);
      srcx += advx;
    }
  };
#pragma unroll
  for (int q = 0; q < kMcSlots - 1; ++q)
    if (q < ntile) issue(q);

  // ---- the stationary rows as bf16 pieces (second MFMA operand: lane = row lr, k = 16 c + 8 kg ..) ----
  bf16x8 a1[NCH], a2[NCH], a3[NCH];
  if (!DI) {
    const float* up = a.statF + stat_c * a.ldstat + 8 * kg;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 r0 = *reinterpret_cast<const float4*>(up + 16 * c);
      const float4 r1 = *reinterpret_cast<const float4*>(up + 16 * c + 4);
      uint32_t p1[4], p2[4], p3[4];
      split3x2(r0.x, r0.y, p1[0], p2[0], p3[0]);
      split3x2(r0.z, r0.w, p1[1], p2[1], p3[1]);
      split3x2(r1.x, r1.y, p1[2], p2[2], p3[2]);
      split3x2(r1.z, r1.w, p1[3], p2[3], p3[3]);
      a1[c] = __builtin_bit_cast(bf16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));
      a2[c] = __builtin_bit_cast(bf16x8, make_uint4(p2[0], p2[1], p2[2], p2[3]));
      a3[c] = __builtin_bit_cast(bf16x8, make_uint4(p3[0], p3[1], p3[2], p3[3]));
    }
  } else {
    const uint16_t* pp = a.statP + stat_c * KD + 8 * kg;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      a1[c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pp + 16 * c));
      a2[c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pp + a.statP_plane + 16 * c));
      a3[c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pp + 2 * a.statP_plane + 16 * c));
    }
  }
  const float cst = (a.cstat && ok) ? a.sgn_stat * a.cstat[stat] : 0.f;

  f32x16 hiO[NB], loO[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) hiO[q][e] = loO[q][e] = 0.f;
  float rs = 0.f;
  // LDS byte offsets of this lane's fragments inside a stage
  const uint32_t fk = (uint32_t)(lr * ROWB);                         // K part row
  const int swk = sc_swz<ROWB>(lr);
  const uint32_t ft = (uint32_t)(PART + lr * 64 + 8 * kg);            // T part row of dd block 0 (block b: + 2048 b)
  const int swt = (lr >> 2) & 3;                                      // ((32 b + lr) >> 2) & 3 is the same
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int slot = 0, slot_in = kMcSlots - 1;            // the slot being read / the slot the next request goes to
  for (int t = 0; t < ntile; ++t) {
    const char* st = lds + slot * kMcStage;
    if (t + kMcSlots - 1 < ntile) issue(slot_in);  // (read last in tile t - 1: every wave is past that barrier)
    slot = slot + 1 == kMcSlots ? 0 : slot + 1;
    slot_in = slot_in + 1 == kMcSlots ? 0 : slot_in + 1;
    const uint32_t mw = reinterpret_cast<const uint32_t*>(st + 2 * PART + 256)[wv * 32 + lr];
    // The instruction stream of a tile is pinned group by group (sched_barrier): the fragments of group g + 1 are
    // requested in front of the six MFMAs of group g -- left alone the scheduler emits read / wait / MFMA triples and
    // the matrix pipe idles for an LDS round trip per fragment.  Groups: four k-chunks of the x tile, then four
    // (q, dd block) products; the first product's fragments are requested under the last chunk, exp / split sit between.
    uint4 fr[2][3];
#define MC_LDK(set_, c_)                                                                                 \
  {                                                                                                      \
    const uint32_t fa = fk + 16 * ((2 * (c_) + kg) ^ swk);                                               \
    fr[set_][0] = *reinterpret_cast<const uint4*>(st + fa);                                              \
    fr[set_][1] = *reinterpret_cast<const uint4*>(st + fa + PK);                                         \
    fr[set_][2] = *reinterpret_cast<const uint4*>(st + fa + 2 * PK);                                     \
  }
#define MC_LDT(set_, g_)                                                                                 \
  {                                                                                                      \
    const uint32_t o0 = ft + 16 * ((2 * ((g_) / NB)) ^ swt) + ((g_) % NB) * 2048;                        \
    const uint32_t o1 = ft + 16 * ((2 * ((g_) / NB) + 1) ^ swt) + ((g_) % NB) * 2048;                    \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                   \
      const uint2 h0 = *reinterpret_cast<const uint2*>(st + o0 + pl * PTS);                              \
      const uint2 h1 = *reinterpret_cast<const uint2*>(st + o1 + pl * PTS);                              \
      fr[set_][pl] = make_uint4(h0.x, h0.y, h1.x, h1.y);                                                 \
    }                                                                                                    \
  }
    // ---- x tile: D[stream index][stationary], hi starts at c_stream + c_stat ----
    f32x16 hi, lo;
    const float* cs = reinterpret_cast<const float*>(st + 2 * PART) + 4 * kg;
    MC_LDK(0, 0)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 cv = *reinterpret_cast<const float4*>(cs + 8 * g);
      hi[4 * g] = cv.x + cst; hi[4 * g + 1] = cv.y + cst; hi[4 * g + 2] = cv.z + cst; hi[4 * g + 3] = cv.w + cst;
      lo[4 * g] = 0.f; lo[4 * g + 1] = 0.f; lo[4 * g + 2] = 0.f; lo[4 * g + 3] = 0.f;
    }
    float4 cwv[4];
    if (DI) {
      const float* cwp = reinterpret_cast<const float*>(st + 2 * PART + 128) + 4 * kg;
#pragma unroll
      for (int g = 0; g < 4; ++g) cwv[g] = *reinterpret_cast<const float4*>(cwp + 8 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, fr[c & 1][0]), b2 = __builtin_bit_cast(bf16x8, fr[c & 1][1]),
                   b3 = __builtin_bit_cast(bf16x8, fr[c & 1][2]);
      if (c + 1 < NCH) MC_LDK((c + 1) & 1, c + 1)
      else MC_LDT(0, 0)
      __builtin_amdgcn_sched_barrier(0);
      lo = SC_MFMA(b3, a1[c], lo);
      hi = SC_MFMA(b1, a1[c], hi);
      lo = SC_MFMA(b1, a3[c], lo);
      lo = SC_MFMA(b2, a2[c], lo);
      lo = SC_MFMA(b2, a1[c], lo);
      lo = SC_MFMA(b1, a2[c], lo);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- e = exp(x) in the accumulator layout (value 4 g + e <-> stream index 8 g + 4 kg + e), masked ----
    float w[16];
#pragma unroll
#ifdef MC_ABL_NOEXP
    for (int i = 0; i < 16; ++i) w[i] = hi[i] + lo[i];
#else
    for (int i = 0; i < 16; ++i) {           // saturating exponent (common.h kMceSat): finite whatever the logits; NaN stays
      const float xa = hi[i] + lo[i];
      w[i] = __expf(xa > kMceSat ? kMceSat : xa);
    }
#endif
    if (__any(mw != 0u)) {
      // dU role: the lane's word holds its row's bits for the tile's columns; dI role: lane lr holds the word of the
      // tile's ROW lr (bits = the wave's 32 columns): value i belongs to row 8 (i / 4) + 4 kg + i % 4, column lr
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 8 * (i >> 2) + 4 * kg + (i & 3);
        const uint32_t bit = DI ? (__shfl(mw, m, 64) >> lr) & 1u : (mw >> m) & 1u;
        if (bit) w[i] = 0.f;
      }
    }
    if (DI) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        rs += w[4 * g] * cwv[g].x; rs += w[4 * g + 1] * cwv[g].y; rs += w[4 * g + 2] * cwv[g].z; rs += w[4 * g + 3] * cwv[g].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) rs += w[i];
    }
    // ---- O[dd][stationary] += X^T[dd][stream] . e[stream][stationary], six terms ----
    // the split of values 8..15 (q = 1) sits in the MFMA gaps of the q = 0 products
    bf16x8 w1[2], w2[2], w3[2];
#define MC_SPLITQ(q_)                                                                                    \
  {                                                                                                      \
    uint32_t p1[4], p2[4], p3[4];                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
      split3x2(w[8 * (q_) + 2 * j], w[8 * (q_) + 2 * j + 1], p1[j], p2[j], p3[j]);                       \
    w1[q_] = __builtin_bit_cast(bf16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));                         \
    w2[q_] = __builtin_bit_cast(bf16x8, make_uint4(p2[0], p2[1], p2[2], p2[3]));                         \
    w3[q_] = __builtin_bit_cast(bf16x8, make_uint4(p3[0], p3[1], p3[2], p3[3]));                         \
  }
    MC_SPLITQ(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 2 * NB; ++g) {
      const int q = g / NB;
      const bf16x8 t1 = __builtin_bit_cast(bf16x8, fr[g & 1][0]), t2 = __builtin_bit_cast(bf16x8, fr[g & 1][1]),
                   t3 = __builtin_bit_cast(bf16x8, fr[g & 1][2]);
      if (g + 1 < 2 * NB) MC_LDT((g + 1) & 1, g + 1)
      __builtin_amdgcn_sched_barrier(0);
      f32x16& hO = hiO[g % NB];
      f32x16& lO = loO[g % NB];
      lO = SC_MFMA(t3, w1[q], lO);
      hO = SC_MFMA(t1, w1[q], hO);
      lO = SC_MFMA(t1, w3[q], lO);
      lO = SC_MFMA(t2, w2[q], lO);
      lO = SC_MFMA(t2, w1[q], lO);
      lO = SC_MFMA(t1, w2[q], lO);
      if (g == 0) {
        MC_SPLITQ(1)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef MC_SPLITQ
#undef MC_LDK
#undef MC_LDT
#ifdef MC_ABL_NOBAR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    // tile t + 1 has landed; with three slots the requests of tile t + 2 (seven by wave 0, six by the others) stay
    // in flight
    if (kMcSlots == 3 && KD == 64 && t + 2 < ntile) {
      if (wv == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#endif
  }
  rs += __shfl_xor(rs, 32, 64);
  if (!ok) return;
  if (kg == 0) a.rsum[sl * a.rs_rows + stat] = rs;
  float* orow = a.O + (sl * a.o_rows + stat) * KD + 4 * kg;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(orow + 32 * b + 8 * g) =
          make_float4(hiO[b][4 * g] + loO[b][4 * g], hiO[b][4 * g + 1] + loO[b][4 * g + 1],
                      hiO[b][4 * g + 2] + loO[b][4 * g + 2], hiO[b][4 * g + 3] + loO[b][4 * g + 3]);
}

struct McRows {
  const float* rs_part; const float* O_part; int nsplit; int64_t Bp;   // [nsplit][B], [nsplit][Bp][64]
  const float* tscore;
  const float* U; int64_t ldu;
  const float* T; int64_t ldt;
  float gscale; const float* row_w;
  float* batch_loss; float* coef_out; float* tneg;                     // [B], [Bp], [Bp]
  float* dtscore; int64_t dts_stride;
  float* dU; int64_t lddu;
  float* dT; int64_t lddt;
  uint16_t* Up;                                                        // [3][Bp][64]
  uint16_t* UgT; int64_t ldug;                                         // [3][64][ldug]
  const float* pool_bad; int nbad;
};

// 32 batch rows per workgroup, LPR = d / 4 lanes x float4 per row: 4 waves x (2 rounds x 4 rows at d = 64, 4 rounds x 2
// rows at d = 128)
template <int LPR>
__global__ __launch_bounds__(256) void k_mc_rows(McRows a, int64_t B) {
  constexpr int D = 4 * LPR, RPW = 64 / LPR, NIT = 8 / RPW;
  __shared__ float tile[32 * (D + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int c4 = (lane & (LPR - 1)) * 4;
  const float pbad = sc_wsum(lane < a.nbad ? a.pool_bad[lane] : 0.f);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int rl = wv * 8 + it * RPW + lane / LPR;
    const int64_t r = r0 + rl;
    const bool live = r < B;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), tt = u, o = u;
    float s = 0.f, t = 0.f;
    if (live) {
      u = *reinterpret_cast<const float4*>(a.U + r * a.ldu + c4);
      tt = *reinterpret_cast<const float4*>(a.T + r * a.ldt + c4);
      for (int p = 0; p < a.nsplit; ++p) {                    // fixed order
        s += a.rs_part[(int64_t)p * B + r];
        const float4 v = *reinterpret_cast<const float4*>(a.O_part + ((int64_t)p * a.Bp + r) * D + c4);
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      }
      t = a.tscore[r];
    }
    // non-finite inputs poison the row's loss like an f32 chain would
    float poison = ((u.x + u.y) + (u.z + u.w) + (tt.x + tt.y) + (tt.z + tt.w)) * 0.f;
#pragma unroll
    for (int m = LPR / 2; m > 0; m >>= 1) poison += __shfl_xor(poison, m, 64);
    s += pbad + poison + t * 0.f;
    const float gw = live ? a.gscale * (a.row_w ? a.row_w[r] : 1.f) : 0.f;
    // (s is finite for finite inputs since the exponent saturates; the limits keep an overflowed sum -- more than
    // 2^20 saturated pairs in a row -- from turning 0 * inf into NaN: coef -> 0, dt -> -g)
    const bool sinf = s == __builtin_inff();
    const float coef = sinf ? 0.f : gw / (1.f + s);
    const float dt = sinf ? -gw : -coef * s;
    if ((lane & (LPR - 1)) == 0) {
      if (live) {
        if (a.batch_loss) a.batch_loss[r] = log1pf(s);
        if (a.dtscore) a.dtscore[r * a.dts_stride] = dt;
      }
      a.coef_out[r] = live ? coef : 0.f;
      a.tneg[r] = live ? -t : -__builtin_inff();
    }
    if (live) {
      if (a.dT) *reinterpret_cast<float4*>(a.dT + r * a.lddt + c4) = make_float4(dt * u.x, dt * u.y, dt * u.z, dt * u.w);
      if (a.dU)
        *reinterpret_cast<float4*>(a.dU + r * a.lddu + c4) =
            make_float4(coef * o.x + dt * tt.x, coef * o.y + dt * tt.y, coef * o.z + dt * tt.z, coef * o.w + dt * tt.w);
    }
    uint32_t pa[3], pb[3];
    split3x2(u.x, u.y, pa[0], pa[1], pa[2]);
    split3x2(u.z, u.w, pb[0], pb[1], pb[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint2*>(a.Up + ((int64_t)p * a.Bp + r) * D + c4) = make_uint2(pa[p], pb[p]);
    float* tp = tile + rl * (D + 1) + c4;
    const float cu = live ? coef : 0.f;
    tp[0] = cu * u.x; tp[1] = cu * u.y; tp[2] = cu * u.z; tp[3] = cu * u.w;
  }
  __syncthreads();
  sc_emit_planes_t(tile, D + 1, 32, D, a.UgT, (int64_t)D * a.ldug, a.ldug, r0, tid, 256);
}


}  // namespace

// Layout of the caller's scorer state (one buffer): everything the three launches of the forward leave for the two
// backward products.  Offsets in bytes.
struct ScLayout {
  int64_t Bp, ldbits, ldbt, nblk;
  int64_t ldpt, ldug;       // row strides (bf16 elements) of the TRANSPOSED planes PT [3][d][ldpt], UgT [3][d][ldug]: a
                            // power-of-two stride (2 KB, 32 KB) would put the 32 rows of a stage into one or two
                            // memory channels -- measured: the dI product 143 us instead of ~15
  int nsplit, CW;
  size_t bits, bitsT, g, t, rs, cnt, hits, nhit, Pp, PT, UgT, dbp, pbad, total;
};

static bool sc_layout(int64_t B, int64_t S, int d, ScLayout* L) {
  if (!(d == 64 || d == 128) || S % 128 != 0 || S < 128 || S > 2048 || B < 1) return false;
  L->Bp = (B + 127) / 128 * 128;
  L->ldbits = (B + 255) / 256 * 256;      // rows of 256-word pieces: the backward kernels fetch them by LDS-DMA
  L->ldbt = (S + 255) / 256 * 256;
  L->nblk = L->Bp / 32;
  L->ldpt = S + 128;
  L->ldug = L->Bp + 128;
  L->CW = S >= 1024 ? 512 : (S >= 256 ? (int)(S / 2) : (int)S);
  L->nsplit = (int)(S / L->CW);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  L->bits = take((size_t)(S / 32) * L->ldbits * 4);
  L->bitsT = take((size_t)L->nblk * L->ldbt * 4);
  L->g = take((size_t)L->Bp * 4);
  L->t = take((size_t)B * 4);
  L->rs = take((size_t)L->nsplit * B * 4);
  L->cnt = take((size_t)L->nsplit * B * 4);
  L->hits = take((size_t)B * kScHits * 4);
  L->nhit = take((size_t)B * 4);
  L->Pp = take((size_t)3 * S * d * 2);
  L->PT = take((size_t)3 * d * L->ldpt * 2);
  L->UgT = take((size_t)3 * d * L->ldug * 2);
  L->dbp = take((size_t)L->nblk * S * 4);
  L->pbad = take(64 * 4);
  L->total = o;
  return true;
}

static int sc_raise_lds() {
  // (per call: the attribute is per device and the call is cheap)
  const int cap = 160 * 1024;
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_hinge<128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_hinge<64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_bits<false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_bits<true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

extern "C" {

#if defined(SC_TRACE) || defined(SC_TRACE_ROWS)
int arx_sc_trace_read(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sc_trace), sizeof(g_sc_trace)); }
#endif

int arx_mw_scorer_supported(int64_t B, int64_t S, int d) {
  ScLayout L;
  return sc_layout(B, S, d, &L) ? 1 : 0;
}

size_t arx_mw_scorer_state_bytes(int64_t B, int64_t S, int d) {
  ScLayout L;
  return sc_layout(B, S, d, &L) ? L.total : 0;
}

/* offsets (bytes) of the state's regions a caller may look at: out[0] act bits (word-major [S / 32][out[1] = ld]),
 * out[2] transposed bits [Bp / 32][out[5] = ld], out[3] g [Bp], out[4] Bp */
int arx_mw_scorer_state_layout(int64_t B, int64_t S, int d, int64_t* out) {
  ScLayout L;
  ARX_CHECK_ARG(out && sc_layout(B, S, d, &L), "arx_mw_scorer_state_layout: shape not supported");
  out[0] = (int64_t)L.bits;
  out[1] = L.ldbits;
  out[2] = (int64_t)L.bitsT;
  out[3] = (int64_t)L.g;
  out[4] = L.Bp;
  out[5] = L.ldbt;
  return ARX_OK;
}

int arx_mw_scorer_fwd_phases(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias,
                             const float* T, int64_t ldt, const float* tbias, int64_t tb_stride, int d,
                             const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                             const int32_t* item2slot, int64_t mask_rows, float gscale, const float* row_w, int64_t B,
                             int64_t S, float* batch_loss, float* tscore_out, float* dtscore, int64_t dtscore_stride,
                             float* dU, int64_t lddu, float* dT, int64_t lddt, void* state, size_t state_bytes,
                             int phases, void* stream) {
  return arx_mw_scorer_fwd_seqw(U, ldu, P, ldp, pbias, T, ldt, tbias, tb_stride, d, user_ids, pos_ptr, pos_items,
                                item2slot, mask_rows, gscale, const_cast<float*>(row_w), nullptr, 0, B, S, batch_loss,
                                tscore_out, dtscore, dtscore_stride, dU, lddu, dT, lddt, state, state_bytes, phases,
                                stream);
}

int arx_mw_scorer_fwd_seqw(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias,
                           const float* T, int64_t ldt, const float* tbias, int64_t tb_stride, int d,
                           const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                           const int32_t* item2slot, int64_t mask_rows, float gscale, float* row_w,
                           const float* seq_w, int64_t seq_rows, int64_t B, int64_t S, float* batch_loss,
                           float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu,
                           float* dT, int64_t lddt, void* state, size_t state_bytes, int phases, void* stream) {
  ARX_CHECK_ARG(U && P && T && user_ids && pos_ptr && pos_items && item2slot && state,
                "arx_mw_scorer_fwd: null pointer");
  ARX_CHECK_ARG(!seq_w || (row_w && seq_rows > 0 && B % seq_rows == 0),
                "arx_mw_scorer_fwd_seqw: sequence weights need row_w [B] to write and seq_rows dividing B");
  ScLayout L;
  const bool ok = sc_layout(B, S, d, &L) && ldu % 4 == 0 && ldp % 4 == 0 && ldt % 4 == 0 && ldu >= d && ldp >= d &&
                  ldt >= d && (!dU || (lddu % 4 == 0 && lddu >= d)) && (!dT || (lddt % 4 == 0 && lddt >= d)) &&
                  ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(T) |
                    reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(dT) |
                    reinterpret_cast<uintptr_t>(state)) & 15) == 0;
  if (!ok) {
    set_error("arx_mw_scorer_fwd: shape not supported (d in {64, 128}, S %% 128 == 0, 128 <= S <= 2048, 16-byte rows)");
    return ARX_EUNSUPPORTED;
  }
  if (state_bytes < L.total) {
    set_error("arx_mw_scorer_fwd: state too small (%zu < %zu)", state_bytes, L.total);
    return ARX_EWORKSPACE;
  }
  if (int rc = sc_raise_lds()) return rc;
  hipStream_t s = as_stream(stream);
  char* st = reinterpret_cast<char*>(state);
  uint32_t* bits = reinterpret_cast<uint32_t*>(st + L.bits);
  float* t = reinterpret_cast<float*>(st + L.t);
  float* rs_part = reinterpret_cast<float*>(st + L.rs);
  float* cnt_part = reinterpret_cast<float*>(st + L.cnt);
  int32_t* hits = reinterpret_cast<int32_t*>(st + L.hits);
  int32_t* nhit = reinterpret_cast<int32_t*>(st + L.nhit);
  uint16_t* Pp = reinterpret_cast<uint16_t*>(st + L.Pp);
  uint16_t* PT = reinterpret_cast<uint16_t*>(st + L.PT);
  const PosMask pm = make_pos_mask(user_ids, pos_ptr, pos_items, item2slot);
  const int64_t mrows = mask_rows > 0 ? mask_rows : B;
  if (phases & 1) {
    const int64_t grid = S / 32 + ceil_div(B, 8) + (seq_w ? ceil_div(seq_rows, 32) : 0);
    const ScTScore ts{U, ldu, T, ldt, tbias, tb_stride > 0 ? tb_stride : 1, t, tscore_out};
    k_sc_prep<false><<<(int)grid, 256, 0, s>>>(P, ldp, S, d, pbias, Pp, PT, L.ldpt, reinterpret_cast<float*>(st + L.pbad),
                                               pm, mrows, B, hits, nhit, seq_w, seq_rows, row_w, ts, nullptr, 0);
    ARX_CHECK_LAUNCH();
  }
  if (phases & 2) {
    const int64_t grid = ceil_div(B, 128) * L.nsplit;
    const size_t lds = (size_t)kScSlots * kScStageBytes + (size_t)L.CW * 4 + (size_t)(L.CW / 32) * 128 * 4;
    if (d == 128)
      k_sc_hinge<128><<<(int)grid, 512, lds, s>>>(B, S, L.CW, U, ldu, t, Pp, pbias, bits, L.ldbits, rs_part, cnt_part);
    else
      k_sc_hinge<64><<<(int)grid, 512, lds, s>>>(B, S, L.CW, U, ldu, t, Pp, pbias, bits, L.ldbits, rs_part, cnt_part);
    ARX_CHECK_LAUNCH();
  }
  if (phases & 4) {
    ScRows a{rs_part, cnt_part, L.nsplit, bits, L.ldbits, reinterpret_cast<uint32_t*>(st + L.bitsT), L.ldbt, t, U, ldu, T, ldt,
             P, ldp, pbias, d, gscale, row_w, batch_loss, reinterpret_cast<float*>(st + L.g), dtscore,
             dtscore_stride > 0 ? dtscore_stride : 1, dU, lddu, dT, lddt, reinterpret_cast<uint16_t*>(st + L.UgT), L.ldug,
             reinterpret_cast<float*>(st + L.dbp), hits, nhit, reinterpret_cast<const float*>(st + L.pbad)};
    static const bool rows_old = getenv("ARX_SC_ROWS_OLD") != nullptr;     // (A/B: the four-rows-in-a-row kernel)
    if (!rows_old && d == 64) k_sc_rows_g<16><<<(int)L.nblk, 512, 0, s>>>(a, pm, mrows, B, S);
    else if (!rows_old && d == 128) k_sc_rows_g<32><<<(int)L.nblk, 512, 0, s>>>(a, pm, mrows, B, S);
    else k_sc_rows<<<(int)L.nblk, 512, 0, s>>>(a, pm, mrows, B, S);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

int arx_mw_scorer_fwd(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, const float* T,
                      int64_t ldt, const float* tbias, int64_t tb_stride, int d, const int32_t* user_ids,
                      const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                      float gscale, const float* row_w, int64_t B, int64_t S, float* batch_loss, float* tscore_out,
                      float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu, float* dT, int64_t lddt,
                      void* state, size_t state_bytes, void* stream) {
  return arx_mw_scorer_fwd_phases(U, ldu, P, ldp, pbias, T, ldt, tbias, tb_stride, d, user_ids, pos_ptr, pos_items,
                                  item2slot, mask_rows, gscale, row_w, B, S, batch_loss, tscore_out, dtscore,
                                  dtscore_stride, dU, lddu, dT, lddt, state, state_bytes, 7, stream);
}

/* dU[m, :] = beta dU[m, :] + g[m] * sum_s act[m][s] P[s, :]   (embed_attribute.py:171 backward, latent side) */
int arx_mw_scorer_bwd_du(int64_t B, int64_t S, int d, const void* state, float beta, float* dU, int64_t lddu,
                         void* stream) {
  ScLayout L;
  ARX_CHECK_ARG(state && dU && sc_layout(B, S, d, &L), "arx_mw_scorer_bwd_du: bad argument / shape");
  ARX_CHECK_ARG(lddu % 4 == 0 && lddu >= d && (reinterpret_cast<uintptr_t>(dU) & 15) == 0,
                "arx_mw_scorer_bwd_du: dU rows must be 16-byte aligned");
  if (int rc = sc_raise_lds()) return rc;
  const char* st = reinterpret_cast<const char*>(state);
  const int64_t grid = ceil_div(B, 256) * (d / 32);
  k_sc_bits<false><<<(int)grid, 512, (size_t)kScSlots * (kScStageBytes + 4096), as_stream(stream)>>>(
      B, d, S, S, reinterpret_cast<const uint32_t*>(st + L.bits), L.ldbits,
      reinterpret_cast<const uint16_t*>(st + L.PT), L.ldpt, beta, dU, lddu, reinterpret_cast<const float*>(st + L.g));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

size_t arx_mw_scorer_bwd_di_workspace_bytes(int64_t B, int64_t S, int d, int64_t step_rows) {
  ScLayout L;
  if (!sc_layout(B, S, d, &L)) return 0;
  int64_t ks = step_rows > 0 ? step_rows : 0;
  if (ks <= 0) {
    const int64_t wg = ceil_div(S, 256) * (d / 32);
    int64_t nsl = ceil_div((int64_t)cu_count(), wg);
    ks = ceil_div(ceil_div(L.Bp, nsl), 128) * 128;
  }
  const int64_t nsl = ceil_div(L.Bp, ks);
  return (size_t)nsl * S * d * 4 + 256;
}

/* dI[s, :] = beta dI[s, :] + sum_r act[r][s] g[r] U[r, :], db[s] = sum_r act[r][s] g[r]
 * (embed_attribute.py:171 backward, pool side).  step_rows > 0 (the sequence model: B = L * step_rows time-major
 * rows, step_rows % 128 == 0): the products of the single steps are kept as well -- dI_steps [L][S][d] (may alias
 * the workspace's slices: pass NULL to use the workspace) and db_steps [L][S] -- for TF-1.0's per-unrolled-step
 * clip norm (seqModel.py:179-180). */
int arx_mw_scorer_bwd_di(int64_t B, int64_t S, int d, const void* state, int64_t step_rows, float beta, float* dI,
                         int64_t lddi, float* db, float* dI_steps, float* db_steps, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return arx_mw_scorer_bwd_di_loss(B, S, d, state, step_rows, beta, dI, lddi, db, dI_steps, db_steps, nullptr, 0.f,
                                   nullptr, nullptr, workspace, workspace_bytes, stream);
}

int arx_mw_scorer_bwd_di_loss(int64_t B, int64_t S, int d, const void* state, int64_t step_rows, float beta,
                              float* dI, int64_t lddi, float* db, float* dI_steps, float* db_steps,
                              const float* batch_loss, float gscale, const float* row_w, float* loss_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  ScLayout L;
  ARX_CHECK_ARG(!loss_out || (batch_loss && (reinterpret_cast<uintptr_t>(batch_loss) & 15) == 0 &&
                              (reinterpret_cast<uintptr_t>(row_w) & 15) == 0),
                "arx_mw_scorer_bwd_di_loss: loss_out needs batch_loss (and row_w) 16-byte aligned");
  ARX_CHECK_ARG(state && dI && sc_layout(B, S, d, &L), "arx_mw_scorer_bwd_di: bad argument / shape");
  ARX_CHECK_ARG(lddi % 4 == 0 && lddi >= d && (reinterpret_cast<uintptr_t>(dI) & 15) == 0,
                "arx_mw_scorer_bwd_di: dI rows must be 16-byte aligned");
  ARX_CHECK_ARG(step_rows == 0 || (step_rows % 128 == 0 && B % step_rows == 0),
                "arx_mw_scorer_bwd_di: step_rows must divide B and be a multiple of 128");
  if (int rc = sc_raise_lds()) return rc;
  int64_t ks = step_rows;
  if (ks <= 0) {
    const int64_t wg = ceil_div(S, 256) * (d / 32);
    const int64_t nsl0 = ceil_div((int64_t)cu_count(), wg);
    ks = ceil_div(ceil_div(L.Bp, nsl0), 128) * 128;
  }
  const int64_t nsl = ceil_div(L.Bp, ks);
  float* part = dI_steps;
  if (!part) {
    const size_t need = (size_t)nsl * S * d * 4;
    if (!workspace || workspace_bytes < need) {
      set_error("arx_mw_scorer_bwd_di: workspace too small (%zu < %zu)", workspace_bytes, need);
      return ARX_EWORKSPACE;
    }
    part = reinterpret_cast<float*>(workspace);
  }
  hipStream_t s = as_stream(stream);
  const char* st = reinterpret_cast<const char*>(state);
  const int64_t grid = ceil_div(S, 256) * (d / 32) * nsl;
  k_sc_bits<true><<<(int)grid, 512, (size_t)kScSlots * (kScStageBytes + 4096), s>>>(
      S, d, ks, L.Bp, reinterpret_cast<const uint32_t*>(st + L.bitsT), L.ldbt,
      reinterpret_cast<const uint16_t*>(st + L.UgT), L.ldug, 0.f, part, d, nullptr);
  ARX_CHECK_LAUNCH();
  const float* dbp = reinterpret_cast<const float*>(st + L.dbp);
  const int dbblocks = db ? (int)ceil_div(S, 64) : 0;
  const bool steps = db_steps && step_rows > 0;
  const int64_t Lsteps = steps ? B / step_rows : 0, bps = steps ? step_rows / 32 : 0;
  const int stepblocks = steps ? (int)ceil_div(Lsteps * S, 256) : 0;
  k_sc_tn_reduce<<<(int)ceil_div(S * (d / 4), 64) + dbblocks + stepblocks + (loss_out ? 1 : 0), 256, 0, s>>>(
      part, (int)nsl, S, d, beta, dI, lddi, dbp, L.nblk, db, dbblocks, batch_loss, row_w, B, gscale, loss_out,
      stepblocks, bps, Lsteps, db_steps);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"

/* ---------------------------------------------------------------------------------------------------------------
 * 'mce' (build-defined sampled softmax, see arx_loss_mce_fwdbwd) on the fused family: no [B, S] logits or weights.
 * ------------------------------------------------------------------------------------------------------------- */
namespace arx {
namespace {

struct McLayout {
  int64_t Bp, ldpt, ldug, ldr;
  int nsplit;
  int64_t CW;
  size_t maskR, mask_bytes, t, tneg, coef, rs, O, hits, nhit, Pp, PT, Up, UgT, pbad, zeros, total;
};

bool mc_layout(int64_t B, int64_t S, int d, McLayout* L) {
  if (!(d == 64 || d == 128) || S % 128 != 0 || S < 128 || S > 2048 || B < 1) return false;
  L->Bp = (B + 127) / 128 * 128;
  L->ldpt = S + 128;                     // (not a power of two: see ScLayout)
  L->ldug = L->Bp + 128;
  L->ldr = L->Bp;
  // column splits of the dU role: >= 3 workgroups per CU (two are resident: 400 row blocks x 2 splits run 25 tile
  // rounds per CU where 400 x 1 run 32), no more -- every split writes its own O partial ([Bp][64] f32)
  int ns = 1;
  while (ns < 4 && (L->Bp / 128) * ns < 3 * (int64_t)cu_count() && S / (2 * ns) >= 128 && (S / (2 * ns)) % 32 == 0) ns *= 2;
  L->nsplit = ns;
  L->CW = S / ns;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  L->maskR = take((size_t)(S / 32) * L->ldr * 4);
  L->mask_bytes = o;
  L->t = take((size_t)B * 4);
  L->tneg = take((size_t)L->Bp * 4);
  L->coef = take((size_t)L->Bp * 4);
  L->rs = take((size_t)ns * B * 4);
  L->O = take((size_t)ns * L->Bp * d * 4);
  L->hits = take((size_t)B * kScHits * 4);
  L->nhit = take((size_t)B * 4);
  L->Pp = take((size_t)3 * S * d * 2);
  L->PT = take((size_t)3 * d * L->ldpt * 2);
  L->Up = take((size_t)3 * L->Bp * d * 2);
  L->UgT = take((size_t)3 * d * L->ldug * 2);
  L->pbad = take(64 * 4);
  L->zeros = take(256);                  // (never written: the caller zeroed the state once)
  L->total = o;
  return true;
}

int mc_raise_lds() {
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mc_flow<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mc_flow<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mc_flow<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mc_flow<true, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  return ARX_OK;
}

// slices of the dI role: time steps for the sequence model, else enough of them for two workgroups per CU (four: 128-row
// slices of four tiles, 46 us at B = 16 384; two: 41; one: 42)
int64_t mc_di_slice(const McLayout& L, int64_t S, int64_t step_rows) {
  if (step_rows > 0) return step_rows;
  const int64_t wg = S / 128;
  const int64_t nsl0 = ceil_div((int64_t)2 * cu_count(), wg);
  return ceil_div(ceil_div(L.Bp, nsl0), 128) * 128;
}

}  // namespace
}  // namespace arx

extern "C" {

int arx_mce_scorer_supported(int64_t B, int64_t S, int d) {
  McLayout L;
  return mc_layout(B, S, d, &L) ? 1 : 0;
}

size_t arx_mce_scorer_state_bytes(int64_t B, int64_t S, int d) {
  McLayout L;
  return mc_layout(B, S, d, &L) ? L.total : 0;
}

int arx_mce_scorer_fwd(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, const float* T,
                       int64_t ldt, const float* tbias, int64_t tb_stride, int d, const int32_t* user_ids,
                       const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                       float gscale, float* row_w, const float* seq_w, int64_t seq_rows, int64_t B, int64_t S,
                       float* batch_loss, float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU,
                       int64_t lddu, float* dT, int64_t lddt, void* state, size_t state_bytes, int phases,
                       void* stream) {
  ARX_CHECK_ARG(U && P && T && user_ids && pos_ptr && pos_items && item2slot && state,
                "arx_mce_scorer_fwd: null pointer");
  ARX_CHECK_ARG(!seq_w || (row_w && seq_rows > 0 && B % seq_rows == 0),
                "arx_mce_scorer_fwd: sequence weights need row_w [B] to write and seq_rows dividing B");
  McLayout L;
  const bool ok = mc_layout(B, S, d, &L) && ldu % 4 == 0 && ldp % 4 == 0 && ldt % 4 == 0 && ldu >= d && ldp >= d &&
                  ldt >= d && (!dU || (lddu % 4 == 0 && lddu >= d)) && (!dT || (lddt % 4 == 0 && lddt >= d)) &&
                  ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(T) |
                    reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(dT) |
                    reinterpret_cast<uintptr_t>(state)) & 15) == 0;
  if (!ok) {
    set_error("arx_mce_scorer_fwd: shape not supported (d in {64, 128}, S %% 128 == 0, 128 <= S <= 2048, 16-byte rows)");
    return ARX_EUNSUPPORTED;
  }
  if (state_bytes < L.total) {
    set_error("arx_mce_scorer_fwd: state too small (%zu < %zu)", state_bytes, L.total);
    return ARX_EWORKSPACE;
  }
  ARX_CHECK_ARG(mask_rows <= 0 || mask_rows == B || (mask_rows % 128 == 0 && B % mask_rows == 0),
                "arx_mce_scorer_fwd: mask_rows must be B or a multiple of 128 that divides B");
  if (int rc = mc_raise_lds()) return rc;
  hipStream_t s = as_stream(stream);
  char* st = reinterpret_cast<char*>(state);
  float* t = reinterpret_cast<float*>(st + L.t);
  uint16_t* Pp = reinterpret_cast<uint16_t*>(st + L.Pp);
  uint16_t* PT = reinterpret_cast<uint16_t*>(st + L.PT);
  uint32_t* maskR = reinterpret_cast<uint32_t*>(st + L.maskR);
  float* rs_part = reinterpret_cast<float*>(st + L.rs);
  float* O_part = reinterpret_cast<float*>(st + L.O);
  const PosMask pm = make_pos_mask(user_ids, pos_ptr, pos_items, item2slot);
  const int64_t mrows = mask_rows > 0 ? mask_rows : B;
  if (phases & 1) {
    const int64_t grid = S / 32 + ceil_div(B, 8) + (seq_w ? ceil_div(seq_rows, 32) : 0);
    const ScTScore ts{U, ldu, T, ldt, tbias, tb_stride > 0 ? tb_stride : 1, t, tscore_out};
    k_sc_prep<true><<<(int)grid, 256, 0, s>>>(P, ldp, S, d, pbias, Pp, PT, L.ldpt, reinterpret_cast<float*>(st + L.pbad),
                                              pm, mrows, B, reinterpret_cast<int32_t*>(st + L.hits),
                                              reinterpret_cast<int32_t*>(st + L.nhit), seq_w, seq_rows, row_w, ts,
                                              maskR, L.ldr);
    ARX_CHECK_LAUNCH();
  }
  if (phases & 2) {
    McFlow a{};
    a.nstat = B; a.span = L.CW; a.nstream = S;
    a.statF = U; a.ldstat = ldu;
    a.XK = Pp; a.xk_plane = S * d;
    a.XT = PT; a.ldx = L.ldpt;
    a.cstat = t; a.sgn_stat = -1.f;
    a.cstream = pbias; a.cw = nullptr;
    a.mask = maskR; a.ldmask = L.ldr;
    a.mask_mod = mrows;
    a.zeros = reinterpret_cast<const float*>(st + L.zeros);
    a.O = O_part; a.o_rows = L.Bp;
    a.rsum = rs_part; a.rs_rows = B;
    if (d == 64) k_mc_flow<false, 64><<<(int)(ceil_div(B, 128) * L.nsplit), 256, kMcSlots * mc_stage(64), s>>>(a);
    else k_mc_flow<false, 128><<<(int)(ceil_div(B, 128) * L.nsplit), 256, kMcSlots * mc_stage(128), s>>>(a);
    ARX_CHECK_LAUNCH();
  }
  if (phases & 4) {
    McRows a{rs_part, O_part, L.nsplit, L.Bp, t, U, ldu, T, ldt, gscale, row_w, batch_loss,
             reinterpret_cast<float*>(st + L.coef), reinterpret_cast<float*>(st + L.tneg), dtscore,
             dtscore_stride > 0 ? dtscore_stride : 1, dU, lddu, dT, lddt, reinterpret_cast<uint16_t*>(st + L.Up),
             reinterpret_cast<uint16_t*>(st + L.UgT), L.ldug, reinterpret_cast<const float*>(st + L.pbad),
             (int)(S / 32)};
    if (d == 64) k_mc_rows<16><<<(int)(L.Bp / 32), 256, 0, s>>>(a, B);
    else k_mc_rows<32><<<(int)(L.Bp / 32), 256, 0, s>>>(a, B);
    ARX_CHECK_LAUNCH();
  }
  return ARX_OK;
}

size_t arx_mce_scorer_bwd_di_workspace_bytes(int64_t B, int64_t S, int d, int64_t step_rows) {
  McLayout L;
  if (!mc_layout(B, S, d, &L)) return 0;
  const int64_t nsl = ceil_div(L.Bp, mc_di_slice(L, S, step_rows));
  return (size_t)nsl * S * d * 4 + (size_t)nsl * S * 4 + 512;
}

/* dI[s, :] = beta dI[s, :] + sum_r w_rs U[r, :], db[s] = sum_r w_rs with w_rs = coef_r m_rs exp(x_rs - t_r), recomputed
 * tile by tile; arguments as arx_mw_scorer_bwd_di_loss (step_rows > 0: B % step_rows == 0, step_rows % 128 == 0). */
int arx_mce_scorer_bwd_di_loss(int64_t B, int64_t S, int d, const void* state, const float* pbias, int64_t mask_rows,
                               int64_t step_rows, float beta, float* dI, int64_t lddi, float* db, float* dI_steps, float* db_steps,
                               const float* batch_loss, float gscale, const float* row_w, float* loss_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
  McLayout L;
  ARX_CHECK_ARG(!loss_out || (batch_loss && (reinterpret_cast<uintptr_t>(batch_loss) & 15) == 0 &&
                              (reinterpret_cast<uintptr_t>(row_w) & 15) == 0),
                "arx_mce_scorer_bwd_di_loss: loss_out needs batch_loss (and row_w) 16-byte aligned");
  ARX_CHECK_ARG(state && dI && mc_layout(B, S, d, &L), "arx_mce_scorer_bwd_di: bad argument / shape");
  ARX_CHECK_ARG(lddi % 4 == 0 && lddi >= d && (reinterpret_cast<uintptr_t>(dI) & 15) == 0,
                "arx_mce_scorer_bwd_di: dI rows must be 16-byte aligned");
  ARX_CHECK_ARG(step_rows == 0 || (step_rows % 128 == 0 && B % step_rows == 0),
                "arx_mce_scorer_bwd_di: step_rows must divide B and be a multiple of 128");
  const int64_t mrows = mask_rows > 0 ? mask_rows : B;
  if (int rc = mc_raise_lds()) return rc;
  const int64_t ks = mc_di_slice(L, S, step_rows);
  ARX_CHECK_ARG(mrows == B || (mrows % 128 == 0 && mrows % ks == 0),
                "arx_mce_scorer_bwd_di: mask_rows as given to the forward; a slice of rows must not wrap the user rows");
  const int64_t nsl = ceil_div(L.Bp, ks);
  const size_t part_bytes = (size_t)nsl * S * d * 4;
  const size_t need = (dI_steps ? 0 : part_bytes) + (size_t)nsl * S * 4 + 256;
  if (!workspace || workspace_bytes < need) {
    set_error("arx_mce_scorer_bwd_di: workspace too small (%zu < %zu)", workspace_bytes, need);
    return ARX_EWORKSPACE;
  }
  char* wsb = reinterpret_cast<char*>(workspace);
  float* part = dI_steps ? dI_steps : reinterpret_cast<float*>(wsb);
  float* dbpart = reinterpret_cast<float*>(wsb + (dI_steps ? 0 : al256(part_bytes)));
  hipStream_t s = as_stream(stream);
  const char* st = reinterpret_cast<const char*>(state);
  McFlow a{};
  a.nstat = S; a.span = ks; a.nstream = L.Bp;
  a.statP = reinterpret_cast<const uint16_t*>(st + L.Pp); a.statP_plane = S * d;
  a.XK = reinterpret_cast<const uint16_t*>(st + L.Up); a.xk_plane = L.Bp * d;
  a.XT = reinterpret_cast<const uint16_t*>(st + L.UgT); a.ldx = L.ldug;
  a.cstat = pbias; a.sgn_stat = 1.f;
  a.cstream = reinterpret_cast<const float*>(st + L.tneg);
  a.cw = reinterpret_cast<const float*>(st + L.coef);
  a.mask = reinterpret_cast<const uint32_t*>(st + L.maskR); a.ldmask = L.ldr;
  a.mask_mod = mrows;
  a.zeros = reinterpret_cast<const float*>(st + L.zeros);
  a.O = part; a.o_rows = S;
  a.rsum = dbpart; a.rs_rows = S;
  if (d == 64) k_mc_flow<true, 64><<<(int)((S / 128) * nsl), 256, kMcSlots * mc_stage(64), s>>>(a);
  else k_mc_flow<true, 128><<<(int)((S / 128) * nsl), 256, kMcSlots * mc_stage(128), s>>>(a);
  ARX_CHECK_LAUNCH();
  const int dbblocks = db ? (int)ceil_div(S, 64) : 0;
  const bool steps = db_steps && step_rows > 0;
  const int64_t Lsteps = steps ? B / step_rows : 0;
  const int stepblocks = steps ? (int)ceil_div(Lsteps * S, 256) : 0;
  k_sc_tn_reduce<<<(int)ceil_div(S * (d / 4), 64) + dbblocks + stepblocks + (loss_out ? 1 : 0), 256, 0, s>>>(
      part, (int)nsl, S, d, beta, dI, lddi, dbpart, nsl, db, dbblocks, batch_loss, row_w, B, gscale, loss_out,
      stepblocks, 1, Lsteps, db_steps);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
