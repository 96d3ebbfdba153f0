// loss.hip -- K6: batch losses (forward + backward fused) and the positive mask.
//
//  mw / warp : embed_attribute.py:605-618, 641-649  log(1 + sum relu(mask*(x - t + 1)))
//  ce        : embed_attribute.py:529-531           sparse softmax cross entropy
//  mask      : embed_attribute.py:651-672, 721-745  scatter_update of a bool variable
//
// One 256-thread workgroup per batch row; a row of S=1024 logits is one float4
// per thread.  The row is read twice (sum, then gradient); the second read hits
// L2, and dlogits may overwrite logits in place.
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "posmask.h"

namespace arx {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum over 256 threads, fixed order (deterministic)
__device__ __forceinline__ float block_sum(float v, float* sh /*[4]*/) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// MW and WARP share the body; WARP takes t from the target column and routes
// dt back into it.
// Positive mask built on the fly (embed_attribute.py:721-745 fused into the loss):
// one bit per logit column in LDS, set for every positive item of the row's user
// that has a slot in this pool.  Replaces the persistent [mb, W] bool variable and
// its two scatter_update launches per step.
// Optional fusion of the target score into the margin-loss kernel (embed_attribute.py:208-220 +
// :604-649): the wave that owns row r also forms t_r = U_r . T_r + tb_r and, once dt_r is known,
// the two rank-one gradients dT_r = dt_r U_r and dU_r = dt_r T_r (written, not accumulated: the
// scorer's dU GEMM adds onto it).  Two ~5 us launches less per step.
struct DotFuse {
  const float* U;      // [B, d] latent rows (NULL: not fused, tscore is an input)
  int64_t ldu;
  const float* T;      // [B, d] target-item embedding rows
  int64_t ldt;
  const float* tb;     // target bias, element r at tb[r * tb_stride] (nullable)
  int64_t tb_stride;
  int d;
  float* tscore_out;   // [B] (nullable)
  int64_t dts_stride;  // dtscore element r at dtscore[r * dts_stride]
  float* dU;           // [B, d] (nullable)
  int64_t lddu;
  float* dT;           // [B, d] (nullable)
  int64_t lddt;
};

__device__ __forceinline__ void build_pos_bits(uint32_t* bits, int64_t W, const PosMask& pm,
                                               int64_t mrow) {
  const int nwords = (int)((W + 31) >> 5);
  for (int i = threadIdx.x; i < nwords; i += 256) bits[i] = 0u;
  __syncthreads();
  const int u = pm.user_ids[mrow];
  const int beg = pm.pos_ptr[u], end = pm.pos_ptr[u + 1];
  for (int p = beg + threadIdx.x; p < end; p += 256) {
    const int j = pos_slot(pm, pm.pos_items[p]);
    if (j >= 0 && j < W) atomicOr(&bits[j >> 5], 1u << (j & 31));
  }
  __syncthreads();
}

// SOFT (build-defined 'mce', sampled softmax; !WARP only): the hinge sum becomes
//   loss = log(1 + sum_s keep_s * e_s),  dx_s = g * keep_s * e_s / (1 + sum),  e_s = exp(min(x_s - t, kMceSat)),
// evaluated around M = max(t, max kept x) for range safety.  The saturation at kMceSat (common.h) is part of the
// build-defined loss: a pair whose logit leads the target score by more than 64 has a softmax weight of 1 - 1e-28
// already; capping its exponent keeps the FUSED family (scorer.hip: k_mc_flow anchors at t, no running maximum)
// finite, and both paths, and the oracle, share ONE definition.
template <bool WARP, bool POS, bool SOFT = false>
__global__ __launch_bounds__(256) void k_loss_margin(
    const float* __restrict__ logits, int64_t ldl, const float* __restrict__ tscore,
    const int32_t* __restrict__ target, const uint8_t* __restrict__ mask, int64_t ldm,
    int64_t mask_rows, float gscale, const float* __restrict__ row_w, int64_t W,
    float* __restrict__ batch_loss,
    float* dlogits, int64_t lddl, float* __restrict__ dtscore, PosMask pm) {
  __shared__ float sh[4];
  extern __shared__ uint32_t bits[];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  const uint8_t* m = (!POS && mask) ? mask + (r % mask_rows) * ldm : nullptr;
  if (POS) build_pos_bits(bits, W, pm, r % mask_rows);
  const int tcol = WARP ? target[r] : -1;
  // a target without a logit index (the reference's item_ind2logit_ind raises KeyError): the row's
  // loss is NaN -- loud -- and its gradient zero; nothing is read or written out of bounds
  const bool bad = WARP && (tcol < 0 || tcol >= W);
  const float t = WARP ? (bad ? 0.f : x[tcol]) : tscore[r];
  if constexpr (SOFT) {
    float mx = t;
    for (int64_t c = threadIdx.x; c < W; c += 256) {
      const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
      if (keep) mx = fmaxf(mx, mce_sat(x[c], t));
    }
    mx = block_max(mx, sh);
    float se = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += 256) {
      const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
      se += keep ? expf(mce_sat(x[c], t) - mx) : 0.f;
    }
    se = block_sum(se, sh);
    const float z = expf(t - mx) + se;
    if (threadIdx.x == 0 && batch_loss) batch_loss[r] = mx - t + logf(z);
    if (!dlogits) return;
    const float gz = gscale * (row_w ? row_w[r] : 1.f) / z;
    float* dxs = dlogits + r * lddl;
    for (int64_t c = threadIdx.x; c < W; c += 256) {
      const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
      dxs[c] = keep ? gz * expf(mce_sat(x[c], t) - mx) : 0.f;
    }
    if (threadIdx.x == 0 && dtscore) dtscore[r] = -gz * se;
    return;
  }
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < W; c += 256) {
    const float v = x[c] - t + 1.f;
    const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
    s += (keep && v > 0.f) ? v : 0.f;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0 && batch_loss) batch_loss[r] = bad ? NAN : logf(1.f + s);
  if (!dlogits) return;
  const float g = bad ? 0.f : gscale * (row_w ? row_w[r] : 1.f) / (1.f + s);
  float* dx = dlogits + r * lddl;
  float cnt = 0.f;
  for (int64_t c = threadIdx.x; c < W; c += 256) {
    const float v = x[c] - t + 1.f;
    const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
    const bool act = keep && v > 0.f;
    cnt += act ? 1.f : 0.f;
    dx[c] = act ? g : 0.f;
  }
  cnt = block_sum(cnt, sh);  // (contains the barrier that orders the writes above)
  if (threadIdx.x == 0) {
    const float dt = -g * cnt;
    if (WARP) { if (!bad) dx[tcol] += dt; }
    else if (dtscore) dtscore[r] = dt;
  }
}

// Sampled-pool rows (W <= 2048, e.g. S = 1024 negatives): ONE WAVE per row, the row lives
// in registers (NV float4 per lane), read once; reductions are wave shuffles (no workgroup
// barrier).  The workgroup-per-row kernel above is bound by the positives chain
// (user -> pos_ptr -> pos_items -> item2slot: four dependent loads, ~5 us per row with
// only 8 rows resident per CU: 48 us at B=16384 for 134 MB = 2.8 TB/s); with a wave per row
// 32 rows per CU are in flight and the chain overlaps the row load.
// Same arithmetic, same summation order per lane is NOT kept (lane-strided float4 instead
// of thread-strided scalars): results agree to fp32 rounding, deterministic run to run.
template <bool WARP, bool POS, int NV, bool SOFT = false>
__global__ __launch_bounds__(256) void k_loss_margin_wave(
    const float* __restrict__ logits, int64_t ldl, const float* __restrict__ tscore,
    const int32_t* __restrict__ target, const uint8_t* __restrict__ mask, int64_t ldm,
    int64_t mask_rows, float gscale, const float* __restrict__ row_w, int64_t B, int64_t W,
    float* __restrict__ batch_loss, float* dlogits, int64_t lddl, float* __restrict__ dtscore,
    PosMask pm, DotFuse df) {
  __shared__ uint32_t bits_all[4][NV * 8];           // NV*256 columns -> NV*8 words per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wv;
  if (r >= B) return;                                 // whole wave (no workgroup barrier below)
  uint32_t* bits = bits_all[wv];
  const float* x = logits + r * ldl;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = (int64_t)i * 256 + lane * 4;
    if (c + 3 < W) {       // streaming load: the logits row is read exactly once
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + c));
      v[i] = make_float4(q.x, q.y, q.z, q.w);
    } else {
      v[i] = make_float4(c < W ? x[c] : 0.f, c + 1 < W ? x[c + 1] : 0.f, c + 2 < W ? x[c + 2] : 0.f, 0.f);
    }
  }
  // keep bits of this lane's columns: bit (4*i + e) <-> column i*256 + lane*4 + e
  uint32_t keep = 0u;
  const int64_t mrow = r % mask_rows;
  if (POS) {
    if (lane < NV * 8) bits[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int u = pm.user_ids[mrow];
    const int beg = pm.pos_ptr[u], end = pm.pos_ptr[u + 1];
    for (int p = beg + lane; p < end; p += 64) {
      const int j = pos_slot(pm, pm.pos_items[p]);
      if (j >= 0 && j < W) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;                // 4 columns inside one 32-bit word
      const uint32_t nib = (bits[c >> 5] >> (c & 31)) & 0xFu;
      keep |= ((~nib) & 0xFu) << (4 * i);
    }
  } else if (mask) {
    const uint8_t* m = mask + mrow * ldm;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = (int64_t)i * 256 + lane * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < W && m[c + e] != 0) keep |= 1u << (4 * i + e);
    }
  } else {
    keep = 0xFFFFFFFFu;
  }
  const int tcol = WARP ? target[r] : -1;
  float t;
  float4 ur = make_float4(0.f, 0.f, 0.f, 0.f), tr = ur;     // fused target score: d <= 256
  if (!WARP && df.U) {
    const int c4 = lane * 4;
    if (c4 < df.d) {
      ur = *reinterpret_cast<const float4*>(df.U + r * df.ldu + c4);
      tr = *reinterpret_cast<const float4*>(df.T + r * df.ldt + c4);
    }
    t = wsum(ur.x * tr.x + ur.y * tr.y + ur.z * tr.z + ur.w * tr.w) + (df.tb ? df.tb[r * df.tb_stride] : 0.f);
    if (lane == 0 && df.tscore_out) df.tscore_out[r] = t;
  } else {
    t = WARP ? ((tcol < 0 || tcol >= W) ? 0.f : x[tcol]) : tscore[r];
  }
  const bool bad = WARP && (tcol < 0 || tcol >= W);     // see k_loss_margin
  float s = 0.f;
  float g, cnt = 0.f;
  float4 d[NV];
  float dt_soft = 0.f;
  if constexpr (SOFT) {                                  // sampled softmax ('mce'), see k_loss_margin
    float mx = t;
#pragma unroll
    for (int i = 0; i < NV; ++i)             // saturation of the build-defined loss (k_loss_margin, common.h)
      v[i] = make_float4(mce_sat(v[i].x, t), mce_sat(v[i].y, t), mce_sat(v[i].z, t), mce_sat(v[i].w, t));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = (int64_t)i * 256 + lane * 4;
      if (((keep >> (4 * i)) & 1u) && c < W) mx = fmaxf(mx, v[i].x);
      if (((keep >> (4 * i + 1)) & 1u) && c + 1 < W) mx = fmaxf(mx, v[i].y);
      if (((keep >> (4 * i + 2)) & 1u) && c + 2 < W) mx = fmaxf(mx, v[i].z);
      if (((keep >> (4 * i + 3)) & 1u) && c + 3 < W) mx = fmaxf(mx, v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = (int64_t)i * 256 + lane * 4;
      d[i].x = (((keep >> (4 * i)) & 1u) && c < W) ? __expf(v[i].x - mx) : 0.f;
      d[i].y = (((keep >> (4 * i + 1)) & 1u) && c + 1 < W) ? __expf(v[i].y - mx) : 0.f;
      d[i].z = (((keep >> (4 * i + 2)) & 1u) && c + 2 < W) ? __expf(v[i].z - mx) : 0.f;
      d[i].w = (((keep >> (4 * i + 3)) & 1u) && c + 3 < W) ? __expf(v[i].w - mx) : 0.f;
      s += (d[i].x + d[i].y) + (d[i].z + d[i].w);
    }
    s = wsum(s);
    const float z = __expf(t - mx) + s;
    if (lane == 0 && batch_loss) batch_loss[r] = mx - t + logf(z);
    if (!dlogits) return;
    g = gscale * (row_w ? row_w[r] : 1.f) / z;
#pragma unroll
    for (int i = 0; i < NV; ++i) d[i] = make_float4(d[i].x * g, d[i].y * g, d[i].z * g, d[i].w * g);
    dt_soft = -g * s;
  } else {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = (int64_t)i * 256 + lane * 4;
    const float e0 = v[i].x - t + 1.f, e1 = v[i].y - t + 1.f, e2 = v[i].z - t + 1.f,
                e3 = v[i].w - t + 1.f;
    s += (((keep >> (4 * i)) & 1u) && c < W && e0 > 0.f) ? e0 : 0.f;
    s += (((keep >> (4 * i + 1)) & 1u) && c + 1 < W && e1 > 0.f) ? e1 : 0.f;
    s += (((keep >> (4 * i + 2)) & 1u) && c + 2 < W && e2 > 0.f) ? e2 : 0.f;
    s += (((keep >> (4 * i + 3)) & 1u) && c + 3 < W && e3 > 0.f) ? e3 : 0.f;
  }
  s = wsum(s);
  if (lane == 0 && batch_loss) batch_loss[r] = bad ? NAN : logf(1.f + s);
  if (!dlogits) return;
  g = bad ? 0.f : gscale * (row_w ? row_w[r] : 1.f) / (1.f + s);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = (int64_t)i * 256 + lane * 4;
    const bool a0 = ((keep >> (4 * i)) & 1u) && c < W && (v[i].x - t + 1.f > 0.f);
    const bool a1 = ((keep >> (4 * i + 1)) & 1u) && c + 1 < W && (v[i].y - t + 1.f > 0.f);
    const bool a2 = ((keep >> (4 * i + 2)) & 1u) && c + 2 < W && (v[i].z - t + 1.f > 0.f);
    const bool a3 = ((keep >> (4 * i + 3)) & 1u) && c + 3 < W && (v[i].w - t + 1.f > 0.f);
    cnt += (a0 ? 1.f : 0.f) + (a1 ? 1.f : 0.f) + (a2 ? 1.f : 0.f) + (a3 ? 1.f : 0.f);
    d[i] = make_float4(a0 ? g : 0.f, a1 ? g : 0.f, a2 ? g : 0.f, a3 ? g : 0.f);
  }
  }
  float* dx = dlogits + r * lddl;
  cnt = SOFT ? 0.f : wsum(cnt);
  const float dt = SOFT ? dt_soft : -g * cnt;
  if (WARP) {     // the target column's owner folds dt in before the row is stored
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;
      if (tcol >= c && tcol < c + 4) {
        const int e = tcol - c;
        if (e == 0) d[i].x += dt; else if (e == 1) d[i].y += dt; else if (e == 2) d[i].z += dt; else d[i].w += dt;
      }
    }
  } else {
    if (lane == 0 && dtscore) dtscore[df.U ? r * df.dts_stride : r] = dt;
    if (df.U && lane * 4 < df.d) {
      if (df.dT)
        *reinterpret_cast<float4*>(df.dT + r * df.lddt + lane * 4) =
            make_float4(dt * ur.x, dt * ur.y, dt * ur.z, dt * ur.w);
      if (df.dU)
        *reinterpret_cast<float4*>(df.dU + r * df.lddu + lane * 4) =
            make_float4(dt * tr.x, dt * tr.y, dt * tr.z, dt * tr.w);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = (int64_t)i * 256 + lane * 4;
    if (c + 3 < W) {
      // streaming store: dlogits is consumed by the two backward GEMMs, never re-read from L2
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f o;
      o.x = d[i].x; o.y = d[i].y; o.z = d[i].z; o.w = d[i].w;
      __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dx + c));
    } else {
      if (c < W) dx[c] = d[i].x;
      if (c + 1 < W) dx[c + 1] = d[i].y;
      if (c + 2 < W) dx[c + 2] = d[i].z;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Streaming full-vocabulary evaluation losses (hmf_model.py:130,144: loss_eval of a sampled-loss
// model = 'warp' / 'ce' over ALL V logits).  The [B, V] logits are never materialised: the scorer
// GEMM runs over chunks of the pool and these kernels fold every chunk into two running values per
// row -- ce: (max, sum exp) of an online log-sum-exp; warp: sum of relu(x - t + 1) -- with the
// target logit t_r formed directly from the target's pool row.  The positive mask of 'warp' is
// applied afterwards by taking the masked columns' terms out again (their logits recomputed).
__global__ __launch_bounds__(256) void k_eval_chunk(const float* __restrict__ x, int64_t ld, int64_t n,
                                                    const float* __restrict__ t, int mode, int first,
                                                    float* __restrict__ acc0, float* __restrict__ acc1,
                                                    int64_t B) {
  const int lane = threadIdx.x & 63;
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (r >= B) return;
  const float* row = x + r * ld;
  if (mode == 0) {                       // ce: online log-sum-exp
    float m = -INFINITY;
    for (int64_t c = lane; c < n; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float l = 0.f;
    for (int64_t c = lane; c < n; c += 64) l += expf(row[c] - m);
    l = wsum(l);
    if (lane == 0) {
      if (first) { acc0[r] = m; acc1[r] = l; }
      else {
        const float m0 = acc0[r], l0 = acc1[r];
        const float mm = fmaxf(m0, m);
        acc0[r] = mm;
        acc1[r] = l0 * expf(m0 - mm) + l * expf(m - mm);
      }
    }
  } else {                               // warp: hinge sum against the target logit
    const float tt = t[r];
    float s = 0.f;
    for (int64_t c = lane; c < n; c += 64) {
      const float v = row[c] - tt + 1.f;
      s += v > 0.f ? v : 0.f;
    }
    s = wsum(s);
    if (lane == 0) acc0[r] = first ? s : acc0[r] + s;
  }
}

// take the terms of the masked columns (the user's positives that have a logit) out of the hinge
// sum again: one wave per row, every distinct column once
__global__ __launch_bounds__(256) void k_eval_unmask(const float* __restrict__ U, int64_t ldu,
                                                     const float* __restrict__ P, int64_t ldp,
                                                     const float* __restrict__ pb, int d,
                                                     const float* __restrict__ t, PosMask pm, int64_t mask_rows,
                                                     int64_t V, float* __restrict__ s_acc, int64_t B) {
  const int lane = threadIdx.x & 63;
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (r >= B) return;
  float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool colok = lane * 4 < d;
  if (colok) u = *reinterpret_cast<const float4*>(U + r * ldu + lane * 4);
  const float tt = t[r];
  const int usr = pm.user_ids[r % mask_rows];
  const int beg = pm.pos_ptr[usr], end = pm.pos_ptr[usr + 1];
  float sub = 0.f;
  for (int p0 = beg; p0 < end; p0 += 64) {
    const int p = p0 + lane;
    int j = -1;
    if (p < end) {
      j = pos_slot(pm, pm.pos_items[p]);
      if (j < 0 || j >= V) j = -1;
    }
    // a column named twice (here or in an earlier batch of 64) counts once: the mask is a set
    if (j >= 0) {
      for (int q = beg; q < p; ++q) {
        if (pos_slot(pm, pm.pos_items[q]) == j) { j = -1; break; }
      }
    }
    unsigned long long hm = __ballot(j >= 0);
    while (hm) {
      const int src = __builtin_ctzll(hm);
      hm &= hm - 1;
      const int jj = __shfl(j, src, 64);
      float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (colok) pr = *reinterpret_cast<const float4*>(P + (int64_t)jj * ldp + lane * 4);
      const float x = wsum(u.x * pr.x + u.y * pr.y + u.z * pr.z + u.w * pr.w) + (pb ? pb[jj] : 0.f);
      const float v = x - tt + 1.f;
      sub += v > 0.f ? v : 0.f;
    }
  }
  if (lane == 0) s_acc[r] = fmaxf(s_acc[r] - sub, 0.f);
}

__global__ void k_eval_finish(int mode, const float* __restrict__ acc0, const float* __restrict__ acc1,
                              const float* __restrict__ t, int64_t B, float* __restrict__ out) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= B) return;
  out[r] = mode == 0 ? acc0[r] + logf(acc1[r]) - t[r] : logf(1.f + acc0[r]);
}

// launch helper: true if the wave-per-row kernel took the call
template <bool WARP, bool POS, bool SOFT = false>
static bool launch_margin_wave(const float* logits, int64_t ldl, const float* tscore,
                               const int32_t* target, const uint8_t* mask, int64_t ldm,
                               int64_t mask_rows, float gscale, const float* row_w, int64_t B,
                               int64_t W, float* batch_loss, float* dlogits, int64_t lddl,
                               float* dtscore, PosMask pm, hipStream_t s, DotFuse df = DotFuse{}) {
  if (W > 2048 || (ldl % 4) || (dlogits && (lddl % 4)) ||
      (reinterpret_cast<uintptr_t>(logits) & 15) || (reinterpret_cast<uintptr_t>(dlogits) & 15))
    return false;
  const int grid = (int)ceil_div(B, 4);
  if (W <= 1024)
    k_loss_margin_wave<WARP, POS, 4, SOFT><<<grid, 256, 0, s>>>(logits, ldl, tscore, target, mask, ldm,
                                                          mask_rows, gscale, row_w, B, W, batch_loss,
                                                          dlogits, lddl, dtscore, pm, df);
  else
    k_loss_margin_wave<WARP, POS, 8, SOFT><<<grid, 256, 0, s>>>(logits, ldl, tscore, target, mask, ldm,
                                                          mask_rows, gscale, row_w, B, W, batch_loss,
                                                          dlogits, lddl, dtscore, pm, df);
  return true;
}


// rs / rs-sig / rs-sig2 / bbpr (embed_attribute.py:551-603): pairwise margin or sigmoid errors
// over the full vocabulary, masked, summed, then a scalar transform (loss_func).
//   kind 0 rs      err = relu(x - t + 1)
//   kind 1 rs-sig  err = 2*sigmoid(masked relu(x - t + 1)) - 1
//   kind 2 rs-sig2 err = sigmoid(x - t)
//   kind 3 bbpr    err = sigmoid(x - t), loss = sum (no transform)
//   loss_func 0 log(1+s) | 1 1 - p^-s | 2 s^p | 3 (1+s)^p | 4 s | 5 s^2
// One workgroup per row, two passes (the second hits L2); the target column receives
// -(sum of the row's gradient) like the warp loss.
__device__ __forceinline__ void rs_err(int kind, float x, float t, bool keep, float& em, float& dem) {
  float err, derr;
  if (kind <= 1) {
    const float pre = x - t + 1.f;
    err = fmaxf(pre, 0.f);
    derr = pre > 0.f ? 1.f : 0.f;
  } else {
    err = 1.f / (1.f + __expf(-(x - t)));
    derr = err * (1.f - err);
  }
  em = keep ? err : 0.f;
  dem = keep ? derr : 0.f;
  if (kind == 1) {
    const float sg = 1.f / (1.f + __expf(-em));
    dem = dem * 2.f * sg * (1.f - sg);
    em = 2.f * sg - 1.f;
  }
}

template <bool POS>
__global__ __launch_bounds__(256) void k_loss_rs(
    const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ target,
    const uint8_t* __restrict__ mask, int64_t ldm, int64_t mask_rows, int kind, int loss_func,
    float exp_p, float gscale, const float* __restrict__ row_w, int64_t W,
    float* __restrict__ batch_loss, float* dlogits, int64_t lddl, PosMask pm) {
  __shared__ float sh[4];
  extern __shared__ uint32_t bits[];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  const uint8_t* m = (!POS && mask) ? mask + (r % mask_rows) * ldm : nullptr;
  if (POS) build_pos_bits(bits, W, pm, r % mask_rows);
  const int tcol = target[r];
  const bool bad = tcol < 0 || tcol >= W;               // see k_loss_margin
  const float t = bad ? 0.f : x[tcol];
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < W; c += 256) {
    const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
    float em, dem;
    rs_err(kind, x[c], t, keep, em, dem);
    s += em;
  }
  s = block_sum(s, sh);
  float l, dl;
  if (kind == 3 || loss_func == 4) { l = s; dl = 1.f; }
  else if (loss_func == 0) { l = logf(1.f + s); dl = 1.f / (1.f + s); }
  else if (loss_func == 1) { const float q = powf(exp_p, -s); l = 1.f - q; dl = logf(exp_p) * q; }
  else if (loss_func == 2) { l = powf(s, exp_p); dl = exp_p * powf(s, exp_p - 1.f); }
  else if (loss_func == 3) { l = powf(1.f + s, exp_p); dl = exp_p * powf(1.f + s, exp_p - 1.f); }
  else { l = s * s; dl = 2.f * s; }
  if (threadIdx.x == 0 && batch_loss) batch_loss[r] = bad ? NAN : l;
  if (!dlogits) return;
  const float g = bad ? 0.f : gscale * (row_w ? row_w[r] : 1.f) * dl;
  float* dx = dlogits + r * lddl;
  float tot = 0.f;
  for (int64_t c = threadIdx.x; c < W; c += 256) {
    const bool keep = POS ? !((bits[c >> 5] >> (c & 31)) & 1u) : (m ? (m[c] != 0) : true);
    float em, dem;
    rs_err(kind, x[c], t, keep, em, dem);
    const float d = dem * g;
    tot += d;
    dx[c] = d;
  }
  tot = block_sum(tot, sh);   // (contains the barrier that orders the writes above)
  if (threadIdx.x == 0 && !bad) dx[tcol] += -tot;
}

__global__ __launch_bounds__(256) void k_loss_ce(
    const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ target,
    float gscale, const float* __restrict__ row_w, int64_t V, float* __restrict__ batch_loss,
    float* dlogits, int64_t lddl) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  const int tcol = target[r];
  const bool bad = tcol < 0 || tcol >= V;               // see k_loss_margin
  const float xt = bad ? NAN : x[tcol];
  float mx = -INFINITY;
  for (int64_t c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, x[c]);
  mx = block_max(mx, sh);
  float se = 0.f;
  for (int64_t c = threadIdx.x; c < V; c += 256) se += expf(x[c] - mx);
  se = block_sum(se, sh);
  if (threadIdx.x == 0 && batch_loss) batch_loss[r] = logf(se) + mx - xt;
  if (!dlogits) return;
  const float g = bad ? 0.f : gscale * (row_w ? row_w[r] : 1.f);
  const float inv = g / se;
  float* dx = dlogits + r * lddl;
  for (int64_t c = threadIdx.x; c < V; c += 256) {
    const float p = expf(x[c] - mx) * inv;
    dx[c] = (c == tcol) ? p - g : p;
  }
}

// k_loss_ce for V <= 4096 (the C1 shape: V = 3100): the row is read ONCE into 16 registers per thread instead of three
// dependent passes over it (9.5 us measured at 64 x 3100 -- three L2 round trips and two block reductions per row).
// Same per-thread accumulation order (c = tid, tid + 256, ...) and the same block reductions: bit-identical results.
__global__ __launch_bounds__(256) void k_loss_ce_regs(
    const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ target,
    float gscale, const float* __restrict__ row_w, int64_t V, float* __restrict__ batch_loss,
    float* dlogits, int64_t lddl) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  const int tcol = target[r];
  const bool bad = tcol < 0 || tcol >= V;               // see k_loss_margin
  const float xt = bad ? NAN : x[tcol];
  const float rw = row_w ? row_w[r] : 1.f;
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int64_t c = threadIdx.x + u * 256;
    v[u] = c < V ? x[c] : -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 16; ++u) mx = fmaxf(mx, v[u]);
  mx = block_max(mx, sh);
  float se = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int64_t c = threadIdx.x + u * 256;
    v[u] = expf(v[u] - mx);
    if (c < V) se += v[u];
  }
  se = block_sum(se, sh);
  if (threadIdx.x == 0 && batch_loss) batch_loss[r] = logf(se) + mx - xt;
  if (!dlogits) return;
  const float g = bad ? 0.f : gscale * rw;
  const float inv = g / se;
  float* dx = dlogits + r * lddl;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int64_t c = threadIdx.x + u * 256;
    if (c < V) {
      const float p = v[u] * inv;
      dx[c] = (c == tcol) ? p - g : p;
    }
  }
}

// per-row logsumexp (softmax normaliser of the recommend path, seqModel.py:514-517)
__global__ __launch_bounds__(256) void k_row_lse(const float* __restrict__ logits, int64_t ldl,
                                                 int64_t V, float* __restrict__ out) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  float mx = -INFINITY;
  for (int64_t c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, x[c]);
  mx = block_max(mx, sh);
  float se = 0.f;
  for (int64_t c = threadIdx.x; c < V; c += 256) se += expf(x[c] - mx);
  se = block_sum(se, sh);
  if (threadIdx.x == 0) out[r] = logf(se) + mx;
}

__global__ __launch_bounds__(256) void k_warp_eval(
    const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ target,
    const uint8_t* __restrict__ mask, int64_t ldm, int64_t mask_rows, int64_t V,
    float* __restrict__ margin_rank, int32_t* __restrict__ true_rank) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ldl;
  const uint8_t* m = mask ? mask + (r % mask_rows) * ldm : nullptr;
  const float t = x[target[r]];
  float s = 0.f, cnt = 0.f;
  for (int64_t c = threadIdx.x; c < V; c += 256) {
    const bool keep = m ? (m[c] != 0) : true;
    const float v = x[c] - t;
    if (keep && v + 1.f > 0.f) s += v + 1.f;
    if (keep && v > 0.f) cnt += 1.f;
  }
  s = block_sum(s, sh);
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) {
    margin_rank[r] = s;
    true_rank[r] = (int32_t)cnt;
  }
}

// one wave per batch row walks the row's positives
__global__ __launch_bounds__(256) void k_pos_mask_scatter(
    const int32_t* __restrict__ user_ids, int64_t B, const int32_t* __restrict__ pos_ptr,
    const int32_t* __restrict__ pos_items, const int32_t* __restrict__ item2slot,
    uint8_t* __restrict__ mask, int64_t ldm, uint8_t value) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < B; r += nwave) {
    const int u = user_ids[r];
    const int beg = pos_ptr[u], end = pos_ptr[u + 1];
    for (int p = beg + lane; p < end; p += 64) {
      const int j = item2slot[pos_items[p]];
      if (j >= 0) mask[r * ldm + j] = value;
    }
  }
}

__global__ void k_slot_map_set(int32_t* __restrict__ map, const int32_t* __restrict__ ids,
                               int64_t S, int clear, uint32_t* __restrict__ bits) {
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int id = ids[s];
  if (id < 0) return;                      // (a short device draw marks its missing positions with -1)
  if (clear) map[id] = -1;
  else atomicMax(&map[id], (int32_t)s);  // duplicate ids: last slot wins (dict semantics)
  if (bits) {                              // the attached "in the pool?" bitmap follows the map
    if (clear) atomicAnd(&bits[id >> 5], ~(1u << (id & 31)));
    else atomicOr(&bits[id >> 5], 1u << (id & 31));
  }
}

// item2slot map -> its attached bitmap (arx_slot_map_attach_bitmap); host side, a handful of entries
struct SlotBits { const void* map; uint32_t* bits; };
static SlotBits g_slot_bits[16];
static int g_slot_bits_n = 0;
static std::mutex g_slot_bits_mu;

static uint32_t* slot_bits_of(const void* map) {
  std::lock_guard<std::mutex> lk(g_slot_bits_mu);
  for (int i = 0; i < g_slot_bits_n; ++i)
    if (g_slot_bits[i].map == map) return g_slot_bits[i].bits;
  return nullptr;
}

static PosMask make_pm(const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                       const int32_t* item2slot) {
  return PosMask{user_ids, pos_ptr, pos_items, item2slot, slot_bits_of(item2slot)};
}

PosMask make_pos_mask(const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                      const int32_t* item2slot) {
  return make_pm(user_ids, pos_ptr, pos_items, item2slot);
}

}  // namespace arx

using namespace arx;

extern "C" {

int arx_pos_mask_scatter(const int32_t* user_ids, int64_t B, const int32_t* pos_ptr,
                         const int32_t* pos_items, const int32_t* item2slot, uint8_t* mask,
                         int64_t ldm, int value, void* stream) {
  ARX_CHECK_ARG(user_ids && pos_ptr && pos_items && item2slot && mask,
                "arx_pos_mask_scatter: null pointer");
  if (B <= 0) return ARX_OK;
  int64_t g = ceil_div(B, 4);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  k_pos_mask_scatter<<<(int)g, 256, 0, as_stream(stream)>>>(user_ids, B, pos_ptr, pos_items,
                                                            item2slot, mask, ldm,
                                                            (uint8_t)(value ? 1 : 0));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_slot_map_set(int32_t* item2slot, const int32_t* ids, int64_t S, int clear,
                     void* stream) {
  ARX_CHECK_ARG(item2slot && ids, "arx_slot_map_set: null pointer");
  if (S <= 0) return ARX_OK;
  k_slot_map_set<<<(int)ceil_div(S, 256), 256, 0, as_stream(stream)>>>(item2slot, ids, S, clear,
                                                                       slot_bits_of(item2slot));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_slot_map_attach_bitmap(const int32_t* item2slot, uint32_t* bits) {
  ARX_CHECK_ARG(item2slot, "arx_slot_map_attach_bitmap: null pointer");
  std::lock_guard<std::mutex> lk(g_slot_bits_mu);
  for (int i = 0; i < g_slot_bits_n; ++i)
    if (g_slot_bits[i].map == item2slot) {
      if (bits) { g_slot_bits[i].bits = bits; return ARX_OK; }
      g_slot_bits[i] = g_slot_bits[--g_slot_bits_n];                 // detach
      return ARX_OK;
    }
  if (!bits) return ARX_OK;
  if (g_slot_bits_n >= 16) return ARX_OK;       // best effort: this map is probed directly
  g_slot_bits[g_slot_bits_n++] = SlotBits{item2slot, bits};
  return ARX_OK;
}

int arx_loss_mw_fwdbwd(const float* logits, int64_t ldl, const float* tscore,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                       const float* row_w,
                       int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                       float* dtscore, void* stream) {
  ARX_CHECK_ARG(logits && tscore, "arx_loss_mw_fwdbwd: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mw_fwdbwd: negative size");
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<false, false>(logits, ldl, tscore, nullptr, mask, ldm,
                                       mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S,
                                       batch_loss, dlogits, lddl, dtscore, PosMask{},
                                       as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  k_loss_margin<false, false><<<(int)B, 256, 0, as_stream(stream)>>>(
      logits, ldl, tscore, nullptr, mask, ldm, mask_rows > 0 ? mask_rows : B, gscale, row_w, S,
      batch_loss, dlogits, lddl, dtscore, PosMask{});
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_mce_fwdbwd(const float* logits, int64_t ldl, const float* tscore,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                       const float* row_w,
                       int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                       float* dtscore, void* stream) {
  ARX_CHECK_ARG(logits && tscore, "arx_loss_mce_fwdbwd: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mce_fwdbwd: negative size");
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<false, false, true>(logits, ldl, tscore, nullptr, mask, ldm,
                                       mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S,
                                       batch_loss, dlogits, lddl, dtscore, PosMask{},
                                       as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  k_loss_margin<false, false, true><<<(int)B, 256, 0, as_stream(stream)>>>(
      logits, ldl, tscore, nullptr, mask, ldm, mask_rows > 0 ? mask_rows : B, gscale, row_w, S,
      batch_loss, dlogits, lddl, dtscore, PosMask{});
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_warp_fwdbwd(const float* logits, int64_t ldl, const int32_t* target,
                         const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                         const float* row_w,
                         int64_t B, int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                         void* stream) {
  ARX_CHECK_ARG(logits && target, "arx_loss_warp_fwdbwd: null pointer");
  ARX_CHECK_ARG(B >= 0 && V > 0, "arx_loss_warp_fwdbwd: bad size");
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<true, false>(logits, ldl, nullptr, target, mask, ldm,
                                      mask_rows > 0 ? mask_rows : B, gscale, row_w, B, V,
                                      batch_loss, dlogits, lddl, nullptr, PosMask{},
                                      as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  k_loss_margin<true, false><<<(int)B, 256, 0, as_stream(stream)>>>(
      logits, ldl, nullptr, target, mask, ldm, mask_rows > 0 ? mask_rows : B, gscale, row_w, V,
      batch_loss, dlogits, lddl, nullptr, PosMask{});
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

#define ARX_POS_LDS_MAX_COLS (1 << 20)   /* 128 KB of mask bits in LDS */

int arx_loss_mw_fwdbwd_pos(const float* logits, int64_t ldl, const float* tscore,
                           const int32_t* user_ids, const int32_t* pos_ptr,
                           const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                           float gscale, const float* row_w, int64_t B, int64_t S,
                           float* batch_loss, float* dlogits, int64_t lddl, float* dtscore,
                           void* stream) {
  ARX_CHECK_ARG(logits && tscore && user_ids && pos_ptr && pos_items && item2slot,
                "arx_loss_mw_fwdbwd_pos: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mw_fwdbwd_pos: negative size");
  if (S > ARX_POS_LDS_MAX_COLS) {
    set_error("arx_loss_mw_fwdbwd_pos: %lld columns exceed the LDS mask (use the mask-array form)", (long long)S);
    return ARX_EUNSUPPORTED;
  }
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<false, true>(logits, ldl, tscore, nullptr, nullptr, 0,
                                      mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S,
                                      batch_loss, dlogits, lddl, dtscore,
                                      make_pm(user_ids, pos_ptr, pos_items, item2slot),
                                      as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  const size_t lds = (size_t)((S + 31) / 32) * 4;
  k_loss_margin<false, true><<<(int)B, 256, lds, as_stream(stream)>>>(
      logits, ldl, tscore, nullptr, nullptr, 0, mask_rows > 0 ? mask_rows : B, gscale, row_w, S,
      batch_loss, dlogits, lddl, dtscore, make_pm(user_ids, pos_ptr, pos_items, item2slot));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_mce_fwdbwd_pos(const float* logits, int64_t ldl, const float* tscore,
                           const int32_t* user_ids, const int32_t* pos_ptr,
                           const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                           float gscale, const float* row_w, int64_t B, int64_t S,
                           float* batch_loss, float* dlogits, int64_t lddl, float* dtscore,
                           void* stream) {
  ARX_CHECK_ARG(logits && tscore && user_ids && pos_ptr && pos_items && item2slot,
                "arx_loss_mce_fwdbwd_pos: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mce_fwdbwd_pos: negative size");
  if (S > ARX_POS_LDS_MAX_COLS) {
    set_error("arx_loss_mce_fwdbwd_pos: %lld columns exceed the LDS mask (use the mask-array form)", (long long)S);
    return ARX_EUNSUPPORTED;
  }
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<false, true, true>(logits, ldl, tscore, nullptr, nullptr, 0,
                                      mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S,
                                      batch_loss, dlogits, lddl, dtscore,
                                      make_pm(user_ids, pos_ptr, pos_items, item2slot),
                                      as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  const size_t lds = (size_t)((S + 31) / 32) * 4;
  k_loss_margin<false, true, true><<<(int)B, 256, lds, as_stream(stream)>>>(
      logits, ldl, tscore, nullptr, nullptr, 0, mask_rows > 0 ? mask_rows : B, gscale, row_w, S,
      batch_loss, dlogits, lddl, dtscore, make_pm(user_ids, pos_ptr, pos_items, item2slot));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_eval_chunk_accum(const float* logits, int64_t ldl, int64_t B, int64_t n, const float* tscore,
                         int mode, int first, float* acc0, float* acc1, void* stream) {
  ARX_CHECK_ARG(logits && acc0 && (mode == 1 ? tscore != nullptr : acc1 != nullptr) && (mode == 0 || mode == 1),
                "arx_eval_chunk_accum: bad argument");
  if (B <= 0 || n <= 0) return ARX_OK;
  k_eval_chunk<<<(int)ceil_div(B, 4), 256, 0, as_stream(stream)>>>(logits, ldl, n, tscore, mode, first, acc0,
                                                                   acc1, B);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_eval_warp_unmask(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, int d,
                         const float* tscore, const int32_t* user_ids, const int32_t* pos_ptr,
                         const int32_t* pos_items, const int32_t* item2col, int64_t mask_rows, int64_t B,
                         int64_t V, float* s_acc, void* stream) {
  ARX_CHECK_ARG(U && P && tscore && user_ids && pos_ptr && pos_items && item2col && s_acc,
                "arx_eval_warp_unmask: null pointer");
  ARX_CHECK_ARG(d > 0 && d <= 256 && d % 4 == 0 && ldu % 4 == 0 && ldp % 4 == 0, "arx_eval_warp_unmask: d %% 4, d <= 256");
  if (B <= 0) return ARX_OK;
  k_eval_unmask<<<(int)ceil_div(B, 4), 256, 0, as_stream(stream)>>>(
      U, ldu, P, ldp, pbias, d, tscore, PosMask{user_ids, pos_ptr, pos_items, item2col},
      mask_rows > 0 ? mask_rows : B, V, s_acc, B);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_eval_finish(int mode, const float* acc0, const float* acc1, const float* tscore, int64_t B,
                    float* batch_loss, void* stream) {
  ARX_CHECK_ARG(acc0 && batch_loss && (mode == 1 || (acc1 && tscore)), "arx_eval_finish: bad argument");
  if (B <= 0) return ARX_OK;
  k_eval_finish<<<(int)ceil_div(B, 256), 256, 0, as_stream(stream)>>>(mode, acc0, acc1, tscore, B, batch_loss);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_mw_fused_pos(const float* logits, int64_t ldl, const float* U, int64_t ldu, const float* T,
                          int64_t ldt, const float* tbias, int64_t tbias_stride, int d,
                          const int32_t* user_ids,
                          const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot,
                          int64_t mask_rows, float gscale, const float* row_w, int64_t B, int64_t S,
                          float* batch_loss, float* dlogits, int64_t lddl, float* tscore_out,
                          float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu, float* dT,
                          int64_t lddt, void* stream) {
  ARX_CHECK_ARG(logits && U && T && user_ids && pos_ptr && pos_items && item2slot,
                "arx_loss_mw_fused_pos: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mw_fused_pos: negative size");
  const bool ok = d > 0 && d <= 256 && d % 4 == 0 && ldu % 4 == 0 && ldt % 4 == 0 &&
                  (!dU || lddu % 4 == 0) && (!dT || lddt % 4 == 0) && S <= 2048 &&
                  ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(T) |
                    reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(dT)) & 15) == 0;
  if (!ok) {
    set_error("arx_loss_mw_fused_pos: shape not supported (d %% 4, d <= 256, S <= 2048, 16-byte rows)");
    return ARX_EUNSUPPORTED;
  }
  if (B == 0) return ARX_OK;
  DotFuse df{U, ldu, T, ldt, tbias, tbias_stride > 0 ? tbias_stride : 1, d, tscore_out,
             dtscore_stride > 0 ? dtscore_stride : 1, dU, lddu, dT, lddt};
  if (!launch_margin_wave<false, true>(logits, ldl, nullptr, nullptr, nullptr, 0,
                                       mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S, batch_loss,
                                       dlogits, lddl, dtscore,
                                       make_pm(user_ids, pos_ptr, pos_items, item2slot),
                                       as_stream(stream), df)) {
    set_error("arx_loss_mw_fused_pos: logits / dlogits layout not supported by the wave kernel");
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_mce_fused_pos(const float* logits, int64_t ldl, const float* U, int64_t ldu, const float* T,
                          int64_t ldt, const float* tbias, int64_t tbias_stride, int d,
                          const int32_t* user_ids,
                          const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot,
                          int64_t mask_rows, float gscale, const float* row_w, int64_t B, int64_t S,
                          float* batch_loss, float* dlogits, int64_t lddl, float* tscore_out,
                          float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu, float* dT,
                          int64_t lddt, void* stream) {
  ARX_CHECK_ARG(logits && U && T && user_ids && pos_ptr && pos_items && item2slot,
                "arx_loss_mce_fused_pos: null pointer");
  ARX_CHECK_ARG(B >= 0 && S >= 0, "arx_loss_mce_fused_pos: negative size");
  const bool ok = d > 0 && d <= 256 && d % 4 == 0 && ldu % 4 == 0 && ldt % 4 == 0 &&
                  (!dU || lddu % 4 == 0) && (!dT || lddt % 4 == 0) && S <= 2048 &&
                  ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(T) |
                    reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(dT)) & 15) == 0;
  if (!ok) {
    set_error("arx_loss_mce_fused_pos: shape not supported (d %% 4, d <= 256, S <= 2048, 16-byte rows)");
    return ARX_EUNSUPPORTED;
  }
  if (B == 0) return ARX_OK;
  DotFuse df{U, ldu, T, ldt, tbias, tbias_stride > 0 ? tbias_stride : 1, d, tscore_out,
             dtscore_stride > 0 ? dtscore_stride : 1, dU, lddu, dT, lddt};
  if (!launch_margin_wave<false, true, true>(logits, ldl, nullptr, nullptr, nullptr, 0,
                                       mask_rows > 0 ? mask_rows : B, gscale, row_w, B, S, batch_loss,
                                       dlogits, lddl, dtscore,
                                       make_pm(user_ids, pos_ptr, pos_items, item2slot),
                                       as_stream(stream), df)) {
    set_error("arx_loss_mce_fused_pos: logits / dlogits layout not supported by the wave kernel");
    return ARX_EUNSUPPORTED;
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_warp_fwdbwd_pos(const float* logits, int64_t ldl, const int32_t* target,
                             const int32_t* user_ids, const int32_t* pos_ptr,
                             const int32_t* pos_items, const int32_t* item2slot,
                             int64_t mask_rows, float gscale, const float* row_w, int64_t B,
                             int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                             void* stream) {
  ARX_CHECK_ARG(logits && target && user_ids && pos_ptr && pos_items && item2slot,
                "arx_loss_warp_fwdbwd_pos: null pointer");
  ARX_CHECK_ARG(B >= 0 && V > 0, "arx_loss_warp_fwdbwd_pos: bad size");
  if (V > ARX_POS_LDS_MAX_COLS) {
    set_error("arx_loss_warp_fwdbwd_pos: %lld columns exceed the LDS mask (use the mask-array form)", (long long)V);
    return ARX_EUNSUPPORTED;
  }
  if (B == 0) return ARX_OK;
  if (launch_margin_wave<true, true>(logits, ldl, nullptr, target, nullptr, 0,
                                     mask_rows > 0 ? mask_rows : B, gscale, row_w, B, V,
                                     batch_loss, dlogits, lddl, nullptr,
                                     make_pm(user_ids, pos_ptr, pos_items, item2slot),
                                     as_stream(stream))) {
    ARX_CHECK_LAUNCH();
    return ARX_OK;
  }
  const size_t lds = (size_t)((V + 31) / 32) * 4;
  auto kern = k_loss_margin<true, true>;
  if (lds > 48 * 1024)
    ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<(int)B, 256, lds, as_stream(stream)>>>(
      logits, ldl, nullptr, target, nullptr, 0, mask_rows > 0 ? mask_rows : B, gscale, row_w, V,
      batch_loss, dlogits, lddl, nullptr, make_pm(user_ids, pos_ptr, pos_items, item2slot));
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_rs_fwdbwd(const float* logits, int64_t ldl, const int32_t* target, const uint8_t* mask,
                       int64_t ldm, const int32_t* user_ids, const int32_t* pos_ptr,
                       const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                       int kind, int loss_func, float exp_p, float gscale, const float* row_w,
                       int64_t B, int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                       void* stream) {
  ARX_CHECK_ARG(logits && target, "arx_loss_rs_fwdbwd: null pointer");
  ARX_CHECK_ARG(B >= 0 && V > 0, "arx_loss_rs_fwdbwd: bad size");
  ARX_CHECK_ARG(kind >= 0 && kind <= 3 && loss_func >= 0 && loss_func <= 5,
                "arx_loss_rs_fwdbwd: kind in 0..3 (rs, rs-sig, rs-sig2, bbpr), loss_func in 0..5");
  if (B == 0) return ARX_OK;
  const int64_t mr = mask_rows > 0 ? mask_rows : B;
  if (user_ids) {
    ARX_CHECK_ARG(pos_ptr && pos_items && item2slot, "arx_loss_rs_fwdbwd: incomplete positives CSR");
    if (V > ARX_POS_LDS_MAX_COLS) {
      set_error("arx_loss_rs_fwdbwd: %lld columns exceed the LDS mask (use the mask-array form)", (long long)V);
      return ARX_EUNSUPPORTED;
    }
    const size_t lds = (size_t)((V + 31) / 32) * 4;
    auto kern = k_loss_rs<true>;
    if (lds > 48 * 1024)
      ARX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<(int)B, 256, lds, as_stream(stream)>>>(logits, ldl, target, nullptr, 0, mr, kind, loss_func,
                                                  exp_p, gscale, row_w, V, batch_loss, dlogits, lddl,
                                                  make_pm(user_ids, pos_ptr, pos_items, item2slot));
  } else {
    k_loss_rs<false><<<(int)B, 256, 0, as_stream(stream)>>>(logits, ldl, target, mask, ldm, mr, kind,
                                                            loss_func, exp_p, gscale, row_w, V,
                                                            batch_loss, dlogits, lddl, PosMask{});
  }
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_ce_fwdbwd(const float* logits, int64_t ldl, const int32_t* target, float gscale,
                       const float* row_w, int64_t B, int64_t V, float* batch_loss,
                       float* dlogits, int64_t lddl, void* stream) {
  ARX_CHECK_ARG(logits && target, "arx_loss_ce_fwdbwd: null pointer");
  ARX_CHECK_ARG(B >= 0 && V > 0, "arx_loss_ce_fwdbwd: bad size");
  if (B == 0) return ARX_OK;
  if (V <= 4096)
    k_loss_ce_regs<<<(int)B, 256, 0, as_stream(stream)>>>(logits, ldl, target, gscale, row_w, V,
                                                          batch_loss, dlogits, lddl);
  else
    k_loss_ce<<<(int)B, 256, 0, as_stream(stream)>>>(logits, ldl, target, gscale, row_w, V,
                                                     batch_loss, dlogits, lddl);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_row_logsumexp(const float* logits, int64_t ldl, int64_t B, int64_t V, float* out,
                      void* stream) {
  ARX_CHECK_ARG(logits && out && V > 0 && B >= 0, "arx_row_logsumexp: bad argument");
  if (B == 0) return ARX_OK;
  k_row_lse<<<(int)B, 256, 0, as_stream(stream)>>>(logits, ldl, V, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_loss_warp_eval(const float* logits, int64_t ldl, const int32_t* target,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, int64_t B, int64_t V,
                       float* margin_rank, int32_t* true_rank, void* stream) {
  ARX_CHECK_ARG(logits && target && margin_rank && true_rank, "arx_loss_warp_eval: null pointer");
  if (B <= 0) return ARX_OK;
  k_warp_eval<<<(int)B, 256, 0, as_stream(stream)>>>(logits, ldl, target, mask, ldm,
                                                     mask_rows > 0 ? mask_rows : B, V,
                                                     margin_rank, true_rank);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

}  // extern "C"
