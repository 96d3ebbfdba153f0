/* arx.h -- C ABI of libarx.so: the MI355X (gfx950) hot path of A-RecSys.
 *
 * The reference (skywaLKer518/A-Recsys) has no FFI: its hot path is a TF-1.0
 * op graph built by Python classes.  This header is the boundary a maintainer
 * binds with ctypes; every entry point cites the reference
 * call site(s) whose arithmetic it replaces (paths relative to the reference
 * root).  Conventions:
 *   - extern "C"; returns 0 (ARX_OK) or a negative ARX_E* code; the message for
 *     the calling thread is available from arx_last_error().
 *   - every buffer is a CALLER-OWNED DEVICE pointer (torch tensors are only the
 *     holders); element counts are int64_t; table-row / token indices int32_t.
 *   - row-major fp32 everywhere, explicit leading dimensions in elements.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *     no hidden allocation: scratch comes from a caller-sized workspace.
 *   - no exceptions / C++ types cross the boundary.
 */
#ifndef ARX_H_
#define ARX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARX_OK 0
#define ARX_EINVAL (-1)       /* bad argument (null pointer, bad size, alignment) */
#define ARX_EHIP (-2)         /* a HIP runtime call / launch failed */
#define ARX_EWORKSPACE (-3)   /* workspace too small */
#define ARX_EUNSUPPORTED (-4) /* shape outside the implemented envelope */

#define ARX_KEY_NONE 0x7fffffff /* "no row": skipped by arx_sparse_adagrad */

/* ---- library ---------------------------------------------------------- */
const char* arx_last_error(void);
int arx_version(void);
/* device facts used by bench.py's roofline (CU count, wave size, arch name) */
int arx_device_info(int* cu_count, int* wave_size, int* lds_bytes, char* arch, int arch_len);

/* ---- a4 / a7: ragged CSR expansion (integer, bit-exact) ----------------
 * attributes/mulhot_index.py:48-52 batch_slice2 and :62-67 batch_segids2, as
 * used by embed_attribute.py:392-396 (batch lookup) and :331-347 (sampled-pool
 * staging).  For r in [0,B): row = row_ids ? row_ids[r] : r,
 *   offsets[r] = sum_{q<r} lens[row_q]            (offsets[B] = total)
 *   token_ids[offsets[r]+j] = vals[starts[row]+j],  segids[offsets[r]+j] = r.
 * Positions in [total, capacity) are filled with pad_token / pad_seg.
 * *total_out (device int32) receives the packed length (embed_attribute.py:338,347).
 * For the gradient scatter the same expansion also yields, per token, the
 * gradient-source row and the 1/len factor of tf.div's gradient: segids are
 * offset by seg_base and coef_out[q] = coef_scale / lens[row] (nullable).
 * workspace: arx_csr_expand_workspace_bytes(B). */
size_t arx_csr_expand_workspace_bytes(int64_t B);
int arx_csr_expand(const int32_t* vals, const int32_t* starts, const int32_t* lens,
                   const int32_t* row_ids, int64_t B,
                   int32_t* token_ids, int32_t* segids, int64_t capacity,
                   int32_t* offsets, int32_t* total_out,
                   int32_t pad_token, int32_t pad_seg,
                   int32_t seg_base, float coef_scale, float* coef_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The same expansion WITHOUT compaction, for the gradient scatter only: slot r*max_len + j holds
 * token j of bag r (segid = seg_base + r, coef = coef_scale / len), slots with j >= len hold
 * pad_token (coef 0).  One launch, no prefix sums: arx_sparse_adagrad* drop the pads in the first
 * pass of their sort.  Buffers hold B*max_len entries (max_len >= every lens[row]). */
int arx_bag_expand_padded(const int32_t* vals, const int32_t* starts, const int32_t* lens,
                          const int32_t* row_ids, int64_t B, int max_len, int32_t pad_token,
                          int32_t seg_base, float coef_scale, int32_t* token_ids, int32_t* segids,
                          float* coef_out, void* stream);
/* One-hot twin of the above for the gradient scatter (a17): keys_out[i] =
 * cat_map ? cat_map[ids[i]] : ids[i]; src_out[i] = row_base + i; coef_out[i] = coef. */
int arx_sparse_site_onehot(const int32_t* cat_map, const int32_t* ids, int64_t n,
                           int32_t row_base, float coef, int32_t* keys_out, int32_t* src_out,
                           float* coef_out, void* stream);

/* Row-sharded item table: table rows are striped over the
 * ranks, owner = id % world, local row = id / world.  rows_out[i] = local row if
 * this rank owns ids[i] else zero_row (an all-zero padding row of the shard);
 * keys_out[i] = local row or ARX_KEY_NONE (so non-owned lookups get no update). */
int arx_shard_route(const int32_t* ids, int64_t n, int world, int rank, int32_t zero_row,
                    int32_t* rows_out, int32_t* keys_out, void* stream);
/* Block layout of the shared negative pool on a row-sharded item table (the pool itself:
 * embed_attribute.py:320-348 update_sampled).  Owner g's pool items, in slot order, are rows
 * [0, counts[g]) of its block; blocks travel padded to `cap` rows.  counts has world + 1 cells:
 * counts[0 .. world) the owners' slot counts (the caller reads them to choose cap), counts[world] the number
 * of NEGATIVE ids (no owner: gidx = -1, no row of any block -- the caller rejects such a pool).  cap == 0:
 * only counts is written.  cap > 0: gidx[s] = owner * cap + position (row of slot s in the gathered
 * [world * cap] blocks), my_slots[0 .. counts[rank]) = this rank's slots (rest = S), pool_rows[...] = their
 * local rows (rest = zero_row).  Any S, world <= 256; one launch. */
int arx_pool_blocks(const int32_t* ids, int64_t S, int world, int rank, int32_t zero_row, int64_t cap,
                    int32_t* counts, int32_t* gidx, int32_t* my_slots, int32_t* pool_rows, void* stream);
/* strided 2-D copy (packs / unpacks the all-to-all blocks of the sharded scorer) */
int arx_copy_2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows,
                int64_t cols, void* stream);
/* dst[i * dst_stride] = src[i * src_stride], i < n: one column of the packed [rows, d+4] rows of
 * the sharded exchanges <-> a dense vector (bias, bias gradient) */
int arx_copy_strided_f32(const float* src, int64_t src_stride, float* dst, int64_t dst_stride,
                         int64_t n, void* stream);
/* dst[c, r] = src[r, c] (rows x cols -> cols x rows); used to put W_x^T / dW^T of the LSTM
 * into the layouts the streaming GEMMs take (seqModel.py:477 backward) */
int arx_transpose_f32(const float* src, int64_t lds, int64_t rows, int64_t cols, float* dst,
                      int64_t ldd, void* stream);
/* dst[r, 0:width] = src[rows[r], 0:width], r < n (zeros for a row index outside [0, src_rows)): rows of ANY width --
 * the block-row <-> pool-slot permutations of the sharded step's all-to-all-of-logits exchange,
 * whose rows are one batch shard (B_loc floats) wide */
int arx_gather_rows_wide(const float* src, int64_t lds, int64_t src_rows, const int32_t* rows, int64_t n,
                         int64_t width, float* dst, int64_t ldd, void* stream);

/* ---- a5: one-hot attribute gather --------------------------------------
 * embed_attribute.py:371-381: rows = cat_map[ids]; E[rows] (+ bias[rows]).
 * out[r, 0:d] = (accumulate ? out : 0) + scale * E[cat_map[ids[r]], :]
 * bias_out[r] likewise from bias[Vf] (both nullable together).
 * cat_map may be NULL (identity).  d % 4 == 0, d <= 1024. */
int arx_gather_onehot_fwd(const float* E, const float* bias, const int32_t* cat_map,
                          const int32_t* ids, int64_t B, int d, float scale, int accumulate,
                          float* out, int64_t ldo, float* bias_out, void* stream);

/* Packed form for the sharded exchanges (no reference counterpart): row r of the
 * output holds [ scale * E[row] (d floats) | scale * bias[row] | pad ], ldo > d -- the bias rides
 * in column d of the row that is sent to the peer, no second buffer and no copy. */
int arx_gather_onehot_packed_fwd(const float* E, const float* bias, const int32_t* cat_map,
                                 const int32_t* ids, int64_t B, int d, float scale, float* out,
                                 int64_t ldo, void* stream);

/* An entity with an id feature AND one multi-hot attribute, both lookups in one launch:
 * out[r] = (accumulate ? out[r] : 0) + scale * ( E_id[cat_map ? cat_map[ids[r]] : ids[r]] + mean of
 * E_tok over the entity's bag ), bias likewise (embed_attribute.py:371-407 + the reduce_mean of
 * :219/:235 with scale = 1/F). */
int arx_gather_id_plus_bag(const float* E_id, const float* bias_id, const int32_t* cat_map,
                           const float* E_tok, const float* bias_tok, const int32_t* vals,
                           const int32_t* starts, const int32_t* lens, const int32_t* ids, int64_t B,
                           int d, float scale, int accumulate, float* out, int64_t ldo,
                           float* bias_out, void* stream);

/* Several lookups of a step in ONE launch (embed_attribute.py:371-407 for the users, the target
 * items and the sampled pool of a step -- three independent launches otherwise).  Site s has a one-hot
 * feature (E_id[s], nullable, with cat_map[s] / bias_id[s]), a multi-hot feature (E_tok[s], nullable,
 * with vals / starts / lens / bias_tok[s]) or both:
 *   out[s][r] = scale[s] * ( E_id[map[id]] + mean over the bag of id of E_tok rows ),  id = ids[s][r]
 * (the same arithmetic, bit for bit, as arx_gather_onehot_fwd / arx_gather_mulhot_mean_fwd /
 * arx_gather_id_plus_bag on that site).  bias_out[s] (nullable): the same combination of the bias
 * cells.  All tables d wide; nsites <= 8. */
int arx_lookup_multi(int nsites, const float* const* E_id, const float* const* bias_id,
                     const int32_t* const* cat_map, const float* const* E_tok,
                     const float* const* bias_tok, const int32_t* const* vals,
                     const int32_t* const* starts, const int32_t* const* lens,
                     const int32_t* const* ids, const int64_t* n, int d, const float* scale,
                     float* const* out, const int64_t* ldo, float* const* bias_out, void* stream);

/* nsites one-hot lookups of equal width d in ONE launch (site s: out[s][r, 0:d] = scale[s] *
 * E[s][cat_map[s] ? cat_map[s][ids[s][r]] : ids[s][r], :], bias_out[s][r] likewise; r < n[s]).
 * The lookups of a step (user ids, target items, sampled pool, input items) are independent. */
int arx_gather_onehot_multi(int nsites, const float* const* E, const float* const* bias,
                            const int32_t* const* cat_map, const int32_t* const* ids,
                            const int64_t* n, int d, const float* scale, float* const* out,
                            const int64_t* ldo, float* const* bias_out, void* stream);
/* ... with a stride for every site's bias output, bias_out[s][r * ldb[s]], and bias input,
 * bias[s][row * ldbi[s]] (NULL: 1 everywhere).  bias_out[s] = out[s] + d with ldb[s] = ldo[s] gives
 * the packed rows of arx_gather_onehot_packed_fwd: the sharded step's three lookups (own user rows;
 * pool block and requested target rows, packed for the exchanges) are one launch.  bias[s] = E[s] + d0
 * with ldbi[s] = d reads column d0 of a table of PACKED rows (d = its full row width): reordering
 * received packed rows and splitting off their bias column is one launch too. */
int arx_gather_onehot_multi_ld(int nsites, const float* const* E, const float* const* bias, const int64_t* ldbi,
                               const int32_t* const* cat_map, const int32_t* const* ids,
                               const int64_t* n, int d, const float* scale, float* const* out,
                               const int64_t* ldo, float* const* bias_out, const int64_t* ldb, void* stream);

/* ---- a5: multi-hot gather + segment-mean (K1) ---------------------------
 * embed_attribute.py:382-407 with mulhot_index.py:48-67 fused in:
 *   bag(r) = vals[starts[id_r] : starts[id_r] + lens[id_r]]
 *   out[r,:] = (accumulate ? out : 0) + scale * (sum_k E[bag_k,:]) / lens[id_r]
 * bias_out[r] likewise (tf.div of unsorted_segment_sum by float length). */
int arx_gather_mulhot_mean_fwd(const float* E, const float* bias, const int32_t* vals,
                               const int32_t* starts, const int32_t* lens,
                               const int32_t* ids, int64_t B, int d, float scale,
                               int accumulate, float* out, int64_t ldo, float* bias_out,
                               void* stream);

/* ---- a9: target dot score ------------------------------------------------
 * embed_attribute.py:219-220: score[r] = sum_d(U[r,:]*T[r,:]) + tbias[r]. */
int arx_dot_score_fwd(const float* U, int64_t ldu, const float* T, int64_t ldt,
                      const float* tbias, int64_t B, int d, float* score, void* stream);
/* dU[r,:] = (acc_dU ? dU : 0) + ds[r]*T[r,:];  dT[r,:] = ds[r]*U[r,:] (dT nullable) */
int arx_dot_score_bwd(const float* U, int64_t ldu, const float* T, int64_t ldt,
                      const float* dscore, int64_t B, int d, float* dU, int64_t lddu,
                      int acc_dU, float* dT, int64_t lddt, void* stream);

/* ---- a8: scorer GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) --------------
 * embed_attribute.py:171,188-193,205 in embedding-space form:
 *   C[M,N] = alpha * op(A)[M,K] . op(B)[K,N] + beta * C + col_bias[n]
 * transA=0: A stored [M,K] (lda>=K); transA=1: A stored [K,M] (lda>=M).
 * transB=0: B stored [K,N] (ldb>=N); transB=1: B stored [N,K] (ldb>=K).
 * col_bias nullable.  Split-K (deterministic, workspace partials) is chosen
 * internally; workspace >= arx_gemm_f32_workspace_bytes(M,N,K). */
size_t arx_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int arx_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                 float* C, int64_t ldc, const float* col_bias,
                 void* workspace, size_t workspace_bytes, void* stream);
/* Same, and additionally a_rowsum[m] = sum_k op(A)[m,k] (nullable).  With
 * op(A) = dlogits^T this is the item-bias gradient (embed_attribute.py:171 `+ bias`
 * back-propagated), produced by the dI GEMM for free instead of a separate pass. */
/* EXPERIMENT (off unless ARX_GEMM_BX6=1 in arx.ops): the same C = A . B^T + col_bias
 * (embed_attribute.py:171 matmul(user, item^T) + bias), f32 in / f32 out, on the bf16 matrix pipe: every
 * f32 operand is split exactly into three bf16 pieces and six of the nine piece products (the other three are
 * below 2^-26 of the product) are accumulated in f32 -- every bit an f32 multiply-add chain carries, at 16/6 of
 * the f32-MFMA peak (gfx950 runs f32-input MFMA at 1/16 of the bf16 rate).  K in {64, 128}, N % 128 == 0,
 * lda / ldb / ldc % 4 == 0; workspace >= arx_gemm_nt_bx6_workspace_bytes(N, K) (the bf16 planes of B). */
size_t arx_gemm_nt_bx6_workspace_bytes(int64_t N, int64_t K);
int arx_gemm_nt_bx6(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    const float* col_bias, float* C, int64_t ldc, void* workspace, size_t workspace_bytes,
                    void* stream);

/* C[M, N] = beta C + A[M, K] . Bt[N, K]^T, six-term (f32-exact on the bf16 matrix pipe), for a SMALL second operand
 * given k-contiguous: the LSTM cell's input gradient dx = dz . W_x^T (seqModel.py:99-103 static_rnn backward; tf's
 * MatMul gradient) with Bt = W_x = rows [0, din) of the cell's [din + h, 4h] weight matrix as they lie.  The rows of
 * A stream through LDS in coalesced segments and are split into bf16 pieces on the fly, the planes of Bt are made in
 * LDS per workgroup: no operand planes in HBM, no workspace.  N % 32 == 0, N <= 128, K in {64, 128, 256}, rows
 * 16-byte aligned (arx_gemm_bt_bx6_supported); other shapes: arx_gemm_f32. */
int arx_gemm_bt_bx6_supported(int64_t M, int64_t N, int64_t K);
int arx_gemm_bt_bx6(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* Bt, int64_t ldb,
                    float beta, float* C, int64_t ldc, void* stream);
int arx_gemm_f32_rowsum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                        const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                        float* C, int64_t ldc, const float* col_bias, float* a_rowsum,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Per-time-step TN products for the LSTM path (seqModel.py:480-493 scores every
 * step against the same pool): for t in [0,steps): C_steps[t] = A_t^T . B_t with
 * A_t = A[t*Kb:(t+1)*Kb, 0:M], B_t = B[t*Kb:(t+1)*Kb, 0:N]; rowsum_steps[t][m] =
 * sum_k A_t[k,m].  Optionally C_sum = beta*C_sum + sum_t C_steps[t] (fixed order) and
 * rowsum_sum likewise.  The per-step products are what tf.clip_by_global_norm
 * sees for a matmul'd table (one dense gradient per unrolled step). */
int arx_gemm_f32_steps_tn(int64_t steps, int64_t M, int64_t N, int64_t Kb, const float* A,
                          int64_t lda, const float* B, int64_t ldb, float* C_steps,
                          float* rowsum_steps, float beta, float* C_sum, int64_t ldc,
                          float* rowsum_sum, void* stream);

/* The LSTM cell's weight gradient in one pass over dz (seqModel.py:99-103: static_rnn of
 * LSTMCell, whose kernel is one [din + h, 4h] matrix applied to concat(x_t, h_{t-1})):
 *   Ct[0:N1, m]     = sum_k A[k, m] . B1[k, 0:N1]
 *   Ct[N1:N1+N2, m] = sum_{k >= shift} A[k, m] . B2[k - shift, 0:N2]
 * i.e. (A^T . [B1 | shifted B2])^T stored row-major [N1 + N2, M] (ldct >= M); a_rowsum[m]
 * (optional) = sum_k A[k, m].  With A = dz [L*B, 4h], B1 = x, B2 = the cell outputs and
 * shift = B this is dW (both halves) and db of tf's LSTMCell backward.  N1 + N2 <= 128,
 * N1 % 4 == N2 % 4 == M % 4 == 0, K % 32 == 0, 16-byte aligned operands; anything else
 * ARX_EUNSUPPORTED.  Deterministic (fixed-order split-K). */
size_t arx_gemm_f32_tn_pair_workspace_bytes(int64_t M, int64_t N, int64_t K);
int arx_gemm_f32_tn_pair(int64_t M, int64_t N1, int64_t N2, int64_t K, const float* A, int64_t lda,
                         const float* B1, int64_t ldb1, const float* B2, int64_t ldb2,
                         int64_t shift, float* Ct, int64_t ldct, float* a_rowsum, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- a14: positive mask ---------------------------------------------------
 * embed_attribute.py:651-672 (mask variable + scatter_update set/reset) and
 * :721-745 (host index list).  For each batch row r and each positive item v
 * of user_ids[r] (CSR pos_ptr/pos_items over users; users without positives
 * have empty rows): j = item2slot[v]; if j >= 0: mask[r*ldm + j] = value.
 * value=0 is set_mask, value=1 is reset_mask.  mask is uint8, 1 = keep. */
int arx_pos_mask_scatter(const int32_t* user_ids, int64_t B, const int32_t* pos_ptr,
                         const int32_t* pos_items, const int32_t* item2slot,
                         uint8_t* mask, int64_t ldm, int value, void* stream);
/* item -> pool-slot map maintenance for the sampled pool (the device twin of
 * utils/prepare_train.py:12-16 item_sampled_id2idx): map[ids[s]] = s (or -1). */
int arx_slot_map_set(int32_t* item2slot, const int32_t* ids, int64_t S, int clear, void* stream);
/* Optional 1-bit-per-item "is in the pool?" table in front of an item2slot map: bits = caller-owned
 * uint32[(items + 32) / 32], zero-initialised, registered for THIS map pointer (bits NULL: detach --
 * REQUIRED before the map's memory is released; with 16 maps registered the call is a no-op and the
 * map is probed directly).  arx_slot_map_set keeps it in step with the map; every `*_pos` loss entry that is
 * given the map then answers "not in the pool" -- the case for almost every positive of a user --
 * from the bit (a table 32x smaller, L2-resident) instead of a random 4-byte read of the map
 * (embed_attribute.py:729-745: the per-user loop over positives, `if i in item_sampled_id2idx`). */
int arx_slot_map_attach_bitmap(const int32_t* item2slot, uint32_t* bits);

/* ---- a10-a12: batch losses, forward + backward fused ---------------------
 * d(total)/d(batch_loss[r]) = gscale * (row_w ? row_w[r] : 1).
 * The mask row of logits row r is r % mask_rows (the LSTM path scores L*B
 * time-major rows against one [B, S] mask, seqModel.py:489-493); mask NULL = all kept.
 * dlogits may alias logits (in place) or be NULL (forward only).
 * mw   : embed_attribute.py:641-649  log(1 + sum_s relu(mask*(x - t + 1)))
 * warp : embed_attribute.py:605-618  same with t = logits[r,target[r]] over V
 * ce   : embed_attribute.py:529-531  sparse softmax cross entropy           */
int arx_loss_mw_fwdbwd(const float* logits, int64_t ldl, const float* tscore,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                       const float* row_w,
                       int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                       float* dtscore, void* stream);
int arx_loss_warp_fwdbwd(const float* logits, int64_t ldl, const int32_t* target,
                         const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                         const float* row_w,
                         int64_t B, int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                         void* stream);
/* item_frequency on device (utils/prepare_train.py:19-35): counts[v] += 1 for every training
 * interaction's item id v in [0, n_items) (integer atomics: order-independent), then, when
 * `weights` is given, weights[v] = (counts[v] / total)^power for counts[v] > 0 else 0 -- the
 * reference's p_item before normalisation (arx_sample_wor does not need it normalised), over ALL
 * n_items ids instead of the reference's hash-ordered list of the seen ones.  `counts` is
 * accumulated into (zero it first; several calls / ranks may add up before the weights are
 * taken with n = 0); `total` = number of interactions counted. */
int arx_item_frequency(const int32_t* item_ids, int64_t n, int64_t n_items, int64_t total, float power,
                       int32_t* counts, float* weights, void* stream);
/* Negative-pool sampler on device (replaces utils/prepare_train.py:7-17
 * np.random.choice(items, S, replace=False, p)): weighted sampling without replacement as an
 * exponential race -- key_i = -ln(u_i)/w_i, the S smallest keys in ascending order have the law
 * of S sequential weighted draws without replacement.  weights [n] need not be normalised;
 * entries <= 0 are never drawn.  out_idx [S]: item positions in draw order (-1 if fewer than S
 * positive weights).  Deterministic in (seed, counter); NOT numpy's random stream.
 * workspace >= arx_sample_wor_workspace_bytes(n). */
size_t arx_sample_wor_workspace_bytes(int64_t n);
int arx_sample_wor(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                   int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);
/* The same draw with a pre-filter for very large item sets: keys above key_cap (> 0) are dropped
 * before the sort.  With key_cap = c * S / sum(weights) about c * S items survive (key_i < t with
 * probability 1 - exp(-w_i t)), so for c = 8 the S smallest keys are all below the cap except with
 * probability < exp(-3S) and the result is the un-capped one -- while the sort's later passes
 * handle ~8 S entries instead of n (100 M-item shard: ~5 ms -> ~0.3 ms per redraw).  Entries of
 * out_idx are -1 if fewer than S keys survive.
 * Draws of S <= 2048 items with a cap never store the n keys: the survivors are appended to a list
 * of 16384 entries and sorted by one workgroup (same keys, same (key, item) order: the same draw; the
 * workspace is arx_sample_wor_keys_workspace_bytes(n, S, key_cap) = 128 KB instead of 28 n bytes).  A
 * cap that lets MORE than 16384 keys through (8 S expected survivors is the contract) yields -1 in
 * every entry. */
int arx_sample_wor_capped(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                          float key_cap, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                          void* stream);
size_t arx_sample_wor_keys_workspace_bytes(int64_t n, int64_t S, float key_cap);
/* The S smallest of m (key, id) pairs in (key, position) order: out[r] = id of the pair of rank r.  Merges the
 * ranks' race lists of ONE negative-pool draw over an item set whose weights are sharded (utils/prepare_train.py:7-17
 * draws the pool from one distribution; arx.dist.draw_global_pool all-gathers every rank's S smallest (key, id) pairs
 * rank-major: ties go to the lower rank, then to the earlier entry -- a stable order, identical on every rank).
 * keys: non-negative floats or +inf.  m <= 16384, 0 < S <= m.  One launch. */
int arx_merge_keyed_take(const float* keys, const int32_t* ids, int64_t m, int64_t S, int32_t* out, void* stream);

/* ... and the race keys of the drawn items (out_keys [S], ascending; +inf where out_idx is -1; NULL:
 * not wanted).  For ONE draw over an item set that is sharded over several ranks (the reference draws
 * its S negatives from one distribution, prepare_train.py:7-17): every rank races its own shard
 * -- weights on a common scale, independent seeds --, the ranks exchange their S smallest (key, id)
 * pairs and keep the S smallest of the union: the S smallest keys of the whole item set, i.e. exactly
 * the single-process draw (arx.dist.draw_global_pool). */
int arx_sample_wor_keys(const float* weights, int64_t n, int64_t S, uint64_t seed, uint64_t counter,
                        float key_cap, int32_t* out_idx, float* out_keys, void* workspace,
                        size_t workspace_bytes, void* stream);

/* rs / rs-sig / rs-sig2 / bbpr losses (embed_attribute.py:551-603 _compute_rs_loss) over full
 * logits [B, V], forward + backward fused.  kind: 0 rs, 1 rs-sig, 2 rs-sig2, 3 bbpr;
 * loss_func (hmf_model.py loss_func): 0 log, 1 exp, 2 poly, 3 poly2, 4 linear, 5 square, with
 * exponent/base exp_p (default 1.005).  The mask is either the byte array `mask` [mask_rows, ldm]
 * (0 = masked out; NULL = keep all) or, when user_ids != NULL, built on the fly in LDS from the
 * positives CSR (as in arx_loss_warp_fwdbwd_pos).  The target column receives minus the row sum
 * of the gradient.  dlogits may be NULL (forward only). */
int arx_loss_rs_fwdbwd(const float* logits, int64_t ldl, const int32_t* target, const uint8_t* mask,
                       int64_t ldm, const int32_t* user_ids, const int32_t* pos_ptr,
                       const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                       int kind, int loss_func, float exp_p, float gscale, const float* row_w,
                       int64_t B, int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                       void* stream);

/* mw / warp with the positive mask built inside the kernel (embed_attribute.py:
 * 721-745 + 651-672 fused): column j of row r is masked iff some positive item v
 * of user_ids[r % mask_rows] has item2slot[v] == j.  No [mb, W] mask array is
 * materialised (the reference's is mb*V bools).  W <= 2^20 columns (LDS bits). */
int arx_loss_mw_fwdbwd_pos(const float* logits, int64_t ldl, const float* tscore,
                           const int32_t* user_ids, const int32_t* pos_ptr,
                           const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                           float gscale, const float* row_w, int64_t B, int64_t S,
                           float* batch_loss, float* dlogits, int64_t lddl, float* dtscore,
                           void* stream);
/* The same 'mw' loss with the target score fused in (embed_attribute.py:208-220): the wave that
 * owns row r forms t_r = U_r . T_r + tbias_r (-> tscore_out), and after the loss gradient
 * dT_r = dt_r * U_r and dU_r = dt_r * T_r (both WRITTEN; add the scorer's dU onto dU afterwards).
 * tbias / dtscore may be strided (element r at [r * stride]: the bias column of packed rows).
 * d % 4 == 0, d <= 256, S <= 2048 (S % 4 == 0), 16-byte aligned rows; else ARX_EUNSUPPORTED. */
int arx_loss_mw_fused_pos(const float* logits, int64_t ldl, const float* U, int64_t ldu, const float* T,
                          int64_t ldt, const float* tbias, int64_t tbias_stride, int d,
                          const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                          const int32_t* item2slot, int64_t mask_rows, float gscale, const float* row_w,
                          int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                          float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU,
                          int64_t lddu, float* dT, int64_t lddt, void* stream);

/* Streaming full-vocabulary evaluation losses (hmf_model.py:130,144; lstm/seqModel.py:510: the
 * loss_eval of a model trained with a sampled loss is 'warp' / 'ce' over ALL V logits).  The
 * caller runs the scorer GEMM over chunks of the pool and folds every [B, n] chunk of logits in:
 *   mode 0 (ce)   acc0 = running max, acc1 = running sum exp(x - max)   (online log-sum-exp)
 *   mode 1 (warp) acc0 += sum_j relu(x_j - t + 1)                       (t = the target's logit)
 * first != 0 starts the scan.  arx_eval_warp_unmask then takes the terms of the masked columns --
 * the row's user's positives that have a logit (item2col >= 0), each distinct column once,
 * embed_attribute.py:729-741 -- out again (their logits are recomputed from U and the pool rows
 * P / pbias).  arx_eval_finish: batch_loss = acc0 + log(acc1) - t (ce) or log(1 + acc0) (warp). */
int arx_eval_chunk_accum(const float* logits, int64_t ldl, int64_t B, int64_t n, const float* tscore,
                         int mode, int first, float* acc0, float* acc1, void* stream);
int arx_eval_warp_unmask(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, int d,
                         const float* tscore, const int32_t* user_ids, const int32_t* pos_ptr,
                         const int32_t* pos_items, const int32_t* item2col, int64_t mask_rows, int64_t B,
                         int64_t V, float* s_acc, void* stream);
int arx_eval_finish(int mode, const float* acc0, const float* acc1, const float* tscore, int64_t B,
                    float* batch_loss, void* stream);

/* ---- a8-a11 fused: the 'mw' scorer of a training step (csrc/scorer.hip) ----------------------------
 * embed_attribute.py:148-206 get_prediction on the sampled pool + :208-220 get_target_score + :641-649 the
 * 'mw' loss, forward and backward, WITHOUT [B, S] logits or dlogits in HBM, all three products on the bf16
 * matrix pipe, f32-exact: every f32 operand is split exactly into three bf16 pieces, six of the nine piece
 * products (the other three are below 2^-26 of the product) are accumulated in f32 with the small terms in
 * their own accumulator; where one operand is the 0/1 hinge-activity matrix (exact in ONE piece) three do.
 *
 * arx_mw_scorer_fwd (three launches):
 *   t_r = U_r . T_r + tbias_r;   x = U . P^T + pbias  (not stored);
 *   act[r, s] = mask[r, s] and (x[r, s] - t_r + 1 > 0), one bit per logit, kept in `state`;
 *   loss_r = log(1 + sum_s act * (x - t + 1));  g_r = gscale * row_w_r / (1 + sum);  dt_r = -g_r * #act
 * with the mask built from the positives CSR as in arx_loss_mw_fwdbwd_pos (user of row r = user_ids[r %
 * mask_rows]).  Outputs: batch_loss [B], tscore_out [B] (nullable), dtscore (stride dtscore_stride; nullable),
 * dU = dt * T and dT = dt * U (rank-one terms of the target score, WRITTEN; nullable).  `state` keeps what the
 * backward products read: the act bits in both orientations, g, the bf16 planes of P (both layouts) and of
 * g * U, the bias-gradient partials.
 * arx_mw_scorer_bwd_du:  dU[r, :] = beta dU[r, :] + g_r sum_s act[r, s] P[s, :]
 * arx_mw_scorer_bwd_di:  dI[s, :] = beta dI[s, :] + sum_r act[r, s] g_r U[r, :],  db[s] = sum_r act[r, s] g_r;
 *   step_rows > 0 (the sequence model: B = L * step_rows time-major rows, step_rows % 128 == 0): the products of
 *   the single time steps are kept too -- dI_steps [L][S][d] (NULL: they live in the workspace only) and
 *   db_steps [L][S] -- because TF-1.0's clip_by_global_norm squares one dense gradient per unrolled step
 *   (seqModel.py:179-180).
 * Shapes: d in {64, 128}, S % 128 == 0, 128 <= S <= 2048, any B >= 1; U / P / T / dU / dT / dI rows 16-byte
 * aligned (ld % 4 == 0).  arx_mw_scorer_supported tells; callers take the unfused path (arx_gemm_f32 +
 * arx_loss_mw_fused_pos) otherwise.  `state`: arx_mw_scorer_state_bytes(B, S, d) bytes, 256-byte aligned,
 * ZEROED once by the caller, private to one (model, stream).  arx_mw_scorer_state_layout: byte offsets of the
 * regions a caller may inspect -- out[0] act bits, word-major: bit (s & 31) of word [(s >> 5) * out[1] + r];
 * out[2] the transposed bits: bit (r & 31) of word [(r >> 5) * out[5] + s]; out[3] g [out[4]]; out[4] = B rounded
 * up to 128 (out: 6 values). */
int arx_mw_scorer_supported(int64_t B, int64_t S, int d);
size_t arx_mw_scorer_state_bytes(int64_t B, int64_t S, int d);
int arx_mw_scorer_state_layout(int64_t B, int64_t S, int d, int64_t* out);
int arx_mw_scorer_fwd(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, const float* T,
                      int64_t ldt, const float* tbias, int64_t tb_stride, int d, const int32_t* user_ids,
                      const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                      float gscale, const float* row_w, int64_t B, int64_t S, float* batch_loss, float* tscore_out,
                      float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu, float* dT, int64_t lddt,
                      void* state, size_t state_bytes, void* stream);
/* the same in parts -- phases: bit 0 the pool planes, the positives' hit lists and (since round 5) the target scores
 * t_r = U_r . T_r + tb_r, written to the state and to tscore_out (needs U and T: no longer ahead of the lookups),
 * bit 1 the hinge GEMM (act bits, partial sums; reads the target scores bit 0 left), bit 2 the row kernel (loss, g,
 * rank-one terms, the g U planes); in this order on one stream.  arx_mw_scorer_fwd == phases 7. */
int arx_mw_scorer_fwd_phases(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias,
                             const float* T, int64_t ldt, const float* tbias, int64_t tb_stride, int d,
                             const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                             const int32_t* item2slot, int64_t mask_rows, float gscale, const float* row_w, int64_t B,
                             int64_t S, float* batch_loss, float* tscore_out, float* dtscore, int64_t dtscore_stride,
                             float* dU, int64_t lddu, float* dT, int64_t lddt, void* state, size_t state_bytes,
                             int phases, void* stream);
/* arx_mw_scorer_fwd_phases for the sequence model (seqModel.py:561-567): with seq_w [B] (time-major, B = L *
 * seq_rows) the row weights are formed on the way -- row_w[t * seq_rows + b] = seq_w[t * seq_rows + b] /
 * (sum_t seq_w[t * seq_rows + b] + 1e-12) is WRITTEN to row_w by the launch of phase bit 0 and read by the row
 * kernel (the arithmetic of arx_seq_weights).  seq_w == NULL: row_w is an input (nullable), as above. */
int arx_mw_scorer_fwd_seqw(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias,
                           const float* T, int64_t ldt, const float* tbias, int64_t tb_stride, int d,
                           const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                           const int32_t* item2slot, int64_t mask_rows, float gscale, float* row_w,
                           const float* seq_w, int64_t seq_rows, int64_t B, int64_t S, float* batch_loss,
                           float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU, int64_t lddu,
                           float* dT, int64_t lddt, void* state, size_t state_bytes, int phases, void* stream);
int arx_mw_scorer_bwd_du(int64_t B, int64_t S, int d, const void* state, float beta, float* dU, int64_t lddu,
                         void* stream);
size_t arx_mw_scorer_bwd_di_workspace_bytes(int64_t B, int64_t S, int d, int64_t step_rows);
int arx_mw_scorer_bwd_di(int64_t B, int64_t S, int d, const void* state, int64_t step_rows, float beta, float* dI,
                         int64_t lddi, float* db, float* dI_steps, float* db_steps, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same, and *loss_out = gscale * sum_r row_w[r] * batch_loss[r] over the B row losses the forward wrote (row_w
 * NULL: ones) -- hmf_model.py:140 reduce_mean with gscale = 1 / B; seqModel.py:571-604 with row_w the normalised
 * example weights -- added up in a fixed order by one more block of the reduce launch.  loss_out NULL:
 * arx_mw_scorer_bwd_di. */
int arx_mw_scorer_bwd_di_loss(int64_t B, int64_t S, int d, const void* state, int64_t step_rows, float beta,
                              float* dI, int64_t lddi, float* db, float* dI_steps, float* db_steps,
                              const float* batch_loss, float gscale, const float* row_w, float* loss_out,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Sampled softmax ('mce').  BUILD-DEFINED: the reference accepts loss 'mce'
 * (embed_attribute.py:527 assert, :717 feed guard, run_hmf.py:31,100, lstm/run.py:447) but its
 * compute_loss has no branch for it (:529-549).  Defined here in the shape of 'mw' (:641-649):
 *   loss_r = log(1 + sum_s m_rs * exp(x_rs - t_r))      (softmax cross-entropy over
 *            [target score || the sampled logits the positive mask keeps]; no log-Q term)
 *   dx_rs  = g_r * m_rs * exp(x_rs - t_r) / (1 + sum),  dt_r = -sum_s dx_rs.
 * Round 6: the exponent SATURATES, exp(min(x_rs - t_r, 64)) in the loss and in the weight: the plain sampled softmax
 * while no kept logit leads the target score by more than 64 (a softmax weight of 1 - 1e-28 there), finite beyond --
 * the same definition on every path (these entries, the fused arx_mce_scorer_* family, the oracle).
 * Same three forms, arguments and restrictions as the arx_loss_mw_* entries above. */
int arx_loss_mce_fwdbwd(const float* logits, int64_t ldl, const float* tscore,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, float gscale,
                       const float* row_w,
                       int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                       float* dtscore, void* stream);
int arx_loss_mce_fwdbwd_pos(const float* logits, int64_t ldl, const float* tscore,
                           const int32_t* user_ids, const int32_t* pos_ptr,
                           const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                           float gscale, const float* row_w, int64_t B, int64_t S,
                           float* batch_loss, float* dlogits, int64_t lddl, float* dtscore,
                           void* stream);
int arx_loss_mce_fused_pos(const float* logits, int64_t ldl, const float* U, int64_t ldu, const float* T,
                          int64_t ldt, const float* tbias, int64_t tbias_stride, int d,
                          const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                          const int32_t* item2slot, int64_t mask_rows, float gscale, const float* row_w,
                          int64_t B, int64_t S, float* batch_loss, float* dlogits, int64_t lddl,
                          float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU,
                          int64_t lddu, float* dT, int64_t lddt, void* stream);

/* 'mce' on the fused family (round 5; the reference's call sites: embed_attribute.py:527,717 accept the loss,
 * lstm/run.py:447 and run_hmf.py:31 select it): the arithmetic of arx_loss_mce_fused_pos WITHOUT [B, S] logits or
 * weights in HBM.  loss_r = log(1 + s_r), s_r = sum_s m_rs e_rs, e_rs = exp(x_rs - t_r) (anchored at the target
 * score, which is a member of the softmax set; no further max shift: e overflows where x_rs - t_r > 88.7, the
 * materialising entry point's max shift does not), backward weight w_rs = coef_r m_rs e_rs, coef_r = gscale row_w_r /
 * (1 + s_r), dt_r = -coef_r s_r.  The weight tile is recomputed where it is used and consumed out of the MFMA
 * accumulators (six-term f32-exact x, e split into three exact bf16 pieces, six-term products):
 *   arx_mce_scorer_fwd (phases bit 0: k_sc_prep as in arx_mw_scorer_fwd_seqw, whose walk of the positives chain
 *     leaves the masked pairs as one bit row per USER row -- mask_rows must be B (or 0) or a multiple of 128 that
 *     divides B; bit 1: one pass over U . P^T that leaves s_r AND O_r = sum_s m e_rs P_s; bit 2: the
 *     row kernel): batch_loss, tscore_out, dtscore = dt, dT = dt U, dU = coef O + dt T (WRITTEN: the latent-side
 *     product is complete, there is no bwd_du), and in `state` the planes of U and coef U, coef, -t.
 *   arx_mce_scorer_bwd_di_loss: dI[s, :] = beta dI[s, :] + sum_r w_rs U[r, :], db[s] = sum_r w_rs; pbias = the pool
 *     bias and mask_rows the mask_rows the forward was given; step_rows / dI_steps / db_steps / loss_out as in arx_mw_scorer_bwd_di_loss.
 * Shapes: d in {64, 128} (128: round 6), S % 128 == 0, 128 <= S <= 2048, B >= 1, rows 16-byte aligned; arx_mce_scorer_supported tells,
 * callers take arx_gemm_f32 + arx_loss_mce_fused_pos otherwise.  `state`: arx_mce_scorer_state_bytes bytes, 256-byte
 * aligned, zeroed once, private to one (model, stream). */
int arx_mce_scorer_supported(int64_t B, int64_t S, int d);
size_t arx_mce_scorer_state_bytes(int64_t B, int64_t S, int d);
int arx_mce_scorer_fwd(const float* U, int64_t ldu, const float* P, int64_t ldp, const float* pbias, const float* T,
                       int64_t ldt, const float* tbias, int64_t tb_stride, int d, const int32_t* user_ids,
                       const int32_t* pos_ptr, const int32_t* pos_items, const int32_t* item2slot, int64_t mask_rows,
                       float gscale, float* row_w, const float* seq_w, int64_t seq_rows, int64_t B, int64_t S,
                       float* batch_loss, float* tscore_out, float* dtscore, int64_t dtscore_stride, float* dU,
                       int64_t lddu, float* dT, int64_t lddt, void* state, size_t state_bytes, int phases,
                       void* stream);
size_t arx_mce_scorer_bwd_di_workspace_bytes(int64_t B, int64_t S, int d, int64_t step_rows);
int arx_mce_scorer_bwd_di_loss(int64_t B, int64_t S, int d, const void* state, const float* pbias, int64_t mask_rows,
                               int64_t step_rows, float beta, float* dI, int64_t lddi, float* db, float* dI_steps, float* db_steps,
                               const float* batch_loss, float gscale, const float* row_w, float* loss_out,
                               void* workspace, size_t workspace_bytes, void* stream);

int arx_loss_warp_fwdbwd_pos(const float* logits, int64_t ldl, const int32_t* target,
                             const int32_t* user_ids, const int32_t* pos_ptr,
                             const int32_t* pos_items, const int32_t* item2slot,
                             int64_t mask_rows, float gscale, const float* row_w, int64_t B,
                             int64_t V, float* batch_loss, float* dlogits, int64_t lddl,
                             void* stream);
int arx_loss_ce_fwdbwd(const float* logits, int64_t ldl, const int32_t* target, float gscale,
                       const float* row_w, int64_t B, int64_t V, float* batch_loss,
                       float* dlogits, int64_t lddl, void* stream);
/* out[r] = log(sum_c exp(logits[r, c])): the softmax normaliser of the recommend path
 * (seqModel.py:514-517 top_k(softmax(full_logits))): top-k of the logits + this gives the
 * softmax values of the top-k without materialising the softmax. */
int arx_row_logsumexp(const float* logits, int64_t ldl, int64_t B, int64_t V, float* out,
                      void* stream);

/* embed_attribute.py:620-639 warp_eval -> margin_rank[B] (float), true_rank[B] (int32) */
int arx_loss_warp_eval(const float* logits, int64_t ldl, const int32_t* target,
                       const uint8_t* mask, int64_t ldm, int64_t mask_rows, int64_t B, int64_t V,
                       float* margin_rank, int32_t* true_rank, void* stream);

/* ---- a17: embedding_lookup gradient scatter fused with sparse Adagrad (K7)
 * hmf_model.py:146-151 / seqModel.py:173-182 restricted to the rows that
 * receive gradient.  n contributions i: table row keys[i] (ARX_KEY_NONE =
 * skip) receives coef[i] * G[src[i], 0:d] (and coef[i]*Gb[src[i]] for the
 * bias table).  Duplicates are summed first (stable sort by key => fixed
 * summation order), then ONE Adagrad application per unique row:
 *   g *= *gscale_dev (nullable; clip_by_global_norm coefficient)
 *   acc[row] += g^2 ; E[row] -= *lr_dev * g / sqrt(acc[row])
 * src NULL => identity; coef NULL => 1.  bias/bias_acc/Gb nullable together.
 * acc == NULL (then bias_acc == NULL too): plain gradient descent, E[row] -= lr * g -- the
 * GradientDescentOptimizer branch of seqModel.py:175-176; the same holds for arx_sparse_adagrad_cat
 * (sorted pass), arx_sparse_adagrad_cat_multi (acc[t] == NULL) and arx_adagrad_dense.
 * key_bits: number of significant key bits (0 => 31) to shorten the sort. */
size_t arx_sparse_adagrad_workspace_bytes(int64_t n);
int arx_sparse_adagrad(float* E, float* acc, float* bias, float* bias_acc, int d,
                       const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                       const float* G, int64_t ldg, const float* Gb,
                       const float* lr_dev, const float* gscale_dev, int key_bits,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Same with a per-table-row ticket counter (aux_cnt: caller-owned int32[table rows],
 * zero-initialised once, left clean): long runs of duplicates are then completed by
 * their last-arriving piece inside the same launch instead of a second pass.
 * aux_cnt == NULL behaves exactly like arx_sparse_adagrad. */
int arx_sparse_adagrad_ticket(float* E, float* acc, float* bias, float* bias_acc, int d,
                              const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                              const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                              const float* gscale_dev, int key_bits, int32_t* aux_cnt,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Fast path of the above for one-hot lookups with few contributions (id-only batches:
 * n = sum(site_n) <= ~64 k).  The lookup sites are described directly -- site s
 * contributes, for j < site_n[s]: key = site_cat_map[s] ? site_cat_map[s][ids[j]] : ids[j],
 * gradient row G[site_row_base[s] + j] (and Gb[...]), factor site_coef[s] -- so no
 * separate key-generation launches and no sort: integer atomics elect the first
 * contribution of every row as its leader, which sums its duplicates in
 * contribution order (bit-deterministic) and applies Adagrad once.
 * Rows with more than 16 duplicates ("hot", e.g. Zipf-popular targets) are summed by
 * a whole workgroup each in a third launch.
 * aux_first / aux_cnt: caller-owned int32[table_rows], initialised to INT_MAX / 0
 * once; aux_hot: int32[aux_hot_len >= 3] zero-initialised once (hot-row list, sized
 * n/16 + 2); the kernels leave all three clean.  keys/src/coef_buf: [n].
 * mode 0 (default): for n <= 16384 the key generation is folded into a
 * single-workgroup LDS radix sort followed by the two Adagrad passes of
 * arx_sparse_adagrad (workspace >= arx_sparse_adagrad_workspace_bytes(n)) -- 3
 * launches, no atomics; larger n and mode 1 use the atomic leader election above. * mode 0: the sorted pass described above; mode 0x10 / 0x20: its first half (contributions + sort:
 * needs the ids only) / second half (apply) on their own -- same arguments, same untouched workspace. */
int arx_sparse_adagrad_cat(float* E, float* acc, float* bias, float* bias_acc, int64_t table_rows,
                           int d, int nsites, const int32_t* const* site_cat_map,
                           const int32_t* const* site_ids, const int64_t* site_n,
                           const int32_t* site_row_base, const float* site_coef,
                           const float* G, int64_t ldg, const float* Gb, const float* lr_dev,
                           const float* gscale_dev, int32_t* aux_first, int32_t* aux_cnt,
                           int32_t* aux_hot, int64_t aux_hot_len, int32_t* keys_buf,
                           int32_t* src_buf, float* coef_buf, int mode, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Several one-hot tables of equal width d in ONE pass (hmf_model.py:146-151 applies one
 * Adagrad op per variable; the per-table kernel chains are launch-bound, so the tables
 * share the key generation, the sort and the apply launches: the sort key carries the
 * table index above the row bits).  Table t: E[t], acc[t], bias[t]/bias_acc[t] (NULL: no
 * bias), table_rows[t], aux_cnt[t] (int32[table_rows[t]] zeros, for all tables or none).
 * Site s updates table site_table[s]; the other site arrays as in arx_sparse_adagrad_cat.
 * Multi-hot lookups join the same pass as pre-expanded segments: the caller runs
 * arx_csr_expand into keys_buf / src_buf / coef_buf BEHIND the one-hot contributions (segment
 * e occupies extra_n[e] entries starting at sum(site_n) + sum(extra_n[:e]), table-local token
 * keys, padded with ARX_KEY_NONE) and names the table of each segment in extra_table[e].
 * ntables <= 4, nsites <= 8, nextra <= 8, rows + table bits <= 30.
 * workspace >= arx_sparse_adagrad_workspace_bytes(sum(site_n) + sum(extra_n)). */
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
                                 size_t workspace_bytes, void* stream);
/* The same pass in two halves: phase 1 = key generation + sort (depends on the lookup ids only --
 * it can run on a side stream under the forward/backward GEMMs), phase 2 = apply (needs G);
 * phase 3 = both.  Both halves take the same arguments and the same, otherwise untouched,
 * workspace. */
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
                                       size_t workspace_bytes, void* stream);

/* arx_sparse_adagrad_cat_multi_phase with a MULTI-HOT table riding on the pass (HET layout,
 * attributes/comb_attribute.py:151-176: an item has an id feature AND a multi-hot attribute, both looked
 * up with the same item ids and fed by the same gradient rows -- embed_attribute.py:383-400).  The
 * lookups of one-hot table 0 (site_table == 0) are also the entity lookups of the bags of table
 * bag_E[bag_rows, d].  The bag index is given PER ROW OF TABLE 0 (the sorted keys of the pass are
 * rows): the bag of the entity that maps to row r is vals[starts[r] .. starts[r] + lens[r]); starts,
 * lens: int32[table_rows[0]].  The entity -> row map (site_cat_map, or the identity) must be one-to-one.
 *   phase 1: the one-hot sort, then -- from its sorted list, whose run heads of table 0 ARE the
 *            distinct entities -- the bags of the distinct entities, sorted by token;
 *   phase 2: the one-hot apply, which writes the merged, 1/len-scaled gradient row of every
 *            distinct entity as a side output, then the token runs over those rows -> ONE Adagrad
 *            update per touched token row (same result as arx_sparse_adagrad_bags on the same sites).
 * Saves the entity sort and the merge pass of arx_sparse_adagrad_bags.  No pre-expanded segments.
 * Each half splits again for callers that overlap the pass with other work (round 3):
 *   phase 5 = the one-hot keys + sort (+ the bag offsets), phase 6 = the token chain (needs 5),
 *   phase 7 = the one-hot apply with its side output (needs 5 and G), phase 8 = the token apply
 *   (needs 6 and 7) -- so the tail of the token chain can run under the one-hot apply.
 * Round 5 (ARX_K7_RIDER = win | split): with `split` the one-hot list gets run records in
 * sorted order as well and the two apply phases cut the work by data flow: phase 7 = the runs of table 0 alone
 * (Adagrad on its rows + the merged rows), phase 8 = ONE launch with the token runs over the merged rows and the runs
 * of the other one-hot tables.  `win` (the default) measured 1-3 us/step faster at C3.
 * bag_workspace >= arx_sparse_adagrad_bags_workspace_bytes(lookups of table 0, max_len, d). */
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
                                      void* bag_workspace, size_t bag_workspace_bytes, void* stream);


/* arx_sparse_adagrad_cat_multi_bags with the riding table's STATIC token order (csc.hip).  The bag index
 * (attributes/attribute.py: the feature CSR built once by _init_attributes, embed_attribute.py:265-318) never
 * changes, so the token-major order of its (token, entity) pairs is built ONCE per table by the caller:
 *   csc_qpos  int32[len(vals)]  CSR position p = starts[r] + j -> place q of the pair in the order sorted by
 *                               (token, p); -1 for tokens outside [0, bag_rows) and for positions past max_len;
 *   csc_qte   int32[2 * csc_nq] {token row, row of table 0} of every place;
 *   csc_flags uint8[F + F / 16], F = csc_nq rounded up to 256: one flag byte per place, then one coarse byte per
 *             16 places; ZERO on entry of phase 5 and zero again on return of phase 6 (1 and 3 hold both);
 *   csc_slot_of int32[table_rows[0]] scratch (row of the merged gradients of an entity of the step).
 * The step then neither expands nor sorts: phase 5 also marks the live pairs from the lookups of table 0, phase 6
 * sweeps the flags out in place order (= the token-sorted list, entries of a token in (entity, position) order --
 * for a standard CSR the order the stable sort produced: same bits) and extracts the run records -- 3 launches for
 * 8 (the marks belong to phase 5 because phase 7 reads csc_slot_of and only waits for phase 5).  Used when csc_qpos != NULL, d >= 32 and lookups * max_len > 8192; otherwise
 * (and with csc_qpos == NULL) identical to arx_sparse_adagrad_cat_multi_bags.
 * phase: the phase number, optionally | 0x100 (this pass applies in the `split` form) or | 0x200 (`win`), whatever
 * ARX_K7_RIDER says -- the same bits in every phase of a pass. */
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
                                          int32_t* csc_slot_of, int64_t csc_nq, void* stream);

/* Multi-hot lookups of ONE table (embed_attribute.py:397-406: embedding_lookup of the bag
 * tokens + unsorted_segment_sum / length; hmf_model.py:146-151 one Adagrad apply per variable),
 * in two merge stages so that an entity that occurs k times in a step (Zipf-popular target
 * items; a target that is also in the pool) costs its bag ONCE:
 *   1. the lookups of all sites (site s: ids site_ids[s][0..site_n[s]), gradient rows
 *      G[site_row_base[s] + j], factor site_coef[s]) are sorted by ENTITY id; every distinct
 *      entity e gets one merged row Gu = (1/lens[e]) * sum coef * G[row]  (bias gradients alike);
 *   2. only the distinct entities' bags (vals[starts[e] .. + lens[e])) are expanded, sorted by
 *      token row and summed per row from Gu (all coefficients 1) -> ONE Adagrad update per
 *      touched row (acc == NULL: gradient descent).
 * Deterministic (stable sorts, fixed-order sums, no float atomics).  n_entities = rows of
 * lens (ids outside [0, n_entities) and tokens outside [0, table_rows) are dropped); max_len >=
 * every bag length.  phase 1 = everything that depends on the ids only (both sorts), phase 2 =
 * merge + apply (needs G), 3 = both; same workspace for both halves, untouched in between,
 * >= arx_sparse_adagrad_bags_workspace_bytes(sum(site_n), max_len, d).  aux_cnt: int32[table_rows]
 * zeros or NULL (only used when sum(site_n) * max_len <= 8192). */
size_t arx_sparse_adagrad_bags_workspace_bytes(int64_t n_lookups, int max_len, int d);
int arx_sparse_adagrad_bags(int phase, float* E, float* acc, float* bias, float* bias_acc,
                            int64_t table_rows, int d, const int32_t* vals, const int32_t* starts,
                            const int32_t* lens, int64_t n_entities, int max_len, int nsites,
                            const int32_t* const* site_ids, const int64_t* site_n,
                            const int32_t* site_row_base, const float* site_coef, const float* G,
                            int64_t ldg, const float* Gb, const float* lr_dev, const float* gscale_dev,
                            int32_t* aux_cnt, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a8: output_feat 2 / 3 -- pooling over a bag in SCORE space ------------------------------
 * embed_attribute.py:194-200.  scores [B, >= offs[W]] are the per-token scores of the pool's
 * packed bag tokens (latent . E[tok] + b[tok]: arx_gemm_f32 over the gathered token rows); bag j
 * is the column range [offs[j], offs[j+1]) (arx_csr_expand's offsets).
 *   mode 2  out[r, j] = max over the bag                           (tf.segment_max, :195)
 *   mode 3  out[r, j] = M + log(1 + sum exp(score - M)), M = *gmax_dev = reduce_max over the WHOLE
 *           table's score matrix (:197-200; arx_max_argmax over chunks of the table)
 * Backward: dscores [B, cap] (columns >= offs[W] zeroed); mode 2 sends the gradient to the arg-max
 * entries (ties share equally, like tf.segment_max's gradient); mode 3 also writes, per row, the
 * part of the gradient that flows through M (resid_rows[r] = sum_j dout / (1 + s)): its total goes
 * to the arg-max element of the table's score matrix (arx_gmax_residual_bwd). */
int arx_segment_pool_fwd(const float* scores, int64_t lds, const int32_t* offs, int64_t B, int64_t W,
                         int mode, const float* gmax_dev, float* out, int64_t ldo, void* stream);
int arx_segment_pool_bwd(const float* scores, int64_t lds, const int32_t* offs, int64_t B, int64_t W,
                         int64_t cap, int mode, const float* gmax_dev, const float* out, int64_t ldo,
                         const float* dout, int64_t ldd, float* dscores, int64_t ldds, float* resid_rows,
                         void* stream);
/* Running maximum and its FIRST arg-max (smallest (global column, row)) over x[rows, cols] whose
 * columns are col_base.. of a wider matrix: best[0] = value, best_idx = {row, global column};
 * first != 0 starts a new scan, otherwise the previous best takes part.  Deterministic. */
int arx_max_argmax(const float* x, int64_t rows, int64_t cols, int64_t ld, int64_t col_base, int first,
                   float* best, int32_t* best_idx, void* scratch /* reduce scratch, see arx_reduce_scratch_bytes */,
                   void* stream);
/* row_grad[0..d) = resid * U[idx[0], :], bias_grad[0] = resid, dU[idx[0], :] += resid * E_row
 * (E_row = the table row idx[1], already gathered); bias_grad / dU nullable. */
int arx_gmax_residual_bwd(const float* resid_dev, const int32_t* idx_dev, const float* U, int64_t ldu,
                          const float* E_row, int d, float* row_grad, float* bias_grad, float* dU,
                          int64_t lddu, void* stream);
/* output_feat 3 under a per-time-step scorer (lstm/seqModel.py:492 -> embed_attribute.py:197-200: one reduce_max
 * per unrolled step): step t's residual is the rank-one row RG[t] = resid_t * U[r*_t] on table row vrows[t] (+ RGb[t]
 * on its bias cell) -- in TF part of that step's DENSE matmul gradient, so clip_by_global_norm (seqModel.py:180) squares
 * it together with the merged token contributions of the same row.  corr[t] = what adding it changes of the squared
 * norm arx_merged_sq_norm computed without it:
 *   per_step != 0 (the steps' gradients X[t] stay apart: an IndexedSlices contribution to the variable exists)
 *       2 <M_t[v_t], RG_t> + |RG_t|^2,  M_t[v] = sum_{k: keys[k] == v} coef[k] X[t][src[k]]
 *   per_step == 0 (one summed gradient X): rows naming the same table row add up first, R_v = sum_{t: v_t = v} RG_t;
 *       corr[t] = 2 <M[v_t], R_v> + |R_v|^2 for the FIRST t naming v, 0 for the others
 * and the d = 1 analogue on (Xb, RGb) with its own per_step flag.  The caller adds sum_t corr[t] (fixed order) to the
 * squared norm.  keys / src / coef: the pool's token list as given to arx_merged_sq_norm. */
int arx_gmax_norm_corr(const int32_t* keys, const int32_t* src, const float* coef, int64_t n, const float* X,
                       int64_t ldx, int d, int per_step, int64_t step_stride, const float* Xb, int per_step_b,
                       int64_t stepb_stride, const int32_t* vrows, const float* RG, int64_t ldrg, const float* RGb,
                       int L, float* corr, void* stream);

/* ---- a16/a19: dense Adagrad, norms, clip ---------------------------------
 * tf.train.AdagradOptimizer dense apply; tf.clip_by_global_norm
 * (seqModel.py:180): coef = max_norm / max(sqrt(sq), max_norm). */
int arx_adagrad_dense(float* w, float* acc, const float* g, int64_t n, const float* lr_dev,
                      const float* gscale_dev, void* stream);
/* Adagrad over the rows of W [rows, d] whose DENSE gradient row in G is not all zero, with arx_adagrad_dense's
 * arithmetic (a row of zeros would not move: acc += 0, w -= 0 -- the sparse update of hmf_model.py:146-151 on a dense
 * gradient table), and every consumed gradient row / bias-gradient cell is ZEROED: G and Gb are left all zero for the
 * next accumulation.  The replicated token table of the sharded HET step (arx.dist.ShardedHMFRepTokens: G = the merged
 * token gradients summed over the ranks).  acc == NULL: gradient descent.  bias / bias_acc / Gb nullable. */
int arx_adagrad_rows_nonzero(float* W, float* acc, float* bias, float* bias_acc, float* G, float* Gb, int64_t rows,
                             int d, const float* lr_dev, void* stream);
/* The same update for up to 8 dense parameters in one launch (the LSTM weights, biases and input
 * projections of a step: seqModel.py:173-182 applies one op per variable).  acc[t] NULL: gradient descent. */
int arx_adagrad_dense_multi(int count, float* const* w, float* const* acc, const float* const* g,
                            const int64_t* n, const float* lr_dev, const float* gscale_dev, void* stream);
/* *out_accum += sum_i w_i * x_i^2 with w_i = row_scale ? row_scale[i / d] : 1.
 * Re-entrant (round 4): the deterministic one-launch reductions (arx_sq_norm_accum[_multi],
 * arx_sq_norm_clip_multi, arx_merged_sq_norm, arx_max_argmax) keep their block partials and the arrival
 * ticket in a CALLER-provided `scratch` of arx_reduce_scratch_bytes() bytes: zero it once after allocation,
 * every call leaves it zeroed; launches that may overlap on the GPU (different streams, different models in
 * one process) take different scratch buffers, launches on one stream may share one.  NULL is an error. */
size_t arx_reduce_scratch_bytes(void);
int arx_sq_norm_accum(const float* x, int64_t n, int d, const float* row_scale,
                      float* out_accum, void* scratch, void* stream);
/* the same accumulation over up to 8 tensors in one launch (the global norm of an LSTM step) */
int arx_sq_norm_accum_multi(int count, const float* const* x, const int64_t* n, const int* d,
                            const float* const* row_scale, float* out_accum, void* scratch, void* stream);
/* arx_sq_norm_accum_multi and arx_clip_coef in ONE launch: the last-arriving workgroup, which already
 * combines the partial sums, also forms coef = max_norm / max(||g||, max_norm) and ||g||
 * (seqModel.py:179-180 clip_by_global_norm).  init != 0: *sqnorm_out = sum (no prior fill);
 * init == 0: accumulated onto *sqnorm_out first (norms of earlier launches). */
int arx_sq_norm_clip_multi(int count, const float* const* x, const int64_t* n, const int* d,
                           const float* const* row_scale, int init, float* sqnorm_out, float max_norm,
                           float* coef_out, float* gnorm_out, void* scratch, void* stream);
int arx_clip_coef(const float* sqnorm_dev, float max_norm, float* coef_out, float* gnorm_out,
                  void* stream);
/* Norm of a table gradient AFTER summing the contributions that land on the same table row
 * (the dense gradient tf.gradients produces for  innerp = E . u^T  gathered by pool item,
 * embed_attribute.py:171-172,188-193, as clip_by_global_norm sees it, seqModel.py:180):
 *   *out_accum += sum_{t<L}  sum_rows || sum_{c: keys[c]==row} coef[c] * X[t*step_stride + src[c]*ldx + 0..d) ||^2
 *              +  sum_{t<Lb} sum_rows (  sum_{c: keys[c]==row} coef[c] * Xb[t*stepb_stride + src[c]] )^2
 * keys/src/coef as for arx_sparse_adagrad (ARX_KEY_NONE = padding); X or Xb may be NULL.
 * Workspace: arx_sparse_adagrad_workspace_bytes(n). */
int arx_merged_sq_norm(const int32_t* keys, const int32_t* src, const float* coef, int64_t n,
                       int key_bits, const float* X, int64_t ldx, int d, int L, int64_t step_stride,
                       const float* Xb, int Lb, int64_t stepb_stride, float* out_accum,
                       void* workspace, size_t workspace_bytes, void* scratch, void* stream);

/* ---- small device utilities ------------------------------------------------ */
int arx_fill_f32(float* p, int64_t n, float v, void* stream);
int arx_fill_i32(int32_t* p, int64_t n, int32_t v, void* stream);
/* out[i] = idx[i] >= 0 ? table[idx[i]] : fill -- drawn positions -> item ids (utils/prepare_train.py:7-17: the
 * population list the sampler draws from need not be 0..n-1); unfilled positions of a short draw stay `fill`. */
int arx_take_i32(const int32_t* table, const int32_t* idx, int64_t n, int32_t fill, int32_t* out, void* stream);
int arx_fill_u8(uint8_t* p, int64_t n, int v, void* stream);
/* y = a*x + b*y  (n elements) */
int arx_axpby(float a, const float* x, float b, float* y, int64_t n, void* stream);
/* y[r,0:d] = a*x[r % xrows, 0:d] + b*y[r,0:d]  (row broadcast, rows = n) */
int arx_add_rows_bcast(float a, const float* x, int64_t ldx, int64_t xrows, float b, float* y,
                       int64_t ldy, int64_t rows, int d, void* stream);
/* out[r] = sum_c x[r*ld + c]  (rows x cols), out (+)= when accumulate */
int arx_row_sum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                int accumulate, void* stream);
/* out[c] = sum_r x[r*ld + c]; deterministic two-stage, partials in the workspace */
size_t arx_col_sum_workspace_bytes(int64_t rows, int64_t cols);
int arx_col_sum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                void* workspace, size_t workspace_bytes, void* stream);
/* mean over rows: *out = scale * sum_i x[i] */
int arx_sum_scaled(const float* x, int64_t n, float scale, float* out, void* stream);
/* tf.nn.dropout (embed_attribute.py:236): y = x * keep_mask / keep_prob, counter RNG */
int arx_dropout_fwd(const float* x, int64_t n, float keep_prob, uint64_t seed, float* y,
                    uint8_t* keep_mask, void* stream);
int arx_dropout_bwd(const float* dy, const uint8_t* keep_mask, int64_t n, float keep_prob,
                    float* dx, void* stream);
/* Same draw with a DEVICE-side step counter mixed into the seed: a launch captured in a hipGraph
 * yields a new mask on every replay once the graph also bumps the counter (arx_counter_add). */
int arx_dropout_fwd_step(const float* x, int64_t n, float keep_prob, uint64_t seed,
                         const uint64_t* step_dev, float* y, uint8_t* keep_mask, void* stream);
int arx_counter_add(uint64_t* counter_dev, uint64_t v, void* stream);
/* up to 8 device-to-device copies of 4-byte words in one launch (placeholder feeds,
 * embed_attribute.py:697-719 add_input: users, items, targets, ... of one step) */
int arx_copy_words(int count, const void* const* src, void* const* dst, const int64_t* n_words,
                   void* stream);
/* elementwise activation for the optional MLP (hmf_model.py:80-94): kind 0=relu 1=tanh */
int arx_act_fwd(const float* x, int64_t n, int kind, float* y, void* stream);
int arx_act_bwd(const float* y, const float* dy, int64_t n, int kind, float* dx, void* stream);
/* y[r,c] += b[c] */
int arx_add_col_bias(float* y, int64_t ld, int64_t rows, int64_t cols, const float* b,
                     void* stream);

/* ---- a16 (next): top-k over logits rows ------------------------------------
 * hmf_model.py:154 tf.nn.top_k(logits, k, sorted=True): descending values,
 * ties broken by lower index.  k <= 1024. */
int arx_topk(const float* logits, int64_t ld, int64_t B, int64_t V, int k, float* values,
             int32_t* indices, void* stream);
/* Streaming full-vocabulary recommend: the [B, V] logits are never materialised --
 * the scorer GEMM runs over a chunk of the vocabulary, arx_topk_chunk keeps the chunk's k best
 * per row (indices offset by idx_base = first column of the chunk), arx_topk_merge folds them
 * into the running result (both lists sorted: descending value, ascending index; on equal
 * values list A -- the earlier chunks, lower indices -- wins, which is tf.nn.top_k's tie rule). */
int arx_topk_chunk(const float* logits, int64_t ld, int64_t B, int64_t V, int k, int32_t idx_base,
                   float* values, int32_t* indices, void* stream);
int arx_topk_merge(const float* va, const int32_t* ia, const float* vb, const int32_t* ib, int64_t B,
                   int ka, int kb, int k, float* vo, int32_t* io, void* stream);

/* The fused form of that streaming scorer (round 5): after the first chunk gave every row its k best (arx_topk_chunk),
 * the scorer GEMM over the REST of the vocabulary writes no logits -- arx_gemm_nt_topk_filter keeps, per row, only the
 * logits above thr[row * ldthr] (the row's k-th best so far; strictly above: an equal one further right loses the tie)
 * as (value, global column = col_base + column) in cand_v / cand_i [M, ldcand]: the kernel splits the columns into
 * `parts` ranges (arx_gemm_nt_topk_parts; one workgroup per 128 rows and range) and range p fills
 * cand[row][p * capp ..) in ascending column order, deterministically.  Pre-fill cand_v with -inf: then
 * arx_topk_chunk over the candidate rows (positions are in column order, so its tie rule holds), arx_take_rows_i32
 * (positions -> columns) and arx_topk_merge finish the top-k.  *overflow is set when a range's segment was too short
 * (results incomplete: fall back to the chunked path).  K in {32, 64, 128}; values bit-identical to arx_gemm_f32's.
 * lse_part (nullable) [M, ldl >= parts]: lse_part[row][p] = log sum exp of the row's logits over column range p -- with
 * the first chunk's row log-sum-exp this is the softmax normaliser of seqModel.py:514-517 top_k(softmax(logits)). */
int arx_gemm_nt_topk_parts(int64_t M, int64_t N, int* parts);
int arx_gemm_nt_topk_filter(const float* A, int64_t lda, int64_t M, const float* Bm, int64_t ldb, int64_t N, int64_t K,
                            const float* col_bias, const float* thr, int64_t ldthr, int32_t col_base, float* cand_v,
                            int32_t* cand_i, int64_t ldcand, int capp, int* overflow, float* lse_part, int64_t ldl,
                            void* stream);
/* The evaluation losses over the full vocabulary (hmf_model.py:130,144; seqModel.py:510: the full-softmax / full-WMRB
 * loss a sampled-loss model is selected on) in ONE pass of the same GEMM, no logits: per row and column range p
 * (arx_gemm_nt_topk_parts) lse_part[row][p] = log sum exp of the logits ('ce': loss = logsumexp_p(lse_part) - t) and /
 * or relu_part[row][p] = sum relu(logit - tscore[row] + 1) ('warp', embed_attribute.py:605-618: loss = log(1 + sum_p
 * relu_part), positives taken out by arx_eval_warp_unmask as for the chunked form).  Either output may be NULL. */
int arx_gemm_nt_eval_parts(const float* A, int64_t lda, int64_t M, const float* Bm, int64_t ldb, int64_t N, int64_t K,
                           const float* col_bias, const float* tscore, float* lse_part, float* relu_part, int64_t ldl,
                           void* stream);
/* out[r][j] = table[r * ld + pos[r * ldp + j]], r < B, j < k */
int arx_take_rows_i32(const int32_t* table, int64_t ld, const int32_t* pos, int64_t ldp, int64_t B, int k,
                      int32_t* out, int64_t ldo, void* stream);

/* ---- a19-a20: LSTM encoder (K9) ---------------------------------------------
 * lstm/seqModel.py:99-103,477 -- tf.contrib.rnn LSTMCell(h), no peepholes,
 * forget_bias=1, gate order i,j,f,o, zero initial state, static_rnn over L
 * steps.  x: [L,B,din] time-major; W: [(din+h), 4h]; b: [4h].
 * hs: [L,B,h] outputs; gates: [L,B,4h] post-activation (i, j=tanh, f, o) and
 * cs: [L,B,h] cell states are saved for backward. */
int arx_lstm_fwd(const float* x, const float* W, const float* b, int64_t L, int64_t B,
                 int din, int h, float forget_bias, float* hs, float* cs, float* gates,
                 void* stream);
/* dhs: [L,B,h] upstream gradient of every output.  Produces dz [L,B,4h]
 * (pre-activation gate gradients, BPTT through h and c).  dx, dW and db are
 * then plain GEMMs / a column sum over dz that the caller issues
 * (arx_gemm_f32 with W_x^T, [x;h_prev]^T . dz, arx_col_sum). */
int arx_lstm_bwd(const float* W, const float* hs, const float* cs, const float* gates,
                 const float* dhs, int64_t L, int64_t B, int din, int h, float* dz,
                 void* stream);
/* The same, and wxt [4h, din] = W[0:din, :]^T (the B operand of dx = dz . W_x^T) written on the way
 * (wxt may be NULL). */
int arx_lstm_bwd_wxt(const float* W, const float* hs, const float* cs, const float* gates,
                     const float* dhs, int64_t L, int64_t B, int din, int h, float* dz, float* wxt,
                     void* stream);

/* *out = scale * sum_i x[i]*y[i] (single workgroup, fixed order): the weighted
 * sum of sequence_loss (seqModel.py:561-563,596). */
int arx_dot_scaled(const float* x, const float* y, int64_t n, float scale, float* out,
                   void* stream);
/* out[i] = c / lens[ids ? ids[i] : i] : per-bag factor of tf.div's gradient, used to
 * weight squared norms of multi-hot lookup gradients (seqModel.py:180). */
int arx_inv_len_scale(const int32_t* lens, const int32_t* ids, int64_t n, float c, float* out,
                      void* stream);
/* lstm/seqModel.py:551-567 sequence_loss_by_example: per-row loss weights
 * out[t,b] = w[t,b] / (sum_t w[t,b] + 1e-12)  (w time-major [L,B]). */
int arx_seq_weights(const float* w, int64_t L, int64_t B, float* out, void* stream);

/* ---- HIP-graph capture of a whole step ---------------------------------------
 * The per-step kernel sequence has static shapes and pointers, so the host
 * captures it once and replays it (MI355X launch-bound regime at B=64). */
int arx_capture_begin(void* stream);
int arx_capture_end(void* stream, void** graph_exec_out);
int arx_graph_launch(void* graph_exec, void* stream);
/* Synchronises the DEVICE first (as does arx_graph_feeds_destroy): on ROCm 7.0 an executable graph destroyed while launches
 * of another executable graph are in flight makes that graph's next launch segfault inside the runtime -- and a host
 * language's collector may destroy a plan at any time.  Rare (a plan dies with its model); one pipeline bubble. */
int arx_graph_destroy(void* graph_exec);
/* Placeholder feeds as nodes of the captured step: arx_copy_words launches issued INSIDE the capture become
 * graph nodes whose (source, destination, length) triples are replaced before a replay -- one submission per
 * step instead of an eager copy + the graph.  arx_capture_end_feeds: as arx_capture_end, and `feeds` = a handle
 * on the n_feed_nodes copy nodes found (the captured graph stays alive with it); arx_graph_feed_dst0:
 * the first destination a node was captured with (to tell the nodes apart); arx_graph_set_feed: what node idx
 * copies at the following launches (count 0: nothing). */
int arx_capture_end_feeds(void* stream, void** graph_exec_out, void** feeds_out, int* n_feed_nodes);
int arx_graph_feed_dst0(void* feeds, int idx, void** dst0);
int arx_graph_set_feed(void* graph_exec, void* feeds, int idx, int count, const void* const* src, void* const* dst,
                       const int64_t* n_words);
int arx_graph_feeds_destroy(void* feeds);

#ifdef __cplusplus
}
#endif
#endif /* ARX_H_ */
