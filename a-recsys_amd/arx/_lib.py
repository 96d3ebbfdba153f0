"""ctypes binding of libarx.so (include/arx.h).

The library is the product: if it is missing or a symbol is absent this module
raises -- there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64; load it FIRST so libarx.so binds to the same
# HIP runtime instance (two runtimes in one process => "no ROCm-capable device").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARX_LIB") or os.path.join(_HERE, "lib", "libarx.so")   # ARX_LIB: A/B builds (tools/)

i32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
f32p = C.c_void_p
u8p = C.c_void_p
vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int32
f32 = C.c_float
sz = C.c_size_t
u64 = C.c_uint64
cint = C.c_int

# name -> (restype, argtypes); mirrors include/arx.h one-to-one
PROTOTYPES = {
    "arx_last_error": (C.c_char_p, []),
    "arx_version": (cint, []),
    "arx_device_info": (cint, [C.POINTER(cint), C.POINTER(cint), C.POINTER(cint), C.c_char_p, cint]),
    "arx_csr_expand_workspace_bytes": (sz, [i64]),
    "arx_csr_expand": (cint, [i32p, i32p, i32p, i32p, i64, i32p, i32p, i64, i32p, i32p, i32, i32,
                              i32, f32, f32p, vp, sz, vp]),
    "arx_bag_expand_padded": (cint, [i32p, i32p, i32p, i32p, i64, cint, i32, i32, f32, i32p, i32p, f32p, vp]),
    "arx_sparse_site_onehot": (cint, [i32p, i32p, i64, i32, f32, i32p, i32p, f32p, vp]),
    "arx_shard_route": (cint, [i32p, i64, cint, cint, i32, i32p, i32p, vp]),
    "arx_pool_blocks": (cint, [i32p, i64, cint, cint, i32, i64, i32p, i32p, i32p, i32p, vp]),
    "arx_copy_2d": (cint, [f32p, i64, f32p, i64, i64, i64, vp]),
    "arx_transpose_f32": (cint, [f32p, i64, i64, i64, f32p, i64, vp]),
    "arx_gather_rows_wide": (cint, [f32p, i64, i64, i32p, i64, i64, f32p, i64, vp]),
    "arx_gather_onehot_fwd": (cint, [f32p, f32p, i32p, i32p, i64, cint, f32, cint, f32p, i64, f32p, vp]),
    "arx_copy_strided_f32": (cint, [f32p, i64, f32p, i64, i64, vp]),
    "arx_gather_onehot_packed_fwd": (cint, [f32p, f32p, i32p, i32p, i64, cint, f32, f32p, i64, vp]),
    "arx_gather_id_plus_bag": (cint, [f32p, f32p, i32p, f32p, f32p, i32p, i32p, i32p, i32p, i64, cint, f32,
                                      cint, f32p, i64, f32p, vp]),
    "arx_lookup_multi": (cint, [cint] + [C.POINTER(vp)] * 9 + [C.POINTER(i64), cint, C.POINTER(f32), C.POINTER(vp),
                                C.POINTER(i64), C.POINTER(vp), vp]),
    "arx_gather_onehot_multi": (cint, [cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                       C.POINTER(i64), cint, C.POINTER(f32), C.POINTER(vp), C.POINTER(i64),
                                       C.POINTER(vp), vp]),
    "arx_gather_onehot_multi_ld": (cint, [cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(vp),
                                          C.POINTER(vp), C.POINTER(i64), cint, C.POINTER(f32), C.POINTER(vp), C.POINTER(i64),
                                          C.POINTER(vp), C.POINTER(i64), vp]),
    "arx_gather_mulhot_mean_fwd": (cint, [f32p, f32p, i32p, i32p, i32p, i32p, i64, cint, f32, cint,
                                          f32p, i64, f32p, vp]),
    "arx_dot_score_fwd": (cint, [f32p, i64, f32p, i64, f32p, i64, cint, f32p, vp]),
    "arx_dot_score_bwd": (cint, [f32p, i64, f32p, i64, f32p, i64, cint, f32p, i64, cint, f32p, i64, vp]),
    "arx_gemm_f32_workspace_bytes": (sz, [i64, i64, i64]),
    "arx_gemm_nt_bx6_workspace_bytes": (sz, [i64, i64]),
    "arx_gemm_nt_bx6": (cint, [i64, i64, i64, f32p, i64, f32p, i64, f32p, f32p, i64, vp, sz, vp]),
    "arx_gemm_f32": (cint, [cint, cint, i64, i64, i64, f32, f32p, i64, f32p, i64, f32, f32p, i64,
                            f32p, vp, sz, vp]),
    "arx_gemm_f32_rowsum": (cint, [cint, cint, i64, i64, i64, f32, f32p, i64, f32p, i64, f32, f32p,
                                   i64, f32p, f32p, vp, sz, vp]),
    "arx_gemm_f32_tn_pair_workspace_bytes": (sz, [i64, i64, i64]),
    "arx_gemm_f32_tn_pair": (cint, [i64, i64, i64, i64, f32p, i64, f32p, i64, f32p, i64, i64, f32p, i64, f32p,
                                    vp, sz, vp]),
    "arx_gemm_f32_steps_tn": (cint, [i64, i64, i64, i64, f32p, i64, f32p, i64, f32p, f32p, f32, f32p,
                                     i64, f32p, vp]),
    "arx_dot_scaled": (cint, [f32p, f32p, i64, f32, f32p, vp]),
    "arx_inv_len_scale": (cint, [i32p, i32p, i64, f32, f32p, vp]),
    "arx_pos_mask_scatter": (cint, [i32p, i64, i32p, i32p, i32p, u8p, i64, cint, vp]),
    "arx_slot_map_set": (cint, [i32p, i32p, i64, cint, vp]),
    "arx_slot_map_attach_bitmap": (cint, [i32p, i32p]),
    "arx_loss_mw_fwdbwd": (cint, [f32p, i64, f32p, u8p, i64, i64, f32, f32p, i64, i64, f32p, f32p,
                                  i64, f32p, vp]),
    "arx_loss_mce_fwdbwd": (cint, [f32p, i64, f32p, u8p, i64, i64, f32, f32p, i64, i64, f32p, f32p,
                                  i64, f32p, vp]),
    "arx_loss_warp_fwdbwd": (cint, [f32p, i64, i32p, u8p, i64, i64, f32, f32p, i64, i64, f32p, f32p,
                                    i64, vp]),
    "arx_loss_mw_fwdbwd_pos": (cint, [f32p, i64, f32p, i32p, i32p, i32p, i32p, i64, f32, f32p, i64,
                                      i64, f32p, f32p, i64, f32p, vp]),
    "arx_loss_mce_fwdbwd_pos": (cint, [f32p, i64, f32p, i32p, i32p, i32p, i32p, i64, f32, f32p, i64,
                                      i64, f32p, f32p, i64, f32p, vp]),
    "arx_loss_mw_fused_pos": (cint, [f32p, i64, f32p, i64, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                     i64, f32, f32p, i64, i64, f32p, f32p, i64, f32p, f32p, i64, f32p, i64,
                                     f32p, i64, vp]),
    "arx_loss_mce_fused_pos": (cint, [f32p, i64, f32p, i64, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                     i64, f32, f32p, i64, i64, f32p, f32p, i64, f32p, f32p, i64, f32p, i64,
                                     f32p, i64, vp]),
    "arx_loss_warp_fwdbwd_pos": (cint, [f32p, i64, i32p, i32p, i32p, i32p, i32p, i64, f32, f32p,
                                        i64, i64, f32p, f32p, i64, vp]),
    "arx_item_frequency": (cint, [i32p, i64, i64, i64, f32, i32p, f32p, vp]),
    "arx_eval_chunk_accum": (cint, [f32p, i64, i64, i64, f32p, cint, cint, f32p, f32p, vp]),
    "arx_eval_warp_unmask": (cint, [f32p, i64, f32p, i64, f32p, cint, f32p, i32p, i32p, i32p, i32p, i64, i64,
                                    i64, f32p, vp]),
    "arx_eval_finish": (cint, [cint, f32p, f32p, f32p, i64, f32p, vp]),
    "arx_mw_scorer_supported": (cint, [i64, i64, cint]),
    "arx_merge_keyed_take": (cint, [f32p, i32p, i64, i64, i32p, vp]),
    "arx_adagrad_rows_nonzero": (cint, [f32p, f32p, f32p, f32p, f32p, f32p, i64, cint, f32p, vp]),
    "arx_mw_scorer_state_bytes": (sz, [i64, i64, cint]),
    "arx_mw_scorer_state_layout": (cint, [i64, i64, cint, C.POINTER(i64)]),
    "arx_mw_scorer_fwd": (cint, [f32p, i64, f32p, i64, f32p, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                 i64, f32, f32p, i64, i64, f32p, f32p, f32p, i64, f32p, i64, f32p, i64, vp, sz, vp]),
    "arx_mw_scorer_fwd_phases": (cint, [f32p, i64, f32p, i64, f32p, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                        i64, f32, f32p, i64, i64, f32p, f32p, f32p, i64, f32p, i64, f32p, i64, vp, sz,
                                        cint, vp]),
    "arx_mw_scorer_fwd_seqw": (cint, [f32p, i64, f32p, i64, f32p, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                      i64, f32, f32p, f32p, i64, i64, i64, f32p, f32p, f32p, i64, f32p, i64, f32p, i64,
                                      vp, sz, cint, vp]),
    "arx_mw_scorer_bwd_di_loss": (cint, [i64, i64, cint, vp, i64, f32, f32p, i64, f32p, f32p, f32p, f32p, f32, f32p,
                                         f32p, vp, sz, vp]),
    "arx_mw_scorer_bwd_du": (cint, [i64, i64, cint, vp, f32, f32p, i64, vp]),
    "arx_mw_scorer_bwd_di_workspace_bytes": (sz, [i64, i64, cint, i64]),
    "arx_mw_scorer_bwd_di": (cint, [i64, i64, cint, vp, i64, f32, f32p, i64, f32p, f32p, f32p, vp, sz, vp]),
    "arx_gemm_bt_bx6_supported": (cint, [i64, i64, i64]),
    "arx_gemm_bt_bx6": (cint, [i64, i64, i64, f32p, i64, f32p, i64, f32, f32p, i64, vp]),
    "arx_mce_scorer_supported": (cint, [i64, i64, cint]),
    "arx_mce_scorer_state_bytes": (sz, [i64, i64, cint]),
    "arx_mce_scorer_fwd": (cint, [f32p, i64, f32p, i64, f32p, f32p, i64, f32p, i64, cint, i32p, i32p, i32p, i32p,
                                  i64, f32, f32p, f32p, i64, i64, i64, f32p, f32p, f32p, i64, f32p, i64, f32p, i64,
                                  vp, sz, cint, vp]),
    "arx_mce_scorer_bwd_di_workspace_bytes": (sz, [i64, i64, cint, i64]),
    "arx_mce_scorer_bwd_di_loss": (cint, [i64, i64, cint, vp, f32p, i64, i64, f32, f32p, i64, f32p, f32p, f32p, f32p, f32,
                                          f32p, f32p, vp, sz, vp]),
    "arx_sample_wor_workspace_bytes": (sz, [i64]),
    "arx_sample_wor_keys_workspace_bytes": (sz, [i64, i64, f32]),
    "arx_sample_wor": (cint, [f32p, i64, i64, u64, u64, i32p, vp, sz, vp]),
    "arx_sample_wor_capped": (cint, [f32p, i64, i64, u64, u64, f32, i32p, vp, sz, vp]),
    "arx_sample_wor_keys": (cint, [f32p, i64, i64, u64, u64, f32, i32p, f32p, vp, sz, vp]),
    "arx_loss_rs_fwdbwd": (cint, [f32p, i64, i32p, u8p, i64, i32p, i32p, i32p, i32p, i64, cint, cint,
                                  f32, f32, f32p, i64, i64, f32p, f32p, i64, vp]),
    "arx_loss_ce_fwdbwd": (cint, [f32p, i64, i32p, f32, f32p, i64, i64, f32p, f32p, i64, vp]),
    "arx_row_logsumexp": (cint, [f32p, i64, i64, i64, f32p, vp]),
    "arx_loss_warp_eval": (cint, [f32p, i64, i32p, u8p, i64, i64, i64, i64, f32p, i32p, vp]),
    "arx_sparse_adagrad_workspace_bytes": (sz, [i64]),
    "arx_sparse_adagrad": (cint, [f32p, f32p, f32p, f32p, cint, i32p, i32p, f32p, i64, f32p, i64,
                                  f32p, f32p, f32p, cint, vp, sz, vp]),
    "arx_sparse_adagrad_ticket": (cint, [f32p, f32p, f32p, f32p, cint, i32p, i32p, f32p, i64, f32p,
                                         i64, f32p, f32p, f32p, cint, i32p, vp, sz, vp]),
    "arx_sparse_adagrad_cat": (cint, [f32p, f32p, f32p, f32p, i64, cint, cint, C.POINTER(vp),
                                      C.POINTER(vp), C.POINTER(i64), C.POINTER(i32),
                                      C.POINTER(f32), f32p, i64, f32p, f32p, f32p, i32p, i32p,
                                      i32p, i64, i32p, i32p, f32p, cint, vp, sz, vp]),
    "arx_sparse_adagrad_cat_multi": (cint, [cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                            C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), cint, cint,
                                            C.POINTER(i32), C.POINTER(vp), C.POINTER(vp),
                                            C.POINTER(i64), C.POINTER(i32), C.POINTER(f32), f32p, i64,
                                            f32p, f32p, f32p, i32p, i32p, f32p, cint, C.POINTER(i64),
                                            C.POINTER(i32), vp, sz, vp]),
    "arx_sparse_adagrad_cat_multi_phase": (cint, [cint, cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                  C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), cint, cint,
                                                  C.POINTER(i32), C.POINTER(vp), C.POINTER(vp),
                                                  C.POINTER(i64), C.POINTER(i32), C.POINTER(f32), f32p, i64,
                                                  f32p, f32p, f32p, i32p, i32p, f32p, cint, C.POINTER(i64),
                                                  C.POINTER(i32), vp, sz, vp]),
    "arx_sparse_adagrad_cat_multi_bags": (cint, [cint, cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                 C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), cint, cint,
                                                 C.POINTER(i32), C.POINTER(vp), C.POINTER(vp),
                                                 C.POINTER(i64), C.POINTER(i32), C.POINTER(f32), f32p, i64,
                                                 f32p, f32p, f32p, i32p, i32p, f32p, vp, sz,
                                                 f32p, f32p, f32p, f32p, i64, i32p, i32p, i32p, cint, i32p,
                                                 vp, sz, vp]),
    "arx_sparse_adagrad_cat_multi_bags_csc": (cint, [cint, cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                     C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), cint, cint,
                                                     C.POINTER(i32), C.POINTER(vp), C.POINTER(vp),
                                                     C.POINTER(i64), C.POINTER(i32), C.POINTER(f32), f32p, i64,
                                                     f32p, f32p, f32p, i32p, i32p, f32p, vp, sz,
                                                     f32p, f32p, f32p, f32p, i64, i32p, i32p, i32p, cint, i32p,
                                                     vp, sz, i32p, i32p, vp, i32p, i64, vp]),
    "arx_sparse_adagrad_bags_workspace_bytes": (sz, [i64, cint, cint]),
    "arx_sparse_adagrad_bags": (cint, [cint, f32p, f32p, f32p, f32p, i64, cint, i32p, i32p, i32p, i64, cint,
                                       cint, C.POINTER(vp), C.POINTER(i64), C.POINTER(i32), C.POINTER(f32),
                                       f32p, i64, f32p, f32p, f32p, i32p, vp, sz, vp]),
    "arx_segment_pool_fwd": (cint, [f32p, i64, i32p, i64, i64, cint, f32p, f32p, i64, vp]),
    "arx_segment_pool_bwd": (cint, [f32p, i64, i32p, i64, i64, i64, cint, f32p, f32p, i64, f32p, i64, f32p, i64,
                                    f32p, vp]),
    "arx_max_argmax": (cint, [f32p, i64, i64, i64, i64, cint, f32p, i32p, vp, vp]),
    "arx_reduce_scratch_bytes": (sz, []),
    "arx_gmax_residual_bwd": (cint, [f32p, i32p, f32p, i64, f32p, cint, f32p, f32p, f32p, i64, vp]),
    "arx_gmax_norm_corr": (cint, [i32p, i32p, f32p, i64, f32p, i64, cint, cint, i64, f32p, cint, i64, i32p, f32p, i64,
                                  f32p, cint, f32p, vp]),
    "arx_adagrad_dense": (cint, [f32p, f32p, f32p, i64, f32p, f32p, vp]),
    "arx_adagrad_dense_multi": (cint, [cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), f32p, f32p, vp]),
    "arx_sq_norm_accum": (cint, [f32p, i64, cint, f32p, f32p, vp, vp]),
    "arx_clip_coef": (cint, [f32p, f32, f32p, f32p, vp]),
    "arx_sq_norm_accum_multi": (cint, [cint, C.POINTER(vp), C.POINTER(i64), C.POINTER(cint), C.POINTER(vp),
                                       f32p, vp, vp]),
    "arx_sq_norm_clip_multi": (cint, [cint, C.POINTER(vp), C.POINTER(i64), C.POINTER(cint), C.POINTER(vp), cint,
                                      f32p, C.c_float, f32p, f32p, vp, vp]),
    "arx_merged_sq_norm": (cint, [i32p, i32p, f32p, i64, cint, f32p, i64, cint, cint, i64, f32p, cint,
                                  i64, f32p, vp, sz, vp, vp]),
    "arx_fill_f32": (cint, [f32p, i64, f32, vp]),
    "arx_fill_i32": (cint, [i32p, i64, i32, vp]),
    "arx_take_i32": (cint, [i32p, i32p, i64, i32, i32p, vp]),
    "arx_fill_u8": (cint, [u8p, i64, cint, vp]),
    "arx_axpby": (cint, [f32, f32p, f32, f32p, i64, vp]),
    "arx_add_rows_bcast": (cint, [f32, f32p, i64, i64, f32, f32p, i64, i64, cint, vp]),
    "arx_row_sum": (cint, [f32p, i64, i64, i64, f32p, cint, vp]),
    "arx_col_sum_workspace_bytes": (sz, [i64, i64]),
    "arx_col_sum": (cint, [f32p, i64, i64, i64, f32p, vp, sz, vp]),
    "arx_sum_scaled": (cint, [f32p, i64, f32, f32p, vp]),
    "arx_dropout_fwd": (cint, [f32p, i64, f32, u64, f32p, u8p, vp]),
    "arx_dropout_bwd": (cint, [f32p, u8p, i64, f32, f32p, vp]),
    "arx_dropout_fwd_step": (cint, [f32p, i64, f32, u64, vp, f32p, u8p, vp]),
    "arx_counter_add": (cint, [vp, u64, vp]),
    "arx_copy_words": (cint, [cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), vp]),
    "arx_act_fwd": (cint, [f32p, i64, cint, f32p, vp]),
    "arx_act_bwd": (cint, [f32p, f32p, i64, cint, f32p, vp]),
    "arx_add_col_bias": (cint, [f32p, i64, i64, i64, f32p, vp]),
    "arx_topk": (cint, [f32p, i64, i64, i64, cint, f32p, i32p, vp]),
    "arx_topk_chunk": (cint, [f32p, i64, i64, i64, cint, i32, f32p, i32p, vp]),
    "arx_topk_merge": (cint, [f32p, i32p, f32p, i32p, i64, cint, cint, cint, f32p, i32p, vp]),
    "arx_gemm_nt_topk_parts": (cint, [i64, i64, C.POINTER(C.c_int)]),
    "arx_gemm_nt_topk_filter": (cint, [f32p, i64, i64, f32p, i64, i64, i64, f32p, f32p, i64, i32, f32p, i32p, i64, cint,
                                       i32p, f32p, i64, vp]),
    "arx_take_rows_i32": (cint, [i32p, i64, i32p, i64, i64, cint, i32p, i64, vp]),
    "arx_gemm_nt_eval_parts": (cint, [f32p, i64, i64, f32p, i64, i64, i64, f32p, f32p, f32p, f32p, i64, vp]),
    "arx_lstm_fwd": (cint, [f32p, f32p, f32p, i64, i64, cint, cint, f32, f32p, f32p, f32p, vp]),
    "arx_lstm_bwd": (cint, [f32p, f32p, f32p, f32p, f32p, i64, i64, cint, cint, f32p, vp]),
    "arx_lstm_bwd_wxt": (cint, [f32p, f32p, f32p, f32p, f32p, i64, i64, cint, cint, f32p, f32p, vp]),
    "arx_seq_weights": (cint, [f32p, i64, i64, f32p, vp]),
    "arx_capture_begin": (cint, [vp]),
    "arx_capture_end": (cint, [vp, C.POINTER(vp)]),
    "arx_capture_end_feeds": (cint, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(cint)]),
    "arx_graph_feed_dst0": (cint, [vp, cint, C.POINTER(vp)]),
    "arx_graph_set_feed": (cint, [vp, vp, cint, cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]),
    "arx_graph_feeds_destroy": (cint, [vp]),
    "arx_graph_launch": (cint, [vp, vp]),
    "arx_graph_destroy": (cint, [vp]),
}


class ArxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libarx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C a-recsys_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("libarx.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

_NO_CHECK = ("arx_last_error", "arx_version", "arx_csr_expand_workspace_bytes",
             "arx_col_sum_workspace_bytes",
             "arx_gemm_f32_workspace_bytes", "arx_gemm_f32_tn_pair_workspace_bytes",
             "arx_sparse_adagrad_workspace_bytes",
             "arx_sample_wor_workspace_bytes", "arx_sample_wor_keys_workspace_bytes",
             "arx_gemm_nt_bx6_workspace_bytes", "arx_reduce_scratch_bytes", "arx_mw_scorer_supported",
             "arx_mw_scorer_state_bytes", "arx_mw_scorer_bwd_di_workspace_bytes", "arx_mce_scorer_supported",
             "arx_mce_scorer_state_bytes", "arx_mce_scorer_bwd_di_workspace_bytes", "arx_gemm_bt_bx6_supported")


def call(name, *args):
    """Call an int-returning entry point, raising ArxError on a negative code."""
    rc = getattr(lib, name)(*args)
    if name not in _NO_CHECK and rc != 0:
        msg = lib.arx_last_error()
        raise ArxError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
    return rc
