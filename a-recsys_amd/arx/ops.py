"""Thin tensor-level wrappers over the C ABI (include/arx.h).

torch tensors are used ONLY as device-memory holders: every function passes
`tensor.data_ptr()` and explicit sizes to libarx.so; no torch arithmetic runs
on the hot path.  All launches go to torch's current HIP stream so that
torch.cuda events / torch.distributed (RCCL) order correctly against them.
"""
from __future__ import annotations

import math

import os

import torch

from . import _lib
from ._lib import call

KEY_NONE = 0x7FFFFFFF


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_dev_index = None


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream()
    spends ~8 us in device-index plumbing per call -- a third of the host time of a B=64 step --
    so the raw-handle accessor is used where this torch build has it."""
    global _dev_index
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return _raw_stream(_dev_index)


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError("%s must be a device tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous() and t.dim() == 1:
        raise ValueError("%s must be contiguous" % name)


def _ld(t):
    """leading dimension (elements) of a 2-D row-major tensor or view"""
    if t.dim() == 1:
        return t.shape[0]
    if t.stride(-1) != 1:
        raise ValueError("inner dimension must be contiguous")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class Workspace(object):
    """Grow-only device scratch buffer (caller-owned workspace of the C ABI)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if nbytes <= 0:
            return 0, 0
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self.buf.data_ptr(), self.buf.numel()


def device_info():
    import ctypes as C
    cu, wave, lds = C.c_int(), C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    call("arx_device_info", C.byref(cu), C.byref(wave), C.byref(lds), arch, 64)
    return {"cu_count": cu.value, "wave_size": wave.value, "lds_bytes": lds.value,
            "arch": arch.value.decode()}


# ---- a4/a7 ---------------------------------------------------------------
def csr_expand(vals, starts, lens, row_ids, capacity, ws, pad_token=KEY_NONE, pad_seg=-1,
               seg_base=0, coef_scale=1.0, want_coef=False, out=None):
    """batch_slice2 / batch_segids2 (mulhot_index.py:48-67) on device.

    Returns (token_ids[capacity], segids[capacity], offsets[B+1], total[1], coef or None)."""
    B = int(row_ids.shape[0]) if row_ids is not None else int(lens.shape[0])
    dev = vals.device
    for t, n in ((vals, 'vals'), (starts, 'starts'), (lens, 'lens'), (row_ids, 'row_ids')):
        _chk(t, torch.int32, n)
    if out is None:
        tok = torch.empty(capacity, dtype=torch.int32, device=dev)
        seg = torch.empty(capacity, dtype=torch.int32, device=dev)
        offs = torch.empty(B + 1, dtype=torch.int32, device=dev)
        tot = torch.empty(1, dtype=torch.int32, device=dev)
        coef = torch.empty(capacity, dtype=torch.float32, device=dev) if want_coef else None
    else:
        tok, seg, offs, tot, coef = out
    wsp, wsn = ws.get(_lib.lib.arx_csr_expand_workspace_bytes(B))
    call("arx_csr_expand", _p(vals), _p(starts), _p(lens), _p(row_ids), B, _p(tok), _p(seg),
         int(capacity), _p(offs), _p(tot), int(pad_token), int(pad_seg), int(seg_base),
         float(coef_scale), _p(coef), wsp, wsn, _stream())
    return tok, seg, offs, tot, coef


def bag_expand_padded(vals, starts, lens, row_ids, max_len, seg_base, coef_scale, keys_out, src_out,
                      coef_out, pad_token=KEY_NONE):
    """Un-compacted bag expansion for K7 (slot r*max_len + j; pads = pad_token): one launch."""
    B = int(row_ids.shape[0]) if row_ids is not None else int(lens.shape[0])
    if int(keys_out.shape[0]) < B * int(max_len):
        raise ValueError("bag_expand_padded: output buffers hold %d < %d entries"
                         % (int(keys_out.shape[0]), B * int(max_len)))
    call("arx_bag_expand_padded", _p(vals), _p(starts), _p(lens), _p(row_ids), B, int(max_len),
         int(pad_token), int(seg_base), float(coef_scale), _p(keys_out), _p(src_out), _p(coef_out),
         _stream())


def sparse_site_onehot(cat_map, ids, row_base, coef, keys_out, src_out, coef_out):
    call("arx_sparse_site_onehot", _p(cat_map), _p(ids), int(ids.shape[0]), int(row_base),
         float(coef), _p(keys_out), _p(src_out), _p(coef_out), _stream())


def shard_route(ids, world, rank, zero_row, rows_out, keys_out):
    call("arx_shard_route", _p(ids), int(ids.shape[0]), int(world), int(rank), int(zero_row),
         _p(rows_out), _p(keys_out), _stream())


def pool_blocks(ids, world, rank, zero_row, cap, counts, gidx=None, my_slots=None, pool_rows=None):
    """Block layout of a pool striped over `world` owners (arx.h); cap == 0: counts only."""
    call("arx_pool_blocks", _p(ids), int(ids.shape[0]), int(world), int(rank), int(zero_row), int(cap),
         _p(counts), _p(gidx), _p(my_slots), _p(pool_rows), _stream())


def copy_2d(src, dst):
    call("arx_copy_2d", _p(src), _ld(src), _p(dst), _ld(dst), int(src.shape[0]), int(src.shape[1]),
         _stream())


def copy_strided(src, dst):
    """dst[i] = src[i] for 1-D fp32 views of any (positive) element stride."""
    call("arx_copy_strided_f32", _p(src), int(src.stride(0)), _p(dst), int(dst.stride(0)),
         int(src.shape[0]), _stream())


# ---- a5 ---------------------------------------------------------------------
def transpose(src, dst):
    """dst[c, r] = src[r, c] for 2-D fp32 views with unit inner stride."""
    rows, cols = int(src.shape[0]), int(src.shape[1])
    call("arx_transpose_f32", _p(src), int(src.stride(0)), rows, cols, _p(dst), int(dst.stride(0)),
         _stream())


def gather_rows_wide(src, rows, dst):
    """dst[r, :] = src[rows[r], :] for 2-D fp32 views of any width (zeros for row indices out of range)."""
    call("arx_gather_rows_wide", _p(src), _ld(src), int(src.shape[0]), _p(rows), int(rows.shape[0]),
         int(dst.shape[1]), _p(dst), _ld(dst), _stream())


def gather_onehot(E, bias, cat_map, ids, out, scale=1.0, accumulate=False, bias_out=None):
    _chk(E, torch.float32, 'E'); _chk(ids, torch.int32, 'ids'); _chk(out, torch.float32, 'out')
    call("arx_gather_onehot_fwd", _p(E), _p(bias), _p(cat_map), _p(ids), int(ids.shape[0]),
         int(E.shape[1]), float(scale), int(bool(accumulate)), _p(out), _ld(out), _p(bias_out),
         _stream())
    return out


def gather_onehot_packed(E, bias, cat_map, ids, out, scale=1.0):
    """out[r] = [scale * E[row] | scale * bias[row] | pad]: packed rows of the sharded exchanges."""
    _chk(E, torch.float32, 'E'); _chk(ids, torch.int32, 'ids'); _chk(out, torch.float32, 'out')
    call("arx_gather_onehot_packed_fwd", _p(E), _p(bias), _p(cat_map), _p(ids), int(ids.shape[0]),
         int(E.shape[1]), float(scale), _p(out), _ld(out), _stream())
    return out


def gather_id_plus_bag(E_id, bias_id, cat_map, E_tok, bias_tok, vals, starts, lens, ids, out, scale=1.0,
                       accumulate=False, bias_out=None):
    """id row + bag mean of one entity in one launch (both scaled by `scale`)."""
    call("arx_gather_id_plus_bag", _p(E_id), _p(bias_id), _p(cat_map), _p(E_tok), _p(bias_tok), _p(vals),
         _p(starts), _p(lens), _p(ids), int(ids.shape[0]), int(E_id.shape[1]), float(scale),
         int(bool(accumulate)), _p(out), _ld(out), _p(bias_out), _stream())


class GatherSet(object):
    """Descriptor arrays of arx_gather_onehot_multi, built once per plan.
    sites: [(E, bias|None, cat_map|None, ids, out, scale, bias_out|None)], equal width d.
    bias_out may be the string 'packed': the bias goes to column d of the site's out rows.
    bias may be an int c: column c of the (packed) table rows themselves."""

    def __init__(self, sites):
        import ctypes as C
        n = len(sites)
        self.n = n
        self.d = int(sites[0][0].shape[1])
        vp = lambda xs: (C.c_void_p * n)(*[(_p(x) or None) for x in xs])
        self.E = vp([s[0] for s in sites])
        colb = [isinstance(s[1], int) for s in sites]
        self.bias = (C.c_void_p * n)(*[(_p(s[0]) + 4 * s[1]) if cb else (_p(s[1]) or None)
                                       for s, cb in zip(sites, colb)])
        self.ldbi = (C.c_int64 * n)(*[self.d if cb else 1 for cb in colb])
        self.cat_map, self.ids = vp([s[2] for s in sites]), vp([s[3] for s in sites])
        packed = [isinstance(s[6], str) for s in sites]
        assert all(s[6] == 'packed' and _ld(s[4]) > self.d for s, pk in zip(sites, packed) if pk)
        self.out = vp([s[4] for s in sites])
        self.bias_out = (C.c_void_p * n)(*[(_p(s[4]) + 4 * self.d) if pk else (_p(s[6]) or None)
                                           for s, pk in zip(sites, packed)])
        self.cnt = (C.c_int64 * n)(*[int(s[3].shape[0]) for s in sites])
        self.ldo = (C.c_int64 * n)(*[_ld(s[4]) for s in sites])
        self.ldb = (C.c_int64 * n)(*[_ld(s[4]) if pk else 1 for s, pk in zip(sites, packed)])
        self.scale = (C.c_float * n)(*[float(s[5]) for s in sites])
        self._keep = sites


def gather_onehot_multi(gs):
    call("arx_gather_onehot_multi_ld", gs.n, gs.E, gs.bias, gs.ldbi, gs.cat_map, gs.ids, gs.cnt, gs.d, gs.scale, gs.out,
         gs.ldo, gs.bias_out, gs.ldb, _stream())


class LookupSet(object):
    """Descriptor arrays of arx_lookup_multi, built once per plan.  sites: [(E_id|None, bias_id|None,
    cat_map|None, E_tok|None, bias_tok|None, vals|None, starts|None, lens|None, ids, out, scale,
    bias_out|None)], equal width d."""

    def __init__(self, sites):
        import ctypes as C
        n = len(sites)
        self.n = n
        self.d = int((sites[0][0] if sites[0][0] is not None else sites[0][3]).shape[1])
        vp = lambda xs: (C.c_void_p * n)(*[(_p(x) or None) for x in xs])
        col = lambda k: vp([s[k] for s in sites])
        self.E_id, self.bias_id, self.cat_map = col(0), col(1), col(2)
        self.E_tok, self.bias_tok, self.vals, self.starts, self.lens = col(3), col(4), col(5), col(6), col(7)
        self.ids, self.out, self.bias_out = col(8), col(9), col(11)
        self.cnt = (C.c_int64 * n)(*[int(s[8].shape[0]) for s in sites])
        self.ldo = (C.c_int64 * n)(*[_ld(s[9]) for s in sites])
        self.scale = (C.c_float * n)(*[float(s[10]) for s in sites])
        self._keep = sites


def lookup_multi(ls):
    call("arx_lookup_multi", ls.n, ls.E_id, ls.bias_id, ls.cat_map, ls.E_tok, ls.bias_tok, ls.vals, ls.starts,
         ls.lens, ls.ids, ls.cnt, ls.d, ls.scale, ls.out, ls.ldo, ls.bias_out, _stream())


def gather_mulhot_mean(E, bias, vals, starts, lens, ids, out, scale=1.0, accumulate=False,
                       bias_out=None):
    _chk(E, torch.float32, 'E'); _chk(ids, torch.int32, 'ids'); _chk(out, torch.float32, 'out')
    call("arx_gather_mulhot_mean_fwd", _p(E), _p(bias), _p(vals), _p(starts), _p(lens), _p(ids),
         int(ids.shape[0]), int(E.shape[1]), float(scale), int(bool(accumulate)), _p(out),
         _ld(out), _p(bias_out), _stream())
    return out


# ---- a9 -----------------------------------------------------------------------
def dot_score(U, T, tbias, score):
    call("arx_dot_score_fwd", _p(U), _ld(U), _p(T), _ld(T), _p(tbias), int(U.shape[0]),
         int(U.shape[1]), _p(score), _stream())
    return score


def dot_score_bwd(U, T, dscore, dU, acc_dU, dT):
    call("arx_dot_score_bwd", _p(U), _ld(U), _p(T), _ld(T), _p(dscore), int(U.shape[0]),
         int(U.shape[1]), _p(dU), _ld(dU), int(bool(acc_dU)), _p(dT),
         _ld(dT) if dT is not None else 0, _stream())


# ---- a8 -----------------------------------------------------------------------
# The scorer products run on the bf16 matrix pipe, f32-exact (three exact bf16 pieces per operand, six MFMA terms,
# f32 accumulation: csrc/gemm_bx6.hip) -- the default since round 4; ARX_SCORER_F32=1 selects the f32-input MFMA
# kernels (the A/B reference, bench.py's *_f32mfma sub-results).  Read once per process.
SCORER_F32 = bool(os.environ.get("ARX_SCORER_F32"))
_BX6 = not SCORER_F32
_bx6_ws = {}


def gemm_nt_bx6(A, B, C, col_bias=None):
    """C[M, N] = A[M, K] . B[N, K]^T + col_bias by six bf16 MFMAs per f32 product term (csrc/gemm_bx6.hip):
    every bit of an f32 multiply-add chain at 16/6 of the f32 MFMA peak.  K in {64, 128}, N % 128 == 0."""
    M, K = int(A.shape[0]), int(A.shape[1])
    N = int(B.shape[0])
    ws = _bx6_ws.setdefault(A.device, Workspace(A.device))
    wsp, wsn = ws.get(_lib.lib.arx_gemm_nt_bx6_workspace_bytes(N, K))
    call("arx_gemm_nt_bx6", M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(col_bias), _p(C), _ld(C), wsp, wsn,
         _stream())
    return C


def gemm(A, B, C, ws, transA=False, transB=False, alpha=1.0, beta=0.0, col_bias=None,
         a_rowsum=None):
    """C[M,N] = alpha * op(A) . op(B) + beta * C + col_bias  (fp32 MFMA);
    a_rowsum[m] = sum_k op(A)[m,k] when given."""
    M, N = int(C.shape[0]), int(C.shape[1])
    K = int(A.shape[0] if transA else A.shape[1])
    kb = int(B.shape[1] if transB else B.shape[0])
    if K != kb:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, kb))
    if (_BX6 and transB and not transA and K in (64, 128) and N % 128 == 0 and alpha == 1.0 and beta == 0.0
            and a_rowsum is None and M >= 4096):
        return gemm_nt_bx6(A, B, C, col_bias)
    wsp, wsn = ws.get(_lib.lib.arx_gemm_f32_workspace_bytes(M, N, K))
    call("arx_gemm_f32_rowsum", int(bool(transA)), int(bool(transB)), M, N, K, float(alpha), _p(A),
         _ld(A), _p(B), _ld(B), float(beta), _p(C), _ld(C), _p(col_bias), _p(a_rowsum), wsp, wsn,
         _stream())
    return C


def gemm_tn_pair_supported(M, N1, N2, K):
    """The shape class of gemm_tn_pair (include/arx.h: arx_gemm_f32_tn_pair)."""
    return (N1 % 4 == 0 and N2 % 4 == 0 and 32 < N1 + N2 <= 128 and M >= 64 and M % 4 == 0 and
            K >= 64 and K % 32 == 0)


def gemm_tn_pair(A, B1, B2, shift, Ct, ws, a_rowsum=None):
    """Ct [N1 + N2, M] = (A^T . [B1 | B2 shifted down by `shift` rows])^T in one pass over A [K, M]
    (the LSTM cell's dW with A = dz, B1 = x, B2 = the cell outputs, shift = B), a_rowsum = A's column sums."""
    K, M = int(A.shape[0]), int(A.shape[1])
    N1, N2 = int(B1.shape[1]), int(B2.shape[1])
    wsp, wsn = ws.get(_lib.lib.arx_gemm_f32_tn_pair_workspace_bytes(M, N1 + N2, K))
    call("arx_gemm_f32_tn_pair", M, N1, N2, K, _p(A), _ld(A), _p(B1), _ld(B1), _p(B2), _ld(B2), int(shift),
         _p(Ct), _ld(Ct), _p(a_rowsum), wsp, wsn, _stream())
    return Ct


def gemm_steps_tn(A, B, C_steps, rowsum_steps, steps, Kb, C_sum=None, beta=0.0, rowsum_sum=None):
    """C_steps[t] = A_t^T . B_t for every time step t (rows t*Kb..), plus the sums."""
    M, N = int(C_steps.shape[1]), int(C_steps.shape[2])
    call("arx_gemm_f32_steps_tn", int(steps), M, N, int(Kb), _p(A), _ld(A), _p(B), _ld(B),
         _p(C_steps), _p(rowsum_steps), float(beta), _p(C_sum),
         _ld(C_sum) if C_sum is not None else 0, _p(rowsum_sum), _stream())


def dot_scaled(x, y, scale, out, n=None):
    call("arx_dot_scaled", _p(x), _p(y), int(x.numel() if n is None else n), float(scale), _p(out),
         _stream())


def inv_len_scale(lens, ids, c, out):
    call("arx_inv_len_scale", _p(lens), _p(ids), int(out.numel()), float(c), _p(out), _stream())


# ---- a14 ------------------------------------------------------------------------
def pos_mask_scatter(user_ids, pos_ptr, pos_items, item2slot, mask, value):
    call("arx_pos_mask_scatter", _p(user_ids), int(user_ids.shape[0]), _p(pos_ptr), _p(pos_items),
         _p(item2slot), _p(mask), _ld(mask), int(value), _stream())


def _slot_map_detach(ptr, _bits_keepalive):
    import ctypes as C
    try:
        _lib.lib.arx_slot_map_attach_bitmap(C.c_void_p(ptr), None)
    except Exception:
        pass


def slot_map_attach_bitmap(item2slot, bits):
    """Register the "in the pool?" bitmap (int32 zeros [(items + 32) // 32]) of an item2slot map.
    The registration is keyed by the map's device address: it is dropped when the map TENSOR is
    collected (weakref finaliser, which also keeps `bits` alive until then), so a stale entry can
    never meet a new allocation at the same address.  Best effort: with 16 maps registered the call
    is a no-op and the losses probe the map directly."""
    import weakref
    call("arx_slot_map_attach_bitmap", _p(item2slot), _p(bits))       # (bits None: detach)
    if bits is not None:
        weakref.finalize(item2slot, _slot_map_detach, item2slot.data_ptr(), bits)


def slot_map_set(item2slot, ids, clear=False):
    call("arx_slot_map_set", _p(item2slot), _p(ids), int(ids.shape[0]), int(bool(clear)), _stream())


# ---- a10-a12 -----------------------------------------------------------------------
def loss_mw(logits, tscore, mask, batch_loss, dlogits, dtscore, gscale, row_w=None,
            mask_rows=0, kind='mw'):
    """kind 'mw' (WMRB hinge, embed_attribute.py:641-649) or 'mce' (build-defined sampled softmax)."""
    B, S = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_%s_fwdbwd" % kind, _p(logits), _ld(logits), _p(tscore), _p(mask),
         _ld(mask) if mask is not None else 0, int(mask_rows), float(gscale), _p(row_w), B, S,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _p(dtscore),
         _stream())


def loss_warp(logits, target, mask, batch_loss, dlogits, gscale, row_w=None, mask_rows=0):
    B, V = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_warp_fwdbwd", _p(logits), _ld(logits), _p(target), _p(mask),
         _ld(mask) if mask is not None else 0, int(mask_rows), float(gscale), _p(row_w), B, V,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _stream())


POS_MASK_MAX_COLS = 1 << 20


def loss_mw_pos(logits, tscore, user_ids, pos_ptr, pos_items, item2slot, batch_loss, dlogits,
                dtscore, gscale, row_w=None, mask_rows=0, kind='mw'):
    B, S = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_%s_fwdbwd_pos" % kind, _p(logits), _ld(logits), _p(tscore), _p(user_ids), _p(pos_ptr),
         _p(pos_items), _p(item2slot), int(mask_rows), float(gscale), _p(row_w), B, S,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _p(dtscore),
         _stream())


def loss_mw_fused_pos(logits, U, T, tbias, user_ids, pos_ptr, pos_items, item2slot, batch_loss, dlogits,
                      tscore_out, dtscore, dU, dT, gscale, row_w=None, mask_rows=0, kind='mw'):
    """'mw' (or 'mce') loss with the target score t = U.T + tbias, dT = dt*U and dU = dt*T formed by the
    same kernel (dU is WRITTEN: accumulate the scorer's dU onto it afterwards)."""
    B, S = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_%s_fused_pos" % kind, _p(logits), _ld(logits), _p(U), _ld(U), _p(T), _ld(T), _p(tbias),
         int(tbias.stride(0)) if tbias is not None else 1, int(U.shape[1]), _p(user_ids), _p(pos_ptr),
         _p(pos_items), _p(item2slot), int(mask_rows), float(gscale), _p(row_w), B, S, _p(batch_loss),
         _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _p(tscore_out), _p(dtscore),
         int(dtscore.stride(0)) if dtscore is not None else 1, _p(dU),
         _ld(dU) if dU is not None else 0, _p(dT), _ld(dT) if dT is not None else 0, _stream())


def eval_chunk_accum(logits, tscore, mode, first, acc0, acc1):
    call("arx_eval_chunk_accum", _p(logits), _ld(logits), int(logits.shape[0]), int(logits.shape[1]), _p(tscore),
         int(mode), int(bool(first)), _p(acc0), _p(acc1), _stream())


def eval_warp_unmask(U, P, pbias, tscore, user_ids, pos_ptr, pos_items, item2col, s_acc, mask_rows=0):
    call("arx_eval_warp_unmask", _p(U), _ld(U), _p(P), _ld(P), _p(pbias), int(U.shape[1]), _p(tscore), _p(user_ids),
         _p(pos_ptr), _p(pos_items), _p(item2col), int(mask_rows), int(U.shape[0]), int(P.shape[0]), _p(s_acc),
         _stream())


def eval_finish(mode, acc0, acc1, tscore, out):
    call("arx_eval_finish", int(mode), _p(acc0), _p(acc1), _p(tscore), int(out.shape[0]), _p(out), _stream())


def mw_scorer_supported(B, S, d):
    """True when the fused 'mw' scorer (csrc/scorer.hip) takes this shape: d in {64, 128}, S % 128 == 0,
    128 <= S <= 2048 -- and the process has not asked for the f32-MFMA reference (ARX_SCORER_F32)."""
    return (not SCORER_F32) and bool(_lib.lib.arx_mw_scorer_supported(int(B), int(S), int(d)))


class MwScorer(object):
    """The 'mw' scorer of a training step on the bf16 matrix pipe, f32-exact (include/arx.h, "a8-a11 fused"):
    fwd() forms target score, act bits, loss and row factors without [B, S] logits; bwd_dU() / bwd_dI() are the two
    backward products out of the bits.  Owns the caller-side state buffer (one per model and stream)."""

    def __init__(self, B, S, d, device):
        import ctypes as C
        self.B, self.S, self.d = int(B), int(S), int(d)
        n = int(_lib.lib.arx_mw_scorer_state_bytes(self.B, self.S, self.d))
        if n == 0:
            raise ValueError("MwScorer: shape not supported (B=%d, S=%d, d=%d)" % (B, S, d))
        self.state = torch.zeros(n, dtype=torch.uint8, device=device)
        lay = (C.c_int64 * 6)()
        call("arx_mw_scorer_state_layout", self.B, self.S, self.d, lay)
        self.Bp = int(lay[4])
        ld, ldt = int(lay[1]), int(lay[5])
        i32 = self.state.view(torch.int32)
        self.act_bits = i32[lay[0] // 4: lay[0] // 4 + (self.S // 32) * ld].view(self.S // 32, ld)     # word-major
        self.act_bits_t = i32[lay[2] // 4: lay[2] // 4 + (self.Bp // 32) * ldt].view(self.Bp // 32, ldt)[:, :self.S]
        self.g = self.state.view(torch.float32)[lay[3] // 4: lay[3] // 4 + self.Bp]
        self.ws = Workspace(device)

    def fwd(self, U, P, pbias, T, tbias, user_ids, pos_ptr, pos_items, item2slot, batch_loss, tscore_out, dtscore,
            dU, dT, gscale, row_w=None, mask_rows=0, phases=7, seq_w=None, seq_rows=0):
        """phases: 1 pool planes + hit lists, 2 hinge GEMM, 4 row kernel (arx_mw_scorer_fwd_phases).  seq_w (the
        sequence model's raw example weights, [L * seq_rows]): row_w is WRITTEN by phase 1 (normalised over time)."""
        call("arx_mw_scorer_fwd_seqw", _p(U), _ld(U), _p(P), _ld(P), _p(pbias), _p(T), _ld(T), _p(tbias),
             int(tbias.stride(0)) if tbias is not None else 1, self.d, _p(user_ids), _p(pos_ptr), _p(pos_items),
             _p(item2slot), int(mask_rows), float(gscale), _p(row_w), _p(seq_w), int(seq_rows), self.B, self.S,
             _p(batch_loss), _p(tscore_out), _p(dtscore), int(dtscore.stride(0)) if dtscore is not None else 1,
             _p(dU), _ld(dU) if dU is not None else 0, _p(dT), _ld(dT) if dT is not None else 0, _p(self.state),
             int(self.state.numel()), int(phases), _stream())

    def bwd_dU(self, dU, beta=1.0):
        """dU = beta dU + g * (act . P)"""
        call("arx_mw_scorer_bwd_du", self.B, self.S, self.d, _p(self.state), float(beta), _p(dU), _ld(dU), _stream())

    def bwd_dI(self, dI, db=None, beta=0.0, step_rows=0, dI_steps=None, db_steps=None, loss=None):
        """dI = beta dI + act^T . (g U), db = act^T . g; step_rows > 0: the per-time-step products too; loss =
        (batch_loss [B], gscale, row_w [B] or None, out [1]): out = gscale * sum_r row_w_r * batch_loss_r, the step's
        scalar, out of the same reduce launch."""
        wsp, wsn = (None, 0)
        if dI_steps is None:
            wsp, wsn = self.ws.get(_lib.lib.arx_mw_scorer_bwd_di_workspace_bytes(self.B, self.S, self.d,
                                                                                 int(step_rows)))
        bl, gs, rw, out = loss if loss is not None else (None, 0.0, None, None)
        call("arx_mw_scorer_bwd_di_loss", self.B, self.S, self.d, _p(self.state), int(step_rows), float(beta), _p(dI),
             _ld(dI), _p(db), _p(dI_steps), _p(db_steps), _p(bl), float(gs), _p(rw), _p(out), wsp, wsn, _stream())


def gemm_bt_bx6_supported(M, N, K):
    """arx_gemm_bt_bx6 takes this shape (N % 32 == 0, N <= 128, K in {64, 128, 256}) and the six-term family is on."""
    return (not SCORER_F32) and bool(_lib.lib.arx_gemm_bt_bx6_supported(int(M), int(N), int(K)))


def gemm_bt_bx6(A, Bt, C, beta=0.0):
    """C = beta C + A . Bt^T, six-term f32-exact, Bt [N, K] k-contiguous (the LSTM's dx = dz . W_x^T with Bt = W_x)."""
    M, K = int(A.shape[0]), int(A.shape[1])
    N = int(Bt.shape[0])
    call("arx_gemm_bt_bx6", M, N, K, _p(A), _ld(A), _p(Bt), _ld(Bt), float(beta), _p(C), _ld(C), _stream())


def mce_scorer_supported(B, S, d):
    """True when the fused 'mce' family (csrc/scorer.hip, k_mc_flow) takes this shape: d in {64, 128}, S % 128 == 0,
    128 <= S <= 2048 -- and the process has not asked for the materialising reference (ARX_SCORER_F32 /
    ARX_MCE_FUSED=0)."""
    if SCORER_F32 or os.environ.get('ARX_MCE_FUSED', '1') == '0':
        return False
    return bool(_lib.lib.arx_mce_scorer_supported(int(B), int(S), int(d)))


class MceScorer(object):
    """The build-defined sampled softmax 'mce' without [B, S] logits or weights in HBM (include/arx.h, "'mce' on the
    fused family"): fwd() leaves loss, target-score terms and the COMPLETE latent gradient (one pass forms s_r and
    O_r = sum_s e_rs P_s); bwd_dI() recomputes the weight tiles for the pool-side product.  Same surface as MwScorer
    (bwd_dU is a no-op kept for the graph nodes that drive both)."""

    def __init__(self, B, S, d, device):
        self.B, self.S, self.d = int(B), int(S), int(d)
        n = int(_lib.lib.arx_mce_scorer_state_bytes(self.B, self.S, self.d))
        if n == 0:
            raise ValueError("MceScorer: shape not supported (B=%d, S=%d, d=%d)" % (B, S, d))
        self.state = torch.zeros(n, dtype=torch.uint8, device=device)
        self.ws = Workspace(device)
        self._pbias, self._mask_rows = None, 0

    def fwd(self, U, P, pbias, T, tbias, user_ids, pos_ptr, pos_items, item2slot, batch_loss, tscore_out, dtscore,
            dU, dT, gscale, row_w=None, mask_rows=0, phases=7, seq_w=None, seq_rows=0):
        self._pbias, self._mask_rows = pbias, int(mask_rows)
        call("arx_mce_scorer_fwd", _p(U), _ld(U), _p(P), _ld(P), _p(pbias), _p(T), _ld(T), _p(tbias),
             int(tbias.stride(0)) if tbias is not None else 1, self.d, _p(user_ids), _p(pos_ptr), _p(pos_items),
             _p(item2slot), int(mask_rows), float(gscale), _p(row_w), _p(seq_w), int(seq_rows), self.B, self.S,
             _p(batch_loss), _p(tscore_out), _p(dtscore), int(dtscore.stride(0)) if dtscore is not None else 1,
             _p(dU), _ld(dU) if dU is not None else 0, _p(dT), _ld(dT) if dT is not None else 0, _p(self.state),
             int(self.state.numel()), int(phases), _stream())

    def bwd_dU(self, dU, beta=1.0):
        """(the forward wrote dU = coef O + dt T: nothing left to add)"""
        if beta != 1.0:
            raise RuntimeError("MceScorer: the latent gradient is written by fwd(); bwd_dU(beta != 1) would drop it")

    def bwd_dI(self, dI, db=None, beta=0.0, step_rows=0, dI_steps=None, db_steps=None, loss=None):
        wsp, wsn = self.ws.get(_lib.lib.arx_mce_scorer_bwd_di_workspace_bytes(self.B, self.S, self.d, int(step_rows)))
        bl, gs, rw, out = loss if loss is not None else (None, 0.0, None, None)
        call("arx_mce_scorer_bwd_di_loss", self.B, self.S, self.d, _p(self.state), _p(self._pbias), self._mask_rows,
             int(step_rows),
             float(beta), _p(dI), _ld(dI), _p(db), _p(dI_steps), _p(db_steps), _p(bl), float(gs), _p(rw), _p(out),
             wsp, wsn, _stream())


def loss_warp_pos(logits, target, user_ids, pos_ptr, pos_items, item2slot, batch_loss, dlogits,
                  gscale, row_w=None, mask_rows=0):
    B, V = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_warp_fwdbwd_pos", _p(logits), _ld(logits), _p(target), _p(user_ids), _p(pos_ptr),
         _p(pos_items), _p(item2slot), int(mask_rows), float(gscale), _p(row_w), B, V,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _stream())


def item_frequency(item_ids, n_items, counts, weights=None, total=0, power=0.5):
    """counts[v] += #(item_ids == v); weights (optional) = (counts / total)^power, 0 where unseen."""
    n = 0 if item_ids is None else int(item_ids.shape[0])
    call("arx_item_frequency", _p(item_ids), n, int(n_items), int(total), float(power), _p(counts),
         _p(weights), _stream())


def sample_wor(weights, S, seed, counter, out, ws, key_cap=0.0, out_keys=None):
    """Weighted sampling without replacement on device (exponential race); out: int32 [S].
    key_cap > 0: pre-filter for very large item sets (arx_sample_wor_capped).  out_keys (float32
    [S]): the race keys of the drawn items, ascending (arx_sample_wor_keys)."""
    n = int(weights.shape[0])
    wsp, wsn = ws.get(_lib.lib.arx_sample_wor_keys_workspace_bytes(n, int(S), float(key_cap)))
    call("arx_sample_wor_keys", _p(weights), n, int(S), int(seed) & (2 ** 64 - 1),
         int(counter) & (2 ** 64 - 1), float(key_cap), _p(out), _p(out_keys), wsp, wsn, _stream())
    return out


RS_KINDS = {'rs': 0, 'rs-sig': 1, 'rs-sig2': 2, 'bbpr': 3}
RS_FUNCS = {'log': 0, 'exp': 1, 'poly': 2, 'poly2': 3, 'linear': 4, 'square': 5}


def loss_rs(logits, target, kind, loss_func, exp_p, batch_loss, dlogits, gscale, mask=None,
            pos=None, row_w=None, mask_rows=0):
    """rs / rs-sig / rs-sig2 / bbpr over full logits.  pos = (user_ids, pos_ptr, pos_items,
    item2slot) builds the mask on the fly; else `mask` (uint8 [rows, V]) or no mask."""
    B, V = int(logits.shape[0]), int(logits.shape[1])
    uid, ptr, items, i2s = pos if pos is not None else (None, None, None, None)
    call("arx_loss_rs_fwdbwd", _p(logits), _ld(logits), _p(target), _p(mask),
         _ld(mask) if mask is not None else 0, _p(uid), _p(ptr), _p(items), _p(i2s), int(mask_rows),
         RS_KINDS[kind], RS_FUNCS[loss_func], float(exp_p), float(gscale), _p(row_w), B, V,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _stream())


def row_logsumexp(logits, out):
    call("arx_row_logsumexp", _p(logits), _ld(logits), int(logits.shape[0]), int(logits.shape[1]),
         _p(out), _stream())
    return out


def loss_ce(logits, target, batch_loss, dlogits, gscale, row_w=None):
    B, V = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_ce_fwdbwd", _p(logits), _ld(logits), _p(target), float(gscale), _p(row_w), B, V,
         _p(batch_loss), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _stream())


def loss_warp_eval(logits, target, mask, margin_rank, true_rank, mask_rows=0):
    B, V = int(logits.shape[0]), int(logits.shape[1])
    call("arx_loss_warp_eval", _p(logits), _ld(logits), _p(target), _p(mask),
         _ld(mask) if mask is not None else 0, int(mask_rows), B, V, _p(margin_rank),
         _p(true_rank), _stream())


# ---- a17 ---------------------------------------------------------------------------
def key_bits_for(nrows):
    return max(1, int(math.ceil(math.log2(max(2, int(nrows))))))


def sparse_adagrad(E, acc, bias, bias_acc, keys, src, coef, G, Gb, lr_dev, ws, gscale_dev=None,
                   n=None, aux_cnt=None):
    """aux_cnt (int32[table rows], zeros): enables the one-launch apply (ticket)."""
    n = int(keys.shape[0]) if n is None else int(n)
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_workspace_bytes(n))
    call("arx_sparse_adagrad_ticket", _p(E), _p(acc), _p(bias), _p(bias_acc), int(E.shape[1]),
         _p(keys), _p(src), _p(coef), n, _p(G), _ld(G), _p(Gb), _p(lr_dev), _p(gscale_dev),
         key_bits_for(E.shape[0]), _p(aux_cnt), wsp, wsn, _stream())


class CatSiteArgs(object):
    """Host-side descriptor arrays of arx_sparse_adagrad_cat, built once per plan."""

    def __init__(self, sites):
        import ctypes as C
        n = len(sites)
        self.n = n
        self.total = sum(int(s[2].shape[0]) for s in sites)
        self.cat_map = (C.c_void_p * n)(*[_p(s[0]) or None for s in sites])
        self.ids = (C.c_void_p * n)(*[_p(s[2]) for s in sites])
        self.count = (C.c_int64 * n)(*[int(s[2].shape[0]) for s in sites])
        self.row_base = (C.c_int32 * n)(*[int(s[3]) for s in sites])
        self.coef = (C.c_float * n)(*[float(s[4]) for s in sites])
        self._keep = sites


def sparse_adagrad_cat(E, acc, bias, bias_acc, site_args, G, Gb, lr_dev, aux_first, aux_cnt,
                       aux_hot, keys_buf, src_buf, coef_buf, ws, gscale_dev=None, mode=0):
    """sites: list of (cat_map|None, _, ids, row_base, coef) packed in CatSiteArgs.
    mode 0: fused keygen + LDS sort + Adagrad passes (n <= 16384) else atomic election."""
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_workspace_bytes(site_args.total))
    call("arx_sparse_adagrad_cat", _p(E), _p(acc), _p(bias), _p(bias_acc), int(E.shape[0]),
         int(E.shape[1]), site_args.n, site_args.cat_map, site_args.ids, site_args.count,
         site_args.row_base, site_args.coef, _p(G), _ld(G), _p(Gb), _p(lr_dev), _p(gscale_dev),
         _p(aux_first), _p(aux_cnt), _p(aux_hot), int(aux_hot.numel()), _p(keys_buf), _p(src_buf),
         _p(coef_buf), int(mode), wsp, wsn, _stream())


class MultiCatArgs(object):
    """Host-side descriptor arrays of arx_sparse_adagrad_cat_multi, built once per plan.
    tables: list of (E, acc, bias|None, bias_acc|None, aux_cnt|None);
    sites: list of (table_index, cat_map|None, ids, row_base, coef)."""

    def __init__(self, tables, sites, extra=()):
        """extra: [(table_index, n)] pre-expanded segments (arx_csr_expand output) that follow
        the one-hot contributions in keys_buf / src_buf / coef_buf."""
        import ctypes as C
        nt, ns = len(tables), len(sites)
        self.nt, self.ns = nt, ns
        # a table may be VIRTUAL -- (None, None, None, None, None, rows): entity ids of a bag table
        # riding on the pass (arx_sparse_adagrad_cat_multi_bags), no rows of its own
        self.d = int(next(t[0] for t in tables if t[0] is not None).shape[1])
        self.n_cat = sum(int(s[2].shape[0]) for s in sites)
        self.nx = len(extra)
        self.extra_n = (C.c_int64 * max(self.nx, 1))(*[int(e[1]) for e in extra])
        self.extra_table = (C.c_int32 * max(self.nx, 1))(*[int(e[0]) for e in extra])
        self.extra_off = []
        off = self.n_cat
        for e in extra:
            self.extra_off.append(off)
            off += int(e[1])
        self.total = off
        self.E = (C.c_void_p * nt)(*[_p(t[0]) for t in tables])
        self.acc = (C.c_void_p * nt)(*[_p(t[1]) for t in tables])
        self.bias = (C.c_void_p * nt)(*[_p(t[2]) or None for t in tables])
        self.bias_acc = (C.c_void_p * nt)(*[_p(t[3]) or None for t in tables])
        self.rows = (C.c_int64 * nt)(*[int(t[0].shape[0]) if t[0] is not None else int(t[5]) for t in tables])
        self.cnt = (C.c_void_p * nt)(*[_p(t[4]) or None for t in tables])
        m = max(ns, 1)
        self.site_table = (C.c_int32 * m)(*[int(s[0]) for s in sites])
        self.cat_map = (C.c_void_p * m)(*[_p(s[1]) or None for s in sites])
        self.ids = (C.c_void_p * m)(*[_p(s[2]) for s in sites])
        self.count = (C.c_int64 * m)(*[int(s[2].shape[0]) for s in sites])
        self.row_base = (C.c_int32 * m)(*[int(s[3]) for s in sites])
        self.coef = (C.c_float * m)(*[float(s[4]) for s in sites])
        self._keep = (tables, sites)


def sparse_adagrad_cat_multi(args, G, Gb, lr_dev, keys_buf, src_buf, coef_buf, ws, gscale_dev=None, phase=3):
    """phase 1: key generation + sort (needs the ids only), phase 2: apply, 3: both.  The two
    halves must use the same workspace object, untouched in between."""
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_workspace_bytes(args.total))
    call("arx_sparse_adagrad_cat_multi_phase", int(phase), args.nt, args.E, args.acc, args.bias,
         args.bias_acc, args.rows, args.cnt, args.d, args.ns, args.site_table, args.cat_map, args.ids,
         args.count, args.row_base, args.coef, _p(G), _ld(G), _p(Gb), _p(lr_dev), _p(gscale_dev),
         _p(keys_buf), _p(src_buf), _p(coef_buf), args.nx, args.extra_n, args.extra_table, wsp, wsn,
         _stream())


def sparse_adagrad_cat_multi_bags(args, G, Gb, lr_dev, keys_buf, src_buf, coef_buf, ws, bag_E, bag_acc,
                                  bag_bias, bag_bias_acc, vals, starts, lens, max_len, bag_ws,
                                  gscale_dev=None, phase=3, bag_aux_cnt=None, csc=None, split=None):
    """arx_sparse_adagrad_cat_multi with a multi-hot table riding on it: the lookups of one-hot
    table 0 are also the lookups of the bags of table bag_E -- an item's id row and its multi-hot
    attribute (HET layout); starts / lens are indexed by the ROW of table 0 (arx.h).  Same phases as
    sparse_adagrad_cat_multi; `ws` and `bag_ws` must be the same objects for both halves."""
    assert args.nx == 0
    n0 = sum(int(args.count[q]) for q in range(args.ns) if int(args.site_table[q]) == 0)
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_workspace_bytes(args.total))
    bwp, bwn = bag_ws.get(_lib.lib.arx_sparse_adagrad_bags_workspace_bytes(max(n0, 1), int(max_len), args.d))
    # csc: (BagCSC, flags, slot_of) -- the table's static token order (no expansion, no token sort in the step)
    cs, flags, slot_of = csc if csc is not None else (None, None, None)
    # split: this pass's apply form (True `split`, False `win`, None: the process default ARX_K7_RIDER) -- arx.h
    form = 0 if split is None else (0x100 if split else 0x200)
    call("arx_sparse_adagrad_cat_multi_bags_csc", int(phase) | form, args.nt, args.E, args.acc, args.bias,
         args.bias_acc, args.rows, args.cnt, args.d, args.ns, args.site_table, args.cat_map, args.ids,
         args.count, args.row_base, args.coef, _p(G), _ld(G), _p(Gb), _p(lr_dev), _p(gscale_dev),
         _p(keys_buf), _p(src_buf), _p(coef_buf), wsp, wsn, _p(bag_E), _p(bag_acc), _p(bag_bias),
         _p(bag_bias_acc), int(bag_E.shape[0]), _p(vals), _p(starts), _p(lens), int(max_len),
         _p(bag_aux_cnt), bwp, bwn, _p(cs.qpos) if cs is not None else None,
         _p(cs.qte) if cs is not None else None, _p(flags), _p(slot_of), cs.nq if cs is not None else 0,
         _stream())


class BagCSC(object):
    """The STATIC token-major order of a multi-hot table's bags (csrc/csc.hip; arx.h,
    arx_sparse_adagrad_cat_multi_bags_csc): built once per (bag index, max_len) on the host from the
    feature CSR of attributes/attribute.py (embed_attribute.py:265-318 uploads it once and never
    changes it).  Pairs are enumerated by (entity, position in bag) and ordered by token with a stable
    sort -- the order the per-step stable radix sort of the expanded bags produced, so the token
    sums keep their bits.  ok == False: a CSR position belongs to two bags (no place of its own)."""

    def __init__(self, vals, starts, lens, max_len, table_rows):
        import numpy as np
        dev = vals.device
        v = vals.detach().cpu().numpy()
        ln = np.clip(lens.detach().cpu().numpy().astype(np.int64), 0, int(max_len))
        n_ent, nv = int(ln.shape[0]), int(v.shape[0])
        st = starts.detach().cpu().numpy().astype(np.int64)[:n_ent]
        tot = int(ln.sum())
        assert nv < (1 << 31) and tot < (1 << 31)
        # every CSR position may belong to ONE bag only (it gets one place): ranges of non-empty bags, in start order,
        # must not overlap and must lie inside vals
        live = np.nonzero(ln > 0)[0]
        o_ = live[np.argsort(st[live], kind='stable')]
        inside = bool(live.size == 0 or (st[o_[0]] >= 0 and st[o_[-1]] + ln[o_[-1]] <= nv))
        self.ok = inside and bool(np.all(st[o_[:-1]] + ln[o_[:-1]] <= st[o_[1:]]))
        self.nq, self.n_ent = 0, n_ent
        if not self.ok:
            self.qpos = self.qte = None
            return
        ent = np.repeat(np.arange(n_ent, dtype=np.int32), ln)
        pos = np.repeat((st - (np.cumsum(ln) - ln)).astype(np.int32), ln) + np.arange(tot, dtype=np.int32)
        tok = v[pos]                                   # CSR position and token of every pair, (entity, position) order
        bad = (tok < 0) | (tok >= int(table_rows))
        if bad.any():                                  # tokens outside the table are dropped (as the step's sort does)
            keep = ~bad
            pos, ent, tok = pos[keep], ent[keep], tok[keep]
        # stable order by token: LSD passes over 16-bit digits (numpy's stable sort is a radix sort for those)
        order = np.argsort((tok & 0xffff).astype(np.uint16), kind='stable')
        if int(table_rows) > (1 << 16):
            order = order[np.argsort((tok[order] >> 16).astype(np.uint16), kind='stable')]
        nq = int(order.shape[0])
        qpos = np.full(max(nv, 1), -1, dtype=np.int32)
        qpos[pos[order]] = np.arange(nq, dtype=np.int32)
        self.nq = nq
        qte = np.empty((max(nq, 1), 2), dtype=np.int32)
        qte[:nq, 0] = tok[order]
        qte[:nq, 1] = ent[order]
        self.qpos = torch.from_numpy(qpos).to(dev)
        self.qte = torch.from_numpy(qte).to(dev)

    def scratch(self):
        """Per-pass device state: the flag bytes (zero between steps) and the entity -> merged-row map."""
        dev = self.qpos.device
        fb = (self.nq + 255) // 256 * 256          # flag bytes + one coarse byte per 16 of them (arx.h)
        return (torch.zeros(fb + fb // 16, dtype=torch.uint8, device=dev),
                torch.zeros(max(self.n_ent, 1), dtype=torch.int32, device=dev))


class BagSiteArgs(object):
    """Host-side descriptor arrays of arx_sparse_adagrad_bags, built once per plan.
    sites: list of (ids, row_base, coef) -- the lookups of one multi-hot table."""

    def __init__(self, sites, max_len):
        import ctypes as C
        n = len(sites)
        self.n = n
        self.total = sum(int(s[0].shape[0]) for s in sites)
        self.max_len = int(max_len)
        self.ids = (C.c_void_p * n)(*[_p(s[0]) for s in sites])
        self.count = (C.c_int64 * n)(*[int(s[0].shape[0]) for s in sites])
        self.row_base = (C.c_int32 * n)(*[int(s[1]) for s in sites])
        self.coef = (C.c_float * n)(*[float(s[2]) for s in sites])
        self._keep = sites


def sparse_adagrad_bags(E, acc, bias, bias_acc, vals, starts, lens, site_args, G, Gb, lr_dev, ws,
                        gscale_dev=None, phase=3, aux_cnt=None):
    """Multi-hot lookups of one table: merge per entity, then per token (arx.h).  phase 1: both
    sorts (ids only), 2: merge + apply, 3: both; same `ws` for both halves."""
    d = int(E.shape[1])
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_bags_workspace_bytes(site_args.total, site_args.max_len, d))
    call("arx_sparse_adagrad_bags", int(phase), _p(E), _p(acc), _p(bias), _p(bias_acc), int(E.shape[0]), d,
         _p(vals), _p(starts), _p(lens), int(lens.shape[0]), site_args.max_len, site_args.n, site_args.ids,
         site_args.count, site_args.row_base, site_args.coef, _p(G), _ld(G), _p(Gb), _p(lr_dev),
         _p(gscale_dev), _p(aux_cnt), wsp, wsn, _stream())


def segment_pool_fwd(scores, offs, W, mode, out, gmax=None):
    call("arx_segment_pool_fwd", _p(scores), _ld(scores), _p(offs), int(scores.shape[0]), int(W), int(mode),
         _p(gmax), _p(out), _ld(out), _stream())


def segment_pool_bwd(scores, offs, W, mode, out, dout, dscores, gmax=None, resid_rows=None):
    call("arx_segment_pool_bwd", _p(scores), _ld(scores), _p(offs), int(scores.shape[0]), int(W),
         int(dscores.shape[1]), int(mode), _p(gmax), _p(out), _ld(out), _p(dout), _ld(dout), _p(dscores),
         _ld(dscores), _p(resid_rows), _stream())


_reduce_scratch = {}


def new_reduce_scratch(device):
    """A zeroed arx_reduce_scratch_bytes() buffer: the caller-owned block partials + arrival ticket of the
    deterministic one-launch reductions (norms, running arg-max).  Launches that may overlap on the GPU need
    different buffers (a Runtime owns one); every call leaves it zeroed."""
    return torch.zeros(int(_lib.lib.arx_reduce_scratch_bytes()), dtype=torch.uint8, device=device)


def _rscratch(x, scratch):
    """scratch given by the caller, else ONE buffer per device (fine for launches on one stream only)."""
    if scratch is not None:
        return _p(scratch)
    s = _reduce_scratch.get(x.device)
    if s is None:
        s = _reduce_scratch[x.device] = new_reduce_scratch(x.device)
    return _p(s)


def max_argmax(x, col_base, first, best, best_idx, scratch=None):
    call("arx_max_argmax", _p(x), int(x.shape[0]), int(x.shape[1]), _ld(x), int(col_base), int(bool(first)),
         _p(best), _p(best_idx), _rscratch(x, scratch), _stream())


def gmax_residual_bwd(resid, idx, U, E_row, row_grad, bias_grad, dU):
    call("arx_gmax_residual_bwd", _p(resid), _p(idx), _p(U), _ld(U), _p(E_row), int(U.shape[1]), _p(row_grad),
         _p(bias_grad), _p(dU), _ld(dU) if dU is not None else 0, _stream())


def gmax_norm_corr(keys, src, coef, n, X, d, per_step, step_stride, Xb, per_step_b, stepb_stride, vrows, RG, RGb, corr):
    """corr[t] = change of a merged table gradient's squared norm from the arg-max residual rows (arx.h)."""
    call("arx_gmax_norm_corr", _p(keys), _p(src), _p(coef), int(n), _p(X), int(X.stride(-2)) if X is not None else 0,
         int(d), int(bool(per_step)), int(step_stride), _p(Xb), int(bool(per_step_b)), int(stepb_stride), _p(vrows),
         _p(RG), int(RG.stride(-2)) if RG is not None else 0, _p(RGb), int(vrows.shape[0]), _p(corr), _stream())


def adagrad_dense(w, acc, g, lr_dev, gscale_dev=None):
    call("arx_adagrad_dense", _p(w), _p(acc), _p(g), int(w.numel()), _p(lr_dev), _p(gscale_dev),
         _stream())


def adagrad_rows_nonzero(W, acc, bias, bias_acc, G, Gb, lr_dev):
    """Adagrad over the rows of W whose dense gradient row in G is not all zero; consumed rows of G / cells of Gb are
    zeroed (arx.h)."""
    call("arx_adagrad_rows_nonzero", _p(W), _p(acc), _p(bias), _p(bias_acc), _p(G), _p(Gb), int(W.shape[0]),
         int(W.shape[1]), _p(lr_dev), _stream())


def adagrad_dense_multi(params, lr_dev, gscale_dev=None):
    """params: [(w, acc|None, g)]: one launch per 8 parameters (arx_adagrad_dense_multi)."""
    import ctypes as C
    for k in range(0, len(params), 8):
        grp = params[k:k + 8]
        m = len(grp)
        ws_ = (C.c_void_p * m)(*[_p(p[0]) for p in grp])
        accs = (C.c_void_p * m)(*[(_p(p[1]) or None) for p in grp])
        gs = (C.c_void_p * m)(*[_p(p[2]) for p in grp])
        ns = (C.c_int64 * m)(*[int(p[0].numel()) for p in grp])
        call("arx_adagrad_dense_multi", m, ws_, accs, gs, ns, _p(lr_dev), _p(gscale_dev), _stream())


def sq_norm_accum(x, out_accum, d=1, row_scale=None, n=None, scratch=None):
    call("arx_sq_norm_accum", _p(x), int(x.numel() if n is None else n), int(d), _p(row_scale),
         _p(out_accum), _rscratch(x, scratch), _stream())


def sq_norm_accum_multi(items, out_accum, scratch=None):
    """items: [(x, d, row_scale|None, n|None)]: out += sum over all of them, <= 8 per launch."""
    import ctypes as C
    for k in range(0, len(items), 8):
        grp = items[k:k + 8]
        m = len(grp)
        xs = (C.c_void_p * m)(*[_p(g[0]) for g in grp])
        ns = (C.c_int64 * m)(*[int(g[0].numel() if g[3] is None else g[3]) for g in grp])
        ds = (C.c_int * m)(*[int(g[1]) for g in grp])
        rs = (C.c_void_p * m)(*[(_p(g[2]) or None) for g in grp])
        call("arx_sq_norm_accum_multi", m, xs, ns, ds, rs, _p(out_accum), _rscratch(out_accum, scratch), _stream())


def sq_norm_clip_multi(items, sqnorm, max_norm, coef_out, gnorm_out=None, init=True, scratch=None):
    """sq_norm_accum_multi + clip_coef in as few launches as tensors / 8: the LAST launch also forms
    coef = max_norm / max(||g||, max_norm) (arx_sq_norm_clip_multi); init: sqnorm is overwritten by the
    first launch instead of accumulated onto (no fill launch)."""
    import ctypes as C
    groups = [items[k:k + 8] for k in range(0, len(items), 8)]
    if not groups:
        if init:
            fill_f32(sqnorm, 0.0)
        clip_coef(sqnorm, max_norm, coef_out, gnorm_out)
        return
    for gi, grp in enumerate(groups):
        m = len(grp)
        xs = (C.c_void_p * m)(*[_p(g[0]) for g in grp])
        ns = (C.c_int64 * m)(*[int(g[0].numel() if g[3] is None else g[3]) for g in grp])
        ds = (C.c_int * m)(*[int(g[1]) for g in grp])
        rs = (C.c_void_p * m)(*[(_p(g[2]) or None) for g in grp])
        first, last = gi == 0, gi == len(groups) - 1
        if last:
            call("arx_sq_norm_clip_multi", m, xs, ns, ds, rs, int(bool(init and first)), _p(sqnorm), float(max_norm),
                 _p(coef_out), _p(gnorm_out), _rscratch(sqnorm, scratch), _stream())
        elif init and first:
            fill_f32(sqnorm, 0.0)
            call("arx_sq_norm_accum_multi", m, xs, ns, ds, rs, _p(sqnorm), _rscratch(sqnorm, scratch), _stream())
        else:
            call("arx_sq_norm_accum_multi", m, xs, ns, ds, rs, _p(sqnorm), _rscratch(sqnorm, scratch), _stream())


def merged_sq_norm(keys, src, coef, table_rows, out_accum, ws, X=None, d=0, L=1, step_stride=0,
                   Xb=None, Lb=1, stepb_stride=0, n=None, scratch=None):
    """out += sum_t sum_rows ||sum_{c -> row} coef_c X_t[src_c]||^2 (+ the d = 1 analogue on Xb):
    the norm of a table gradient after merging contributions per table row."""
    n = int(keys.shape[0]) if n is None else int(n)
    wsp, wsn = ws.get(_lib.lib.arx_sparse_adagrad_workspace_bytes(n))
    call("arx_merged_sq_norm", _p(keys), _p(src), _p(coef), n, key_bits_for(table_rows),
         _p(X), int(X.stride(-2)) if X is not None else 0, int(d), int(L), int(step_stride),
         _p(Xb), int(Lb), int(stepb_stride), _p(out_accum), wsp, wsn, _rscratch(out_accum, scratch), _stream())


def clip_coef(sqnorm, max_norm, coef_out, gnorm_out=None):
    call("arx_clip_coef", _p(sqnorm), float(max_norm), _p(coef_out), _p(gnorm_out), _stream())


# ---- utilities -------------------------------------------------------------------------
def fill_f32(t, v):
    call("arx_fill_f32", _p(t), int(t.numel()), float(v), _stream())


def fill_i32(t, v):
    call("arx_fill_i32", _p(t), int(t.numel()), int(v), _stream())


def fill_u8(t, v):
    call("arx_fill_u8", _p(t), int(t.numel()), int(v), _stream())


def axpby(a, x, b, y, n=None):
    call("arx_axpby", float(a), _p(x), float(b), _p(y), int(y.numel() if n is None else n), _stream())


def add_rows_bcast(a, x, b, y):
    call("arx_add_rows_bcast", float(a), _p(x), _ld(x), int(x.shape[0]), float(b), _p(y), _ld(y),
         int(y.shape[0]), int(y.shape[1]), _stream())


def row_sum(x, out, accumulate=False):
    call("arx_row_sum", _p(x), _ld(x), int(x.shape[0]), int(x.shape[1]), _p(out),
         int(bool(accumulate)), _stream())


def col_sum(x, out, ws):
    rows, cols = int(x.shape[0]), int(x.shape[1])
    wsp, wsn = ws.get(_lib.lib.arx_col_sum_workspace_bytes(rows, cols))
    call("arx_col_sum", _p(x), _ld(x), rows, cols, _p(out), wsp, wsn, _stream())


def sum_scaled(x, scale, out, n=None):
    call("arx_sum_scaled", _p(x), int(x.numel() if n is None else n), float(scale), _p(out), _stream())


def dropout_fwd(x, keep_prob, seed, y, keep_mask):
    call("arx_dropout_fwd", _p(x), int(x.numel()), float(keep_prob), int(seed), _p(y),
         _p(keep_mask), _stream())


def dropout_fwd_step(x, keep_prob, seed, step_dev, y, keep_mask):
    """dropout whose seed also takes a device-side step counter (int64 tensor [1])."""
    call("arx_dropout_fwd_step", _p(x), int(x.numel()), float(keep_prob), int(seed) & (2 ** 64 - 1),
         _p(step_dev), _p(y), _p(keep_mask), _stream())


def merge_keyed_take(keys, ids, S, out):
    """out[r] = id of the r-th smallest (key, position) pair, r < S (arx.h): the merge of sorted race lists.
    keys: float32 >= 0 (race keys: the kernel orders their bit patterns), ids / out: int32."""
    if keys.dtype != torch.float32 or ids.dtype != torch.int32 or out.dtype != torch.int32:
        raise TypeError("merge_keyed_take: keys float32, ids / out int32 expected (got %s, %s, %s)"
                        % (keys.dtype, ids.dtype, out.dtype))
    if ids.shape[0] != keys.shape[0] or out.shape[0] < int(S):
        raise ValueError("merge_keyed_take: ids must match keys, out must hold S entries")
    call("arx_merge_keyed_take", _p(keys), _p(ids), int(keys.shape[0]), int(S), _p(out), _stream())


def take_i32(table, idx, out, fill=-1):
    call("arx_take_i32", _p(table), _p(idx), int(idx.shape[0]), int(fill), _p(out), _stream())


def copy_words(pairs):
    """[(src, dst), ...] device tensors of 4-byte elements, equal sizes per pair; <= 8 per launch."""
    import ctypes as C
    for k in range(0, len(pairs), 8):
        grp = pairs[k:k + 8]
        n = len(grp)
        src, dst, cnt = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
        for a, (s_, d_) in enumerate(grp):
            if s_.element_size() != 4 or d_.element_size() != 4 or s_.numel() != d_.numel():
                raise ValueError("copy_words: 4-byte tensors of equal size expected")
            src[a], dst[a], cnt[a] = s_.data_ptr(), d_.data_ptr(), s_.numel()
        call("arx_copy_words", n, src, dst, cnt, _stream())


def counter_add(counter_dev, v=1):
    call("arx_counter_add", _p(counter_dev), int(v), _stream())


def dropout_bwd(dy, keep_mask, keep_prob, dx):
    call("arx_dropout_bwd", _p(dy), _p(keep_mask), int(dy.numel()), float(keep_prob), _p(dx), _stream())


def act_fwd(x, kind, y):
    call("arx_act_fwd", _p(x), int(x.numel()), int(kind), _p(y), _stream())


def act_bwd(y, dy, kind, dx):
    call("arx_act_bwd", _p(y), _p(dy), int(y.numel()), int(kind), _p(dx), _stream())


def topk_chunk(logits, k, idx_base, values, indices):
    call("arx_topk_chunk", _p(logits), _ld(logits), int(logits.shape[0]), int(logits.shape[1]), int(k),
         int(idx_base), _p(values), _p(indices), _stream())


def topk_merge(va, ia, vb, ib, k, vo, io):
    call("arx_topk_merge", _p(va), _p(ia), _p(vb), _p(ib), int(va.shape[0]), int(va.shape[1]),
         int(vb.shape[1]), int(k), _p(vo), _p(io), _stream())


def gemm_nt_topk_parts(M, N):
    """Column ranges arx_gemm_nt_topk_filter splits N columns into for M rows (sizes the candidate rows)."""
    import ctypes as C
    n = C.c_int(0)
    call("arx_gemm_nt_topk_parts", int(M), int(N), C.byref(n))
    return int(n.value)


def gemm_nt_topk_filter(A, Bm, col_bias, thr, col_base, cand_v, cand_i, capp, overflow, lse_part=None):
    """Scorer GEMM A . Bm^T + col_bias that keeps only the logits above thr[row]; lse_part [M, >= parts]: also the
    per-column-range log-sum-exp of every row (arx.h)."""
    call("arx_gemm_nt_topk_filter", _p(A), _ld(A), int(A.shape[0]), _p(Bm), _ld(Bm), int(Bm.shape[0]), int(A.shape[1]),
         _p(col_bias), _p(thr), int(thr.stride(0)), int(col_base), _p(cand_v), _p(cand_i), int(cand_v.stride(0)),
         int(capp), _p(overflow), _p(lse_part), int(lse_part.stride(0)) if lse_part is not None else 0, _stream())


def gemm_nt_eval_parts(A, Bm, col_bias, tscore, lse_part, relu_part):
    """Full-vocabulary evaluation sums out of the scorer GEMM, no logits (arx.h)."""
    ref = lse_part if lse_part is not None else relu_part
    call("arx_gemm_nt_eval_parts", _p(A), _ld(A), int(A.shape[0]), _p(Bm), _ld(Bm), int(Bm.shape[0]), int(A.shape[1]),
         _p(col_bias), _p(tscore), _p(lse_part), _p(relu_part), int(ref.stride(0)), _stream())


def take_rows_i32(table, pos, out):
    call("arx_take_rows_i32", _p(table), int(table.stride(0)), _p(pos), int(pos.stride(0)), int(pos.shape[0]),
         int(pos.shape[1]), _p(out), int(out.stride(0)), _stream())


def topk(logits, k, values, indices):
    call("arx_topk", _p(logits), _ld(logits), int(logits.shape[0]), int(logits.shape[1]), int(k),
         _p(values), _p(indices), _stream())


def lstm_fwd(x, W, b, L, B, din, h, forget_bias, hs, cs, gates):
    call("arx_lstm_fwd", _p(x), _p(W), _p(b), int(L), int(B), int(din), int(h), float(forget_bias),
         _p(hs), _p(cs), _p(gates), _stream())


def lstm_bwd(W, hs, cs, gates, dhs, L, B, din, h, dz, wxt=None):
    """dz of the whole unrolled cell; wxt (optional, [4h, din]) receives W[:din]^T on the way."""
    call("arx_lstm_bwd_wxt", _p(W), _p(hs), _p(cs), _p(gates), _p(dhs), int(L), int(B), int(din),
         int(h), _p(dz), _p(wxt), _stream())


def seq_weights(w, L, B, out):
    call("arx_seq_weights", _p(w), int(L), int(B), _p(out), _stream())


class CapturedGraph(object):
    """A hipGraph of one step (arx_capture_* in include/arx.h).  end(feeds=...): the placeholder feeds issued inside
    the capture (copy_words) stay addressable -- set_feeds() swaps their sources before a replay."""

    def __init__(self):
        import ctypes as C
        self._exec = C.c_void_p(0)
        self._feeds = C.c_void_p(0)
        self.feed_groups = None      # per captured copy node: the destination pointers of its (<= 8) feeds, in order
        self._node_of = []           # group index -> node index
        self._fed = False            # the nodes hold live sources (else: they copy nothing)

    def begin(self):
        call("arx_capture_begin", _stream())

    def end(self, feeds=None):
        """feeds: the [(src, dst), ...] list copy_words() was called with inside this capture (None: no feed nodes)."""
        import ctypes as C
        if not feeds:
            call("arx_capture_end", _stream(), C.byref(self._exec))
            return
        n = C.c_int(0)
        call("arx_capture_end_feeds", _stream(), C.byref(self._exec), C.byref(self._feeds), C.byref(n))
        groups = [feeds[k:k + 8] for k in range(0, len(feeds), 8)]
        first = {}
        for i in range(n.value):
            d0 = C.c_void_p(0)
            call("arx_graph_feed_dst0", self._feeds, i, C.byref(d0))
            first[d0.value] = i
        node_of = [first.get(g[0][1].data_ptr()) for g in groups]
        if n.value != len(groups) or None in node_of or len(set(node_of)) != len(groups):
            raise RuntimeError("captured step: %d feed node(s) found for %d feed group(s)" % (n.value, len(groups)))
        self._node_of = node_of
        self.feed_groups = [tuple((d.data_ptr(), d.numel()) for _, d in g) for g in groups]
        self._fed = True

    def feeds_match(self, feeds):
        """Do these pending feeds address exactly the destinations the capture's feed nodes write?"""
        if self.feed_groups is None:
            return False
        groups = [feeds[k:k + 8] for k in range(0, len(feeds), 8)]
        return [tuple((d.data_ptr(), d.numel()) for _, d in g) for g in groups] == self.feed_groups

    def set_feeds(self, feeds):
        """The sources the feed nodes copy at the next launches; None / []: they copy nothing."""
        import ctypes as C
        if self.feed_groups is None or (not feeds and not self._fed):
            return
        for gi, ni in enumerate(self._node_of):
            grp = feeds[gi * 8:(gi + 1) * 8] if feeds else []
            n = len(grp)
            src, dst, cnt = (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))(), (C.c_int64 * max(n, 1))()
            for a, (s_, d_) in enumerate(grp):
                if s_.element_size() != 4 or d_.element_size() != 4 or s_.numel() != d_.numel():
                    raise ValueError("set_feeds: 4-byte tensors of equal size expected")
                src[a], dst[a], cnt[a] = s_.data_ptr(), d_.data_ptr(), s_.numel()
            call("arx_graph_set_feed", self._exec, self._feeds, ni, n, src, dst, cnt)
        self._fed = bool(feeds)

    def launch(self):
        call("arx_graph_launch", self._exec, _stream())

    def __del__(self):
        try:
            if self._exec:
                _lib.lib.arx_graph_destroy(self._exec)
            if self._feeds:
                _lib.lib.arx_graph_feeds_destroy(self._feeds)
        except Exception:
            pass
