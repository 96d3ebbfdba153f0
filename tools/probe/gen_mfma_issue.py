#!/usr/bin/env python
"""Generates tools/probe/mfma_issue.hip: an issue-slot probe for v_mfma_f32_32x32x16_bf16 on gfx950.

The bf16-pipe scorer kernels (csrc/gemm_bx6.hip) sat at ~24 % MFMA-busy in round 3.  This probe measures, one
variable at a time and with the instruction stream written out by hand (one asm block per loop body, so the
compiler cannot re-order or fold anything), what ONE wave per SIMD can issue between MFMAs for free:

  * the accumulator pattern of a 6-MFMA chunk (same accumulator back to back / never closer than 2 apart / six
    independent accumulators),
  * fillers per MFMA gap: v_fma_f32, v_pk_fma_f32, integer VALU, the act-bit expansion (VALU that writes the
    bf16 operand of an MFMA three places later), the hinge-epilogue ops, ds_read_b128,
  * a loader wave sharing the SIMD (global_load_dwordx4 -> ds_write_b128, one barrier per 48 MFMAs).

Output of the built program: shader cycles per MFMA (s_memtime around the loop) per variant.
  python tools/probe/gen_mfma_issue.py > tools/probe/mfma_issue.hip && hipcc --offload-arch=gfx950 -O3 ...
"""
import sys

ACC3 = [0, 1, 2, 0, 1, 0, 1, 0, 2, 1, 0, 1]         # X Y H X Y X | Y X H Y X Y
ACC2 = [0, 0, 0, 0, 0, 1] * 2                        # L L L L L H
ACC6 = [0, 1, 2, 3, 4, 5] * 2


def body(acc, fillers, barrier_every=0):
    """One loop body of 12 MFMAs.  fillers(j) -> list of asm lines placed after MFMA j."""
    out = []
    for j in range(12):
        out.append("v_mfma_f32_32x32x16_bf16 %%[c%d], %%[a%d], %%[b%d], %%[c%d]" % (acc[j], j % 6, j % 6, acc[j]))
        out += fillers(j)
    return out


def body_consume(acc, fillers, dist=1, vmem=False):
    """12 MFMAs = 2 chunks of 6; each chunk's three A operands come from ds_read_b128 issued `dist` chunk(s) earlier
    (q0..q2 / q3..q5 alternate), s_waitcnt lgkmcnt before the chunk's first MFMA: the kernel's real dependency."""
    out = []
    for g in range(2):
        rd = (g + dist) & 1 if dist == 1 else g & 1
        base = 3 * rd
        for k in range(3):
            out.append("ds_read_b128 %%[q%d], %%[la] offset:%d" % (base + k, 1024 * (3 * g + k)))
        out.append("s_waitcnt lgkmcnt(3)")
        use = 3 * (g & 1)
        for j in range(6):
            out.append("v_mfma_f32_32x32x16_bf16 %%[c%d], %%[q%d], %%[b%d], %%[c%d]" % (acc[6 * g + j], use + j % 3, j, acc[6 * g + j]))
            out += fillers(6 * g + j)
    return out


def body_pingpong(n_epi, vgap=2, reinit=True, inplace=False):
    """24 MFMAs: 12 on tile X (v[32:47] hi, v[48:63] lo) with the hinge epilogue of tile Y (v[0:15], v[16:31])
    between them, then 12 on Y with the epilogue of X -- the accumulators the VALU reads WERE written by MFMAs one
    half earlier, as in k_sc_hinge."""
    out = []
    for half in range(2):
        hi, lo = (32, 48) if half == 0 else (0, 16)
        yh, yl = (0, 16) if half == 0 else (32, 48)
        for j in range(12):
            acc = "v[%d:%d]" % (lo, lo + 15) if j % 6 != 1 else "v[%d:%d]" % (hi, hi + 15)
            out.append("v_mfma_f32_32x32x16_bf16 %s, %%[a%d], %%[b%d], %s" % (acc, j % 6, j % 6, acc))
            if j < n_epi:
                i = 15 - j
                if inplace:          # hipcc's allocation: nv lands in the accumulator register it was read from
                    out += ["v_sub_f32 v%d, -v%d, v%d" % (yh + i, yh + i, yl + i),
                            "v_alignbit_b32 %%[i0], %%[i0], v%d, 31" % (yh + i),
                            "v_max_f32 v%d, -v%d, 0" % (yh + i, yh + i),
                            "v_add_f32 %%[f7], %%[f7], v%d" % (yh + i)]
                else:
                    out += ["v_sub_f32 %%[f0], -v%d, v%d" % (yh + i, yl + i),
                            "v_alignbit_b32 %[i0], %[i0], %[f0], 31",
                            "v_max_f32 %[f1], -%[f0], 0",
                            "v_add_f32 %[f7], %[f7], %[f1]"]
                if reinit:
                    out += ["v_sub_f32 v%d, %%[k0], %%[k1]" % (yh + i), "v_mov_b32 v%d, 0" % (yl + i)]
    return out


def body_manyb(n_epi=0, nb=24, base=128, lds=True):
    """48 MFMAs of one hinge tile as the compiler lays it out: per chunk of 6 the A operands are three LDS fragments
    (q regs, read one chunk ahead), the B operands walk through nb DIFFERENT register quads pinned at v[base ..]
    (the U pieces a1 / a2 / a3 of 8 chunks: 24 quads = 96 registers) in the kernel's order; accumulators X pinned
    v[32:63]; optional epilogue on tile Y v[0:31]."""
    out = []
    pat = [(2, 0, 1), (0, 0, 0), (0, 2, 1), (1, 1, 1), (1, 0, 1), (0, 1, 1)]     # (frag, piece, lo?) per MFMA
    for c in range(8):
        rd = 3 * ((c + 1) & 1)
        if lds:
            for k in range(3):
                out.append("ds_read_b128 %%[q%d], %%[la] offset:%d" % (rd + k, 1024 * (3 * (c % 5) + k)))
            out.append("s_waitcnt lgkmcnt(3)")
        use = 3 * (c & 1)
        for m, (fr, pc, lo) in enumerate(pat):
            bq = base + 4 * ((3 * c + pc) % nb)
            acc = "v[48:63]" if lo else "v[32:47]"
            out.append("v_mfma_f32_32x32x16_bf16 %s, %%[q%d], v[%d:%d], %s" % (acc, use + fr, bq, bq + 3, acc))
            j = 6 * c + m
            if j % 3 == 0 and j // 3 < n_epi:
                i = 15 - j // 3
                out += ["v_sub_f32 %%[f0], -v%d, v%d" % (i, 16 + i),
                        "v_alignbit_b32 %[i0], %[i0], %[f0], 31",
                        "v_max_f32 %[f1], -%[f0], 0",
                        "v_add_f32 %[f7], %[f7], %[f1]",
                        "v_sub_f32 v%d, %%[k0], %%[k1]" % i, "v_mov_b32 v%d, 0" % (16 + i)]
    return out


def f_none(j):
    return []


def f_fma(n):
    def f(j):
        return ["v_fma_f32 %%[f%d], %%[f%d], %%[k0], %%[k1]" % ((j * n + q) % 8, (j * n + q) % 8) for q in range(n)]
    return f


def f_pk(n):
    def f(j):
        return ["v_pk_fma_f32 %%[p%d], %%[p%d], %%[pk], %%[pk]" % ((j * n + q) % 4, (j * n + q) % 4) for q in range(n)]
    return f


def f_int(n):
    def f(j):
        r = []
        for q in range(n):
            i = (j * n + q) % 8
            r.append(("v_and_b32 %%[i%d], %%[i%d], %%[k2]" if q % 2 == 0 else "v_mul_u32_u24 %%[i%d], %%[i%d], %%[k3]") % (i, i))
        return r
    return f


def f_expand(per):
    """The act-bit expansion: 8 bits of a word -> four dwords of two bf16 0/1 each (1 + 8 ops) + 2 ops of word
    bookkeeping, writing the A operand of the MFMA three places later.  per = MFMAs per expansion."""
    def f(j):
        if j % per:
            return []
        d = (j + 3) % 6
        r = ["v_lshrrev_b32 %[i0], 8, %[i0]", "v_and_b32 %[i1], 0xff, %[i0]", "v_lshl_or_b32 %[i1], %[i1], 15, %[i1]"]
        for q, (m, k) in enumerate(((0x00010001, 0x3F80), (0x00040004, 0x0FE0), (0x00100010, 0x03F8), (0x00400040, 0x00FE))):
            r.append("v_and_b32 %%[i2], 0x%x, %%[i1]" % m)
            r.append("v_mul_u32_u24 v%d, 0x%x, %%[i2]" % (100 + 4 * d + q, k))     # (A[d] is pinned to v[100+4d : 103+4d])
        return r
    return f


def f_hinge(n):
    def f(j):
        r = []
        for q in range(n):
            i = (j * n + q) % 8
            r += ["v_sub_f32 %%[f%d], %%[k0], %%[f%d]" % (i, i),
                  "v_alignbit_b32 %%[i0], %%[i0], %%[f%d], 31" % i,
                  "v_max_f32 %%[f%d], %%[f%d], %%[k1]" % ((i + 1) % 8, i),
                  "v_add_f32 %[f7], %[f7], %[k0]"]
        return r
    return f


def f_acc_read(n):
    """VALU that READS registers of an idle accumulator tile (pinned v[140:171]) -- the hinge epilogue's loads"""
    def f(j):
        return ["v_add_f32 %%[f%d], v%d, v%d" % ((j * n + q) % 8, 140 + (j * n + q) % 16, 156 + (j * n + q) % 16) for q in range(n)]
    return f


def f_acc_write(n):
    """VALU that WRITES registers of an idle accumulator tile (pinned v[140:171]) -- the re-initialisation"""
    def f(j):
        return ["v_sub_f32 v%d, %%[k0], %%[f%d]" % (140 + (j * n + q) % 32, (j * n + q) % 8) for q in range(n)]
    return f


def f_epi(n, base=140):
    """the real epilogue per value: read both tiles' element, alignbit, max, add, re-init hi, zero lo (6 VALU)"""
    def f(j):
        r = []
        for q in range(n):
            i = (j * n + q) % 16
            r += ["v_sub_f32 %%[f0], -v%d, v%d" % (base + i, base + 16 + i),
                  "v_alignbit_b32 %[i0], %[i0], %[f0], 31",
                  "v_max_f32 %[f1], -%[f0], 0",
                  "v_add_f32 %[f7], %[f7], %[f1]",
                  "v_sub_f32 v%d, %%[k0], %%[k1]" % (base + i),
                  "v_mov_b32 v%d, 0" % (base + 16 + i)]
        return r
    return f


def f_ds(per, wait=True):
    def f(j):
        if j % per:
            return []
        r = ["ds_read_b128 %%[q%d], %%[la] offset:%d" % ((j // per) % 4, 1024 * (j % 8))]
        if wait and (j // per) % 4 == 3:
            r.append("s_waitcnt lgkmcnt(2)")
        return r
    return f


def f_nop(n):
    def f(j):
        return ["s_nop 0"] * n
    return f


def combine(*fs):
    def f(j):
        r = []
        for x in fs:
            r += x(j)
        return r
    return f


VARIANTS = [
    # name, acc pattern, fillers, loader wave (0 / 1), threads
    ("six accumulators, bare", ACC6, f_none, 0),
    ("3 acc rotation (same acc >= 2 apart), bare", ACC3, f_none, 0),
    ("L L L L L H (same acc back to back), bare", ACC2, f_none, 0),
    ("LLLLLH + 1 s_nop / MFMA", ACC2, f_nop(1), 0),
    ("LLLLLH + 1 v_fma / MFMA", ACC2, f_fma(1), 0),
    ("LLLLLH + 2 v_fma / MFMA", ACC2, f_fma(2), 0),
    ("LLLLLH + 4 v_fma / MFMA", ACC2, f_fma(4), 0),
    ("LLLLLH + 1 ds_read_b128 / 2 MFMA", ACC2, f_ds(2), 0),
    ("3 acc + 1 s_nop / MFMA", ACC3, f_nop(1), 0),
    ("3 acc + 1 v_fma / MFMA", ACC3, f_fma(1), 0),
    ("3 acc + 2 v_fma / MFMA", ACC3, f_fma(2), 0),
    ("3 acc + 3 v_fma / MFMA", ACC3, f_fma(3), 0),
    ("3 acc + 4 v_fma / MFMA", ACC3, f_fma(4), 0),
    ("3 acc + 5 v_fma / MFMA", ACC3, f_fma(5), 0),
    ("3 acc + 6 v_fma / MFMA", ACC3, f_fma(6), 0),
    ("3 acc + 8 v_fma / MFMA", ACC3, f_fma(8), 0),
    ("6 acc + 4 v_fma / MFMA", ACC6, f_fma(4), 0),
    ("6 acc + 6 v_fma / MFMA", ACC6, f_fma(6), 0),
    ("3 acc + 1 v_pk_fma / MFMA", ACC3, f_pk(1), 0),
    ("3 acc + 2 v_pk_fma / MFMA", ACC3, f_pk(2), 0),
    ("3 acc + 2 int VALU / MFMA", ACC3, f_int(2), 0),
    ("3 acc + 4 int VALU / MFMA", ACC3, f_int(4), 0),
    ("3 acc + 1 hinge value (4 VALU) / MFMA", ACC3, f_hinge(1), 0),
    ("3 acc + expansion (11 VALU) / 3 MFMA", ACC3, f_expand(3), 0),
    ("3 acc + expansion (11 VALU) / 2 MFMA", ACC3, f_expand(2), 0),
    ("3 acc + expansion (11 VALU) / 1 MFMA", ACC3, f_expand(1), 0),
    ("3 acc + 1 ds_read_b128 / MFMA", ACC3, f_ds(1), 0),
    ("3 acc + 1 ds_read_b128 / 2 MFMA", ACC3, f_ds(2), 0),
    ("3 acc + ds_read / 2 MFMA + expansion / 3 MFMA", ACC3, combine(f_ds(2), f_expand(3)), 0),
    ("3 acc + ds_read / 2 MFMA + 1 hinge value / MFMA", ACC3, combine(f_ds(2), f_hinge(1)), 0),
    ("3 acc bare + loader wave on the SIMD", ACC3, f_none, 1),
    ("3 acc + ds_read / 2 MFMA + loader wave", ACC3, f_ds(2), 1),
    ("3 acc + ds_read / 2 + 1 hinge value + loader wave", ACC3, combine(f_ds(2), f_hinge(1)), 1),
    ("3 acc + ds_read / 2 + expansion / 3 + loader wave", ACC3, combine(f_ds(2), f_expand(3)), 1),
    ("LLLLLH + ds_read / 2 MFMA + loader wave", ACC2, f_ds(2), 1),
    # accumulators in VGPRs instead of AGPRs (what hipcc picks for a kernel whose budget is <= 256 registers)
    ("VGPR acc: 3 acc bare", ACC3, f_none, 0, "v"),
    ("VGPR acc: 3 acc + 2 v_fma / MFMA", ACC3, f_fma(2), 0, "v"),
    ("VGPR acc: 3 acc + 4 v_fma / MFMA", ACC3, f_fma(4), 0, "v"),
    ("VGPR acc: LLLLLH + 2 v_fma / MFMA", ACC2, f_fma(2), 0, "v"),
    ("VGPR acc: 3 acc + 1 hinge value (4 VALU) / MFMA", ACC3, f_hinge(1), 0, "v"),
    ("VGPR acc: 3 acc + ds_read / 2 + 1 hinge value", ACC3, combine(f_ds(2), f_hinge(1)), 0, "v"),
    ("VGPR acc: 3 acc + expansion / 3 MFMA", ACC3, f_expand(3), 0, "v"),
    ("VGPR acc: 3 acc + ds_read / 2 + hinge + loader", ACC3, combine(f_ds(2), f_hinge(1)), 1, "v"),
    ("CONSUMED frags (3 ds_read / 6 MFMA, 1 chunk ahead), bare", ACC3, f_none, 0, "v", "consume"),
    ("CONSUMED frags + 2 v_fma / MFMA", ACC3, f_fma(2), 0, "v", "consume"),
    ("CONSUMED frags + real epilogue 1 value / 3 MFMA", ACC3, lambda j: f_epi(1)(j) if j % 3 == 0 else [], 0, "v", "consume"),
    ("CONSUMED frags, bare + loader wave", ACC3, f_none, 1, "v", "consume"),
    ("CONSUMED frags + epilogue + loader wave", ACC3, lambda j: f_epi(1)(j) if j % 3 == 0 else [], 1, "v", "consume"),
    ("VGPR acc + 2 VALU READING an idle acc tile / MFMA", ACC3, f_acc_read(2), 0, "v"),
    ("VGPR acc + 2 VALU WRITING an idle acc tile / MFMA", ACC3, f_acc_write(2), 0, "v"),
    ("VGPR acc + real epilogue, 1 value (6 VALU) / 3 MFMA", ACC3, lambda j: f_epi(1)(j) if j % 3 == 0 else [], 0, "v"),
    ("VGPR acc + real epilogue, 1 value (6 VALU) / 3 MFMA + ds/2", ACC3, combine(f_ds(2), lambda j: f_epi(1)(j) if j % 3 == 0 else []), 0, "v"),
    ("AGPR acc + real epilogue, 1 value (6 VALU) / 3 MFMA", ACC3, lambda j: f_epi(1)(j) if j % 3 == 0 else [], 0, "a"),
    ("AGPR acc + 2 VALU WRITING an idle VGPR tile / MFMA", ACC3, f_acc_write(2), 0, "a"),
    # accumulators PINNED to v[32:79]; the idle tile next to them (v[0:31], as hipcc allocates the two sets of
    # k_sc_hinge) or far away (v[140:171])
    ("PINNED acc v[32:79] + epilogue on v[0:31] (adjacent), 1 value / 3 MFMA", ACC3, lambda j: f_epi(1, 0)(j) if j % 3 == 0 else [], 0, "pin"),
    ("PINNED acc v[32:79] + epilogue on v[140:171] (far), 1 value / 3 MFMA", ACC3, lambda j: f_epi(1, 140)(j) if j % 3 == 0 else [], 0, "pin"),
    ("PINNED acc v[32:79] + epilogue on v[80:111], 1 value / 3 MFMA", ACC3, lambda j: f_epi(1, 80)(j) if j % 3 == 0 else [], 0, "pin"),
    ("PINNED acc v[32:79] bare", ACC3, f_none, 0, "pin"),
    ("PING-PONG tiles X / Y, no epilogue", None, None, 0, "pp", 0),
    ("PING-PONG tiles X / Y, epilogue of the OTHER tile: 12 values / 12 MFMA, with re-init", None, None, 0, "pp", 12),
    ("PING-PONG tiles X / Y, epilogue 12 values / 12 MFMA, READ only (no re-init)", None, None, 0, "pp", -12),
    ("PING-PONG tiles X / Y, epilogue 6 values / 12 MFMA, with re-init", None, None, 0, "pp", 6),
    ("TILE as compiled: 24 B quads v[128:223], LDS frags, no epilogue", None, None, 0, "mb", (0, 24, 128, True)),
    ("TILE as compiled: 24 B quads, LDS frags, epilogue 16 values", None, None, 0, "mb", (16, 24, 128, True)),
    ("TILE: 24 B quads, frags NOT reloaded, no epilogue", None, None, 0, "mb", (0, 24, 128, False)),
    ("TILE: 3 B quads (reused), LDS frags, no epilogue", None, None, 0, "mb", (0, 3, 128, True)),
    ("TILE: 24 B quads at v[64:159], LDS frags, epilogue 16", None, None, 0, "mb", (16, 24, 64, True)),
    ("TILE as compiled + loader wave", None, None, 1, "mb", (16, 24, 128, True)),
    ("PING-PONG, epilogue 12 values / 12 MFMA IN PLACE (nv written over the accumulator element)", None, None, 0, "pp", 112),
]

HEAD = r'''// GENERATED by tools/probe/gen_mfma_issue.py -- do not edit.  Issue-slot probe for v_mfma_f32_32x32x16_bf16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// loader waves (threads >= 256): per stage 12 x global_load_dwordx4 (L2-resident planes) -> 12 x ds_write_b128,
// one barrier per stage -- what the movers of k_nt_hinge_bx6 do
__device__ __forceinline__ void loader(const uint4* __restrict__ src, uint4* lds, int stages, int lt) {
  const uint4* g = src + lt;
  for (int st = 0; st < stages; ++st) {
    uint4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = g[(j * 256 + st * 64) & 8191];
#pragma unroll
    for (int j = 0; j < 12; ++j) lds[2048 + ((j * 256 + lt) & 2047)] = v[j];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}
'''

KERNEL = r'''
__global__ __launch_bounds__(%(threads)d) void k%(idx)d(float* out, unsigned long long* cyc, const uint4* src, int iters) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += %(threads)d) lds[i] = src[i & 255];
  __syncthreads();
  if (tid >= 256) {
    if (%(loader)d) loader(src, lds, iters / 4, tid - 256);
    return;
  }
  uint32_t a[6][4], b[6][4];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const uint4 x = src[(tid + 64 * j) & 255], y = src[(tid + 64 * j + 32) & 255];
    a[j][0] = x.x; a[j][1] = x.y; a[j][2] = x.z; a[j][3] = x.w;
    b[j][0] = y.x; b[j][1] = y.y; b[j][2] = y.z; b[j][3] = y.w;
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 A[6], B[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    A[j] = (u32x4){a[j][0], a[j][1], a[j][2], a[j][3]};
    B[j] = (u32x4){b[j][0], b[j][1], b[j][2], b[j][3]};
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0};
  float f0 = 1.f + tid, f1 = 2.f, f2 = 3.f, f3 = 4.f, f4 = 5.f, f5 = 6.f, f6 = 7.f, f7 = 8.f;
  f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, pk = {1.0001f, 0.5f};
  uint32_t i0 = src[tid & 255].x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
  u32x4 q0 = A[0], q1 = A[1], q2 = A[2], q3 = A[3], q4 = A[4], q5 = A[5];
  const float k0 = 1.0001f, k1 = 0.5f;
  const uint32_t k2 = 0x00ffff0fu, k3 = 0x3f81u;
  const uint32_t la = (tid & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(
%(asm)s
        : [c0] "+%(a0)s"(c0), [c1] "+%(a1)s"(c1), [c2] "+%(a2)s"(c2), [c3] "+%(a3)s"(c3), [c4] "+%(a4)s"(c4), [c5] "+%(a4)s"(c5),
          [a0] "+{v[100:103]}"(A[0]), [a1] "+{v[104:107]}"(A[1]), [a2] "+{v[108:111]}"(A[2]), [a3] "+{v[112:115]}"(A[3]),
          [a4] "+{v[116:119]}"(A[4]), [a5] "+{v[120:123]}"(A[5]),
          [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), [f6] "+v"(f6), [f7] "+v"(f7),
          [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3),
          [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [i4] "+v"(i4), [i5] "+v"(i5), [i6] "+v"(i6), [i7] "+v"(i7),
          [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3), [q4] "+v"(q4), [q5] "+v"(q5)
        : [b0] "v"(B[0]), [b1] "v"(B[1]), [b2] "v"(B[2]), [b3] "v"(B[3]), [b4] "v"(B[4]), [b5] "v"(B[5]),
          [k0] "v"(k0), [k1] "v"(k1), [k2] "v"(k2), [k3] "v"(k3), [pk] "v"(pk), [la] "v"(la)
        : "memory"%(clob)s);
    if (%(loader)d && (it & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.x + p2.x + p3.x + p0.y + p1.y + p2.y + p3.y;
  s += (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + q0.x + q1.y + q2.z + q3.w + q4.x + q5.y);
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e] + c4[e] + c5[e];
#pragma unroll
  for (int j = 0; j < 6; ++j) s += (float)(A[j].x + A[j].w);
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
'''

MAIN_HEAD = r'''
static float* d_out;
static unsigned long long* d_cyc;
static uint4* d_src;
typedef void (*kern_t)(float*, unsigned long long*, const uint4*, int);

static void run(kern_t k, const char* what, int cus, int threads, int mfmas = 12) {
  const int iters = 400;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 65536, 0, d_out, d_cyc, d_src, 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 65536, 0, d_out, d_cyc, d_src, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0;
  (void)hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
  const double nm = iters * (double)mfmas;
  const double tf = (double)cus * 4 * nm * 2.0 * 32 * 32 * 16 / ms / 1e9;
  printf("%-52s %7.1f cyc/MFMA  %7.1f TF bf16  (%.3f ms)  err=%d\n", what, c / nm, tf, ms, (int)hipGetLastError());
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("CUs %d clock %d kHz\n", cus, p.clockRate);
  (void)hipMalloc(&d_out, 4);
  (void)hipMalloc(&d_cyc, 8);
  (void)hipMalloc(&d_src, 8192 * 16);
  {
    static uint32_t h[8192 * 4];
    for (int i = 0; i < 8192 * 4; ++i) h[i] = 0x3F803F80u ^ ((i * 2654435761u) & 0x007F007Fu);
    (void)hipMemcpy(d_src, h, sizeof(h), hipMemcpyHostToDevice);
  }
'''


def main():
    w = sys.stdout.write
    w(HEAD)
    for idx, var in enumerate(VARIANTS):
        name, acc, fill, ld = var[:4]
        accc = var[4] if len(var) > 4 else "a"
        if accc == "mb":
            lines = body_manyb(*var[5])
        elif accc == "pp":
            lines = body_pingpong(abs(var[5]) % 100, reinit=var[5] > 0, inplace=abs(var[5]) >= 100)
        else:
            lines = body_consume(acc, fill) if len(var) > 5 else body(acc, fill)
        asm = "\n".join('        "%s\\n\\t"' % l for l in lines)
        clob = ""
        import re as _re
        used = sorted(set(int(x) for x in _re.findall(r"(?<![\[:\w])v(\d+)", asm)))
        if accc == "pin":
            a0, a1, a2, a3 = "{v[32:47]}", "{v[48:63]}", "{v[64:79]}", "v"
            used = [r for r in used if not (32 <= r < 80)]
        elif accc in ("pp", "mb"):
            a0, a1, a2, a3 = "{v[0:15]}", "{v[16:31]}", "{v[32:47]}", "{v[48:63]}"
            used = [r for r in used if not (0 <= r < 64)]
        else:
            a0 = a1 = a2 = a3 = accc
        used = [r for r in used if not (100 <= r < 124)]
        if used:
            clob = ", " + ", ".join('"v%d"' % r for r in used)
        w(KERNEL % dict(idx=idx, threads=512 if ld else 256, loader=ld, asm=asm, a0=a0, a1=a1, a2=a2, a3=a3, a4=("v" if accc in ("pin", "pp", "mb") else a3), clob=clob))
    w(MAIN_HEAD)
    for idx, var in enumerate(VARIANTS):
        name, acc, fill, ld = var[:4]
        w('  run(k%d, "%s", cus, %d, %d);\n' % (idx, name, 512 if ld else 256, 24 if (len(var) > 4 and var[4] == "pp") else 48 if (len(var) > 4 and var[4] == "mb") else 12))
    w("  return 0;\n}\n")


if __name__ == "__main__":
    main()
