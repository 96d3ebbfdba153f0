"""Does a HIP stream's CU mask (hipExtStreamCreateWithCUMask) hold for (a) plain launches and (b) the nodes of a LINEAR
hipGraph launched into that stream?  Times a compute-bound kernel (f32 matmul 4096^3) unmasked / masked to 1/8, 1/4, 1/2
of the CUs, eager and as a captured graph; prints which mask layouts (low bits / one bit in eight) slow it how much."""
import ctypes
import os
import sys
import time

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = ctypes.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dev = torch.device('cuda', 0)
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)
fn = lambda: torch.mm(a, b, out=c)
base = timeit(fn, torch.cuda.current_stream())
print("unmasked eager %.3f ms" % base)
ALL = (1 << 256) - 1
layouts = {
    "low32": (1 << 32) - 1, "low64": (1 << 64) - 1, "low128": (1 << 128) - 1,
    "every8th": sum(1 << i for i in range(0, 256, 8)), "every4th": sum(1 << i for i in range(0, 256, 4)),
    "all256": ALL, "all_but_low16": ALL ^ 0xFFFF, "low16": 0xFFFF,
    "two_per_xcd_a": sum(1 << i for i in range(16)),      # if bit i -> XCD i % 8: bits 0..15 = 2 CUs on every XCD
}
for name, bits in layouts.items():
    st = masked_stream(bits)
    t = timeit(fn, st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            fn()
    tg = timeit(g.replay, st)
    # replay() launches into the CURRENT stream: under torch.cuda.stream(st) that is the masked one
    print("%-16s bits %3d  eager %.3f ms (x%.2f)  graph-in-masked-stream %.3f ms (x%.2f)"
          % (name, bin(bits).count("1"), t, t / base, tg, tg / base))
