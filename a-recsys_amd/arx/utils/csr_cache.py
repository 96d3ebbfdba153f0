"""Flat binary cache of a preprocessed dataset -- what
input_attribute.read_data stores instead of the reference's Python-2 pickle of live objects
(input_attribute.py:59-62).

File layout (little endian):
    bytes 0..7    magic  b'ARXCSR01'
    bytes 8..15   uint64 length H of the JSON header
    bytes 16..    H bytes of UTF-8 JSON: {"meta": {...}, "arrays": {name: {dtype, shape, offset}}}
    then every array's raw bytes at `offset` (absolute, 64-byte aligned)
Arrays are plain C-order buffers, so a loader can np.memmap them (or hipMemcpy them straight
into the device-side attribute maps) without parsing anything.
"""
from __future__ import annotations

import json

import numpy as np

MAGIC = b'ARXCSR01'
ALIGN = 64


def save(path, arrays, meta=None):
    """arrays: {name: ndarray}; meta: JSON-serialisable dict."""
    items = [(k, np.ascontiguousarray(v)) for k, v in arrays.items()]
    for k, a in items:
        if a.dtype == object:
            raise TypeError("array %r has dtype object" % k)
    desc = {k: {"dtype": a.dtype.str, "shape": list(a.shape), "offset": 0} for k, a in items}

    def header_bytes():
        return json.dumps({"meta": meta or {}, "arrays": desc}, sort_keys=True).encode('utf-8')

    # offsets depend on the header length, which depends on the offsets' digits: iterate to a fixpoint
    hlen = len(header_bytes())
    while True:
        off = 16 + hlen
        for k, a in items:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            desc[k]["offset"] = off
            off += a.nbytes
        h = header_bytes()
        if len(h) == hlen:
            break
        hlen = len(h)
    with open(path, 'wb') as f:
        f.write(MAGIC)
        f.write(np.uint64(hlen).tobytes())
        f.write(h)
        for k, a in items:
            f.write(b'\0' * (desc[k]["offset"] - f.tell()))
            f.write(a.tobytes())


def load(path, mmap=False):
    """-> (arrays dict, meta dict)."""
    with open(path, 'rb') as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s: not an ARXCSR01 file" % path)
        hlen = int(np.frombuffer(f.read(8), dtype='<u8')[0])
        head = json.loads(f.read(hlen).decode('utf-8'))
        out = {}
        for k, d in head["arrays"].items():
            dt, shape = np.dtype(d["dtype"]), tuple(d["shape"])
            if mmap:
                out[k] = np.memmap(path, dtype=dt, mode='r', offset=d["offset"], shape=shape)
            else:
                f.seek(d["offset"])
                n = int(np.prod(shape, dtype=np.int64)) if shape else 1
                out[k] = np.frombuffer(f.read(n * dt.itemsize), dtype=dt).reshape(shape).copy()
    return out, head["meta"]
