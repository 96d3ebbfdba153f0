"""arx -- MI355X-native hot path of A-RecSys (embedding lookup -> scorer -> sampled loss).

Host code is Python mirroring the reference's classes (Attributes,
EmbeddingAttribute, LatentProductModel, SeqModel); all arithmetic on the path
runs in libarx.so (hand-written HIP for gfx950) through a ctypes C ABI
(include/arx.h).  Importing this package loads the library and fails loudly if
it is missing -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when libarx.so is absent)

__all__ = ["_lib"]
