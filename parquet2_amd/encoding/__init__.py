"""`parquet2::encoding` (src/encoding/mod.rs): the hybrid-RLE / bit-packing part and DELTA_BINARY_PACKED."""
import ctypes as C

from .._lib import OK, as_u8, lib, ptr
from . import bitpacked, hybrid_rle, uleb128  # noqa: F401
from . import delta_bitpacked, zigzag_leb128  # noqa: F401,E402


def ceil8(value):
    """src/encoding/mod.rs:23-27"""
    return lib().pq2_ceil8(value)


def get_length(values):
    """First 4 bytes as a little-endian u32, or None (src/encoding/mod.rs:14-21)."""
    a = as_u8(values)
    out = C.c_uint64()
    return out.value if lib().pq2_get_length(ptr(a), a.size, C.byref(out)) == OK else None
