"""`parquet2::encoding::zigzag_leb128` (src/encoding/zigzag_leb128.rs)."""
import ctypes as C

import numpy as np

from .._lib import as_u8, lib, ptr
from ..error import raise_for_status


def decode(values):
    """Returns (value, consumed) (zigzag_leb128.rs:5-8)."""
    a = as_u8(values)
    v, c = C.c_int64(), C.c_uint64()
    raise_for_status(lib().pq2_zigzag_leb128_decode(ptr(a), a.size, C.byref(v), C.byref(c)), "zigzag_leb128::decode")
    return v.value, c.value


def encode(value):
    """Returns (container of 10 bytes, produced) (zigzag_leb128.rs:10-15)."""
    buf = np.zeros(10, dtype=np.uint8)
    n = lib().pq2_zigzag_leb128_encode(value, buf.ctypes.data_as(C.POINTER(C.c_uint8)))
    return bytes(buf), n
