"""`parquet2::encoding::uleb128` (src/encoding/uleb128.rs)."""
import ctypes as C

import numpy as np

from .._lib import as_u8, lib, ptr
from ..error import raise_for_status


def decode(values):
    """Returns (value, consumed) (uleb128.rs:3-23)."""
    a = as_u8(values)
    v, c = C.c_uint64(), C.c_uint64()
    raise_for_status(lib().pq2_uleb128_decode(ptr(a), a.size, C.byref(v), C.byref(c)), "uleb128::decode")
    return v.value, c.value


def encode(value, container=None):
    """Encodes `value`; returns the bytes (or, given a 10-byte `container`, fills it and returns the count)."""
    buf = np.zeros(10, dtype=np.uint8)
    n = lib().pq2_uleb128_encode(value, buf.ctypes.data_as(C.POINTER(C.c_uint8)))
    if container is None:
        return bytes(buf[:n])
    container[:n] = bytes(buf[:n])
    return n
