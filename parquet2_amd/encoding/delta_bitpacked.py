"""`parquet2::encoding::delta_bitpacked` (src/encoding/delta_bitpacked/): DELTA_BINARY_PACKED integers.

`Decoder` decodes on the GPU (delta_kernel): the blocks of a stream one after the other, the values of a block
lane-parallel with a 64-bit prefix sum over the deltas.  `encode` is the host-side write path.
"""
import ctypes as C

import numpy as np

from .._lib import as_u8, default_context, lib, ptr
from ..error import raise_for_status
from . import uleb128

# more values than this in one stream are refused by the library (PQ2_MAX_PAGE_VALUES)
_MAX_VALUES = 0x7FFFFFFF


class Decoder:
    """`delta_bitpacked::Decoder::try_new(values)` (decoder.rs:136-180) as an iterator of i64 (:216-243).

    The stream is decoded eagerly; an error that the reference would return from `try_new` or from a later
    `next()` is raised here by the constructor.  `consumed_bytes()` is the reference's (:182-185) once every
    value has been taken.
    """

    def __init__(self, values, ctx=None):
        ctx = ctx or default_context()
        values = as_u8(values)
        # total_count is the third varint of the header (:138-150); the device checks the whole header again
        at = 0
        total = 0
        for _ in range(3):
            total, used = uleb128.decode(values[at:])
            at += used
        self._values, self._consumed = ctx.delta_bitpacked_decode_host(values, min(total, _MAX_VALUES))
        self._pos = 0

    try_new = classmethod(lambda cls, values, ctx=None: cls(values, ctx))

    def consumed_bytes(self):
        return self._consumed

    def collect(self):
        return self._values[self._pos:]

    def __iter__(self):
        return self

    def __next__(self):
        if self._pos >= len(self._values):
            raise StopIteration
        v = int(self._values[self._pos])
        self._pos += 1
        return v

    def size_hint(self):
        n = len(self._values) - self._pos
        return n, n

    def __len__(self):
        return len(self._values) - self._pos


def encode(iterator, buffer=None):
    """`delta_bitpacked::encode(iterator, &mut buffer)` (encoder.rs:11-78), on the host.  Returns the bytes, or
    appends them to the given bytearray."""
    v = np.ascontiguousarray(list(iterator) if not isinstance(iterator, np.ndarray) else iterator, dtype=np.int64)
    out = np.zeros(lib().pq2_delta_bitpacked_encode_bound(v.size), dtype=np.uint8)
    written = C.c_uint64()
    raise_for_status(lib().pq2_delta_bitpacked_encode(ptr(v), v.size, ptr(out), out.size, C.byref(written)),
                     "delta_bitpacked::encode")
    data = out[:written.value].tobytes()
    if buffer is None:
        return data
    buffer.extend(data)
    return None
