"""`parquet2::encoding::bitpacked` (src/encoding/bitpacked/): LSB-first bit packing of u32.

`Decoder` unpacks on the GPU (bitunpack_kernel); `encode` is the host-side
write path.
"""
import numpy as np

from .._lib import as_u8, default_context, lib, ptr
from ..error import raise_for_status


class Decoder:
    """`bitpacked::Decoder::<T>::try_new(packed, num_bits, length)` (decode.rs:28-86), T = u32 (the default) or
    u64 (`dtype=np.uint64`: `Unpackable for u64`, mod.rs:153-166).

    The values are unpacked eagerly on the GPU; iterating yields Python ints,
    `collect()` returns the ndarray.
    """

    def __init__(self, packed, num_bits, length, ctx=None, dtype=np.uint32):
        ctx = ctx or default_context()
        if np.dtype(dtype) == np.uint64:
            self._values = ctx.bitunpack_u64_host(packed, num_bits, length)
        else:
            self._values = ctx.bitunpack_u32_host(packed, num_bits, length)
        self._pos = 0

    try_new = classmethod(lambda cls, packed, num_bits, length, ctx=None, dtype=np.uint32:
                          cls(packed, num_bits, length, ctx, dtype))

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


def encode(unpacked, num_bits, packed=None):
    """`bitpacked::encode::<u32>` (encode.rs:11-36): packs blocks of 32 values, zero-padding the last one.

    Returns the packed bytes (ndarray) or fills the given `packed` uint8 ndarray (which must be zeroed).
    """
    v = np.ascontiguousarray(unpacked, dtype=np.uint32)
    need = ((v.size + 31) // 32) * 4 * num_bits
    out = packed if packed is not None else np.zeros(need, dtype=np.uint8)
    raise_for_status(lib().pq2_bitpacked_encode_u32(ptr(v), v.size, num_bits, ptr(out), out.size),
                     "bitpacked::encode")
    return out
