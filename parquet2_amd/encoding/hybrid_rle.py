"""`parquet2::encoding::hybrid_rle` (src/encoding/hybrid_rle/): Parquet's RLE / bit-packing hybrid.

Value decoding (`HybridRleDecoder`, `BitmapIter`) runs on the GPU through the
C ABI; the run iterator (`Decoder`) and the encoders are host-side.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from .._lib import OK, Run, as_u8, default_context, lib, ptr
from ..error import error_for_status, raise_for_status


@dataclass(frozen=True)
class Bitpacked:
    """`HybridEncoded::Bitpacked(slice)` (mod.rs:15-21)."""
    data: bytes


@dataclass(frozen=True)
class Rle:
    """`HybridEncoded::Rle(slice, run_length)`."""
    data: bytes
    run_length: int


class HybridEncoded:
    Bitpacked = Bitpacked
    Rle = Rle


class Decoder:
    """`hybrid_rle::Decoder::new(values, num_bits)`: iterator over the runs of a stream (decoder.rs:8-65)."""

    def __init__(self, values, num_bits):
        self._data = as_u8(values)
        self._num_bits = num_bits
        cap = max(1, self._data.size)
        n = C.c_uint64()
        runs = (Run * min(cap, 1 << 16))()
        self._status = lib().pq2_hybrid_rle_runs(ptr(self._data), self._data.size, num_bits, runs, len(runs),
                                                 C.byref(n))
        if n.value > len(runs):
            runs = (Run * n.value)()
            self._status = lib().pq2_hybrid_rle_runs(ptr(self._data), self._data.size, num_bits, runs, len(runs),
                                                     C.byref(n))
        self._runs = runs
        self._n = n.value
        self._i = 0

    new = classmethod(lambda cls, values, num_bits: cls(values, num_bits))

    def num_bits(self):
        return self._num_bits

    def __iter__(self):
        return self

    def __next__(self):
        if self._i >= self._n:
            if self._status != OK:  # the reference panics at this run
                st, self._status = self._status, OK
                raise error_for_status(st, "hybrid_rle::Decoder")
            raise StopIteration
        r = self._runs[self._i]
        self._i += 1
        data = self._data[r.offset:r.offset + r.len].tobytes()
        return Bitpacked(data) if r.kind == 0 else Rle(data, r.run_length)

    def run_offsets(self):
        """(kind, offset, len, run_length) of every remaining run, without copying slices."""
        return [(r.kind, r.offset, r.len, r.run_length) for r in self._runs[self._i:self._n]]


class HybridRleDecoder:
    """`HybridRleDecoder::try_new(data, num_bits, num_values)` (mod.rs:35-116).

    Decodes the whole stream on the GPU at construction (the consumers of the
    reference drain it with `collect`); iterating yields exactly `num_values`
    ints, `collect()` returns the uint32 ndarray.  Inputs on which the
    reference panics raise `ReferencePanic`.
    """

    def __init__(self, data, num_bits, num_values, ctx=None):
        ctx = ctx or default_context()
        values, status = ctx.decode_page_host(data, num_bits, num_values)
        raise_for_status(status, "HybridRleDecoder: %s" % lib().pq2_strerror(status).decode())
        self._values = values
        self._pos = 0

    try_new = classmethod(lambda cls, data, num_bits, num_values, ctx=None: cls(data, num_bits, num_values, ctx))

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


class BitmapIter:
    """`BitmapIter::new(slice, offset, len)`: `len` bools starting at bit `offset` (bitmap.rs:14-68).

    Reading past the slice repeats its last byte (0 for an empty slice), as the
    reference does.  The bits are unpacked on the GPU (width-1 bit-unpack).
    """

    def __init__(self, slice_, offset, length, ctx=None):
        data = as_u8(slice_)
        if offset // 8 > data.size:
            raise error_for_status(5, "BitmapIter::new: offset / 8 > slice.len()")
        data = data[offset // 8:]
        bit0 = offset % 8
        need = (bit0 + length + 7) // 8
        if need > data.size:
            fill = data[-1] if data.size else 0
            data = np.concatenate([data, np.full(need - data.size, fill, dtype=np.uint8)])
        if length == 0:
            self._values = np.zeros(0, dtype=bool)
        else:
            ctx = ctx or default_context()
            bits = ctx.bitunpack_u32_host(data[:need], 1, bit0 + length)
            self._values = bits[bit0:].astype(bool)
        self._pos = 0

    new = classmethod(lambda cls, slice_, offset, length, ctx=None: cls(slice_, offset, length, ctx))

    def collect(self):
        return self._values[self._pos:]

    def __iter__(self):
        return self

    def __next__(self):
        if self._pos >= len(self._values):
            raise StopIteration
        v = bool(self._values[self._pos])
        self._pos += 1
        return v

    def size_hint(self):
        n = len(self._values) - self._pos
        return n, n


def _encode(fn_name, values, dtype, extra, cap):
    v = np.ascontiguousarray(values, dtype=dtype)
    out = np.zeros(cap(v.size), dtype=np.uint8)
    written = C.c_uint64()
    args = [ptr(v), v.size] + extra + [ptr(out), out.size, C.byref(written)]
    raise_for_status(getattr(lib(), fn_name)(*args), fn_name)
    return out[:written.value].tobytes()


def encode_u32(values, num_bits, writer=None):
    """`hybrid_rle::encode_u32` (encoder.rs:9-29): a single bit-packed run.  Returns the bytes
    (or appends them to the bytearray `writer`)."""
    buf = _encode("pq2_hybrid_rle_encode_u32", values, np.uint32, [num_bits],
                  lambda n: 16 + 4 * max(num_bits & 0xff, 1) * ((n + 31) // 32))
    if writer is not None:
        writer += buf
    return buf


def encode_bool(values, writer=None):
    """`hybrid_rle::encode_bool` (encoder.rs:75-93)."""
    buf = _encode("pq2_hybrid_rle_encode_bool", values, np.uint8, [], lambda n: 16 + n // 8)
    if writer is not None:
        writer += buf
    return buf


def bitpacked_encode(values, writer=None):
    """`hybrid_rle::bitpacked_encode` = `bitmap::encode_bool` (bitmap.rs:71-102): bools, LSB first."""
    buf = _encode("pq2_bitmap_encode_bool", values, np.uint8, [], lambda n: 8 + n // 8)
    if writer is not None:
        writer += buf
    return buf
