"""Synthetic hybrid-RLE pages for tests and benchmarks (binding of libpq2synth.so, include/pq2synth.h)."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libpq2synth.so")

SIGNATURES = {
    "pq2s_encode_bound": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "pq2s_encode_hybrid": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]),
    "pq2s_gen_dict_values": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_void_p]),
    "pq2s_gen_def_values": (None, [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_void_p]),
    "pq2s_gen_uniform_values": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "pq2s_gen_adversarial": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
}

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -m parquet2_amd.build`" % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = restype, argtypes
        _LIB = h
    return _LIB


def _p(a):
    return C.c_void_p(a.ctypes.data)


def seed_for(config_id, page_index):
    """Seed convention shared by tests and bench: 0x5EED0000 + config * 0x100 + page."""
    return 0x5EED0000 + config_id * 0x100 + page_index


def encode_hybrid(values, num_bits, max_groups=63):
    """Spec-conforming hybrid encoder with genuine RLE runs; returns uint8 ndarray."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.zeros(lib().pq2s_encode_bound(v.size, num_bits), dtype=np.uint8)
    n = lib().pq2s_encode_hybrid(_p(v), v.size, num_bits, max_groups, _p(out), out.size)
    if n == 0 and v.size:
        raise RuntimeError("pq2s_encode_hybrid: buffer too small")
    return out[:n].copy()


def dict_values(seed, n, num_bits=12, mean_repeat=64.0, mean_distinct=256.0):
    v = np.zeros(n, dtype=np.uint32)
    lib().pq2s_gen_dict_values(seed, n, num_bits, mean_repeat, mean_distinct, _p(v))
    return v


def def_values(seed, n, mean_valid=380.0, mean_null=20.0):
    v = np.zeros(n, dtype=np.uint32)
    lib().pq2s_gen_def_values(seed, n, mean_valid, mean_null, _p(v))
    return v


def uniform_values(seed, n, num_bits):
    v = np.zeros(n, dtype=np.uint32)
    lib().pq2s_gen_uniform_values(seed, n, num_bits, _p(v))
    return v


def adversarial(seed, n, num_bits):
    """Returns (stream uint8 ndarray, expected uint32 ndarray)."""
    expected = np.zeros(n, dtype=np.uint32)
    cap = 64 + (n // 4 + 2) * (2 + 4 + num_bits)
    out = np.zeros(cap, dtype=np.uint8)
    m = lib().pq2s_gen_adversarial(seed, n, num_bits, _p(out), out.size, _p(expected))
    if m == 0 and n:
        raise RuntimeError("pq2s_gen_adversarial: buffer too small")
    return out[:m].copy(), expected


def dict_page(seed, n, num_bits=12, mean_repeat=64.0, mean_distinct=256.0):
    """Config-2-shaped page: (stream, values).  The stream carries no width byte."""
    v = dict_values(seed, n, num_bits, mean_repeat, mean_distinct)
    return encode_hybrid(v, num_bits), v


def def_levels_page(seed, n, mean_valid=380.0, mean_null=20.0):
    """Config-3-shaped page (bit width 1): (stream, values)."""
    v = def_values(seed, n, mean_valid, mean_null)
    return encode_hybrid(v, 1), v


def bench_pattern(n_bytes):
    """`0b01011011u8.rotate_left(i)` for i in 0..n (benches/decode_bitpacking.rs:9-11)."""
    i = np.arange(n_bytes, dtype=np.uint32) % 8
    b = np.uint32(0x5B)
    return (((b << i) | (b >> (8 - i))) & 0xFF).astype(np.uint8)


class PageBatch:
    """Streams concatenated into one input buffer + the descriptor array of the C ABI."""

    def __init__(self, align=16):
        from ._lib import PAGE_DESC_DTYPE
        self._dtype = PAGE_DESC_DTYPE
        self.align = align
        self.chunks = []
        self.rows = []
        self.in_bytes = 0
        self.out_values = 0

    def add(self, stream, num_bits, num_values, out_align=4):
        if not isinstance(stream, np.ndarray):
            stream = np.frombuffer(bytes(stream), dtype=np.uint8)
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        pad = (-self.in_bytes) % self.align
        if pad:
            self.chunks.append(np.zeros(pad, dtype=np.uint8))
            self.in_bytes += pad
        self.out_values += (-self.out_values) % out_align
        self.rows.append((self.in_bytes, stream.size, num_bits, num_values, self.out_values))
        self.chunks.append(stream)
        self.in_bytes += stream.size
        self.out_values += num_values
        return len(self.rows) - 1

    def finish(self):
        """Returns (input uint8 ndarray padded by 16 bytes, descs ndarray, out_values)."""
        data = np.concatenate(self.chunks + [np.zeros(16, dtype=np.uint8)]) if self.chunks else \
            np.zeros(16, dtype=np.uint8)
        descs = np.array(self.rows, dtype=self._dtype) if self.rows else np.zeros(0, dtype=self._dtype)
        return data, descs, self.out_values
