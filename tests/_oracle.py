"""ctypes binding of the CPU oracle (oracle/libpq2oracle.so).

Test infrastructure: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

OK, OUT_OF_SPEC, INVALID_PARAMETER, FEATURE_NOT_SUPPORTED, WOULD_OVER_ALLOCATE, REFERENCE_WOULD_PANIC = range(6)

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class BitmapRun(C.Structure):
    _fields_ = [("kind", C.c_int32), ("offset", C.c_uint64), ("bytes", C.c_uint64), ("length", C.c_uint64),
                ("is_set", C.c_uint32)]


class Interval(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64)]


class FilteredRun(C.Structure):
    _fields_ = [("kind", C.c_int32), ("values_off", C.c_uint64), ("values_len", C.c_uint64), ("offset", C.c_uint64),
                ("length", C.c_uint64), ("is_set", C.c_uint32), ("set_count", C.c_uint64)]


class Split(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("rep_off", "rep_len", "def_off", "def_len", "val_off", "val_len")]


class HybridEncoded(C.Structure):
    _fields_ = [("kind", C.c_int32), ("data", C.c_void_p), ("data_len", C.c_size_t), ("run_length", C.c_size_t)]


class RunDecoder(C.Structure):
    _fields_ = [("values", C.c_void_p), ("len", C.c_size_t), ("num_bits", C.c_size_t)]


def build(native=False, out_dir=None):
    """Compiles the oracle with gcc; returns the path of the shared object."""
    out_dir = out_dir or ORACLE_DIR
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target, "OUT=" + out_dir])
    return os.path.join(out_dir, "libpq2oracle_native.so" if native else "libpq2oracle.so")


def _as_u8(data):
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else \
        np.ascontiguousarray(data, dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(0, dtype=np.uint8)
    return a


def _ptr(a, typ=u8p):
    # a zero-length numpy array still has a valid (dangling but non-null) data pointer
    return a.ctypes.data_as(typ)


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(ORACLE_DIR, "libpq2oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        sz = C.c_size_t
        lib.pq2o_ceil8.restype = sz
        lib.pq2o_ceil8.argtypes = [sz]
        lib.pq2o_get_length.argtypes = [u8p, sz, C.POINTER(sz)]
        lib.pq2o_get_bit_width.restype = C.c_uint32
        lib.pq2o_get_bit_width.argtypes = [C.c_int16]
        lib.pq2o_uleb128_decode.argtypes = [u8p, sz, u64p, C.POINTER(sz)]
        lib.pq2o_uleb128_encode.restype = sz
        lib.pq2o_uleb128_encode.argtypes = [C.c_uint64, u8p]
        for name, typ in (("8", u8p), ("16", C.POINTER(C.c_uint16)), ("32", u32p), ("64", u64p)):
            getattr(lib, "pq2o_unpack" + name).argtypes = [u8p, typ, sz]
            getattr(lib, "pq2o_pack" + name).argtypes = [typ, u8p, sz]
        lib.pq2o_bitpacked_encode_u32.restype = None
        lib.pq2o_bitpacked_encode_u32.argtypes = [u32p, sz, sz, u8p, sz]
        lib.pq2o_bitunpack_u32.argtypes = [u8p, sz, sz, sz, u32p]
        lib.pq2o_bitunpack_u64.argtypes = [u8p, sz, sz, sz, u64p]
        lib.pq2o_zigzag_leb128_decode.argtypes = [u8p, sz, C.POINTER(C.c_int64), C.POINTER(sz)]
        lib.pq2o_delta_bitpacked_decode.argtypes = [u8p, sz, C.POINTER(C.c_int64), sz, C.POINTER(sz), C.POINTER(sz)]
        lib.pq2o_zigzag_leb128_encode.restype = sz
        lib.pq2o_zigzag_leb128_encode.argtypes = [C.c_int64, C.POINTER(C.c_uint8)]
        lib.pq2o_delta_bitpacked_encode.restype = sz
        lib.pq2o_delta_bitpacked_encode.argtypes = [C.POINTER(C.c_int64), sz, u8p]
        lib.pq2o_run_decoder_new.restype = None
        lib.pq2o_run_decoder_new.argtypes = [C.POINTER(RunDecoder), u8p, sz, sz]
        lib.pq2o_run_decoder_next.argtypes = [C.POINTER(RunDecoder), C.POINTER(HybridEncoded)]
        lib.pq2o_hybrid_rle_decode_u32.argtypes = [u8p, sz, C.c_uint32, sz, u32p]
        lib.pq2o_bitmap_encode_bool.restype = sz
        lib.pq2o_bitmap_encode_bool.argtypes = [u8p, sz, u8p]
        lib.pq2o_hybrid_rle_encode_u32.restype = sz
        lib.pq2o_hybrid_rle_encode_u32.argtypes = [u32p, sz, C.c_uint32, u8p]
        lib.pq2o_hybrid_rle_encode_bool.restype = sz
        lib.pq2o_hybrid_rle_encode_bool.argtypes = [u8p, sz, u8p]
        lib.pq2o_split_buffer_v1.argtypes = [u8p, sz, C.c_int, C.c_int, C.POINTER(Split)]
        lib.pq2o_split_buffer_v2.argtypes = [sz, sz, sz, C.POINTER(Split)]
        lib.pq2o_dict_indices_locate.argtypes = [u8p, sz, u32p, C.POINTER(sz)]
        lib.pq2o_hybrid_rle_bitmap_runs.argtypes = [u8p, sz, sz, C.POINTER(BitmapRun), sz, C.POINTER(sz)]
        lib.pq2o_hybrid_rle_boolean_iter.argtypes = [u8p, sz, sz, u8p, sz, C.POINTER(sz)]
        lib.pq2o_filtered_bitmap_runs.argtypes = [C.POINTER(BitmapRun), sz, u8p, C.POINTER(Interval), sz,
                                                  C.POINTER(FilteredRun), sz, C.POINTER(sz)]
        lib.pq2o_time_hybrid_rle_decode.restype = C.c_double
        lib.pq2o_time_hybrid_rle_decode.argtypes = [u8p, sz, C.c_uint32, sz, u32p, C.c_int]
        lib.pq2o_time_bitunpack_u32.restype = C.c_double
        lib.pq2o_time_bitunpack_u32.argtypes = [u8p, sz, sz, sz, u32p, C.c_int]

    # -- scalar helpers ---------------------------------------------------
    def ceil8(self, v):
        return self.lib.pq2o_ceil8(v)

    def get_bit_width(self, max_level):
        return self.lib.pq2o_get_bit_width(max_level)

    def get_length(self, data):
        a = _as_u8(data)
        out = C.c_size_t()
        r = self.lib.pq2o_get_length(_ptr(a), a.size, C.byref(out))
        return None if r != 0 else out.value

    def uleb128_decode(self, data):
        a = _as_u8(data)
        v, c = C.c_uint64(), C.c_size_t()
        st = self.lib.pq2o_uleb128_decode(_ptr(a), a.size, C.byref(v), C.byref(c))
        return st, v.value, c.value

    def uleb128_encode(self, value):
        buf = np.zeros(10, dtype=np.uint8)
        n = self.lib.pq2o_uleb128_encode(value, _ptr(buf))
        return bytes(buf[:n])

    # -- bit packing ------------------------------------------------------
    def unpack(self, bits, data, num_bits):
        dt = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[bits]
        ct = {8: C.c_uint8, 16: C.c_uint16, 32: C.c_uint32, 64: C.c_uint64}[bits]
        a = _as_u8(data)
        out = np.zeros(bits, dtype=dt)
        st = getattr(self.lib, "pq2o_unpack%d" % bits)(_ptr(a), out.ctypes.data_as(C.POINTER(ct)), num_bits)
        return st, out

    def pack(self, bits, values, num_bits):
        dt = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[bits]
        ct = {8: C.c_uint8, 16: C.c_uint16, 32: C.c_uint32, 64: C.c_uint64}[bits]
        v = np.ascontiguousarray(values, dtype=dt)
        out = np.zeros(bits * bits // 8, dtype=np.uint8)
        st = getattr(self.lib, "pq2o_pack%d" % bits)(v.ctypes.data_as(C.POINTER(ct)), _ptr(out), num_bits)
        return st, out

    def bitpacked_encode_u32(self, values, num_bits, packed_len):
        v = np.ascontiguousarray(values, dtype=np.uint32)
        out = np.zeros(packed_len, dtype=np.uint8)
        self.lib.pq2o_bitpacked_encode_u32(_ptr(v, u32p), v.size, num_bits, _ptr(out), out.size)
        return out

    def bitunpack_u32(self, packed, num_bits, length):
        a = _as_u8(packed)
        out = np.zeros(length, dtype=np.uint32)
        st = self.lib.pq2o_bitunpack_u32(_ptr(a), a.size, num_bits, length, _ptr(out, u32p))
        return st, out

    def bitunpack_u64(self, packed, num_bits, length):
        a = _as_u8(packed)
        out = np.zeros(length, dtype=np.uint64)
        st = self.lib.pq2o_bitunpack_u64(_ptr(a), a.size, num_bits, length, _ptr(out, u64p))
        return st, out

    def zigzag_leb128_decode(self, data):
        a = _as_u8(data)
        v, c = C.c_int64(), C.c_size_t()
        st = self.lib.pq2o_zigzag_leb128_decode(_ptr(a), a.size, C.byref(v), C.byref(c))
        return st, v.value, c.value

    def delta_bitpacked_decode(self, data, cap):
        """(status, values int64 ndarray of the header's count, consumed bytes)"""
        a = _as_u8(data)
        out = np.zeros(max(cap, 1), dtype=np.int64)
        n, c = C.c_size_t(), C.c_size_t()
        st = self.lib.pq2o_delta_bitpacked_decode(_ptr(a), a.size, out.ctypes.data_as(C.POINTER(C.c_int64)), cap,
                                                  C.byref(n), C.byref(c))
        return st, out[:min(n.value, cap)], c.value

    def zigzag_leb128_encode(self, value):
        buf = (C.c_uint8 * 10)()
        n = self.lib.pq2o_zigzag_leb128_encode(C.c_int64(value), buf)
        return bytes(buf[:n])

    def delta_bitpacked_encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64)
        out = np.zeros(32 + ((v.size + 127) // 128) * 1035, dtype=np.uint8)
        n = self.lib.pq2o_delta_bitpacked_encode(v.ctypes.data_as(C.POINTER(C.c_int64)), v.size, _ptr(out))
        return out[:n].tobytes()

    # -- hybrid rle -------------------------------------------------------
    def runs(self, data, num_bits):
        """hybrid_rle::Decoder collected: list of ("bitpacked", off, len) / ("rle", off, len, run_length)."""
        a = _as_u8(data)
        dec = RunDecoder()
        self.lib.pq2o_run_decoder_new(C.byref(dec), _ptr(a), a.size, num_bits)
        base = a.ctypes.data
        out = []
        while True:
            run = HybridEncoded()
            r = self.lib.pq2o_run_decoder_next(C.byref(dec), C.byref(run))
            if r == 0:
                return OK, out
            if r < 0:
                return -r, out
            off = (run.data or base) - base
            if run.kind == 0:
                out.append(("bitpacked", off, run.data_len))
            else:
                out.append(("rle", off, run.data_len, run.run_length))

    def hybrid_rle_decode(self, data, num_bits, num_values):
        a = _as_u8(data)
        out = np.zeros(num_values, dtype=np.uint32)
        st = self.lib.pq2o_hybrid_rle_decode_u32(_ptr(a), a.size, num_bits, num_values, _ptr(out, u32p))
        return st, out

    def bitmap_encode_bool(self, bools):
        b = np.ascontiguousarray(bools, dtype=np.uint8)
        out = np.zeros(b.size // 8 + 2, dtype=np.uint8)
        n = self.lib.pq2o_bitmap_encode_bool(_ptr(b), b.size, _ptr(out))
        return bytes(out[:n])

    def encode_u32(self, values, num_bits):
        v = np.ascontiguousarray(values, dtype=np.uint32)
        out = np.zeros(10 + (v.size * max(num_bits, 1) + 7) // 8 + 256, dtype=np.uint8)
        n = self.lib.pq2o_hybrid_rle_encode_u32(_ptr(v, u32p), v.size, num_bits, _ptr(out))
        return bytes(out[:n])

    def encode_bool(self, bools):
        b = np.ascontiguousarray(bools, dtype=np.uint8)
        out = np.zeros(12 + b.size // 8, dtype=np.uint8)
        n = self.lib.pq2o_hybrid_rle_encode_bool(_ptr(b), b.size, _ptr(out))
        return bytes(out[:n])

    # -- page glue --------------------------------------------------------
    def split_buffer_v1(self, buf, has_rep, has_def):
        a = _as_u8(buf)
        s = Split()
        st = self.lib.pq2o_split_buffer_v1(_ptr(a), a.size, int(has_rep), int(has_def), C.byref(s))
        return st, s

    def split_buffer_v2(self, length, rep_len, def_len):
        s = Split()
        st = self.lib.pq2o_split_buffer_v2(length, rep_len, def_len, C.byref(s))
        return st, s

    def dict_indices_locate(self, values):
        a = _as_u8(values)
        bw, off = C.c_uint32(), C.c_size_t()
        st = self.lib.pq2o_dict_indices_locate(_ptr(a), a.size, C.byref(bw), C.byref(off))
        return st, bw.value, off.value

    # -- deserialize adapters ---------------------------------------------
    def bitmap_runs(self, data, num_values, cap=None):
        a = _as_u8(data)
        cap = cap if cap is not None else max(a.size, 1)
        runs = (BitmapRun * cap)()
        n = C.c_size_t()
        st = self.lib.pq2o_hybrid_rle_bitmap_runs(_ptr(a), a.size, num_values, runs, cap, C.byref(n))
        return st, [runs[i] for i in range(min(n.value, cap))], n.value

    def boolean_iter(self, data, num_values):
        a = _as_u8(data)
        out = np.zeros(max(num_values, 1), dtype=np.uint8)
        n = C.c_size_t()
        st = self.lib.pq2o_hybrid_rle_boolean_iter(_ptr(a), a.size, num_values, _ptr(out), out.size, C.byref(n))
        return st, out[:n.value]

    def filtered_runs(self, runs, stream, intervals, cap=256):
        a = _as_u8(stream)
        r = (BitmapRun * max(len(runs), 1))(*runs)
        iv = (Interval * max(len(intervals), 1))(*[Interval(s, l) for s, l in intervals])
        out = (FilteredRun * cap)()
        n = C.c_size_t()
        st = self.lib.pq2o_filtered_bitmap_runs(r, len(runs), _ptr(a), iv, len(intervals), out, cap, C.byref(n))
        return st, [out[i] for i in range(min(n.value, cap))]

    # -- timing -----------------------------------------------------------
    def time_hybrid_rle_decode(self, data, num_bits, num_values, reps):
        a = _as_u8(data)
        out = np.zeros(num_values, dtype=np.uint32)
        return self.lib.pq2o_time_hybrid_rle_decode(_ptr(a), a.size, num_bits, num_values, _ptr(out, u32p), reps)

    def time_bitunpack_u32(self, packed, num_bits, length, reps):
        a = _as_u8(packed)
        out = np.zeros(length, dtype=np.uint32)
        return self.lib.pq2o_time_bitunpack_u32(_ptr(a), a.size, num_bits, length, _ptr(out, u32p), reps)


_ORACLE = None


def oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
