"""ctypes binding of libpq2hip.so (the C ABI declared in include/pq2hip.h).

There is no CPU fallback: decoding needs the HIP library and a GPU.  Loading
fails loudly if the shared object is missing, and `Context()` raises
`DeviceError` when no device is usable.
"""
import ctypes as C
import os

import numpy as np

from .error import DeviceError, raise_for_status

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQ2_LIB") or os.path.join(PKG, "lib", "libpq2hip.so")  # PQ2_LIB: A/B builds

OK, OUT_OF_SPEC, INVALID_PARAMETER, FEATURE_NOT_SUPPORTED, WOULD_OVER_ALLOCATE, REFERENCE_WOULD_PANIC, DEVICE_ERROR = \
    range(7)
BLOCK_VALUES = 1024

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)


class PageDesc(C.Structure):
    """pq2_page_desc"""
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("num_bits", C.c_uint32),
                ("num_values", C.c_uint64), ("out_off", C.c_uint64)]


PAGE_DESC_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("num_bits", "<u4"), ("num_values", "<u8"),
                            ("out_off", "<u8")])
assert PAGE_DESC_DTYPE.itemsize == C.sizeof(PageDesc) == 32

# pq2_dict_ref
# pq2_raw_page
RAW_PAGE_DTYPE = np.dtype([("buf_off", "<u8"), ("buf_len", "<u4"), ("num_values", "<u4"), ("max_rep_level", "<u2"),
                           ("max_def_level", "<u2"), ("flags", "<u4"), ("out_off", "<u8")])
assert RAW_PAGE_DTYPE.itemsize == 32
RAW_DICT_INDICES = 1
# pq2_interval
INTERVAL_DTYPE = np.dtype([("start", "<u8"), ("length", "<u8")])
DICT_REF_DTYPE = np.dtype([("dict_off", "<u8"), ("dict_len", "<u4"), ("reserved", "<u4")])
assert DICT_REF_DTYPE.itemsize == 16


class Split(C.Structure):
    """pq2_split"""
    _fields_ = [(n, C.c_uint64) for n in ("rep_off", "rep_len", "def_off", "def_len", "val_off", "val_len")]


class Run(C.Structure):
    """pq2_run"""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_uint32), ("offset", C.c_uint64), ("len", C.c_uint64),
                ("run_length", C.c_uint64)]


# every symbol include/pq2hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pq2_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "pq2_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "pq2_strerror": (C.c_char_p, [C.c_int32]),
    "pq2_last_error": (C.c_char_p, [C.c_void_p]),
    "pq2_ctx_enable_profiling": (C.c_int32, [C.c_void_p, C.c_int32]),
    "pq2_ctx_last_schedule": (C.c_char_p, [C.c_void_p]),
    "pq2_ctx_pages_total": (C.c_uint64, [C.c_void_p]),
    "pq2_ctx_last_density": (C.c_uint32, [C.c_void_p]),
    "pq2_ctx_scout_rounds": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pq2_ctx_set_schedule": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "pq2_ctx_kernel_times": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p, u64p]),
    "pq2_decode_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_uint64, C.c_void_p, C.c_void_p]),
    "pq2_fetch_status": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pq2_decode_batch_device_noalloc": (C.c_int32, [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                     C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pq2_decode_validity_bitmap_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                             C.c_void_p]),
    "pq2_decode_dict_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                  C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                                  C.c_void_p]),
    "pq2_decode_batch_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                           C.c_uint64, C.c_void_p]),
    "pq2_bitunpack_u32_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p,
                                              C.c_void_p]),
    "pq2_bitunpack_u32_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]),
    "pq2_decode_optional_dict_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                           C.c_void_p]),
    "pq2_filtered_validity_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                       C.c_void_p, C.c_void_p]),
    "pq2_decode_selected_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                      C.c_void_p]),
    "pq2_decode_raw_pages_v1_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                    C.c_void_p]),
    "pq2_bitunpack_u64_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p,
                                              C.c_void_p]),
    "pq2_bitunpack_u64_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]),
    "pq2_delta_bitpacked_decode_batch_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                             C.c_void_p, C.c_void_p]),
    "pq2_delta_bitpacked_decode_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p,
                                                     u64p]),
    "pq2_delta_bitpacked_encode_bound": (C.c_uint64, [C.c_uint64]),
    "pq2_delta_bitpacked_encode": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
    "pq2_zigzag_leb128_decode": (C.c_int32, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int64), u64p]),
    "pq2_zigzag_leb128_encode": (C.c_uint64, [C.c_int64, u8p]),
    "pq2_partition_pages": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, u64p]),
    "pq2_partition_by_cost": (C.c_int32, [u64p, C.c_uint64, C.c_uint32, u64p]),
    "pq2_page_costs": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
    "pq2_multi_decode": (C.c_int32, [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_uint64, C.c_void_p]),
    "pq2_ceil8": (C.c_uint64, [C.c_uint64]),
    "pq2_get_length": (C.c_int32, [C.c_void_p, C.c_uint64, u64p]),
    "pq2_get_bit_width": (C.c_uint32, [C.c_int16]),
    "pq2_uleb128_decode": (C.c_int32, [C.c_void_p, C.c_uint64, u64p, u64p]),
    "pq2_uleb128_encode": (C.c_uint64, [C.c_uint64, u8p]),
    "pq2_split_buffer_v1": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(Split)]),
    "pq2_split_buffer_v2": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Split)]),
    "pq2_dict_indices_locate": (C.c_int32, [C.c_void_p, C.c_uint64, u32p, u64p]),
    "pq2_hybrid_rle_runs": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Run), C.c_uint64, u64p]),
    "pq2_bitpacked_encode_u32": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]),
    "pq2_hybrid_rle_encode_u32": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, u64p]),
    "pq2_hybrid_rle_encode_bool": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
    "pq2_bitmap_encode_bool": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
}

_LIB = None


def lib():
    """The loaded shared library (loaded once)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -m parquet2_amd.build` (needs hipcc); "
                "parquet2_amd has no CPU decode path" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64.  Loading it
        # first makes the dynamic linker resolve this library's libamdhip64.so.7 to that same copy; two
        # runtimes in one process leave the second one without GPUs.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = handle
    return _LIB


def as_u8(data):
    """bytes-like / ndarray -> contiguous uint8 ndarray (zero-copy where possible)."""
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)


def ptr(a):
    return C.c_void_p(a.ctypes.data)


class Context:
    """One decoder context bound to one GPU (pq2_ctx).  Not thread-safe."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        st = lib().pq2_ctx_create(device_id, C.byref(self._h))
        if st != OK:
            self._h = None
            raise DeviceError("pq2_ctx_create(device %d) failed: %s (no CPU fallback exists)" %
                              (device_id, lib().pq2_strerror(st).decode()))
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            lib().pq2_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise_for_status(st, lib().pq2_last_error(self._h).decode() or lib().pq2_strerror(st).decode())

    # ---- kernel timing ---------------------------------------------------------
    def enable_profiling(self, on=True):
        self._check(lib().pq2_ctx_enable_profiling(self._h, int(on)))

    def last_schedule(self):
        """Name of the kernel that carried the last batch's values."""
        return lib().pq2_ctx_last_schedule(self._h).decode()

    def set_schedule(self, name):
        """"auto" | "scout" | "ring" | "tile" | "page": how the batches of this context are scheduled."""
        self._check(lib().pq2_ctx_set_schedule(self._h, name.encode()))

    def scout_rounds(self, n_pages):
        """Rounds the resolve step of the scout kernels took for every page of the last batch (diagnostics)."""
        out = np.zeros(n_pages, dtype=np.uint32)
        self._check(lib().pq2_ctx_scout_rounds(self._h, C.c_void_p(out.ctypes.data), n_pages))
        return out

    def kernel_times(self):
        """(parse_ms, expand_ms, batches, expand_launches) summed since the last call."""
        a, b, n, k = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
        self._check(lib().pq2_ctx_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(n), C.byref(k)))
        return a.value, b.value, n.value, k.value

    # ---- device-resident batch decode --------------------------------------
    def decode_batch_device(self, d_in_ptr, in_bytes, descs, d_out_ptr, out_values, stream=0, want_status=True):
        """descs: ndarray of PAGE_DESC_DTYPE.  Returns per-page statuses (int32 ndarray) or None."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        status = np.zeros(len(descs), dtype=np.int32) if want_status else None
        st = lib().pq2_decode_batch_device(self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs), len(descs),
                                           C.c_void_p(d_out_ptr), out_values,
                                           ptr(status) if want_status else None, C.c_void_p(stream))
        self._check(st)
        return status

    def decode_validity_bitmap_batch_device(self, d_in_ptr, in_bytes, descs, d_bitmap_ptr, bitmap_bytes, stream=0,
                                            want_counts=True):
        """One-bit definition levels -> validity bitmaps on the device (pq2_decode_validity_bitmap_batch_device;
        descs' out_off counts bytes of the bitmap buffer).  Returns (rows, set_counts, statuses), or None when
        want_counts is false (the call then only enqueues)."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        n = len(descs)
        rows = np.zeros(n, dtype=np.uint32) if want_counts else None
        sets = np.zeros(n, dtype=np.uint32) if want_counts else None
        status = np.zeros(n, dtype=np.int32) if want_counts else None
        st = lib().pq2_decode_validity_bitmap_batch_device(
            self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs), n, C.c_void_p(d_bitmap_ptr), bitmap_bytes,
            ptr(rows) if want_counts else None, ptr(sets) if want_counts else None,
            ptr(status) if want_counts else None, C.c_void_p(stream))
        self._check(st)
        return (rows, sets, status) if want_counts else None

    def decode_dict_batch_device(self, d_in_ptr, in_bytes, descs, d_dict_ptr, dict_bytes, dict_refs, value_size,
                                 d_out_ptr, out_values, stream=0, want_status=True):
        """Indices -> dictionary values on the device (pq2_decode_dict_batch_device).  descs: PAGE_DESC_DTYPE,
        dict_refs: DICT_REF_DTYPE, one per page."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        dict_refs = np.ascontiguousarray(dict_refs, dtype=DICT_REF_DTYPE)
        assert len(dict_refs) == len(descs)
        status = np.zeros(len(descs), dtype=np.int32) if want_status else None
        st = lib().pq2_decode_dict_batch_device(self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs), len(descs),
                                                C.c_void_p(d_dict_ptr), dict_bytes, ptr(dict_refs), value_size,
                                                C.c_void_p(d_out_ptr), out_values,
                                                ptr(status) if want_status else None, C.c_void_p(stream))
        self._check(st)
        return status

    def decode_optional_dict_batch_device(self, d_in_ptr, in_bytes, def_descs, idx_descs, max_def_levels, d_dict_ptr,
                                          dict_bytes, dict_refs, value_size, d_out_ptr, d_validity_ptr, out_rows,
                                          stream=0, want_status=True):
        """Nullable dictionary pages -> values + validity bytes (pq2_decode_optional_dict_batch_device)."""
        def_descs = np.ascontiguousarray(def_descs, dtype=PAGE_DESC_DTYPE)
        idx_descs = np.ascontiguousarray(idx_descs, dtype=PAGE_DESC_DTYPE)
        dict_refs = np.ascontiguousarray(dict_refs, dtype=DICT_REF_DTYPE)
        max_def_levels = np.ascontiguousarray(max_def_levels, dtype=np.uint32)
        assert len(def_descs) == len(idx_descs) == len(dict_refs) == len(max_def_levels)
        status = np.zeros(len(def_descs), dtype=np.int32) if want_status else None
        st = lib().pq2_decode_optional_dict_batch_device(
            self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(def_descs), ptr(idx_descs), ptr(max_def_levels),
            len(def_descs), C.c_void_p(d_dict_ptr), dict_bytes, ptr(dict_refs), value_size, C.c_void_p(d_out_ptr),
            C.c_void_p(d_validity_ptr), out_rows, ptr(status) if want_status else None, C.c_void_p(stream))
        self._check(st)
        return status

    def decode_selected_batch_device(self, d_in_ptr, in_bytes, descs, intervals, interval_offsets, d_out_ptr, out_values,
                                     stream=0, want_status=True):
        """SliceFilteredIter over every page's values (pq2_decode_selected_batch_device).  intervals: INTERVAL_DTYPE,
        interval_offsets: n_pages + 1 indices into it."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        intervals = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE)
        interval_offsets = np.ascontiguousarray(interval_offsets, dtype=np.uint64)
        assert len(interval_offsets) == len(descs) + 1
        status = np.zeros(len(descs), dtype=np.int32) if want_status else None
        st = lib().pq2_decode_selected_batch_device(self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs), len(descs),
                                                    ptr(intervals), ptr(interval_offsets), C.c_void_p(d_out_ptr),
                                                    out_values, ptr(status) if want_status else None,
                                                    C.c_void_p(stream))
        self._check(st)
        return status

    def filtered_validity_batch_device(self, d_in_ptr, in_bytes, descs, intervals, interval_offsets, d_validity_ptr,
                                       validity_rows, d_counts_ptr, stream=0):
        """FilteredHybridBitmapIter over every page's definition levels (pq2_filtered_validity_batch_device): validity
        bytes of the selected rows, and per interval the valid rows skipped in front of it and inside it."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        intervals = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE)
        interval_offsets = np.ascontiguousarray(interval_offsets, dtype=np.uint64)
        assert len(interval_offsets) == len(descs) + 1
        status = np.zeros(len(descs), dtype=np.int32)
        st = lib().pq2_filtered_validity_batch_device(self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs), len(descs),
                                                      ptr(intervals), ptr(interval_offsets), C.c_void_p(d_validity_ptr),
                                                      validity_rows, C.c_void_p(d_counts_ptr), ptr(status),
                                                      C.c_void_p(stream))
        self._check(st)
        return status

    def decode_raw_pages_v1_device(self, d_buf_ptr, buf_bytes, pages, max_index_bits, d_rep_ptr, d_def_ptr, d_idx_ptr,
                                   out_values, stream=0):
        """Whole decompressed V1 data page buffers in device memory -> repetition levels, definition levels and
        dictionary indices (pq2_decode_raw_pages_v1_device).  pages: RAW_PAGE_DTYPE.  Returns the statuses."""
        pages = np.ascontiguousarray(pages, dtype=RAW_PAGE_DTYPE)
        status = np.zeros(len(pages), dtype=np.int32)
        st = lib().pq2_decode_raw_pages_v1_device(self._h, C.c_void_p(d_buf_ptr), buf_bytes, ptr(pages), len(pages),
                                                  max_index_bits, C.c_void_p(d_rep_ptr), C.c_void_p(d_def_ptr),
                                                  C.c_void_p(d_idx_ptr), out_values, ptr(status), C.c_void_p(stream))
        self._check(st)
        return status

    def fetch_status(self, n_pages):
        status = np.zeros(n_pages, dtype=np.int32)
        self._check(lib().pq2_fetch_status(self._h, ptr(status), n_pages))
        return status

    # ---- host-buffer batch decode -------------------------------------------
    def decode_batch_host(self, data, descs, out_values=None, out=None):
        """Returns (values ndarray of uint32, statuses ndarray of int32).  `out`: a uint32 array to decode into
        (page-locked memory lets the copies of the chunks overlap)."""
        data = as_u8(data)
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        if out_values is None:
            out_values = int((descs["out_off"] + descs["num_values"]).max()) if len(descs) else 0
        if out is None:
            out = np.zeros(out_values, dtype=np.uint32)
        assert out.dtype == np.uint32 and out.flags.c_contiguous and out.size >= out_values
        status = np.zeros(len(descs), dtype=np.int32)
        st = lib().pq2_decode_batch_host(self._h, ptr(data), data.size, ptr(descs), len(descs), ptr(out), out_values,
                                         ptr(status))
        self._check(st)
        return out, status

    def decode_page_host(self, data, num_bits, num_values):
        """HybridRleDecoder::try_new(data, num_bits, num_values).collect() for one stream."""
        data = as_u8(data)
        descs = np.zeros(1, dtype=PAGE_DESC_DTYPE)
        descs[0] = (0, data.size, num_bits, num_values, 0)
        out, status = self.decode_batch_host(data, descs, num_values)
        return out, int(status[0])

    # ---- bit-unpacking ---------------------------------------------------------
    def bitunpack_u32_host(self, packed, num_bits, length):
        packed = as_u8(packed)
        out = np.zeros(length, dtype=np.uint32)
        st = lib().pq2_bitunpack_u32_host(self._h, ptr(packed), packed.size, num_bits, length, ptr(out))
        self._check(st)
        return out if packed.size else out[:0]

    def bitunpack_u64_host(self, packed, num_bits, length):
        """bitpacked::Decoder::<u64>::try_new(packed, num_bits, length)?.collect()"""
        packed = as_u8(packed)
        out = np.zeros(length, dtype=np.uint64)
        st = lib().pq2_bitunpack_u64_host(self._h, ptr(packed), packed.size, num_bits, length, ptr(out))
        self._check(st)
        return out if packed.size else out[:0]

    def bitunpack_u64_device(self, d_packed_ptr, packed_len, num_bits, length, d_out_ptr, stream=0):
        self._check(lib().pq2_bitunpack_u64_device(self._h, C.c_void_p(d_packed_ptr), packed_len, num_bits, length,
                                                   C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    # ---- DELTA_BINARY_PACKED -----------------------------------------------------
    def delta_bitpacked_decode_batch_device(self, d_in_ptr, in_bytes, descs, d_out_ptr, out_values, stream=0):
        """delta_bitpacked::Decoder over every stream of the batch (pq2_delta_bitpacked_decode_batch_device); a
        descriptor's num_values is the room at its out_off, in int64.  Returns (statuses, num_values, consumed_bytes)."""
        descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
        status = np.zeros(len(descs), dtype=np.int32)
        counts = np.zeros(len(descs), dtype=np.uint64)
        consumed = np.zeros(len(descs), dtype=np.uint64)
        st = lib().pq2_delta_bitpacked_decode_batch_device(self._h, C.c_void_p(d_in_ptr), in_bytes, ptr(descs),
                                                           len(descs), C.c_void_p(d_out_ptr), out_values, ptr(counts),
                                                           ptr(consumed), ptr(status), C.c_void_p(stream))
        self._check(st)
        return status, counts, consumed

    def delta_bitpacked_decode_host(self, values, cap):
        """One stream in host memory: (values as int64, consumed_bytes); raises as the reference's decoder errs."""
        values = as_u8(values)
        out = np.zeros(max(cap, 1), dtype=np.int64)
        n, consumed = C.c_uint64(0), C.c_uint64(0)
        st = lib().pq2_delta_bitpacked_decode_host(self._h, ptr(values), values.size, ptr(out), cap, C.byref(n),
                                                   C.byref(consumed))
        self._check(st)
        return out[:n.value], consumed.value

    def bitunpack_u32_device(self, d_packed_ptr, packed_len, num_bits, length, d_out_ptr, stream=0):
        self._check(lib().pq2_bitunpack_u32_device(self._h, C.c_void_p(d_packed_ptr), packed_len, num_bits, length,
                                                   C.c_void_p(d_out_ptr), C.c_void_p(stream)))


_DEFAULT_CTX = {}


def default_context(device_id=0):
    """Process-wide context per device, used by the iterator-style mirrors."""
    ctx = _DEFAULT_CTX.get(device_id)
    if ctx is None:
        ctx = _DEFAULT_CTX[device_id] = Context(device_id)
    return ctx


def decode_batch_device_noalloc(device_id, d_in_ptr, in_bytes, d_descs_ptr, n_pages, max_bits, d_out_ptr, out_values,
                                d_status_ptr, stream=0):
    """pq2_decode_batch_device_noalloc: descriptors and statuses are device arrays of the caller; enqueue only."""
    st = lib().pq2_decode_batch_device_noalloc(device_id, C.c_void_p(d_in_ptr), in_bytes, C.c_void_p(d_descs_ptr),
                                               n_pages, max_bits, C.c_void_p(d_out_ptr), out_values,
                                               C.c_void_p(d_status_ptr), C.c_void_p(stream))
    if st != OK:
        raise_for_status(st, "pq2_decode_batch_device_noalloc: " + lib().pq2_strerror(st).decode())


def multi_decode(contexts, data, descs, out_values=None, out=None):
    """pq2_multi_decode: one batch over several contexts (one per GPU), a host thread each.  Returns
    (values ndarray of uint32, statuses ndarray of int32)."""
    data = as_u8(data)
    descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
    if out_values is None:
        out_values = int((descs["out_off"] + descs["num_values"]).max()) if len(descs) else 0
    if out is None:
        out = np.zeros(out_values, dtype=np.uint32)
    assert out.dtype == np.uint32 and out.flags.c_contiguous and out.size >= out_values
    status = np.zeros(len(descs), dtype=np.int32)
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    st = lib().pq2_multi_decode(handles, len(contexts), ptr(data), data.size, ptr(descs), len(descs), ptr(out),
                                out_values, ptr(status))
    if st != OK:
        msg = b"; ".join(lib().pq2_last_error(c._h) or b"" for c in contexts).decode()
        raise_for_status(st, "pq2_multi_decode: " + msg)
    return out, status


def page_costs(descs, data=None):
    """What the page split balances (pq2_page_costs): the bytes a page moves through HBM and, where the streams are
    at hand (`data`: the input buffer in host memory), 256 for every run header the page is estimated to hold."""
    descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
    costs = np.zeros(len(descs), dtype=np.uint64)
    if data is not None:
        data = np.ascontiguousarray(data, dtype=np.uint8)
    st = lib().pq2_page_costs(ptr(data) if data is not None else None, data.size if data is not None else 0,
                              ptr(descs), len(descs), costs.ctypes.data_as(u64p))
    if st != OK:
        raise_for_status(st, "pq2_page_costs")
    return costs.astype(np.int64)


def partition_pages(descs, n_parts, data=None):
    """Contiguous page ranges balanced on page cost: returns n_parts + 1 bounds.  Without `data` the cost is stream
    bytes + value bytes (pq2_partition_pages); with the input buffer, run headers count too (pq2_page_costs,
    pq2_partition_by_cost: what pq2_multi_decode does)."""
    descs = np.ascontiguousarray(descs, dtype=PAGE_DESC_DTYPE)
    bounds = np.zeros(n_parts + 1, dtype=np.uint64)
    if data is None:
        st = lib().pq2_partition_pages(ptr(descs), len(descs), n_parts, bounds.ctypes.data_as(u64p))
    else:
        costs = page_costs(descs, data).astype(np.uint64)
        st = lib().pq2_partition_by_cost(costs.ctypes.data_as(u64p), len(descs), n_parts, bounds.ctypes.data_as(u64p))
    if st != OK:
        raise_for_status(st, "pq2_partition_pages")
    return bounds.astype(np.int64)
