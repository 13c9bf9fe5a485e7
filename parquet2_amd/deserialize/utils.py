"""`parquet2::deserialize::utils` (src/deserialize/utils.rs)."""
import ctypes as C
from collections import deque

from .._lib import as_u8, lib, ptr
from ..encoding.hybrid_rle import HybridRleDecoder
from ..error import OutOfSpec, raise_for_status
from ..indexes import Interval
from ..page import split_buffer
from ..read.levels import get_bit_width
from .hybrid_rle import HybridDecoderBitmapIter


def dict_indices_decoder(page, ctx=None):
    """Decoder of a dictionary page's indices: the first byte of the values buffer is the bit width
    (utils.rs:13-27)."""
    _, _, indices_buffer = split_buffer(page)
    a = as_u8(indices_buffer)
    bit_width, off = C.c_uint32(), C.c_uint64()
    st = lib().pq2_dict_indices_locate(ptr(a), a.size, C.byref(bit_width), C.byref(off))
    if st == 1:
        raise OutOfSpec("Bit width of dictionary pages cannot be larger than 32")
    raise_for_status(st, "dict_indices_decoder: empty values buffer")
    return HybridRleDecoder(indices_buffer[off.value:], bit_width.value, page.num_values(), ctx)


class DefLevelsDecoder:
    """utils.rs:30-58: `Bitmap` runs when max_def_level == 1, decoded `Levels` otherwise."""

    def __init__(self, kind, iter_, max_level=None):
        self.kind = kind  # "Bitmap" | "Levels"
        self.iter = iter_
        self.max_level = max_level

    @classmethod
    def try_new(cls, page, ctx=None):
        _, def_levels, _ = split_buffer(page)
        max_def_level = page.descriptor.max_def_level
        if max_def_level == 1:
            return cls("Bitmap", HybridDecoderBitmapIter(def_levels, page.num_values()))
        it = HybridRleDecoder(def_levels, get_bit_width(max_def_level), page.num_values(), ctx)
        return cls("Levels", it, max_def_level)


class OptionalValues:
    """Zips validity with values (utils.rs:60-90)."""

    def __init__(self, validity, values):
        self._validity = iter(validity)
        self._values = iter(values)

    new = classmethod(lambda cls, validity, values: cls(validity, values))

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._values, None) if next(self._validity) else None


class SliceFilteredIter:
    """Keeps the items that fall inside the selected row intervals (utils.rs:92-148)."""

    def __init__(self, iter_, selected_rows):
        self._iter = iter(iter_)
        self._rows = deque(Interval(i.start, i.length) for i in selected_rows)
        self._current_remaining = 0
        self._current = 0
        self._total = sum(i.length for i in self._rows)

    new = classmethod(lambda cls, iter_, selected_rows: cls(iter_, selected_rows))

    def __iter__(self):
        return self

    def __next__(self):
        if self._current_remaining == 0:
            if not self._rows:
                raise StopIteration
            interval = self._rows.popleft()
            for _ in range(interval.start - self._current):  # iter.nth(start - current)
                next(self._iter)
            item = next(self._iter)
            self._current = interval.start + interval.length
            self._current_remaining = interval.length - 1
            self._total -= 1
            return item
        self._current_remaining -= 1
        self._total -= 1
        return next(self._iter)

    def size_hint(self):
        return self._total, self._total
