"""`parquet2::deserialize::filtered_rle` (src/deserialize/filtered_rle.rs): intersects the runs of a
max-def-level-1 column with the selected row intervals.  Host-side, run-granular."""
from collections import deque
from dataclasses import dataclass

from ..encoding.hybrid_rle import BitmapIter
from ..indexes import Interval
from .hybrid_rle import Bitmap, HybridDecoderBitmapIter, Repeated


@dataclass(frozen=True)
class FBitmap:
    values: bytes
    offset: int
    length: int

    def len(self):
        return self.length


@dataclass(frozen=True)
class FRepeated:
    is_set: bool
    length: int

    def len(self):
        return self.length


@dataclass(frozen=True)
class Skipped:
    set_count: int

    def len(self):
        return 0


class FilteredHybridEncoded:
    """`FilteredHybridEncoded::{Bitmap, Repeated, Skipped}` (filtered_rle.rs:14-28)."""
    Bitmap = FBitmap
    Repeated = FRepeated
    Skipped = Skipped


def _is_set_count(values, offset, length):
    """filtered_rle.rs:30-34"""
    return int(BitmapIter(values, offset, length).collect().sum()) if length else 0


class FilteredHybridBitmapIter:
    """filtered_rle.rs:53-274"""

    def __init__(self, iter_, selected_rows):
        self._iter = iter(iter_)
        self._rows = deque(Interval(i.start, i.length) for i in selected_rows)
        self._current = None           # (run, offset)
        self._interval = None
        self._items_in_runs = 0
        self._total = sum(i.length for i in self._rows)

    new = classmethod(lambda cls, iter_, selected_rows: cls(iter_, selected_rows))

    def len(self):
        return self._total

    def __iter__(self):
        return self

    def _advance_interval(self, length):
        if self._interval is not None:
            self._interval.start += length
            self._interval.length -= length
            self._total -= length

    def __next__(self):
        while True:
            if self._interval is None:
                if not self._rows:
                    raise StopIteration
                self._interval = self._rows.popleft()
                continue
            if self._current is None:
                self._current = (next(self._iter), 0)
                continue
            break
        interval = self._interval
        run, offset = self._current
        full = run.length
        run_length = full - offset
        interval_start = interval.start - self._items_in_runs
        is_bitmap = isinstance(run, Bitmap)

        if interval_start > 0:
            max_skip = full - offset
            to_skip = min(interval_start, max_skip)
            if is_bitmap:
                set_count = _is_set_count(run.values, offset, to_skip)
            else:
                set_count = to_skip if run.is_set else 0
            self._items_in_runs += to_skip
            self._current = None if to_skip == max_skip else (run, offset + to_skip)
            return Skipped(set_count)

        new_offset = offset + interval_start
        if run_length > interval.length:
            length = interval.length
            self._items_in_runs += length
            self._total -= length
            self._interval = self._rows.popleft() if self._rows else None
            self._current = (run, offset + length)
        else:
            self._items_in_runs += run_length
            length = min(run_length, full - new_offset)
            self._advance_interval(length)
            self._current = None
        if is_bitmap:
            return FBitmap(run.values, new_offset, length)
        return FRepeated(run.is_set, length)


def FilteredHybridRleDecoderIter(values, num_values, selected_rows):
    """`FilteredHybridBitmapIter<HybridDecoderBitmapIter>` (filtered_rle.rs:8-10)."""
    return FilteredHybridBitmapIter(HybridDecoderBitmapIter(values, num_values), selected_rows)
