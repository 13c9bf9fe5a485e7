"""`parquet2::deserialize::hybrid_rle` (src/deserialize/hybrid_rle.rs): run adapters for
definition levels whose maximum is 1 (bit-packed runs are bitmaps)."""
from dataclasses import dataclass

from ..encoding import hybrid_rle


@dataclass(frozen=True)
class Bitmap:
    """`HybridEncoded::Bitmap(slice, length)` (hybrid_rle.rs:5-13)."""
    values: bytes
    length: int

    def len(self):
        return self.length


@dataclass(frozen=True)
class Repeated:
    """`HybridEncoded::Repeated(is_set, length)`."""
    is_set: bool
    length: int

    def len(self):
        return self.length


class HybridEncoded:
    Bitmap = Bitmap
    Repeated = Repeated


class HybridRleIter:
    """Adapter over the runs of `hybrid_rle::Decoder` yielding `Bitmap` / `Repeated`, `length` items in
    total and no zero padding (hybrid_rle.rs:37-120)."""

    def __init__(self, iter_, length):
        self._iter = iter(iter_)
        self._length = length
        self._consumed = 0

    new = classmethod(lambda cls, iter_, length: cls(iter_, length))

    def len(self):
        return self._length - self._consumed

    def number_of_elements(self):
        return self.len()

    def is_empty(self):
        return self.len() == 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._consumed == self._length:
            raise StopIteration
        run = next(self._iter)
        if isinstance(run, hybrid_rle.Bitpacked):
            additional = min(len(run.data) * 8, self.len())
            self._consumed += additional
            return Bitmap(run.data, additional)
        additional = min(run.run_length, self.len())
        self._consumed += additional
        return Repeated(run.data[0] == 1, additional)


def HybridDecoderBitmapIter(values, num_values):
    """`HybridRleIter<hybrid_rle::Decoder>` as built by DefLevelsDecoder (utils.rs:46-47)."""
    return HybridRleIter(hybrid_rle.Decoder(values, 1), num_values)


class HybridRleBooleanIter:
    """Runs -> bools (hybrid_rle.rs:125-202).  As in the reference, `current_run` is never cleared,
    so the iterator ends with the first run."""

    def __init__(self, iter_):
        self._iter = iter_
        self._current = None

    new = classmethod(lambda cls, iter_: cls(iter_))

    def __iter__(self):
        return self

    def __next__(self):
        if self._current is None:
            run = next(self._iter)
            if isinstance(run, Bitmap):
                self._current = iter(hybrid_rle.BitmapIter(run.values, 0, run.length))
            else:
                self._current = iter([run.is_set] * run.length)
        return next(self._current)

    def size_hint(self):
        n = self._iter.number_of_elements()
        return n, n


def HybridRleDecoderIter(values, num_values):
    return HybridRleBooleanIter(HybridDecoderBitmapIter(values, num_values))
