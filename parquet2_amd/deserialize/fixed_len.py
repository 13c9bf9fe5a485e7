"""`parquet2::deserialize::fixed_len` (src/deserialize/fixed_len.rs)."""
from ..error import FeatureNotSupported
from ..page import Encoding, Repetition, split_buffer
from . import utils
from .native import Dictionary


class FixexBinaryIter:
    """fixed_len.rs:12-35 (the reference spells it `FixexBinaryIter`)."""

    def __init__(self, values, size):
        self._values, self._size = memoryview(values), size

    def __iter__(self):
        return self

    def __next__(self):
        if len(self._values) == 0:
            raise StopIteration
        item = bytes(self._values[:self._size])
        self._values = self._values[self._size:]
        return item


class FixedLenBinaryPageState:
    """fixed_len.rs:50-110"""

    def __init__(self, kind, validity=None, values=None, dictionary=None):
        self.kind, self.validity, self.values, self.dictionary = kind, validity, values, dictionary

    @classmethod
    def try_new(cls, page, dict_, size, ctx=None):
        is_optional = page.descriptor.repetition == Repetition.Optional
        enc = page.encoding()
        if enc in (Encoding.PlainDictionary, Encoding.RleDictionary) and dict_ is not None:
            if is_optional:
                return cls("OptionalDictionary", validity=utils.DefLevelsDecoder.try_new(page, ctx),
                           dictionary=Dictionary(page, dict_, ctx))
            return cls("RequiredDictionary", dictionary=Dictionary(page, dict_, ctx))
        if enc == Encoding.Plain:
            _, _, values = split_buffer(page)
            if is_optional:
                return cls("Optional", validity=utils.DefLevelsDecoder.try_new(page, ctx),
                           values=FixexBinaryIter(values, size))
            return cls("Required", values=FixexBinaryIter(values, size))
        raise FeatureNotSupported("Viewing page for encoding %s for fixed-length binary type" % enc)
