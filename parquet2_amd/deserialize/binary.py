"""`parquet2::deserialize::binary` (src/deserialize/binary.rs)."""
from ..error import FeatureNotSupported
from ..page import Encoding, Repetition, split_buffer
from . import utils
from .native import Dictionary


class BinaryIter:
    """PLAIN byte arrays: 4-byte LE length + bytes (src/encoding/plain_byte_array.rs)."""

    def __init__(self, values, length=None):
        self._values, self._left = memoryview(values), length

    def __iter__(self):
        return self

    def __next__(self):
        if len(self._values) < 4 or self._left == 0:
            raise StopIteration
        n = int.from_bytes(self._values[:4], "little")
        item = bytes(self._values[4:4 + n])
        self._values = self._values[4 + n:]
        if self._left is not None:
            self._left -= 1
        return item


class BinaryPageState:
    """binary.rs:23-72"""

    def __init__(self, kind, validity=None, values=None, dictionary=None):
        self.kind, self.validity, self.values, self.dictionary = kind, validity, values, dictionary

    @classmethod
    def try_new(cls, page, dict_, ctx=None):
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
                return cls("Optional", validity=utils.DefLevelsDecoder.try_new(page, ctx), values=BinaryIter(values))
            return cls("Required", values=BinaryIter(values, page.num_values()))
        raise FeatureNotSupported("Viewing page for encoding %s for binary type" % enc)
