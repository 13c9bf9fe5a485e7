"""`parquet2::deserialize::native` (src/deserialize/native.rs)."""
import numpy as np

from ..error import FeatureNotSupported, OutOfSpec
from ..page import Encoding, Repetition, split_buffer
from . import utils


def native_cast(page, dtype):
    """PLAIN values viewed as `dtype` (native.rs:15-26)."""
    _, _, values = split_buffer(page)
    dt = np.dtype(dtype)
    if len(values) % dt.itemsize != 0:
        raise OutOfSpec("A primitive page data's len must be a multiple of the type")
    return np.frombuffer(values, dtype=dt.newbyteorder("<"))


class Dictionary:
    """native.rs:28-49: decoded indices + the dictionary they point into."""

    def __init__(self, page, dict_, ctx=None):
        self.indexes = utils.dict_indices_decoder(page, ctx)
        self.dict = dict_

    try_new = classmethod(lambda cls, page, dict_, ctx=None: cls(page, dict_, ctx))

    def len(self):
        return self.indexes.size_hint()[0]


class NativePageState:
    """native.rs:51-99: `Optional | Required | RequiredDictionary | OptionalDictionary`."""

    def __init__(self, kind, validity=None, values=None, dictionary=None):
        self.kind, self.validity, self.values, self.dictionary = kind, validity, values, dictionary

    @classmethod
    def try_new(cls, page, dict_, dtype, ctx=None):
        is_optional = page.descriptor.repetition == Repetition.Optional
        enc = page.encoding()
        if enc in (Encoding.PlainDictionary, Encoding.RleDictionary) and dict_ is not None:
            if is_optional:
                return cls("OptionalDictionary", validity=utils.DefLevelsDecoder.try_new(page, ctx),
                           dictionary=Dictionary(page, dict_, ctx))
            return cls("RequiredDictionary", dictionary=Dictionary(page, dict_, ctx))
        if enc == Encoding.Plain:
            if is_optional:
                return cls("Optional", validity=utils.DefLevelsDecoder.try_new(page, ctx),
                           values=native_cast(page, dtype))
            return cls("Required", values=native_cast(page, dtype))
        raise FeatureNotSupported("Viewing page for encoding %s for native type %s" % (enc, np.dtype(dtype).name))
