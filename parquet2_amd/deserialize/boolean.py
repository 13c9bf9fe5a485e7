"""`parquet2::deserialize::boolean` (src/deserialize/boolean.rs)."""
from ..encoding.hybrid_rle import BitmapIter
from ..error import InvalidParameter
from ..page import Encoding, Repetition, split_buffer
from . import utils


class BooleanPageState:
    """boolean.rs:13-42: PLAIN booleans are a bitmap."""

    def __init__(self, kind, validity=None, values=None):
        self.kind, self.validity, self.values = kind, validity, values

    @classmethod
    def try_new(cls, page, ctx=None):
        is_optional = page.descriptor.repetition == Repetition.Optional
        if page.encoding() == Encoding.Plain:
            _, _, values = split_buffer(page)
            if is_optional:
                return cls("Optional", validity=utils.DefLevelsDecoder.try_new(page, ctx),
                           values=BitmapIter(values, 0, len(values) * 8, ctx))
            return cls("Required", values=BitmapIter(values, 0, page.num_values(), ctx))
        raise InvalidParameter("Viewing page for encoding %s for boolean type not supported" % page.encoding())
