"""The slice of `parquet2::page` the decode path needs (src/page/mod.rs).

`DataPage` is a plain holder of what the decoders read from a decompressed
page: the buffer, the header kind (V1 / V2 with its level byte lengths), the
value count, the encoding and the column's maximum levels.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

from ._lib import Split, as_u8, lib, ptr
from .error import raise_for_status


class Encoding:
    """`parquet2::encoding::Encoding` values used on this path (src/parquet_bridge.rs:291-330)."""
    Plain = "Plain"
    PlainDictionary = "PlainDictionary"
    Rle = "Rle"
    RleDictionary = "RleDictionary"


class Repetition:
    Required = "Required"
    Optional = "Optional"
    Repeated = "Repeated"


@dataclass
class Descriptor:
    """`metadata::Descriptor` (src/metadata/column_descriptor.rs:9-18), levels only."""
    max_def_level: int = 0
    max_rep_level: int = 0
    repetition: str = Repetition.Required


@dataclass
class DataPageHeaderV1:
    num_values: int
    encoding: str = Encoding.Plain


@dataclass
class DataPageHeaderV2:
    num_values: int
    definition_levels_byte_length: int
    repetition_levels_byte_length: int
    encoding: str = Encoding.Plain


@dataclass
class DataPage:
    header: object
    buffer: bytes
    descriptor: Descriptor = field(default_factory=Descriptor)
    selected_rows: Optional[list] = None

    def num_values(self):
        return self.header.num_values

    def encoding(self):
        return self.header.encoding


def _split(st, s, buffer):
    raise_for_status(st, "split_buffer: the number of bytes declared in the levels is higher than the page size")
    b = bytes(buffer) if not isinstance(buffer, (bytes, memoryview)) else buffer
    return (b[s.rep_off:s.rep_off + s.rep_len], b[s.def_off:s.def_off + s.def_len], b[s.val_off:s.val_off + s.val_len])


def split_buffer_v1(buffer, has_rep, has_def):
    """(rep, def, values) of a V1 page: 4-byte LE length prefixes (src/page/mod.rs:350-398)."""
    a = as_u8(buffer)
    s = Split()
    st = lib().pq2_split_buffer_v1(ptr(a), a.size, int(bool(has_rep)), int(bool(has_def)), C.byref(s))
    return _split(st, s, buffer)


def split_buffer_v2(buffer, rep_level_buffer_length, def_level_buffer_length):
    """(rep, def, values) of a V2 page: lengths from the header (src/page/mod.rs:400-411)."""
    s = Split()
    st = lib().pq2_split_buffer_v2(len(buffer), rep_level_buffer_length, def_level_buffer_length, C.byref(s))
    return _split(st, s, buffer)


def split_buffer(page):
    """`page::split_buffer(&DataPage)` (src/page/mod.rs:413-431)."""
    if isinstance(page.header, DataPageHeaderV2):
        return split_buffer_v2(page.buffer, page.header.repetition_levels_byte_length,
                               page.header.definition_levels_byte_length)
    return split_buffer_v1(page.buffer, page.descriptor.max_rep_level > 0, page.descriptor.max_def_level > 0)
