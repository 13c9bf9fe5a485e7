"""parquet2_amd: MI355X (gfx950) decoder for Parquet's RLE / bit-packing hybrid encoding behind the
decoder API of jorgecarleitao/parquet2 (`encoding::{hybrid_rle, bitpacked, uleb128}`, `deserialize`).

The value decode runs in hand-written HIP kernels reached through the C ABI of
include/pq2hip.h (parquet2_amd/lib/libpq2hip.so); this package is the host-side
mirror of the reference's module tree.  There is no CPU decode path.
"""
from . import deserialize, encoding, error, indexes, page, read  # noqa: F401
from ._lib import (BLOCK_VALUES, DICT_REF_DTYPE, INTERVAL_DTYPE, PAGE_DESC_DTYPE, RAW_DICT_INDICES, RAW_PAGE_DTYPE, Context, default_context, lib, multi_decode, page_costs,
                   partition_pages)  # noqa: F401

__version__ = "0.1.0"
