"""`parquet2::read::levels` (src/read/levels.rs)."""
from .._lib import lib


def get_bit_width(max_level):
    """Bits needed to store levels up to `max_level` (src/read/levels.rs:3-5)."""
    return lib().pq2_get_bit_width(max_level)
