"""`parquet2::deserialize` (src/deserialize/mod.rs:9-15): page -> decoder-state adapters."""
from .binary import BinaryPageState  # noqa: F401
from .boolean import BooleanPageState  # noqa: F401
from .filtered_rle import FilteredHybridBitmapIter, FilteredHybridEncoded, FilteredHybridRleDecoderIter  # noqa: F401
from .fixed_len import FixedLenBinaryPageState  # noqa: F401
from .hybrid_rle import (HybridDecoderBitmapIter, HybridEncoded, HybridRleBooleanIter, HybridRleDecoderIter,  # noqa: F401
                         HybridRleIter)
from .native import Dictionary, NativePageState, native_cast  # noqa: F401
from .utils import DefLevelsDecoder, OptionalValues, SliceFilteredIter, dict_indices_decoder  # noqa: F401
