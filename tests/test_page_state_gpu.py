"""The page -> decoder-state adapters (src/deserialize/{native,binary,fixed_len,boolean,utils}.rs) driven the way
the reference's own integration tests drive them: `page_to_vec` (tests/it/read/primitive.rs:87-116,
tests/it/read/binary.rs, fixed_binary.rs, boolean.rs) over `deserialize_optional` (tests/it/read/utils.rs:7-66).
Every level stream and index stream is decoded on the GPU through the C ABI; the expected rows come from the
values the pages were written from, and the streams are also decoded with the CPU oracle.

The reference's tests of these adapters read fixture files that are not in its tree: beyond the definitions cited
above, parity of this layer is unpinned."""
import struct

import numpy as np
import pytest

from parquet2_amd import synth
from parquet2_amd._lib import OK
from parquet2_amd.deserialize import (BinaryPageState, BooleanPageState, DefLevelsDecoder, FixedLenBinaryPageState,
                                      HybridRleDecoderIter, NativePageState, OptionalValues, SliceFilteredIter,
                                      native_cast)
from parquet2_amd.deserialize.hybrid_rle import Bitmap, Repeated
from parquet2_amd.encoding.hybrid_rle import BitmapIter, encode_u32
from parquet2_amd.error import FeatureNotSupported, InvalidParameter, OutOfSpec, ReferencePanic
from parquet2_amd.indexes import Interval
from parquet2_amd.page import (DataPage, DataPageHeaderV1, DataPageHeaderV2, Descriptor, Encoding, Repetition)
from parquet2_amd.read.levels import get_bit_width

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["per_page", "ring", "scout"])
def ctx(request):
    import os

    from parquet2_amd import Context

    os.environ["PQ2_MODE"] = {"per_page": "page", "ring": "ring", "scout": "scout"}[request.param]
    try:
        c = Context(0)
    finally:
        del os.environ["PQ2_MODE"]
    yield c
    c.close()


# ---- writers of test pages -------------------------------------------------------------------------------------

def levels_stream(levels, max_level, mixed=True):
    w = get_bit_width(max_level)
    lv = np.asarray(levels, dtype=np.uint32)
    return bytes(synth.encode_hybrid(lv, w)) if mixed else bytes(encode_u32(lv, w))


def make_page(values_bytes, num_values, encoding, max_def, def_stream=b"", v2=False, selected_rows=None,
              repetition=None):
    """A decompressed data page: V1 prefixes each level stream with its 4-byte length (src/page/mod.rs:350-398),
    V2 carries the lengths in the header (src/page/mod.rs:400-411)."""
    desc = Descriptor(max_def_level=max_def, max_rep_level=0,
                      repetition=repetition or (Repetition.Optional if max_def > 0 else Repetition.Required))
    if v2:
        header = DataPageHeaderV2(num_values, len(def_stream), 0, encoding)
        buf = def_stream + values_bytes
    else:
        header = DataPageHeaderV1(num_values, encoding)
        buf = (struct.pack("<I", len(def_stream)) + def_stream if max_def > 0 else b"") + values_bytes
    return DataPage(header, buf, desc, selected_rows)


def index_bytes(indices, bits, mixed=True):
    idx = np.asarray(indices, dtype=np.uint32)
    body = bytes(synth.encode_hybrid(idx, bits)) if mixed and bits > 0 else bytes(encode_u32(idx, bits))
    return bytes([bits]) + body


# ---- the reference's consumer loops ----------------------------------------------------------------------------

def deserialize_optional(validity, values, ctx):
    """tests/it/read/utils.rs:7-66."""
    values = iter(values)
    out = []
    if validity.kind == "Bitmap":
        for run in validity.iter:
            if isinstance(run, Bitmap):
                for x in BitmapIter(run.values, 0, run.length, ctx):
                    out.append(next(values, None) if x else None)
            else:
                assert isinstance(run, Repeated)
                if run.is_set:
                    for _ in range(run.length):
                        try:
                            out.append(next(values))
                        except StopIteration:
                            break
                else:
                    out.extend([None] * run.length)
        return out
    assert validity.kind == "Levels"
    for x in validity.iter:
        out.append(next(values, None) if x == validity.max_level else None)
    return out


def page_to_vec(page, dict_, dtype, ctx):
    """tests/it/read/primitive.rs:38-116 (PageState::try_new + page_to_vec)."""
    assert page.descriptor.max_rep_level == 0
    if page.selected_rows is not None:
        is_optional = page.descriptor.repetition == Repetition.Optional
        if page.encoding() != Encoding.Plain:
            raise FeatureNotSupported("Viewing page for encoding %s for native type" % page.encoding())
        if is_optional:
            from parquet2_amd.page import split_buffer

            _, def_levels, _ = split_buffer(page)
            validity = HybridRleDecoderIter(def_levels, page.num_values())
            return list(SliceFilteredIter(OptionalValues(validity, native_cast(page, dtype)), page.selected_rows))
        return list(SliceFilteredIter(native_cast(page, dtype), page.selected_rows))
    state = NativePageState.try_new(page, dict_, dtype, ctx)
    if state.kind == "Optional":
        return deserialize_optional(state.validity, state.values, ctx)
    if state.kind == "Required":
        return list(state.values)
    if state.kind == "RequiredDictionary":
        return [state.dictionary.dict[x] for x in state.dictionary.indexes]
    assert state.kind == "OptionalDictionary"
    return deserialize_optional(state.validity, (state.dictionary.dict[x] for x in state.dictionary.indexes), ctx)


def rows_of(levels, max_level, values):
    """What the page was written from: row i holds the next value iff its level is the maximum."""
    it = iter(values)
    return [next(it) if lv == max_level else None for lv in levels]


def same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None) and (x is None or x == y), (i, x, y)


# ---- native ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("dtype", ["<i4", "<i8", "<f4", "<f8"])
def test_native_required_plain(ctx, dtype, v2):
    rng = np.random.default_rng(5)
    vals = rng.integers(-1 << 30, 1 << 30, 5003).astype(dtype)
    page = make_page(vals.tobytes(), vals.size, Encoding.Plain, 0, v2=v2)
    state = NativePageState.try_new(page, None, dtype, ctx)
    assert state.kind == "Required"
    same(page_to_vec(page, None, dtype, ctx), list(vals))


@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("max_def", [1, 2, 3, 7])
def test_native_optional_plain(ctx, oracle, max_def, v2):
    """max_def_level 1 -> DefLevelsDecoder::Bitmap (runs, utils.rs:46-47); above 1 -> ::Levels (decoded values,
    utils.rs:49-56)."""
    n = 40000 + max_def
    rng = np.random.default_rng(max_def)
    valid = synth.def_values(7 + max_def, n, 30.0, 9.0).astype(np.uint32)
    levels = np.where(valid == 1, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    vals = rng.integers(-1 << 40, 1 << 40, int((levels == max_def).sum())).astype("<i8")
    stream = levels_stream(levels, max_def)
    st, lv_o = oracle.hybrid_rle_decode(np.frombuffer(stream, dtype=np.uint8), get_bit_width(max_def), n)
    assert st == OK and np.array_equal(lv_o, levels)
    page = make_page(vals.tobytes(), n, Encoding.Plain, max_def, stream, v2=v2)
    state = NativePageState.try_new(page, None, "<i8", ctx)
    assert state.kind == "Optional"
    assert state.validity.kind == ("Bitmap" if max_def == 1 else "Levels")
    if max_def > 1:
        assert state.validity.max_level == max_def
        assert np.array_equal(state.validity.iter.collect(), levels)
    same(page_to_vec(page, None, "<i8", ctx), rows_of(levels, max_def, vals))


@pytest.mark.parametrize("encoding", [Encoding.PlainDictionary, Encoding.RleDictionary])
@pytest.mark.parametrize("bits", [0, 1, 5, 12, 20])
def test_native_required_dictionary(ctx, oracle, bits, encoding):
    n = 30011
    table = np.random.default_rng(bits).integers(-1 << 30, 1 << 30, max(1, 1 << min(bits, 14))).astype("<i4")
    idx = (synth.dict_values(bits, n, max(bits, 1), 20.0, 64.0) % table.size).astype(np.uint32) if bits else \
        np.zeros(n, dtype=np.uint32)
    values = index_bytes(idx, bits)
    st, idx_o = oracle.hybrid_rle_decode(np.frombuffer(values[1:], dtype=np.uint8), bits, n)
    assert st == OK and np.array_equal(idx_o, idx)
    page = make_page(values, n, encoding, 0)
    state = NativePageState.try_new(page, table, "<i4", ctx)
    assert state.kind == "RequiredDictionary" and state.dictionary.len() == n
    same(page_to_vec(page, table, "<i4", ctx), list(table[idx]))


@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("max_def", [1, 2])
def test_native_optional_dictionary(ctx, max_def, v2):
    n = 50021
    rng = np.random.default_rng(11 + max_def)
    valid = synth.def_values(3, n, 100.0, 15.0).astype(np.uint32)
    levels = np.where(valid == 1, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    n_valid = int((levels == max_def).sum())
    table = rng.integers(0, 1 << 62, 4096).astype("<i8")
    idx = synth.dict_values(9, n_valid, 12, 40.0, 200.0).astype(np.uint32)
    # the index stream of an optional page holds the non-null rows only; its decoder is told num_values rows
    # (utils.rs:13-27), so the stream ends before the count does and the rest reads as zeros
    page = make_page(index_bytes(idx, 12), n, Encoding.RleDictionary, max_def, levels_stream(levels, max_def), v2=v2)
    state = NativePageState.try_new(page, table, "<i8", ctx)
    assert state.kind == "OptionalDictionary"
    assert state.validity.kind == ("Bitmap" if max_def == 1 else "Levels")
    assert np.array_equal(state.dictionary.indexes.collect()[:n_valid], idx)
    assert not state.dictionary.indexes.collect()[n_valid:].any()
    same(page_to_vec(page, table, "<i8", ctx), rows_of(levels, max_def, table[idx]))


def test_native_error_arms(ctx):
    vals = np.arange(16, dtype="<i4")
    # an encoding the adapter does not view (native.rs:92-96)
    with pytest.raises(FeatureNotSupported):
        NativePageState.try_new(make_page(vals.tobytes(), 16, Encoding.Rle, 0), None, "<i4", ctx)
    # a dictionary encoding without its dictionary falls to the same arm
    with pytest.raises(FeatureNotSupported):
        NativePageState.try_new(make_page(index_bytes(vals, 5), 16, Encoding.RleDictionary, 0), None, "<i4", ctx)
    with pytest.raises(FeatureNotSupported):
        NativePageState.try_new(make_page(index_bytes(vals, 5), 16, Encoding.PlainDictionary, 1,
                                          levels_stream(np.ones(16), 1)), None, "<i4", ctx)
    # values whose length is no multiple of the type (native.rs:15-26)
    with pytest.raises(OutOfSpec):
        NativePageState.try_new(make_page(vals.tobytes()[:-1], 16, Encoding.Plain, 0), None, "<i4", ctx)
    # index bit width above 32; an empty values buffer, whose first byte the reference indexes (utils.rs:13-27)
    with pytest.raises(OutOfSpec):
        NativePageState.try_new(make_page(bytes([33, 3, 0]), 16, Encoding.RleDictionary, 0), vals, "<i4", ctx)
    with pytest.raises(ReferencePanic):
        NativePageState.try_new(make_page(b"", 16, Encoding.RleDictionary, 0), vals, "<i4", ctx)
    # a V1 page whose level length runs past the buffer is an error (src/page/mod.rs:350-398); a V2 page whose
    # header's lengths do is sliced without a check (src/page/mod.rs:400-411)
    bad = DataPage(DataPageHeaderV1(16, Encoding.Plain), struct.pack("<I", 99) + b"\x03\x01",
                   Descriptor(1, 0, Repetition.Optional))
    with pytest.raises(OutOfSpec):
        NativePageState.try_new(bad, None, "<i4", ctx)
    bad = DataPage(DataPageHeaderV2(16, 40, 0, Encoding.Plain), b"\x03\x01" + vals.tobytes()[:8],
                   Descriptor(1, 0, Repetition.Optional))
    with pytest.raises(ReferencePanic):
        NativePageState.try_new(bad, None, "<i4", ctx)
    # a level stream of width 9 whose RLE run has one of its two value bytes: the reference panics when it
    # reads the run (Levels arm); with none left the header is dropped and the levels read as zeros
    assert not any(NativePageState.try_new(make_page(vals.tobytes(), 16, Encoding.Plain, 300, bytes([32])), None,
                                           "<i4", ctx).validity.iter)
    page = make_page(vals.tobytes(), 16, Encoding.Plain, 300, bytes([32, 1]))
    with pytest.raises(ReferencePanic):
        NativePageState.try_new(page, None, "<i4", ctx)


@pytest.mark.parametrize("optional", [False, True])
def test_native_selected_rows(ctx, optional):
    """PageState::Filtered (tests/it/read/primitive.rs:46-80): SliceFilteredIter over OptionalValues over
    HybridRleDecoderIter, which ends with the first run of the level stream (hybrid_rle.rs:125-202)."""
    n = 4000
    vals = np.arange(n, dtype="<i4") * 3
    rows = [Interval(5, 10), Interval(100, 1), Interval(900, 250)]
    if optional:
        levels = (np.arange(n) % 7 != 0).astype(np.uint32)
        page = make_page(vals[:int(levels.sum())].tobytes(), n, Encoding.Plain, 1, levels_stream(levels, 1, mixed=False),
                         selected_rows=rows)
        full = rows_of(levels, 1, vals)
    else:
        page = make_page(vals.tobytes(), n, Encoding.Plain, 0, selected_rows=rows)
        full = list(vals)
    want = [x for iv in rows for x in full[iv.start:iv.start + iv.length]]
    same(page_to_vec(page, None, "<i4", ctx), want)
    with pytest.raises(FeatureNotSupported):
        page_to_vec(make_page(index_bytes(vals % 8, 3), n, Encoding.RleDictionary, 0, selected_rows=rows), vals,
                    "<i4", ctx)


# ---- binary, fixed-length binary, boolean ----------------------------------------------------------------------

def plain_binary(items):
    return b"".join(struct.pack("<I", len(x)) + x for x in items)


@pytest.mark.parametrize("max_def", [0, 1, 2])
def test_binary_page_states(ctx, max_def):
    rng = np.random.default_rng(20 + max_def)
    n = 6007
    levels = np.full(n, max_def, dtype=np.uint32) if max_def == 0 else \
        np.where(rng.random(n) < 0.8, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    n_valid = int((levels == max_def).sum())
    items = [bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(n_valid)]
    def_stream = levels_stream(levels, max_def) if max_def else b""
    page = make_page(plain_binary(items), n, Encoding.Plain, max_def, def_stream)
    state = BinaryPageState.try_new(page, None, ctx)
    if max_def == 0:
        assert state.kind == "Required"
        same(list(state.values), items)
    else:
        assert state.kind == "Optional" and state.validity.kind == ("Bitmap" if max_def == 1 else "Levels")
        same(deserialize_optional(state.validity, state.values, ctx), rows_of(levels, max_def, items))
    # the same rows through a dictionary of byte arrays
    table = [bytes([k, k ^ 0x55]) * (k % 5) for k in range(200)]
    idx = rng.integers(0, 200, n_valid).astype(np.uint32)
    page = make_page(index_bytes(idx, 8), n, Encoding.RleDictionary, max_def, def_stream)
    state = BinaryPageState.try_new(page, table, ctx)
    looked_up = (table[x] for x in state.dictionary.indexes)
    if max_def == 0:
        assert state.kind == "RequiredDictionary"
        same(list(looked_up), [table[i] for i in idx])
    else:
        assert state.kind == "OptionalDictionary"
        same(deserialize_optional(state.validity, looked_up, ctx), rows_of(levels, max_def, [table[i] for i in idx]))
    with pytest.raises(FeatureNotSupported):
        BinaryPageState.try_new(make_page(b"", n, Encoding.Rle, max_def, def_stream), None, ctx)
    with pytest.raises(FeatureNotSupported):
        BinaryPageState.try_new(make_page(index_bytes(idx, 8), n, Encoding.PlainDictionary, max_def, def_stream), None,
                                ctx)


@pytest.mark.parametrize("max_def", [0, 1, 3])
def test_fixed_len_binary_page_states(ctx, max_def):
    rng = np.random.default_rng(30 + max_def)
    n, size = 5003, 12
    levels = np.full(n, max_def, dtype=np.uint32) if max_def == 0 else \
        np.where(rng.random(n) < 0.7, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    n_valid = int((levels == max_def).sum())
    raw = rng.integers(0, 256, n_valid * size, dtype=np.uint8).tobytes()
    items = [raw[i * size:(i + 1) * size] for i in range(n_valid)]
    def_stream = levels_stream(levels, max_def) if max_def else b""
    state = FixedLenBinaryPageState.try_new(make_page(raw, n, Encoding.Plain, max_def, def_stream), None, size, ctx)
    if max_def == 0:
        assert state.kind == "Required"
        same(list(state.values), items)
    else:
        assert state.kind == "Optional"
        same(deserialize_optional(state.validity, state.values, ctx), rows_of(levels, max_def, items))
    table = [bytes([k]) * size for k in range(64)]
    idx = rng.integers(0, 64, n_valid).astype(np.uint32)
    state = FixedLenBinaryPageState.try_new(make_page(index_bytes(idx, 6), n, Encoding.PlainDictionary, max_def,
                                                      def_stream), table, size, ctx)
    looked_up = (table[x] for x in state.dictionary.indexes)
    if max_def == 0:
        assert state.kind == "RequiredDictionary"
        same(list(looked_up), [table[i] for i in idx])
    else:
        assert state.kind == "OptionalDictionary"
        same(deserialize_optional(state.validity, looked_up, ctx), rows_of(levels, max_def, [table[i] for i in idx]))
    with pytest.raises(FeatureNotSupported):
        FixedLenBinaryPageState.try_new(make_page(raw, n, Encoding.Rle, max_def, def_stream), None, size, ctx)


@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("max_def", [0, 1, 2])
def test_boolean_page_states(ctx, max_def, v2):
    """PLAIN booleans are a bitmap, unpacked on the GPU (boolean.rs:13-42)."""
    from parquet2_amd.encoding.hybrid_rle import bitpacked_encode

    rng = np.random.default_rng(40 + max_def)
    n = 70001
    levels = np.full(n, max_def, dtype=np.uint32) if max_def == 0 else \
        np.where(rng.random(n) < 0.9, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    n_valid = int((levels == max_def).sum())
    bools = rng.random(n_valid) < 0.4
    def_stream = levels_stream(levels, max_def) if max_def else b""
    page = make_page(bitpacked_encode(bools.astype(np.uint8)), n, Encoding.Plain, max_def, def_stream, v2=v2)
    state = BooleanPageState.try_new(page, ctx)
    if max_def == 0:
        assert state.kind == "Required"
        same(list(state.values), [bool(b) for b in bools])
    else:
        assert state.kind == "Optional"
        # the values iterator covers every bit of the buffer (boolean.rs:27-28), the rows take the first n_valid
        assert state.values.size_hint()[0] == (n_valid + 7) // 8 * 8
        same(deserialize_optional(state.validity, state.values, ctx), rows_of(levels, max_def, [bool(b) for b in bools]))
    for enc in (Encoding.Rle, Encoding.RleDictionary, Encoding.PlainDictionary):
        with pytest.raises(InvalidParameter):
            BooleanPageState.try_new(make_page(b"\x01", 8, enc, max_def, def_stream), ctx)


# ---- utils.rs --------------------------------------------------------------------------------------------------

def test_def_levels_decoder_runs_and_levels(ctx, oracle):
    """Bitmap arm: the runs of the stream, clipped to num_values, without zero padding (hybrid_rle.rs:37-120);
    Levels arm: exactly num_values levels, zero-padded behind the stream (mod.rs:35-116)."""
    levels = np.r_[np.ones(1000), np.zeros(24), np.tile([1, 0, 1, 1, 0, 1, 1, 1], 40), np.ones(77)].astype(np.uint32)
    stream = levels_stream(levels, 1)
    for n in (levels.size, levels.size - 50, 500, 0):
        page = make_page(b"", n, Encoding.Plain, 1, stream)
        d = DefLevelsDecoder.try_new(page, ctx)
        assert d.kind == "Bitmap"
        got = []
        for run in d.iter:
            got.extend(BitmapIter(run.values, 0, run.length, ctx) if isinstance(run, Bitmap) else
                       [run.is_set] * run.length)
        assert got == [bool(x) for x in levels[:n]]
        runs_o = oracle.bitmap_runs(np.frombuffer(stream, dtype=np.uint8), n)
        assert sum(r.length for r in runs_o[1]) == n
    lv3 = (np.arange(9000) % 4).astype(np.uint32)
    stream = levels_stream(lv3, 3)
    for n in (9000, 8999, 9100, 1):
        d = DefLevelsDecoder.try_new(make_page(b"", n, Encoding.Plain, 3, stream), ctx)
        assert d.kind == "Levels" and d.max_level == 3
        st, exp = oracle.hybrid_rle_decode(np.frombuffer(stream, dtype=np.uint8), 2, n)
        assert st == OK and np.array_equal(d.iter.collect(), exp)
        assert np.array_equal(exp[:min(n, 9000)], lv3[:min(n, 9000)]) and not exp[9000:].any()


def test_optional_values(ctx):
    """utils.rs:60-90: a value is taken only where validity is set; the iterator ends with validity."""
    validity = BitmapIter(bytes([0b10110101, 0b00000011]), 0, 10, ctx)
    got = list(OptionalValues.new(validity, iter([10, 11, 12, 13, 14, 15, 16, 17])))
    assert got == [10, None, 11, None, 12, 13, None, 14, 15, 16]
    assert list(OptionalValues(iter([True, True, False, True]), iter([1]))) == [1, None, None, None]
    assert list(OptionalValues(iter([]), iter([1, 2]))) == []
