"""Pages written by pyarrow's C++ Parquet writer (tests/golden/pyarrow_pages.json) through the GPU path: the page
states of src/deserialize over V1 and V2 pages, RLE_DICTIONARY index streams behind their width byte, definition
and repetition levels.  Expected levels and rows come from what was written (see
tests/golden/make_pyarrow_pages.py), so this pins the decode of conforming streams independently of the oracle."""
import base64

import numpy as np
import pytest

from _golden_pages import bit_width, leaves_of, load_pages
from parquet2_amd import synth
from parquet2_amd._lib import OK
from parquet2_amd.deserialize import BinaryPageState, DefLevelsDecoder, NativePageState, dict_indices_decoder
from parquet2_amd.encoding.hybrid_rle import HybridRleDecoder
from parquet2_amd.page import (DataPage, DataPageHeaderV1, DataPageHeaderV2, Descriptor, Encoding, Repetition,
                               split_buffer)
from test_page_state_gpu import deserialize_optional

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


def as_page(p):
    buf = base64.b64decode(p["buffer"])
    enc = {"Plain": Encoding.Plain, "RleDictionary": Encoding.RleDictionary}[p["encoding"]]
    desc = Descriptor(p["max_def"], p["max_rep"], Repetition.Optional if p["max_def"] else Repetition.Required)
    if p["v2"]:
        return DataPage(DataPageHeaderV2(p["num_values"], p["def_len"], p["rep_len"], enc), buf, desc)
    return DataPage(DataPageHeaderV1(p["num_values"], enc), buf, desc)


def dictionary_of(p):
    if p["dictionary"] is None:
        return None
    return leaves_of(dict(p, leaves=None), decoded_indices=range(_dict_len(p)))


def _dict_len(p):
    raw = base64.b64decode(p["dictionary"])
    if p["kind"] != "str":
        return len(raw) // np.dtype("<" + p["kind"]).itemsize
    n = pos = 0
    while pos < len(raw):
        pos += 4 + int.from_bytes(raw[pos:pos + 4], "little")
        n += 1
    return n


def rows_of(p):
    """Flat columns: the rows that were written (None where the level is below the maximum)."""
    it = iter(p["leaves"])
    if p["max_def"] == 0:
        return list(it)
    return [next(it) if int(c) == p["max_def"] else None for c in p["def_levels"]]


def test_levels_and_indices_of_every_page(ctx):
    for i, p in enumerate(load_pages()):
        page = as_page(p)
        n = p["num_values"]
        rep, _, values = split_buffer(page)
        if p["max_rep"]:
            got = HybridRleDecoder.try_new(rep, bit_width(p["max_rep"]), n, ctx).collect()
            assert "".join(map(str, got.tolist())) == p["rep_levels"], (i, p["column"])
        n_leaves = n
        if p["max_def"]:
            d = DefLevelsDecoder.try_new(page, ctx)
            if p["max_def"] == 1:
                assert d.kind == "Bitmap"
                got = deserialize_optional(d, iter(lambda: 1, None), ctx)
                levels = "".join("0" if x is None else "1" for x in got)
            else:
                assert d.kind == "Levels" and d.max_level == p["max_def"]
                levels = "".join(map(str, d.iter.collect().tolist()))
            assert levels == p["def_levels"], (i, p["column"])
            n_leaves = p["def_levels"].count(str(p["max_def"]))
        if p["encoding"] == "RleDictionary":
            idx = dict_indices_decoder(page, ctx).collect()
            assert idx.size == n
            assert leaves_of(p, decoded_indices=idx[:n_leaves].tolist()) == p["leaves"], (i, p["column"])
        else:
            assert leaves_of(p, plain=values) == p["leaves"]


def test_flat_columns_through_the_page_states(ctx):
    """page_to_vec (tests/it/read/primitive.rs:87-116, binary.rs) over the flat columns: rows as written."""
    seen = set()
    for i, p in enumerate(load_pages()):
        if p["max_rep"] or p["max_def"] > 1:
            continue
        page, table = as_page(p), dictionary_of(p)
        if p["kind"] == "str":
            state = BinaryPageState.try_new(page, table, ctx)
        else:
            state = NativePageState.try_new(page, table, "<" + p["kind"], ctx)
        seen.add(state.kind)
        if state.kind == "Required":
            got = list(state.values)
        elif state.kind == "RequiredDictionary":
            got = [table[x] for x in state.dictionary.indexes]
        elif state.kind == "Optional":
            got = deserialize_optional(state.validity, state.values, ctx)
        else:
            got = deserialize_optional(state.validity, (table[x] for x in state.dictionary.indexes), ctx)
        if p["kind"] == "str":
            got = [x.decode() if isinstance(x, bytes) else x for x in got]
        assert got == rows_of(p), (i, p["column"], state.kind)
    assert seen == {"Required", "RequiredDictionary", "Optional", "OptionalDictionary"}


def test_all_pages_as_one_batch(ctx, oracle):
    """Every level and index stream of the fixture as one batch through pq2_decode_batch: what the states above
    decode page by page, here in a single launch, compared with the written levels and with the oracle."""
    batch = synth.PageBatch(align=1)
    want = []
    for p in load_pages():
        page = as_page(p)
        n = p["num_values"]
        rep, def_, values = split_buffer(page)
        if p["max_rep"]:
            batch.add(np.frombuffer(rep, dtype=np.uint8), bit_width(p["max_rep"]), n, out_align=1)
            want.append(np.array(list(map(int, p["rep_levels"])), dtype=np.uint32))
        if p["max_def"]:
            batch.add(np.frombuffer(def_, dtype=np.uint8), bit_width(p["max_def"]), n, out_align=1)
            want.append(np.array(list(map(int, p["def_levels"])), dtype=np.uint32))
        if p["encoding"] == "RleDictionary":
            stream = np.frombuffer(values, dtype=np.uint8)
            st, exp = oracle.hybrid_rle_decode(stream[1:], int(stream[0]), n)
            assert st == OK
            batch.add(stream[1:], int(stream[0]), n, out_align=1)
            want.append(exp)
    data, descs, out_values = batch.finish()
    got, status = ctx.decode_batch_host(data, descs, out_values)
    assert (status == OK).all() and len(want) == len(descs)
    for d, w in zip(descs, want):
        o = int(d["out_off"])
        assert np.array_equal(got[o:o + w.size], w)
