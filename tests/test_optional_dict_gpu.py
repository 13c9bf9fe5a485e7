"""Parity of pq2_decode_optional_dict_batch_device (definition levels + dictionary indices -> nullable values on the
GPU) against the CPU oracle's decoded streams combined as the reference's consumer does
(deserialize_levels, tests/it/read/utils.rs:52-66, over an OptionalDictionary page,
tests/it/read/primitive.rs:104-110): the k-th row whose level equals the maximum takes the dictionary value of the
k-th index.  The reference's tests of this step read fixture files that are not in the tree: parity unpinned beyond
that definition."""
import numpy as np
import pytest

from parquet2_amd import DICT_REF_DTYPE, PAGE_DESC_DTYPE, synth
from parquet2_amd._lib import OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC
from parquet2_amd.encoding.hybrid_rle import encode_u32
from parquet2_amd.read.levels import get_bit_width

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["per_tile", "per_page", "ring", "scout"])
def ctx(request):
    import os

    from parquet2_amd import Context

    os.environ["PQ2_MODE"] = {"per_tile": "tile", "per_page": "page", "ring": "ring", "scout": "scout"}[request.param]
    try:
        c = Context(0)
    finally:
        del os.environ["PQ2_MODE"]
    yield c
    c.close()


class Batch:
    def __init__(self):
        self.chunks, self.at, self.rows = [], 0, 0
        self.defs, self.idxs, self.maxes, self.refs = [], [], [], []

    def _put(self, stream):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        self.chunks.append(stream)
        off = self.at
        self.at += stream.size
        return off, stream.size

    def add(self, def_stream, max_level, rows, idx_stream, idx_bits, idx_values, dict_ref, gap=0):
        o, l = self._put(def_stream)
        self.rows += gap
        self.defs.append((o, l, get_bit_width(max_level), rows, self.rows))
        o, l = self._put(idx_stream)
        self.idxs.append((o, l, idx_bits, idx_values, 0))
        self.maxes.append(max_level)
        self.refs.append(dict_ref)
        self.rows += rows

    def run(self, ctx, dict_bytes, value_size):
        import torch

        data = np.concatenate(self.chunks + [np.zeros(16, dtype=np.uint8)])
        d_in = torch.from_numpy(data).cuda()
        d_dict = torch.from_numpy(np.ascontiguousarray(dict_bytes)).cuda()
        words = value_size // 4
        d_out = torch.full((self.rows * words,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        d_val = torch.full((self.rows,), 7, dtype=torch.uint8, device="cuda")
        status = ctx.decode_optional_dict_batch_device(
            d_in.data_ptr(), d_in.numel(), np.array(self.defs, dtype=PAGE_DESC_DTYPE),
            np.array(self.idxs, dtype=PAGE_DESC_DTYPE), self.maxes, d_dict.data_ptr(), d_dict.numel(),
            np.array(self.refs, dtype=DICT_REF_DTYPE), value_size, d_out.data_ptr(), d_val.data_ptr(), self.rows,
            torch.cuda.current_stream().cuda_stream)
        return d_out.cpu().numpy().view(np.uint32).reshape(-1, words), d_val.cpu().numpy(), status


def expected(oracle, def_stream, max_level, rows, idx_stream, idx_bits, idx_values, table):
    st, levels = oracle.hybrid_rle_decode(def_stream, get_bit_width(max_level), rows)
    assert st == OK
    st, idx = oracle.hybrid_rle_decode(idx_stream, idx_bits, idx_values)
    assert st == OK
    valid = levels == max_level
    rank = np.cumsum(valid) - 1
    present = valid & (rank < idx_values)
    consumed = idx[rank[present]]
    bad = consumed >= len(table)
    out = np.zeros((rows, table.shape[1]), dtype=np.uint32)
    vals = np.zeros((int(present.sum()), table.shape[1]), dtype=np.uint32)
    vals[~bad] = table[consumed[~bad]]
    out[present] = vals
    return out, present.astype(np.uint8), bool(bad.any())


@pytest.mark.parametrize("value_size", [4, 8, 12])
def test_nullable_pages_match_the_consumer_loop(ctx, oracle, value_size):
    rng = np.random.default_rng(300 + value_size)
    words = value_size // 4
    table = rng.integers(0, 1 << 32, size=(1 << 9, words), dtype=np.uint64).astype(np.uint32)
    b = Batch()
    pages = []
    for p, (max_level, rows) in enumerate([(1, 65536), (1, 70001), (2, 30000), (3, 4099), (1, 5), (7, 100000)]):
        if max_level == 1:
            def_stream, _ = synth.def_levels_page(synth.seed_for(3, 80 + p), rows)
        else:  # levels 0..max in stretches, as nested or optional-in-optional columns have them
            lv = np.repeat(rng.integers(0, max_level + 1, size=rows // 7 + 1), 7)[:rows].astype(np.uint32)
            def_stream = np.frombuffer(encode_u32(lv, get_bit_width(max_level)), dtype=np.uint8)
        idx_stream, _ = synth.dict_page(synth.seed_for(2, 90 + p), rows, 9, 12.0, 50.0)
        idx_values = rows if p != 2 else rows // 3  # page 2: fewer indices than valid rows
        args = (def_stream, max_level, rows, idx_stream, 9, idx_values)
        b.add(*args, (0, len(table), 0), gap=p % 3)
        pages.append(args)
    got, validity, status = b.run(ctx, table.view(np.uint8).reshape(-1), value_size)
    assert (status == OK).all()
    covered = np.zeros(b.rows, dtype=bool)
    for d, args in zip(b.defs, pages):
        out, present, bad = expected(oracle, *args, table)
        o, rows = d[4], d[3]
        assert not bad
        assert (validity[o:o + rows] == present).all() and (got[o:o + rows] == out).all(), args[1:3]
        covered[o:o + rows] = True
    assert (got[~covered] == 0x5A5A5A5A).all() and (validity[~covered] == 7).all()
    assert 0.02 < 1.0 - validity[:65536].mean() < 0.08  # the 5 % nulls of the definition-level pages


def test_large_pages_in_segments(ctx, oracle):
    """Few large pages: the rows of a page are shared out over workgroups of 65 536 rows each, which have to agree
    on how many valid rows lie before them.  Pages that end inside a segment, a page of exactly two segments, a
    dictionary that misses an index only a row of a late segment consumes, an index stream that is too short."""
    rng = np.random.default_rng(77)
    table = rng.integers(0, 1 << 32, size=(1 << 9, 2), dtype=np.uint64).astype(np.uint32)
    b = Batch()
    pages = []
    shapes = [(1, 700001, None, 512), (1, 131072, None, 512), (3, 300000, None, 512), (1, 400000, 150000, 512),
              (1, 262145, None, 512)]
    for p, (max_level, rows, idx_values, dict_len) in enumerate(shapes):
        if max_level == 1:
            def_stream, _ = synth.def_levels_page(synth.seed_for(3, 120 + p), rows)
        else:
            lv = np.repeat(rng.integers(0, max_level + 1, size=rows // 5 + 1), 5)[:rows].astype(np.uint32)
            def_stream = np.frombuffer(encode_u32(lv, get_bit_width(max_level)), dtype=np.uint8)
        idx_stream, idx = synth.dict_page(synth.seed_for(2, 130 + p), rows, 9, 12.0, 50.0)
        if p == 4:  # the largest index sits where only the page's last segment gets to it
            idx = idx.copy()
            idx[idx == idx.max()] = 0
            idx[245000] = 511
            idx_stream = synth.encode_hybrid(idx, 9)
            dict_len = 511
        args = (def_stream, max_level, rows, idx_stream, 9, rows if idx_values is None else idx_values)
        b.add(*args, (0, dict_len, 0), gap=p % 3)
        pages.append((args, dict_len))
    got, validity, status = b.run(ctx, table.view(np.uint8).reshape(-1), 8)
    covered = np.zeros(b.rows, dtype=bool)
    for k, (d, (args, dict_len)) in enumerate(zip(b.defs, pages)):
        out, present, bad = expected(oracle, *args, table[:dict_len])
        o, rows = d[4], d[3]
        assert status[k] == (OUT_OF_SPEC if bad else OK), k
        assert (validity[o:o + rows] == present).all() and (got[o:o + rows] == out).all(), k
        covered[o:o + rows] = True
    assert status[4] == OUT_OF_SPEC
    assert (got[~covered] == 0x5A5A5A5A).all() and (validity[~covered] == 7).all()


def test_many_pages_one_workgroup_each(ctx, oracle):
    """300 pages: one workgroup per page (no segments)."""
    rng = np.random.default_rng(78)
    table = rng.integers(0, 1 << 32, size=(1 << 8, 1), dtype=np.uint64).astype(np.uint32)
    b = Batch()
    pages = []
    for p in range(300):
        rows = int(rng.integers(1, 20000))
        def_stream, _ = synth.def_levels_page(synth.seed_for(3, 200 + p), rows, 50.0, 5.0)
        idx_stream, _ = synth.dict_page(synth.seed_for(2, 500 + p), rows, 8, 6.0, 20.0)
        args = (def_stream, 1, rows, idx_stream, 8, rows)
        b.add(*args, (0, 256, 0), gap=p % 2)
        pages.append(args)
    got, validity, status = b.run(ctx, table.view(np.uint8).reshape(-1), 4)
    assert (status == OK).all()
    for d, args in zip(b.defs, pages):
        out, present, bad = expected(oracle, *args, table)
        o, rows = d[4], d[3]
        assert (validity[o:o + rows] == present).all() and (got[o:o + rows] == out).all()


def test_statuses(ctx, oracle):
    rows = 20000
    table = (np.arange(256, dtype=np.uint32) * 5 + 2).reshape(-1, 1)
    def_stream, levels = synth.def_levels_page(synth.seed_for(3, 99), rows)
    idx_stream, idx = synth.dict_page(synth.seed_for(2, 99), rows, 8, 10.0, 40.0)
    n_valid = int((levels == 1).sum())
    consumed_max = int(idx[:n_valid].max())
    b = Batch()
    b.add(def_stream, 1, rows, idx_stream, 8, rows, (0, 256, 0))                    # fine
    b.add(def_stream, 1, rows, idx_stream, 8, rows, (0, consumed_max, 0))            # a consumed index is too large
    b.add(def_stream, 1, rows, np.array([0x10, 1], dtype=np.uint8), 12, rows, (0, 256, 0))  # index stream panics
    b.add(np.array([0x10], dtype=np.uint8), 1, 64, idx_stream, 8, 64, (0, 256, 0))   # levels: zeros, all null
    # only indices that no valid row consumes are too large: not an error
    few = 10
    short_levels = np.zeros(rows, dtype=np.uint32)
    short_levels[:few] = 1
    big = np.full(rows, 255, dtype=np.uint32)
    big[:few] = 3
    b.add(np.frombuffer(encode_u32(short_levels, 1), dtype=np.uint8), 1, rows,
          np.frombuffer(encode_u32(big, 8), dtype=np.uint8), 8, rows, (0, 100, 0))
    got, validity, status = b.run(ctx, table.view(np.uint8).reshape(-1), 4)
    assert list(status) == [OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC, OK, OK]
    out, present, bad = expected(oracle, def_stream, 1, rows, idx_stream, 8, rows, table)
    assert (got[:rows] == out).all() and (validity[:rows] == present).all()
    o = b.defs[3][4]
    assert (validity[o:o + 64] == 0).all() and (got[o:o + 64] == 0).all()
    o = b.defs[4][4]
    assert (validity[o:o + rows] == short_levels).all() and (got[o:o + few, 0] == table[3, 0]).all()


@pytest.mark.parametrize("value_size,idx_bits", [(4, 8), (8, 9), (4, 12), (12, 12)])
def test_flat_nullable_batches_read_their_levels_as_bitmaps(oracle, value_size, idx_bits):
    """From 256 pages on, a batch of flat nullable pages (every maximum level 1) has its level streams turned into
    validity bitmaps: same values, validity and statuses.  Where the dictionaries have at most 16 KiB (the first three
    cases) the index streams are decoded by ring_kernel straight into the pages' rows (no index ever lies in device
    memory); beyond (12 bits x 12 bytes) the indices are written and optional_bits_kernel places them.  Pages of one
    row to several spans of 2048 valid rows, outputs that are not aligned, gaps of up to 300 rows between the pages
    that must stay untouched, pages with long stretches of nulls, index streams shorter than the valid rows, and the
    failures: an index stream and a level stream the reference panics on, a dictionary that misses a consumed
    index."""
    from parquet2_amd import Context

    rng = np.random.default_rng(1000 + value_size)
    words = value_size // 4
    table = rng.integers(0, 1 << 32, size=(1 << idx_bits, words), dtype=np.uint64).astype(np.uint32)
    b = Batch()
    pages, want_status = [], []
    sizes = [1, 3, 4, 5, 4095, 4096, 4097, 8192, 12288, 12289, 300000] + [int(rng.integers(1, 30000)) for _ in range(300)]
    panics = np.frombuffer(b"\x80" * 9 + b"\x02" + b"\x01", dtype=np.uint8)
    for p, rows in enumerate(sizes):
        null_run = 5.0 if p % 7 else 400.0  # (some pages with long stretches of nulls)
        def_stream, levels = synth.def_levels_page(synth.seed_for(3, 3000 + p), rows, 50.0, null_run)
        idx_stream, idx = synth.dict_page(synth.seed_for(2, 4000 + p), rows, idx_bits, 6.0, 20.0)
        idx_values, dict_len, st = rows, len(table), OK
        n_valid = int((levels == 1).sum())
        if p % 11 == 3:
            idx_values = n_valid // 2  # fewer indices than valid rows: the rows behind them stay null
        elif p == 20:
            idx_stream, st = np.array([0x10, 1], dtype=np.uint8), REFERENCE_WOULD_PANIC
        elif p == 21:
            def_stream, st = panics, REFERENCE_WOULD_PANIC
        elif p == 22 and n_valid > 0:
            dict_len, st = int(idx[:n_valid].max()), OUT_OF_SPEC
        args = (def_stream, 1, rows, idx_stream, 12 if p == 20 else idx_bits, idx_values)
        b.add(*args, (0, dict_len, 0), gap=(p % 3) * 150 + p % 2)
        pages.append((args, dict_len))
        want_status.append(st)
    ctx = Context(0)
    try:
        got, validity, status = b.run(ctx, table.view(np.uint8).reshape(-1), value_size)
        assert "bitmap" in ctx.last_schedule(), ctx.last_schedule()
        assert ("rows" in ctx.last_schedule()) == (len(table) * value_size <= 16384), ctx.last_schedule()
    finally:
        ctx.close()
    assert list(status) == want_status
    covered = np.zeros(b.rows, dtype=bool)
    for k, (d, (args, dict_len)) in enumerate(zip(b.defs, pages)):
        o, rows = d[4], d[3]
        covered[o:o + rows] = True
        if want_status[k] == REFERENCE_WOULD_PANIC:
            continue
        out, present, bad = expected(oracle, *args, table[:dict_len])
        assert bad == (want_status[k] == OUT_OF_SPEC), k
        assert (validity[o:o + rows] == present).all() and (got[o:o + rows] == out).all(), (k, rows)
    assert (got[~covered] == 0x5A5A5A5A).all() and (validity[~covered] == 7).all()
