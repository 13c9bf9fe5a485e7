"""Parity of pq2_decode_dict_batch_device (dictionary indices -> dictionary values on the GPU) against the CPU
oracle's indices gathered with numpy: the consumer loop of a dictionary-encoded page in the reference
(tests/it/read/primitive.rs:99-108; PrimitivePageDict::value, tests/it/read/dictionary/primitive.rs:20-27).
No test of the reference holds a literal vector for this step (its dictionary fixtures are files that are not in the
tree): parity unpinned beyond the definition `out[i] = dict[index[i]]`, index >= len -> OutOfSpec."""
import numpy as np
import pytest

from parquet2_amd import DICT_REF_DTYPE, synth
from parquet2_amd._lib import OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC
from parquet2_amd.error import InvalidParameter

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


def run(ctx, data, descs, out_values, dict_bytes, refs, value_size):
    import torch

    d_in = torch.from_numpy(data).cuda()
    d_dict = torch.from_numpy(dict_bytes).cuda()
    words = value_size // 4
    d_out = torch.full((max(out_values, 1) * words,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    status = ctx.decode_dict_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_dict.data_ptr(), d_dict.numel(), refs,
                                          value_size, d_out.data_ptr(), out_values,
                                          torch.cuda.current_stream().cuda_stream)
    return d_out.cpu().numpy().view(np.uint32).reshape(-1, words)[:max(out_values, 1)], status


@pytest.mark.parametrize("value_size", [4, 8, 12])
def test_gather_matches_oracle(ctx, oracle, value_size):
    rng = np.random.default_rng(100 + value_size)
    words = value_size // 4
    batch = synth.PageBatch()
    pages, dicts, refs = [], [], []
    dict_off = 0
    for p, (w, n) in enumerate([(1, 70001), (3, 4096), (8, 9000), (12, 300000), (12, 123), (17, 50000), (0, 777)]):
        if w == 0:
            stream = np.zeros(0, dtype=np.uint8)
        else:
            stream, _ = synth.dict_page(synth.seed_for(2, 40 + p), n, w, 20.0, 90.0)
        batch.add(stream, w, n + 50, out_align=1 + (p % 3))  # 50 values past the stream: index 0
        dict_len = max(1 << w, 1)
        table = rng.integers(0, 1 << 32, size=(dict_len, words), dtype=np.uint64).astype(np.uint32)
        dict_off += (-dict_off) % 8
        refs.append((dict_off, dict_len, 0))
        dicts.append((dict_off, table))
        dict_off += table.nbytes
        pages.append((stream, w, n + 50, table))
    data, descs, out_values = batch.finish()
    dict_bytes = np.zeros(dict_off + 8, dtype=np.uint8)
    for off, table in dicts:
        dict_bytes[off:off + table.nbytes] = table.view(np.uint8).reshape(-1)
    got, status = run(ctx, data, descs, out_values, dict_bytes, np.array(refs, dtype=DICT_REF_DTYPE), value_size)
    assert (status == OK).all()
    covered = np.zeros(out_values, dtype=bool)
    for d, (stream, w, n, table) in zip(descs, pages):
        st, idx = oracle.hybrid_rle_decode(stream, w, n)
        assert st == OK
        o = int(d["out_off"])
        assert (got[o:o + n] == table[idx]).all(), "page w=%d differs" % w
        covered[o:o + n] = True
    assert (got[~covered] == 0x5A5A5A5A).all()  # the gaps between pages are never written


@pytest.mark.parametrize("value_size", [4, 8, 12])
def test_lookup_inside_the_decode_kernel(ctx, oracle, value_size):
    """Pages of at most 16 bits with dictionaries of at most 48 KiB: where ring_kernel decodes the batch, the lookup
    happens inside it (no buffer of indices).  Every kind of span: long RLE runs, mixed runs, width 0, pages that end
    inside the stream and behind it, values at every alignment the value size allows, a dictionary that misses
    indices, a stream on which the reference panics."""
    rng = np.random.default_rng(200 + value_size)
    words = value_size // 4
    batch = synth.PageBatch(align=1)
    pages, tables, refs = [], [], []
    dict_off = 0
    shapes = [(1, 200001, 400.0, 30.0), (2, 50000, 3.0, 5.0), (3, 4096, 20.0, 90.0), (7, 9000, 20.0, 90.0),
              (10, 65536, 64.0, 256.0), (12, 300000, 64.0, 256.0), (12, 123, 2.0, 9.0), (0, 777, 0, 0),
              (9, 30000, 1000.0, 8.0), (11, 70000, 5.0, 600.0), (5, 1, 1.0, 1.0), (4, 40000, 1.5, 2.0),
              (13, 50000, 30.0, 100.0), (14, 120000, 64.0, 256.0), (16, 33000, 8.0, 40.0)]
    for p, (w, n, rep, dist) in enumerate(shapes * 2):
        stream = synth.dict_page(synth.seed_for(2, 80 + p), n, w, rep, dist)[0] if w else np.zeros(0, dtype=np.uint8)
        want = n + 50 if p % 3 == 0 else n if p % 3 == 1 else max(1, n // 2)
        if p == 16:  # an RLE run whose value byte is missing: the reference panics (decoder.rs:59-60)
            stream = np.array([0x10, 0x01], dtype=np.uint8)
            w, want = 12, 64
        batch.chunks.append(np.zeros(1 + p % 4, np.uint8))
        batch.in_bytes += 1 + p % 4
        batch.add(stream, w, want, out_align=1 + (p % 3))
        dict_len = max(1 << w, 1)
        if value_size * dict_len > 49152:
            dict_len = 49152 // value_size
        if p == 9:
            dict_len = 20  # misses most indices of a 7-bit page
        table = rng.integers(0, 1 << 32, size=(dict_len, words), dtype=np.uint64).astype(np.uint32)
        dict_off += (-dict_off) % 8
        refs.append((dict_off, dict_len, 0))
        tables.append((dict_off, table))
        dict_off += table.nbytes
        pages.append((stream, w, want, table))
    # enough pages for the schedule batches of many pages get
    for p in range(100):
        stream, _ = synth.dict_page(synth.seed_for(2, 300 + p), 20000 + p, 8, 10.0, 40.0)
        batch.add(stream, 8, 20000 + p, out_align=1)
        refs.append(refs[3])
        pages.append((stream, 8, 20000 + p, tables[3][1]))
    data, descs, out_values = batch.finish()
    dict_bytes = np.zeros(dict_off + 8, dtype=np.uint8)
    for off, table in tables:
        dict_bytes[off:off + table.nbytes] = table.view(np.uint8).reshape(-1)
    got, status = run(ctx, data, descs, out_values, dict_bytes, np.array(refs, dtype=DICT_REF_DTYPE), value_size)
    if ctx.last_schedule().startswith("ring_kernel"):
        assert ctx.last_schedule() == "ring_kernel+dictionary"
    elif "expand_kernel" in ctx.last_schedule():  # one round of expand_kernel: the lookup happens inside it as well
        assert ctx.last_schedule() in ("scout+expand_kernel+dictionary", "expand_kernel+dictionary")
    covered = np.zeros(out_values, dtype=bool)
    seen = set()
    for k, (d, (stream, w, n, table)) in enumerate(zip(descs, pages)):
        st, idx = oracle.hybrid_rle_decode(stream, w, n)
        o = int(d["out_off"])
        covered[o:o + n] = True
        if st != OK:
            assert status[k] == st, k
            seen.add(st)
            continue
        ok = idx < len(table)
        assert status[k] == (OK if ok.all() else OUT_OF_SPEC), k
        seen.add(int(status[k]))
        want = np.zeros((n, words), dtype=np.uint32)
        want[ok] = table[idx[ok]]
        bad = np.nonzero((got[o:o + n] != want).any(axis=1))[0]
        assert bad.size == 0, ("page", k, "w", w, "first mismatch at", bad[:5])
    assert seen == {OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC}
    assert (got[~covered] == 0x5A5A5A5A).all()  # the gaps between pages are never written


def test_index_beyond_the_dictionary_fails_the_page_only(ctx, oracle):
    batch = synth.PageBatch()
    n = 20000
    streams = [synth.dict_page(synth.seed_for(2, 60 + p), n, 8, 10.0, 40.0) for p in range(3)]
    for s, _ in streams:
        batch.add(s, 8, n)
    data, descs, out_values = batch.finish()
    table = np.arange(256, dtype=np.uint32) * 3 + 1
    short = int(streams[1][1].max())  # page 1's dictionary misses its largest index
    dict_bytes = np.concatenate([table, table[:short]]).view(np.uint8)
    refs = np.array([(0, 256, 0), (1024, short, 0), (0, 256, 0)], dtype=DICT_REF_DTYPE)
    got, status = run(ctx, data, descs, out_values, dict_bytes, refs, 4)
    assert list(status) == [OK, OUT_OF_SPEC, OK]
    for p in (0, 2):
        o = int(descs[p]["out_off"])
        assert (got[o:o + n, 0] == table[streams[p][1]]).all()
    o = int(descs[1]["out_off"])
    idx = streams[1][1]
    ok = idx < short
    assert (got[o:o + n, 0][ok] == table[idx[ok]]).all() and (got[o:o + n, 0][~ok] == 0).all()


def test_empty_dictionary_and_failed_streams(ctx):
    batch = synth.PageBatch()
    batch.add(np.zeros(0, dtype=np.uint8), 5, 100)       # zeros from an empty stream, empty dictionary
    batch.add(np.array([0x10], dtype=np.uint8), 9, 64)   # header that ends the stream: zeros
    batch.add(np.array([0x10, 0x01], dtype=np.uint8), 12, 8)  # RLE value cut short: the reference panics
    batch.add(np.zeros(0, dtype=np.uint8), 3, 0)         # no values at all
    data, descs, out_values = batch.finish()
    table = np.array([7, 8, 9], dtype=np.uint32)
    refs = np.array([(0, 0, 0), (0, 3, 0), (0, 3, 0), (0, 0, 0)], dtype=DICT_REF_DTYPE)
    got, status = run(ctx, data, descs, out_values, table.view(np.uint8), refs, 4)
    assert list(status) == [OUT_OF_SPEC, OK, REFERENCE_WOULD_PANIC, OK]
    o = int(descs[1]["out_off"])
    assert (got[o:o + 64, 0] == 7).all()


def test_bad_arguments(ctx):
    import torch

    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    descs = np.zeros(1, dtype=synth.PageBatch()._dtype)
    descs[0] = (0, 0, 3, 4, 0)
    for value_size, ref in ((5, (0, 1, 0)), (8, (4, 1, 0)), (4, (60, 2, 0))):
        with pytest.raises(InvalidParameter):
            ctx.decode_dict_batch_device(d.data_ptr(), 64, descs, d.data_ptr(), 64,
                                         np.array([ref], dtype=DICT_REF_DTYPE), value_size, d.data_ptr(), 4)


def test_dense_headers_send_a_dictionary_batch_round_the_lookup_inside(oracle):
    """200 dictionary pages of short runs (a header per six values): the default schedule looks at the headers, does not
    give the batch to ring_kernel with the lookup inside, and sends it through the scout kernels with the lookup
    inside expand_kernel; 200 pages of long runs stay with ring_kernel.  Same values either way."""
    from parquet2_amd import Context

    rng = np.random.default_rng(17)
    n, w = 60000, 8
    table = rng.integers(0, 1 << 32, size=(1 << w, 1), dtype=np.uint64).astype(np.uint32)
    ctx = Context(0)
    try:
        for kind in ("short", "long"):
            batch = synth.PageBatch()
            want = []
            for p in range(200):
                stream, vals = (synth.adversarial(synth.seed_for(4, 300 + p), n, w) if kind == "short"
                                else synth.dict_page(synth.seed_for(2, 300 + p), n, w))
                batch.add(stream, w, n)
                st, idx = oracle.hybrid_rle_decode(stream, w, n)
                assert st == OK and np.array_equal(idx, vals)
                want.append(table[idx, 0])
            data, descs, out_values = batch.finish()
            refs = np.zeros(200, dtype=DICT_REF_DTYPE)
            refs["dict_len"] = len(table)
            got, status = run(ctx, data, descs, out_values, table.view(np.uint8).reshape(-1), refs, 4)
            assert (status == OK).all()
            assert (ctx.last_schedule() == "scout+expand_kernel+dictionary") == (kind == "short"), (kind, ctx.last_schedule())
            for p in range(200):
                o = int(descs[p]["out_off"])
                assert np.array_equal(got[o:o + n, 0], want[p]), (kind, p)
    finally:
        ctx.close()
