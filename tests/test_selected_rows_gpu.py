"""Parity of pq2_decode_selected_batch_device (SliceFilteredIter over a page's decoded values, on the GPU) against the
reference's own vector (src/deserialize/utils.rs:156-176 `basic`) and the host mirror of SliceFilteredIter driven by
the CPU oracle's values."""
import numpy as np
import pytest

from parquet2_amd import INTERVAL_DTYPE, synth
from parquet2_amd._lib import OK
from parquet2_amd.deserialize.utils import SliceFilteredIter
from parquet2_amd.encoding.hybrid_rle import encode_u32
from parquet2_amd.error import InvalidParameter
from parquet2_amd.indexes import Interval

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


def run(ctx, pages, selections, gaps=None):
    """pages: [(stream, bits, n)], selections: per page list of (start, length).  Returns the per-page results."""
    import torch

    batch = synth.PageBatch()
    for s, w, n in pages:
        batch.add(s, w, n)
    data, descs, _ = batch.finish()
    offs, flat, out_off = [0], [], 0
    for i, sel in enumerate(selections):
        out_off += gaps[i] if gaps else 0
        descs[i]["out_off"] = out_off
        flat += sel
        offs.append(len(flat))
        out_off += sum(l for _, l in sel)
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.full((max(out_off, 1),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    status = ctx.decode_selected_batch_device(d_in.data_ptr(), d_in.numel(), descs,
                                              np.array(flat, dtype=INTERVAL_DTYPE).reshape(-1), offs, d_out.data_ptr(),
                                              out_off, torch.cuda.current_stream().cuda_stream)
    got = d_out.cpu().numpy().view(np.uint32)
    return [got[int(d["out_off"]):int(d["out_off"]) + sum(l for _, l in sel)] for d, sel in zip(descs, selections)], \
        status, got


def test_reference_basic_vector(ctx, kats):
    c = kats["slice_filtered_iter"]
    lo, hi = c["iter_range_inclusive"]
    values = np.arange(lo, hi + 1, dtype=np.uint32)
    stream = np.frombuffer(encode_u32(values, 7), dtype=np.uint8)
    (got,), status, _ = run(ctx, [(stream, 7, len(values))], [[tuple(i) for i in c["intervals"]]])
    assert status[0] == OK
    expected = list(range(0, 2)) + list(range(20, 31)) + [31]  # utils.rs:170-175
    assert got.tolist() == expected
    mirror = list(SliceFilteredIter.new(iter(values.tolist()), [Interval(*i) for i in c["intervals"]]))
    assert got.tolist() == mirror


def test_against_the_host_iterator(ctx, oracle):
    rng = np.random.default_rng(11)
    pages, selections, gaps = [], [], []
    for p, (w, n) in enumerate([(12, 300000), (1, 65536), (3, 5000), (17, 40001), (8, 100)]):
        stream, _ = (synth.def_levels_page(synth.seed_for(3, 70 + p), n) if w == 1
                     else synth.dict_page(synth.seed_for(2, 70 + p), n, w, 15.0, 70.0))
        pages.append((stream, w, n))
        cuts = np.sort(rng.choice(np.arange(1, n), size=min(2 * (3 + 5 * p), n - 1) // 2 * 2, replace=False))
        sel = [(int(a), int(b - a)) for a, b in zip(cuts[0::2], cuts[1::2])]
        if p == 1:
            sel = [(0, n)]  # everything
        if p == 4:
            sel = []        # nothing
        selections.append(sel)
        gaps.append(p % 4)
    results, status, got = run(ctx, pages, selections, gaps)
    assert (status == OK).all()
    total = 0
    for (stream, w, n), sel, res in zip(pages, selections, results):
        st, values = oracle.hybrid_rle_decode(stream, w, n)
        assert st == OK
        mirror = list(SliceFilteredIter.new(iter(values.tolist()), [Interval(a, b) for a, b in sel]))
        assert res.tolist() == mirror
        total += len(mirror)
    assert int((got != 0x5A5A5A5A).sum()) <= total  # nothing is written between the pages' selections


def test_few_rows_of_large_pages(ctx, oracle):
    """Most tiles of 4096 values hold no selected row and are not expanded; the selected rows still come out right
    (intervals at tile boundaries, in the first and the last tile, across several tiles)."""
    rng = np.random.default_rng(17)
    pages, selections = [], []
    for p in range(70):
        w = (3, 12, 20)[p % 3]
        n = int(rng.integers(30000, 90000))
        stream, _ = synth.dict_page(synth.seed_for(2, 300 + p), n, w, 15.0, 70.0)
        pages.append((stream, w, n))
        kind = p % 5
        if kind == 0:
            sel = [(0, 1), (n - 1, 1)]
        elif kind == 1:
            sel = [(4095, 2), (3 * 4096, 4096), (5 * 4096 + 7, 1)]
        elif kind == 2:
            sel = [(int(rng.integers(0, n - 10000)), 9000)]
        elif kind == 3:
            sel = []
        else:
            sel = [(4096, 1), (n - 5000, 4000)]
        selections.append(sel)
    results, status, _ = run(ctx, pages, selections)
    assert (status == OK).all()
    for (stream, w, n), sel, res in zip(pages, selections, results):
        st, values = oracle.hybrid_rle_decode(stream, w, n)
        assert st == OK
        expected = np.concatenate([values[a:a + b] for a, b in sel]) if sel else values[:0]
        assert (res == expected).all()


def test_interval_lists_the_reference_cannot_take(ctx):
    stream = np.frombuffer(encode_u32(np.arange(64, dtype=np.uint32), 6), dtype=np.uint8)
    for sel in ([(5, 0)], [(10, 5), (12, 3)], [(10, 5), (3, 2)], [(60, 5)], [(65, 1)]):
        with pytest.raises(InvalidParameter):
            run(ctx, [(stream, 6, 64)], [sel])
