"""pq2_filtered_validity_batch_device against the host mirror of FilteredHybridBitmapIter
(src/deserialize/filtered_rle.rs:53-274; the mirror is pinned by the reference's twelve vectors in
tests/test_host_logic.py) driven over the same level streams, and against the CPU oracle's levels."""
import numpy as np
import pytest

from parquet2_amd import INTERVAL_DTYPE, synth
from parquet2_amd._lib import OK
from parquet2_amd.deserialize import FilteredHybridEncoded, FilteredHybridRleDecoderIter
from parquet2_amd.encoding.hybrid_rle import BitmapIter
from parquet2_amd.error import InvalidParameter
from parquet2_amd.indexes import Interval

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ring", "scout"])
def ctx(request):
    import os

    from parquet2_amd import Context

    os.environ["PQ2_MODE"] = request.param
    try:
        c = Context(0)
    finally:
        del os.environ["PQ2_MODE"]
    yield c
    c.close()


def from_the_iterator(stream, rows, selection):
    """What a reader collects from FilteredHybridBitmapIter: the bits of the selected rows, and per interval the set
    bits skipped in front of it and yielded inside it."""
    it = FilteredHybridRleDecoderIter(bytes(stream), rows, [Interval(s, l) for s, l in selection])
    bits = []
    counts = np.zeros((len(selection), 2), dtype=np.uint32)
    k, left = 0, selection[0][1] if selection else 0
    for item in it:
        if isinstance(item, FilteredHybridEncoded.Skipped):
            counts[k, 0] += item.set_count
            continue
        if isinstance(item, FilteredHybridEncoded.Bitmap):
            b = BitmapIter(item.values, item.offset, item.length).collect().astype(np.uint8)
        else:
            b = np.full(item.length, 1 if item.is_set else 0, dtype=np.uint8)
        bits.append(b)
        counts[k, 1] += int(b.sum())
        left -= len(b)
        assert left >= 0
        if left == 0 and k + 1 < len(selection):
            k += 1
            left = selection[k][1]
    return (np.concatenate(bits) if bits else np.zeros(0, np.uint8)), counts


def run(ctx, pages, selections):
    import torch

    batch = synth.PageBatch(align=1)
    for s, n in pages:
        batch.add(s, 1, n)
    data, descs, _ = batch.finish()
    offs, flat, out_off = [0], [], 0
    for i, sel in enumerate(selections):
        out_off += i % 3  # gaps between the pages' bytes, every alignment
        descs[i]["out_off"] = out_off
        flat += sel
        offs.append(len(flat))
        out_off += sum(l for _, l in sel)
    d_in = torch.from_numpy(data).cuda()
    d_val = torch.full((max(out_off, 1),), 9, dtype=torch.uint8, device="cuda")
    d_cnt = torch.full((max(2 * len(flat), 1),), -1, dtype=torch.int32, device="cuda")
    status = ctx.filtered_validity_batch_device(d_in.data_ptr(), d_in.numel(), descs,
                                                np.array(flat, dtype=INTERVAL_DTYPE).reshape(-1), offs, d_val.data_ptr(),
                                                out_off, d_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return d_val.cpu().numpy(), d_cnt.cpu().numpy().view(np.uint32).reshape(-1, 2)[:len(flat)], status, descs, offs


def test_matches_the_iterator_and_the_levels(ctx, oracle):
    rng = np.random.default_rng(21)
    pages, selections = [], []
    for p in range(70):
        rows = int(rng.integers(1, 300_000)) if p % 7 else int(rng.integers(1, 40))
        stream, _ = synth.def_levels_page(synth.seed_for(3, 400 + p), rows, 60.0 if p % 2 else 400.0, 7.0 if p % 2 else 20.0)
        if p % 11 == 3:
            stream = stream[:len(stream) // 2]  # the stream holds fewer rows than the page: zeros behind it
        sel, at = [], 0
        for _ in range(int(rng.integers(0, 9))):
            if at >= rows:
                break
            start = at + int(rng.integers(0, max(1, (rows - at) // 3)))
            if start >= rows:
                break
            length = int(rng.integers(1, max(2, min(rows - start, 50_000))))
            length = min(length, rows - start)
            sel.append((start, length))
            at = start + length
        pages.append((stream, rows))
        selections.append(sel)
    selections[0] = [(0, pages[0][1])]  # a page selected whole
    validity, counts, status, descs, offs = run(ctx, pages, selections)
    assert (status == OK).all()
    covered = np.zeros(validity.size, dtype=bool)
    for i, ((stream, rows), sel) in enumerate(zip(pages, selections)):
        st, levels = oracle.hybrid_rle_decode(stream, 1, rows)
        assert st == OK
        want_bits = np.concatenate([levels[s:s + l] for s, l in sel]).astype(np.uint8) if sel else np.zeros(0, np.uint8)
        o = int(descs[i]["out_off"])
        assert (validity[o:o + want_bits.size] == want_bits).all(), i
        covered[o:o + want_bits.size] = True
        prev = 0
        for k, (s, l) in enumerate(sel):
            assert counts[offs[i] + k, 0] == int(levels[prev:s].sum()), (i, k)
            assert counts[offs[i] + k, 1] == int(levels[s:s + l].sum()), (i, k)
            prev = s + l
        if i < 25:  # the iterator itself (pure Python: a sample of the pages)
            bits, cnt = from_the_iterator(stream, rows, sel)
            if i % 11 == 3:
                # a level stream that ends early: the reference's iterator ends with the stream's last run, the
                # library goes on with null rows (as HybridRleDecoder pads with zeros, mod.rs:95)
                assert bits.size <= want_bits.size and (bits == want_bits[:bits.size]).all(), i
            else:
                assert (bits == want_bits).all() and (cnt == counts[offs[i]:offs[i + 1]]).all(), i
    assert (validity[~covered] == 9).all()  # nothing else is written


def test_bad_arguments(ctx):
    import torch

    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    c = torch.zeros(16, dtype=torch.int32, device="cuda")
    descs = np.zeros(1, dtype=synth.PageBatch()._dtype)
    descs[0] = (0, 2, 1, 16, 0)
    for bits, sel in ((2, [(0, 4)]), (1, [(4, 2), (3, 2)]), (1, [(10, 7)]), (1, [(0, 0)])):
        descs[0]["num_bits"] = bits
        with pytest.raises(InvalidParameter):
            ctx.filtered_validity_batch_device(d.data_ptr(), 64, descs, np.array(sel, dtype=INTERVAL_DTYPE).reshape(-1),
                                               [0, len(sel)], d.data_ptr(), 64, c.data_ptr())
