"""pq2_multi_decode: one batch split by cost over several contexts, a host thread each.  The box of these tests
has one GPU, so the contexts share device 0 (and every visible device is added when there are more): the result
must equal what one context decodes, and the oracle's."""
import numpy as np
import pytest

from parquet2_amd import Context, multi_decode, partition_pages, synth
from parquet2_amd._lib import OK, REFERENCE_WOULD_PANIC
from parquet2_amd.error import InvalidParameter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def contexts():
    import torch

    devices = [0, 0, 0] + list(range(1, torch.cuda.device_count()))
    ctxs = [Context(d) for d in devices]
    yield ctxs
    for c in ctxs:
        c.close()


def mixed_batch(n_pages, seed, big=False):
    rng = np.random.default_rng(seed)
    batch = synth.PageBatch()
    streams = []
    for p in range(n_pages):
        w = int(rng.integers(1, 33))
        n = int(rng.integers(1 << 16, 1 << 19)) if big and p % 3 == 0 else int(rng.integers(0, 30000))
        vals = synth.dict_values(seed * 1000 + p, n, w, float(rng.integers(2, 60)), 200.0)
        stream = synth.encode_hybrid(vals, w)
        if p % 17 == 5 and w > 8:
            stream = np.concatenate([stream, np.array([20, 1], dtype=np.uint8)])   # an RLE run short of a value byte
            n += 10
        batch.add(stream, w, n)
        streams.append((stream, w, n))
    return batch.finish(), streams


@pytest.mark.parametrize("big", [False, True])
def test_equals_one_context_and_the_oracle(contexts, oracle, big):
    (data, descs, out_values), streams = mixed_batch(96 if big else 300, 5 + big, big)
    one, st_one = contexts[0].decode_batch_host(data, descs, out_values)
    for k in (1, 2, len(contexts)):
        got, st = multi_decode(contexts[:k], data, descs, out_values)
        assert np.array_equal(st, st_one)
        for d, s, (stream, w, n) in zip(descs, st, streams):
            o = int(d["out_off"])
            if s == OK:
                assert np.array_equal(got[o:o + n], one[o:o + n])
    assert (st_one == REFERENCE_WOULD_PANIC).any()
    for i in range(0, len(streams), 7):
        stream, w, n = streams[i]
        st_o, exp = oracle.hybrid_rle_decode(stream, w, n)
        assert st_o == st_one[i]
        if st_o == OK:
            o = int(descs[i]["out_off"])
            assert np.array_equal(got[o:o + n], exp)


def test_more_contexts_than_pages_and_empty_batches(contexts):
    (data, descs, out_values), streams = mixed_batch(2, 9)
    got, st = multi_decode(contexts, data, descs, out_values)
    one, st_one = contexts[0].decode_batch_host(data, descs, out_values)
    assert np.array_equal(st, st_one)
    for d in descs:                       # the values between two pages are unspecified
        o, n = int(d["out_off"]), int(d["num_values"])
        assert np.array_equal(got[o:o + n], one[o:o + n])
    got, st = multi_decode(contexts, np.zeros(0, dtype=np.uint8), descs[:0], 0)
    assert got.size == 0 and st.size == 0


def test_interleaved_outputs_are_refused(contexts):
    (data, descs, out_values), _ = mixed_batch(40, 11)
    descs = descs.copy()
    b = partition_pages(descs, 2)
    first, other = int(b[1]) - 1, int(b[1])
    # swap the output places of the last page of part 0 and the first page of part 1, sizes made equal
    n = int(min(descs["num_values"][first], descs["num_values"][other]))
    descs["num_values"][[first, other]] = n
    descs["out_off"][[first, other]] = descs["out_off"][[other, first]]
    if n and (partition_pages(descs, 2) == b).all():
        with pytest.raises(InvalidParameter):
            multi_decode(contexts[:2], data, descs, out_values)


def test_the_same_context_twice_is_refused(contexts):
    (data, descs, out_values), _ = mixed_batch(12, 13)
    with pytest.raises(InvalidParameter):
        multi_decode([contexts[0], contexts[1], contexts[0]], data, descs, out_values)


def test_every_context_decodes_its_part_of_the_partition(contexts):
    """The page ranges the contexts get are those of pq2_page_costs + pq2_partition_by_cost over the host's streams."""
    from parquet2_amd._lib import lib

    (data, descs, out_values), _ = mixed_batch(150, 17, True)
    k = min(3, len(contexts))
    bounds = partition_pages(descs, k, data)
    before = [lib().pq2_ctx_pages_total(c._h) for c in contexts[:k]]
    multi_decode(contexts[:k], data, descs, out_values)
    after = [lib().pq2_ctx_pages_total(c._h) for c in contexts[:k]]
    assert [a - b for a, b in zip(after, before)] == [int(bounds[r + 1] - bounds[r]) for r in range(k)]
    assert sum(a - b for a, b in zip(after, before)) == len(descs)
