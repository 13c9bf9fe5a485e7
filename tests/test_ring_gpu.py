"""Streams that press on the flow control of ring_kernel (one workgroup per page, the stream staged once through
an LDS ring, header positions and block starts handed over through bounded LDS queues): everything is checked
against the CPU oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest

from parquet2_amd import synth
from parquet2_amd._lib import OK

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import os

    from parquet2_amd import Context

    os.environ["PQ2_MODE"] = "ring"
    try:
        c = Context(0)
    finally:
        del os.environ["PQ2_MODE"]
    yield c
    c.close()


def uleb(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def rle(count, value, w):
    return uleb(count << 1) + int(value).to_bytes((w + 7) // 8, "little")


def bitpacked(values, w):
    """One bit-packed run of len(values) (a multiple of 8) values."""
    from parquet2_amd.encoding import bitpacked as bp

    assert len(values) % 8 == 0
    payload = bp.encode(np.asarray(values, dtype=np.uint32), w)[:len(values) * w // 8]
    return uleb(((len(values) // 8) << 1) | 1) + bytes(payload)


def check(ctx, oracle, pages):
    """pages: list of (stream bytes, w, n)."""
    batch = synth.PageBatch(align=1)
    for stream, w, n in pages:
        batch.add(np.frombuffer(stream, dtype=np.uint8), w, n, out_align=1)
    data, descs, out_values = batch.finish()
    got, status = ctx.decode_batch_host(data, descs, out_values)
    for i, (stream, w, n) in enumerate(pages):
        st_o, exp = oracle.hybrid_rle_decode(np.frombuffer(stream, dtype=np.uint8), w, n)
        assert status[i] == st_o, (i, w, n, status[i], st_o)
        if st_o == OK:
            o = int(descs[i]["out_off"])
            bad = np.nonzero(got[o:o + n] != exp)[0]
            assert bad.size == 0, (i, w, n, "first mismatch at", bad[:5], got[o:o + n][bad[:5]], exp[bad[:5]])


def test_runs_without_values_longer_than_the_ring(ctx, oracle):
    """Zero-length runs yield nothing and are not listed: the walk passes tens of KiB of them between two runs
    that do, far more than the ring holds, while a wave waits for the next listed run."""
    rng = np.random.default_rng(1)
    pages = []
    for w in (1, 8, 12, 16, 20, 32):
        parts = [rle(5000, 3, w)]
        for k in range(6):
            parts.append(rle(0, 1, w) * int(rng.integers(9000, 40000)))      # 18 KiB .. 200 KiB of nothing
            parts.append(uleb(1) * int(rng.integers(1, 3000)))               # bit-packed runs of zero groups
            vals = rng.integers(0, 1 << min(w, 31), 8 * int(rng.integers(1, 90)), dtype=np.uint64).astype(np.uint32)
            parts.append(bitpacked(vals, w))
            parts.append(rle(int(rng.integers(1, 9000)), int(rng.integers(0, 1 << min(w, 31))), w))
        stream = b"".join(parts)
        n = 5000 + 6 * 9800
        pages.append((stream, w, n))
        pages.append((stream, w, 7000))  # the page ends long before the stream does
    check(ctx, oracle, pages)


def test_single_runs_of_a_million_values(ctx, oracle):
    """A page that is one bit-packed run (as the reference's encoder writes them) leaves the walk nothing to do
    but fill the ring; a page that is one RLE run takes thousands of block starts through a queue of 64."""
    from parquet2_amd.encoding.hybrid_rle import encode_u32

    n = (1 << 20) + 77
    pages = []
    for w in (1, 7, 12, 16, 17, 32):
        vals = synth.uniform_values(w, n, w)
        pages.append((bytes(encode_u32(vals, w)), w, n))
        pages.append((rle(n - 5, (1 << w) - 1 if w < 32 else 0xDEADBEEF, w) + rle(100, 1, w), w, n + 50))
    # a long RLE run in front of mixed runs, and behind them
    vals = synth.dict_values(9, 300000, 12, 30.0, 90.0)
    mixed = bytes(synth.encode_hybrid(vals, 12))
    pages.append((rle(2_000_000, 77, 12) + mixed + rle(1_500_000, 5, 12), 12, 2_000_000 + 300000 + 1_400_000))
    check(ctx, oracle, pages)


@pytest.mark.parametrize("w", [2, 3, 9, 16, 17, 31])
def test_more_runs_per_tile_than_the_queue_holds(ctx, oracle, w):
    """Runs of one value: 4096 listed runs per tile against a queue of 512 header positions."""
    rng = np.random.default_rng(w)
    n = 70000
    parts = []
    for k in range(n // 9 + 1):
        parts.append(rle(1, int(rng.integers(0, 1 << min(w, 31))), w))
        parts.append(bitpacked(rng.integers(0, 1 << min(w, 31), 8, dtype=np.uint64).astype(np.uint32), w))
    stream = b"".join(parts)
    pages = [(stream, w, n), (stream, w, n - 12345), (stream[:len(stream) // 2], w, n)]
    # and pages of the adversarial generator among 70 others, so that the workgroup-per-page schedule is crowded
    for p in range(70):
        s, _ = synth.adversarial(synth.seed_for(4, p), 30000 + 997 * p, w)
        pages.append((bytes(s), w, 30000 + 997 * p))
    check(ctx, oracle, pages)


def test_long_bit_packed_runs_beside_short_ones(ctx, oracle):
    """Bit-packed runs far longer than the 63 groups writers usually stop at (their payload alone exceeds several
    ring slots) between runs of a few values, for widths on both sides of the ring sizes."""
    rng = np.random.default_rng(3)
    pages = []
    for w in (4, 12, 16, 24, 32):
        parts = []
        total = 0
        for k in range(40):
            m = 8 * int(rng.integers(1, 2500))
            parts.append(bitpacked(rng.integers(0, 1 << min(w, 31), m, dtype=np.uint64).astype(np.uint32), w))
            c = int(rng.integers(1, 40))
            parts.append(rle(c, int(rng.integers(0, 1 << min(w, 31))), w))
            total += m + c
        stream = b"".join(parts)
        pages.append((stream, w, total))
        pages.append((stream, w, total + 5000))        # zero padding behind the stream
        pages.append((stream[:-3], w, total))          # the last run loses its value bytes: the reference panics
    check(ctx, oracle, pages)


def test_many_ragged_pages(ctx, oracle):
    """400 pages of every width and of sizes around the tile and span boundaries in one batch."""
    rng = np.random.default_rng(4)
    pages = []
    sizes = [1, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 12289, 40000]
    for p in range(400):
        w = int(rng.integers(0, 33))
        n = sizes[p % len(sizes)] + int(rng.integers(0, 3))
        if w == 0:
            pages.append((bytes([3]), 0, n))
            continue
        vals = synth.dict_values(1000 + p, n, w, float(rng.integers(2, 80)), float(rng.integers(8, 400)))
        stream = bytes(synth.encode_hybrid(vals, w))
        pages.append((stream, w, n + (int(rng.integers(0, 900)) if p % 5 == 0 else 0)))
    check(ctx, oracle, pages)


@pytest.mark.parametrize("schedule", ["ring", "auto"])
def test_mixed_batch_of_index_and_level_pages(oracle, schedule):
    """BASELINE configs[4]'s page mix on one GPU: 160 pages of 2^20 values, three in four dictionary-index pages of
    width 12, one in four definition-level pages of width 1, each replica at its own address.  Every page against
    the generator's values; one page of each distinct stream against the oracle.  (ring_kernel takes level pages
    whole tiles at a time only in batches of nothing but level pages: this batch is the other case.)"""
    import torch

    from parquet2_amd import PAGE_DESC_DTYPE, Context

    n = 1 << 20
    n_pages, distinct = 160, 4
    idx = [synth.dict_page(synth.seed_for(2, p), n, 12) for p in range(distinct)]
    lev = [synth.def_levels_page(synth.seed_for(3, p), n) for p in range(distinct)]
    for (s, v), w in [(p, 12) for p in idx] + [(p, 1) for p in lev]:
        st, want = oracle.hybrid_rle_decode(bytes(s), w, n)
        assert st == OK and np.array_equal(want, v)  # the generator's values are the oracle's
    stride = (max(len(s) for s, _ in idx + lev) + 16 + 255) // 256 * 256
    h = np.zeros((2 * distinct, stride), np.uint8)
    for i, (s, _) in enumerate(idx + lev):
        h[i, :len(s)] = s
    which = np.array([(p // 4) % distinct + (distinct if p % 4 == 3 else 0) for p in range(n_pages)])
    d_in = torch.from_numpy(h).cuda()[torch.from_numpy(which).cuda()].contiguous()
    d_out = torch.full((n_pages * n,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    descs = np.zeros(n_pages, dtype=PAGE_DESC_DTYPE)
    descs["in_off"] = np.arange(n_pages, dtype=np.uint64) * stride
    descs["in_len"] = [len((idx + lev)[k][0]) for k in which]
    descs["num_bits"] = [12 if k < distinct else 1 for k in which]
    descs["num_values"] = n
    descs["out_off"] = np.arange(n_pages, dtype=np.uint64) * n
    ctx = Context(0)
    try:
        ctx.set_schedule(schedule)
        st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), n_pages * n, 0)
        torch.cuda.synchronize()
        assert st.tolist() == [OK] * n_pages
        if schedule == "ring":
            assert ctx.last_schedule() == "ring_kernel"
        expect = [torch.from_numpy(v.astype(np.uint32).view(np.int32)).cuda() for _, v in idx + lev]
        got = d_out.view(n_pages, n)
        for p in range(n_pages):
            assert torch.equal(got[p], expect[which[p]]), "page %d (%s)" % (p, "levels" if which[p] >= distinct else "indices")
    finally:
        ctx.close()
