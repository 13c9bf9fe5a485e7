"""Parity of the HIP decode path (through the C ABI) against the CPU oracle and the reference's
known-answer vectors.  Everything here is bit-exact: the path is integer/byte work."""
import numpy as np
import pytest

from parquet2_amd import PAGE_DESC_DTYPE, synth
from parquet2_amd._lib import INVALID_PARAMETER, OK, REFERENCE_WOULD_PANIC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["one_round", "four_rounds", "per_page", "ring", "scout"])
def ctx(request):
    """Every test runs five times, once per way a batch can be scheduled: parse_kernel + expand_kernel in one
    round, the same with the round pipeline forced on (parse of part r + 1 on one stream while part r is expanded
    on another), decode_kernel, one workgroup per page with the hand-off through device memory, ring_kernel, one
    workgroup per page with the stream staged once through an LDS ring (batches of many pages), and the scout
    kernels, which walk the headers of all pages in parallel over tiles of the streams, in front of expand_kernel
    (batches of few pages).  The context reads the knobs when it is created."""
    import os

    from parquet2_amd import Context

    knobs = {"one_round": {"PQ2_MODE": "tile"},
             "four_rounds": {"PQ2_MODE": "tile", "PQ2_MIN_ROUNDED_TILES": "1", "PQ2_ROUND_SPLIT": "0.1,0.3,0.6"},
             "per_page": {"PQ2_MODE": "page"}, "ring": {"PQ2_MODE": "ring"},
             "scout": {"PQ2_MODE": "scout"}}[request.param]
    os.environ.update(knobs)
    try:
        c = Context(0)
    finally:
        for k in knobs:
            del os.environ[k]
    yield c
    c.close()


def check_page(ctx, oracle, data, w, n, msg=""):
    st_o, exp = oracle.hybrid_rle_decode(data, w, n)
    got, st = ctx.decode_page_host(data, w, n)
    assert st == st_o, (msg, st, st_o)
    if st_o == OK:
        assert got.shape == exp.shape
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (msg, "first mismatch at", bad[:5], got[bad[:5]], exp[bad[:5]])


# ---- the reference's own vectors ---------------------------------------------------------------

def test_reference_hybrid_rle_kats(ctx, kats):
    for c in kats["hybrid_rle_decode"]:
        got, st = ctx.decode_page_host(bytes(c["data"]), c["num_bits"], c["num_values"])
        expected = list(range(c["expected_range"])) if "expected_range" in c else c["expected"]
        assert st == OK and got.tolist() == expected, c["src"]


def test_reference_roundtrip_kat(ctx, kats):
    from parquet2_amd.encoding.hybrid_rle import HybridRleDecoder, encode_u32

    c = kats["hybrid_rle_round_trip"]
    data = list(range(c["range"]))
    buf = encode_u32(data, c["num_bits"])
    assert list(HybridRleDecoder.try_new(buf, c["num_bits"], len(data), ctx)) == data


def test_reference_bitpacked_kats(ctx, kats):
    from parquet2_amd.encoding import bitpacked
    from parquet2_amd.error import OutOfSpec

    for c in kats["bitpacked_decode"]:
        if c["type"] != "u32":
            continue
        got = bitpacked.Decoder.try_new(bytes(c["packed"]), c["num_bits"], c["length"], ctx).collect()
        assert got.tolist() == c["expected"], c["src"]
    c = kats["bitpacked_case1"]
    assert ctx.bitunpack_u32_host(bytes(c["packed"]), c["num_bits"], len(c["unpacked"])).tolist() == c["unpacked"]
    for c in kats["bitpacked_errors"]["cases"]:
        if c["ok"]:
            ctx.bitunpack_u32_host(bytes(c["packed"]), c["num_bits"], c["length"])
        else:
            with pytest.raises(OutOfSpec):
                ctx.bitunpack_u32_host(bytes(c["packed"]), c["num_bits"], c["length"])


def test_reference_run_decoder_values(ctx, kats):
    for c in kats["run_decoder"]:
        if c["kind"] != "bitpacked":
            continue
        got = ctx.bitunpack_u32_host(bytes(c["slice"]), c["bit_width"], c["length"])
        assert got.tolist() == c["values"], c["src"]


def test_unpack_all_ones_every_width(ctx, kats):
    data = bytes([0xFF]) * 4096
    for w in range(1, 33):
        got = ctx.bitunpack_u32_host(data, w, 4096 * 8 // w)
        assert (got == (0xFFFFFFFF >> (32 - w))).all(), w


def test_bitmap_iter_and_boolean_iter(ctx, oracle):
    from parquet2_amd.deserialize import HybridRleDecoderIter
    from parquet2_amd.encoding.hybrid_rle import BitmapIter

    bm = bytes([0b10011101, 0b10011101])
    bits = [(bm[i // 8] >> (i % 8)) & 1 for i in range(16)]
    assert [int(b) for b in BitmapIter.new(bm, 0, 14, ctx)] == bits[:14]
    assert [int(b) for b in BitmapIter.new(bm, 3, 10, ctx)] == bits[3:13]
    # past the slice the last byte repeats (src/encoding/hybrid_rle/bitmap.rs:55-60)
    assert [int(b) for b in BitmapIter.new(bm, 8, 16, ctx)] == bits[8:16] * 2
    assert [int(b) for b in BitmapIter.new(b"", 0, 5, ctx)] == [0] * 5
    stream = bytes([0b00000101, 0b11101011, 0b00000010])
    st, exp = oracle.boolean_iter(stream, 10)
    assert [int(b) for b in HybridRleDecoderIter(stream, 10)] == exp.tolist()
    two_runs = bytes([0b00010000, 1, 0b00010000, 0])
    assert [int(b) for b in HybridRleDecoderIter(two_runs, 16)] == [1] * 8  # first run only, as the reference


# ---- generated streams, every width ----------------------------------------------------------------

@pytest.mark.parametrize("w", list(range(0, 33)))
def test_every_width_mixed_runs(ctx, oracle, w):
    rng = np.random.default_rng(100 + w)
    for n in (1, 7, 8, 9, 31, 33, 255, 1023, 1024, 1025, 4097, 20000):
        vals = synth.dict_values(n * 37 + w, n, w, 20.0, 40.0) if w else np.zeros(n, np.uint32)
        stream = synth.encode_hybrid(vals, w) if w else bytes([3])
        check_page(ctx, oracle, stream, w, n, "mixed w=%d n=%d" % (w, n))
        if w:
            st, exp = oracle.hybrid_rle_decode(stream, w, n)
            assert (exp == vals).all()
        # more values requested than the stream holds: zero padding
        check_page(ctx, oracle, stream, w, n + int(rng.integers(1, 3000)), "padded w=%d n=%d" % (w, n))
        # fewer: truncation inside a run
        check_page(ctx, oracle, stream, w, max(1, n // 2), "truncated w=%d n=%d" % (w, n))


@pytest.mark.parametrize("w", [1, 2, 3, 7, 8, 10, 12, 16, 17, 24, 31, 32])
def test_reference_encoder_streams(ctx, oracle, w):
    """Streams as the reference writes them: one bit-packed run whose payload may be shorter than the
    header announces (src/encoding/hybrid_rle/encoder.rs:59-70)."""
    from parquet2_amd.encoding.hybrid_rle import encode_u32

    for n in (1, 9, 15, 33, 100, 1000, 5001):
        vals = synth.uniform_values(n + w, n, w)
        buf = encode_u32(vals, w)
        assert buf == oracle.encode_u32(vals, w)
        check_page(ctx, oracle, buf, w, n, "reference encoder w=%d n=%d" % (w, n))
        got, st = ctx.decode_page_host(buf, w, n)
        assert (got == vals).all()
        check_page(ctx, oracle, buf[:len(buf) // 2], w, n, "cut payload")


def test_garbage_streams(ctx, oracle):
    """Random bytes: whatever the stream holds, statuses and values match the oracle and nothing faults."""
    rng = np.random.default_rng(42)
    batch = synth.PageBatch(align=1)
    expect = []
    for i in range(600):
        w = int(rng.integers(0, 33))
        length = int(rng.integers(0, 400))
        data = rng.integers(0, 256, length, dtype=np.uint8)
        if rng.random() < 0.5 and length:  # bias towards short headers so that runs chain
            data[rng.integers(0, length, length // 3)] &= 0x1F
        n = int(rng.integers(0, 5000))
        batch.add(data, w, n, out_align=1)
        expect.append(oracle.hybrid_rle_decode(data, w, n))
    data, descs, out_values = batch.finish()
    got, status = ctx.decode_batch_host(data, descs, out_values)
    n_panics = 0
    for i, (st_o, exp) in enumerate(expect):
        assert status[i] == st_o, (i, descs[i])
        n_panics += st_o != OK
        if st_o == OK:
            o, n = int(descs[i]["out_off"]), int(descs[i]["num_values"])
            assert (got[o:o + n] == exp).all(), (i, descs[i])
    assert n_panics > 0  # the reference's panic paths were exercised


def test_panic_paths(ctx, oracle):
    cases = [
        (bytes([0b00010000]), 8, 4),                      # RLE header is the last byte: dropped, zeros
        (bytes([0b00010000, 1]), 16, 4),                  # RLE value needs 2 bytes, 1 left: split_at panics
        (bytes([0x80] * 9 + [0x02, 1, 1]), 8, 4),         # 10th varint byte > 1
        (bytes([0x81] + [0xFF] * 8 + [0x01, 1, 1]), 8, 4),  # bit-packed byte count overflows usize
        (bytes([0x81] + [0xFF] * 7 + [0x00, 1, 1]), 1, 9),  # huge but legal: clamped to the remaining bytes
        (bytes([0x80] * 9 + [0x01, 7, 0]), 8, 4),         # 10-byte header of an RLE run of 2^62 values
        (bytes([0x80, 0x80]), 5, 3),                      # varint cut by the end of the stream
    ]
    seen = set()
    for data, w, n in cases:
        st_o, _ = oracle.hybrid_rle_decode(data, w, n)
        seen.add(st_o)
        check_page(ctx, oracle, data, w, n, repr(data))
    assert seen == {OK, REFERENCE_WOULD_PANIC}
    got, st = ctx.decode_page_host(bytes([3, 1]), 33, 4)
    assert st == INVALID_PARAMETER


def test_unmasked_rle_values_and_header_drop(ctx, oracle):
    # RLE value bytes with bits above num_bits set come out unmasked (src/encoding/hybrid_rle/mod.rs:51-55)
    data = bytes([0b00010000, 0xFF, 0b00000110, 0xAB])
    got, st = ctx.decode_page_host(data, 3, 11)
    assert st == OK and got.tolist() == [0xFF] * 8 + [0xAB] * 3
    data = bytes([20, 0xFF, 0xFF, 0xFF, 0xFF])
    got, st = ctx.decode_page_host(data, 27, 10)
    assert st == OK and got.tolist() == [0xFFFFFFFF] * 10
    # a trailing header with nothing after it is dropped, the rest is zeros (decoder.rs:44-46)
    data = bytes([0b00000110, 5, 0b00010000])
    got, st = ctx.decode_page_host(data, 8, 12)
    assert st == OK and got.tolist() == [5] * 3 + [0] * 9
    check_page(ctx, oracle, data, 8, 12)
    # zero-length runs are legal
    data = bytes([0, 9, 1, 0b00000100, 7])
    check_page(ctx, oracle, data, 8, 5)


def test_alignment_of_streams_and_outputs(ctx, oracle):
    """Streams at every byte alignment, outputs at every dword alignment, in one batch."""
    batch = synth.PageBatch(align=1)
    expect = []
    for k in range(64):
        w = 1 + (k * 5) % 32
        n = 900 + 131 * k
        vals = synth.dict_values(k, n, w, 12.0, 30.0)
        stream = synth.encode_hybrid(vals, w)
        batch.chunks.append(np.zeros(1 + k % 3, np.uint8))  # shift the next stream's address
        batch.in_bytes += 1 + k % 3
        batch.out_values += k % 4
        batch.add(stream, w, n, out_align=1)
        expect.append(vals)
    data, descs, out_values = batch.finish()
    assert len(set(int(d["in_off"]) % 16 for d in descs)) > 8
    assert len(set(int(d["out_off"]) % 4 for d in descs)) == 4
    import torch

    d_in = torch.from_numpy(data).cuda()
    d_out = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    status = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values,
                                     torch.cuda.current_stream().cuda_stream)
    got = d_out.cpu().numpy().view(np.uint32)
    assert (status == OK).all()
    covered = np.zeros(out_values, bool)
    for d, vals in zip(descs, expect):
        o, n = int(d["out_off"]), int(d["num_values"])
        assert (got[o:o + n] == vals).all()
        covered[o:o + n] = True
    assert (got[~covered] == 0x5A5A5A5A).all()  # the gaps between pages are never written
    got_host, status = ctx.decode_batch_host(data, descs, out_values)
    assert (status == OK).all() and (got_host[covered] == got[covered]).all()


# ---- configuration-shaped cases -------------------------------------------------------------------

def test_config2_dict_index_pages(ctx, oracle):
    """1M-value dictionary-index pages, bit width 12, mixed runs; the width byte in front (a17)."""
    from parquet2_amd.deserialize import dict_indices_decoder
    from parquet2_amd.page import DataPage, DataPageHeaderV1, Encoding

    n = 1 << 20
    batch = synth.PageBatch()
    expect = []
    for p in range(3):
        stream, vals = synth.dict_page(synth.seed_for(2, p), n, 12)
        batch.add(stream, 12, n)
        expect.append((stream, vals))
    data, descs, out_values = batch.finish()
    got, status = ctx.decode_batch_host(data, descs, out_values)
    assert (status == OK).all()
    for d, (stream, vals) in zip(descs, expect):
        o = int(d["out_off"])
        assert (got[o:o + n] == vals).all()
    st, exp = oracle.hybrid_rle_decode(expect[0][0], 12, n)
    assert st == OK and (exp == expect[0][1]).all() and (got[:n] == exp).all()
    page = DataPage(DataPageHeaderV1(n, Encoding.RleDictionary), bytes([12]) + expect[1][0].tobytes())
    assert (dict_indices_decoder(page, ctx).collect() == expect[1][1]).all()


def test_config3_definition_levels(ctx, oracle):
    """Many small bit-width-1 pages with long RLE runs and ~5 % nulls."""
    n = 65536
    batch = synth.PageBatch()
    expect = []
    for p in range(300):
        stream, vals = synth.def_levels_page(synth.seed_for(3, p), n)
        batch.add(stream, 1, n)
        expect.append(vals)
    data, descs, out_values = batch.finish()
    got, status = ctx.decode_batch_host(data, descs, out_values)
    assert (status == OK).all()
    assert (got == np.concatenate(expect)).all()
    nulls = 1.0 - got.mean()
    assert 0.03 < nulls < 0.07
    st, exp = oracle.hybrid_rle_decode(data[int(descs[7]["in_off"]):][:int(descs[7]["in_len"])], 1, n)
    assert (exp == expect[7]).all()


@pytest.fixture(scope="module")
def config3_full():
    """10 000 distinct pages of 65 536 definition levels: the batch, and the levels it was written from."""
    n, n_pages = 65536, 10000
    batch = synth.PageBatch()
    levels = np.empty(n * n_pages, dtype=np.uint32)
    for p in range(n_pages):
        stream, vals = synth.def_levels_page(0xC3000000 + p, n)
        batch.add(stream, 1, n)
        levels[p * n:(p + 1) * n] = vals
    return batch.finish(), levels


def test_config3_full_size(ctx, oracle, config3_full):
    """The whole configuration: 10 000 pages x 65 536 levels of width 1 in one batch, every value compared with
    the levels the pages were written from (on the device, in slices), a sample of pages with the oracle."""
    import torch

    (data, descs, out_values), levels = config3_full
    n = 65536
    assert out_values == levels.size
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.full((out_values,), -1, dtype=torch.int32, device="cuda")
    st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values,
                                 torch.cuda.current_stream().cuda_stream)
    assert (st == OK).all()
    step = 1 << 26
    for lo in range(0, out_values, step):
        want = torch.from_numpy(levels[lo:lo + step].view(np.int32)).cuda()
        assert torch.equal(d_out[lo:lo + step], want), "first difference in values %d.." % lo
    for p in (0, 4999, 9999):
        d = descs[p]
        st_o, exp = oracle.hybrid_rle_decode(data[int(d["in_off"]):][:int(d["in_len"])], 1, n)
        assert st_o == OK
        assert (d_out[int(d["out_off"]):int(d["out_off"]) + n].cpu().numpy().view(np.uint32) == exp).all()


@pytest.mark.parametrize("w", [3, 17, 31])
def test_config4_adversarial_short_runs(ctx, oracle, w):
    n = 1 << 20
    stream, expected = synth.adversarial(synth.seed_for(4, w), n, w)
    got, st = ctx.decode_page_host(stream, w, n)
    assert st == OK and (got == expected).all()
    st, exp = oracle.hybrid_rle_decode(stream, w, n)
    assert st == OK and (exp == expected).all()


@pytest.mark.parametrize("w", list(range(1, 33)))
def test_config1_bitunpack_1m(ctx, oracle, w):
    """bitpacked::Decoder<u32> over 2^20 values: the bench pattern and uniform values."""
    from parquet2_amd.encoding import bitpacked

    n = 1 << 20
    pattern = synth.bench_pattern(n * w // 8)
    st, exp = oracle.bitunpack_u32(pattern, w, n)
    assert st == OK and (ctx.bitunpack_u32_host(pattern, w, n) == exp).all()
    vals = synth.uniform_values(w, n, w)
    packed = bitpacked.encode(vals, w)
    assert (ctx.bitunpack_u32_host(packed, w, n) == vals).all()
    # ragged tail: not a multiple of 4, 32 or 1024 values
    m = n - 1027
    assert (ctx.bitunpack_u32_host(packed, w, m) == vals[:m]).all()


# ---- device-resident entry points -------------------------------------------------------------------

def test_device_entry_points(ctx, oracle):
    import torch

    n = 300000
    batch = synth.PageBatch()
    expect = []
    for p, w in enumerate((1, 4, 12, 20, 32, 0)):
        vals = synth.dict_values(p, n + p, w, 50.0, 100.0) if w else np.zeros(n + p, np.uint32)
        batch.add(synth.encode_hybrid(vals, w) if w else np.zeros(3, np.uint8), w, n + p)
        expect.append(vals)
    batch.add(bytes([0b00010000, 1]), 16, 40)  # panics in the reference
    data, descs, out_values = batch.finish()
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.full((out_values,), 0xDEAD, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    status = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values, stream)
    assert status.tolist() == [OK] * 6 + [REFERENCE_WOULD_PANIC]
    got = d_out.cpu().numpy().view(np.uint32)
    for d, vals in zip(descs, expect):
        o = int(d["out_off"])
        assert (got[o:o + len(vals)] == vals).all()
    # asynchronous form + fetch_status, on a side stream, twice in a row (context reuse)
    side = torch.cuda.Stream()
    for _ in range(2):
        d_out.fill_(0x7777)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values,
                                           side.cuda_stream, want_status=False) is None
        assert ctx.fetch_status(len(descs)).tolist() == [OK] * 6 + [REFERENCE_WOULD_PANIC]
        got = d_out.cpu().numpy().view(np.uint32)
        for d, vals in zip(descs, expect):
            o = int(d["out_off"])
            assert (got[o:o + len(vals)] == vals).all()
    # bit-unpack, device-resident
    vals = synth.uniform_values(9, n, 13)
    from parquet2_amd.encoding import bitpacked
    packed = torch.from_numpy(np.concatenate([bitpacked.encode(vals, 13), np.zeros(16, np.uint8)])).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.bitunpack_u32_device(packed.data_ptr(), packed.numel() - 16, 13, n, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) == vals).all()


def test_bad_descriptors_are_rejected(ctx):
    from parquet2_amd.error import InvalidParameter

    data = np.zeros(64, np.uint8)
    descs = np.zeros(1, dtype=PAGE_DESC_DTYPE)
    descs[0] = (60, 10, 3, 5, 0)  # stream runs past the input buffer
    with pytest.raises(InvalidParameter):
        ctx.decode_batch_host(data, descs, 5)
    descs[0] = (0, 10, 3, 50, 0)  # output does not fit
    with pytest.raises(InvalidParameter):
        ctx.decode_batch_host(data, descs, 5)
    out, status = ctx.decode_batch_host(data, np.zeros(0, dtype=PAGE_DESC_DTYPE), 0)
    assert out.size == 0 and status.size == 0
    # a rejected batch leaves no statuses to fetch: not those of the batch before it either
    import torch

    good = np.zeros(1, dtype=PAGE_DESC_DTYPE)
    good[0] = (0, 10, 3, 5, 0)
    d_in = torch.zeros(80, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert ctx.decode_batch_device(d_in.data_ptr(), 64, good, d_out.data_ptr(), 64).tolist() == [OK]
    with pytest.raises(InvalidParameter):
        ctx.decode_batch_device(d_in.data_ptr(), 64, descs, d_out.data_ptr(), 5, want_status=False)
    with pytest.raises(InvalidParameter):
        ctx.fetch_status(1)


def test_idempotent_and_checksum_at_scale(ctx, oracle):
    """Size-independent properties on a batch of config-2 pages: decoding twice gives identical output, and
    a checksum over all pages equals the sum of the per-page oracle checksums."""
    import torch

    n = 1 << 20
    pages = [synth.dict_page(synth.seed_for(2, 10 + p), n, 12) for p in range(4)]
    reps = 8
    batch = synth.PageBatch()
    for r in range(reps):
        for stream, _ in pages:
            batch.add(stream, 12, n)
    data, descs, out_values = batch.finish()
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.zeros(out_values, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values, s)
    assert (st == OK).all()
    first = d_out.clone()
    ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values, s)
    assert torch.equal(first, d_out)
    per_page = [int(v.astype(np.uint64).sum()) for _, v in pages]
    assert int(d_out.to(torch.int64).sum().item()) == reps * sum(per_page)
    view = d_out.view(reps, len(pages), n)
    assert all(torch.equal(view[0], view[r]) for r in range(1, reps))
    assert (view[0, 2].cpu().numpy().view(np.uint32) == pages[2][1]).all()


def test_host_path_in_chunks(ctx, oracle):
    """A batch above the single-shot size goes through the three lanes of the host path in chunks of whole pages:
    ragged pages, empty pages, gaps between the outputs (which must stay untouched) and a page that panics."""
    rng = np.random.default_rng(5)
    batch = synth.PageBatch(align=1)
    expect = []
    for i in range(90):
        w = int(rng.integers(1, 21))
        n = 0 if i % 17 == 5 else int(rng.integers(100_000, 700_000))
        if i == 40:  # an RLE run with one of its two value bytes (decoder.rs:59-60)
            w, stream, vals = 12, np.array([16, 7], dtype=np.uint8), None
        else:
            stream, vals = synth.dict_page(synth.seed_for(7, i), n, w, mean_repeat=40.0, mean_distinct=300.0)
        batch.add(stream, w, n, out_align=int(rng.integers(1, 9)))
        expect.append(vals)
    data, descs, out_values = batch.finish()
    assert data.size + 4 * out_values > (64 << 20)
    got, status = ctx.decode_batch_host(data, descs, out_values)
    covered = np.zeros(out_values, dtype=bool)
    for i, vals in enumerate(expect):
        o, n = int(descs[i]["out_off"]), int(descs[i]["num_values"])
        if vals is None:
            assert status[i] == REFERENCE_WOULD_PANIC
        else:
            assert status[i] == OK
            assert (got[o:o + n] == vals).all(), i
        covered[o:o + n] = True
    assert (got[~covered] == 0).all()  # the host buffer between the pages is not written
    # spot-check the generator against the oracle on one page of the batch
    i = 3
    st_o, exp = oracle.hybrid_rle_decode(data[int(descs[i]["in_off"]):int(descs[i]["in_off"]) + int(descs[i]["in_len"])],
                                         int(descs[i]["num_bits"]), int(descs[i]["num_values"]))
    assert st_o == OK and (exp == expect[i]).all()
