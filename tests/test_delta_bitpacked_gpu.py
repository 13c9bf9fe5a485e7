"""Parity of the device delta_bitpacked::Decoder (pq2_delta_bitpacked_decode_batch_device / _host, delta_kernel)
against the reference's own vectors (src/encoding/delta_bitpacked/decoder.rs:246-365) and the CPU oracle, on
generated streams of many block shapes and on streams that are cut, corrupted or absurd."""
import numpy as np
import pytest

from parquet2_amd.error import InvalidParameter, OutOfSpec, ReferencePanic

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def ctx():
    from parquet2_amd import Context

    c = Context(0)
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


def zigzag(v):
    return uleb(((v << 1) ^ (v >> 63)) & M64)


def encode_delta(values, block_size=128, num_mini_blocks=4, width_floor=0):
    """A DELTA_BINARY_PACKED stream of the i64 `values` (the layout decoder.rs:121-135 describes).  Mini-blocks
    behind the last value get width 0 and no bytes; `width_floor` widens every used mini-block to at least that."""
    values = [int(v) for v in values]
    per_mini = block_size // num_mini_blocks
    out = bytearray(uleb(block_size) + uleb(num_mini_blocks) + uleb(len(values)) + zigzag(values[0] if values else 0))
    deltas = []
    for a, b in zip(values, values[1:]):
        d = (b - a) & M64
        deltas.append(d - (1 << 64) if d >> 63 else d)
    for at in range(0, len(deltas), block_size):
        block = deltas[at:at + block_size]
        low = min(block)
        out += zigzag(low)
        rel = [(d - low) & M64 for d in block]
        widths, payload = [], bytearray()
        for m in range(num_mini_blocks):
            mini = rel[m * per_mini:(m + 1) * per_mini]
            if not mini:
                widths.append(0)
                continue
            w = max(max(x.bit_length() for x in mini), width_floor)
            widths.append(w)
            big = 0
            for i, x in enumerate(mini):
                big |= x << (i * w)
            payload += big.to_bytes((per_mini * w + 7) // 8, "little")
        out += bytes(widths) + payload
    return bytes(out)


def decode_batch(ctx, streams, caps, pad_front=0):
    """Through the device entry point: the streams back to back from byte `pad_front` of one buffer."""
    import torch

    offs, at = [], pad_front
    for s in streams:
        offs.append(at)
        at += len(s)
    blob = np.zeros(at + 16, dtype=np.uint8)
    for o, s in zip(offs, streams):
        blob[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    from parquet2_amd import PAGE_DESC_DTYPE

    descs = np.zeros(len(streams), dtype=PAGE_DESC_DTYPE)
    out_at = 0
    for i, (o, s, cap) in enumerate(zip(offs, streams, caps)):
        descs[i] = (o, len(s), 0, cap, out_at)
        out_at += cap
    d_in = torch.from_numpy(blob).cuda()
    d_out = torch.full((max(out_at, 1),), -0x0123456789ABCDEF, dtype=torch.int64, device="cuda")
    status, counts, consumed = ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), blob.size, descs,
                                                                      d_out.data_ptr(), out_at)
    return status, counts, consumed, d_out.cpu().numpy(), descs


def check_against_oracle(ctx, oracle, streams, caps, pad_front=0):
    status, counts, consumed, out, descs = decode_batch(ctx, streams, caps, pad_front)
    seen = {}
    for i, (s, cap) in enumerate(zip(streams, caps)):
        st, exp, used = oracle.delta_bitpacked_decode(s, cap)
        assert status[i] == st, (i, status[i], st, s[:40])
        seen[st] = seen.get(st, 0) + 1
        o = int(descs[i]["out_off"])
        if st == 0:
            assert counts[i] == len(exp) and consumed[i] == used, (i, counts[i], len(exp), consumed[i], used)
            assert (out[o:o + len(exp)] == exp).all(), i
            # nothing behind the stream's values is touched
            assert (out[o + len(exp):o + cap] == -0x0123456789ABCDEF).all(), i
    return seen


def test_reference_vectors(ctx, kats):
    from parquet2_amd.encoding import delta_bitpacked

    assert len(kats["delta_bitpacked_decode"]) == 4
    for c in kats["delta_bitpacked_decode"]:
        d = delta_bitpacked.Decoder.try_new(bytes(c["data"]), ctx)
        assert d.size_hint() == (len(c["expected"]), len(c["expected"]))
        assert list(d) == c["expected"], c["src"]
        assert d.consumed_bytes() == c["consumed"], c["src"]
    streams = [bytes(c["data"]) for c in kats["delta_bitpacked_decode"]]
    for pad in (0, 1, 7, 13):  # every alignment class of the first byte matters to the dword loads
        status, counts, consumed, out, descs = decode_batch(ctx, streams, [70] * 4, pad)
        for i, c in enumerate(kats["delta_bitpacked_decode"]):
            o = int(descs[i]["out_off"])
            assert status[i] == 0 and counts[i] == len(c["expected"]) and consumed[i] == c["consumed"]
            assert out[o:o + counts[i]].tolist() == c["expected"]


def test_reference_round_trips(ctx, kats):
    """encode -> Decoder gives the data back (src/encoding/delta_bitpacked/mod.rs:13-90), the decode on the device."""
    from parquet2_amd.encoding import delta_bitpacked

    for c in kats["delta_bitpacked_round_trip"]:
        buffer = delta_bitpacked.encode(c["data"])
        d = delta_bitpacked.Decoder.try_new(buffer, ctx)
        assert d.collect().tolist() == c["data"], c["src"]
        if c.get("consumed_is_len"):
            assert d.consumed_bytes() == len(buffer)
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 129, 4000):
        v = rng.integers(-(1 << 63), (1 << 63) - 1, size=n, dtype=np.int64, endpoint=True)
        assert (delta_bitpacked.Decoder.try_new(delta_bitpacked.encode(v), ctx).collect() == v).all()


# the first four are also what the cut and corrupted streams below are made of; blocks of at most 512 values in at
# most 16 mini-blocks take the kernel's chunked path, the others its general one
SHAPES = [(128, 4), (128, 1), (128, 16), (256, 8), (1024, 4), (8192, 128), (1280, 160), (128 * 75, 75 * 2), (640, 5),
          (512, 16), (512, 4), (384, 16), (384, 3), (512, 32)]


@pytest.mark.parametrize("block_size,num_mini_blocks", SHAPES)
def test_generated_streams(ctx, oracle, block_size, num_mini_blocks):
    rng = np.random.default_rng(block_size * 1000 + num_mini_blocks)
    streams, caps, originals = [], [], []
    per_mini = block_size // num_mini_blocks
    lengths = [0, 1, 2, per_mini, per_mini + 1, per_mini + 2, block_size, block_size + 1, block_size + 2,
               2 * block_size + 1, 3 * block_size + per_mini + 1, 5 * block_size - 3, 20011]
    for k, n in enumerate(lengths):
        kind = k % 5
        if kind == 0:  # small steps up and down
            v = np.cumsum(rng.integers(-3, 100, size=n, dtype=np.int64))
        elif kind == 1:  # any i64: the deltas wrap, the widths reach 64
            v = rng.integers(-(1 << 63), (1 << 63) - 1, size=n, dtype=np.int64, endpoint=True)
        elif kind == 2:  # constant steps: every width is 0
            v = np.arange(n, dtype=np.int64) * 7 - 100
        elif kind == 3:  # widths that change from one mini-block to the next
            scale = 1 << rng.integers(0, 40, size=n // max(per_mini, 1) + 1)
            v = np.cumsum(rng.integers(0, 1 << 20, size=n, dtype=np.int64) % np.repeat(scale, per_mini)[:n])
        else:
            v = rng.integers(-(1 << 31), 1 << 31, size=n, dtype=np.int64)
        streams.append(encode_delta(v, block_size, num_mini_blocks, width_floor=int(rng.integers(0, 3))))
        caps.append(n + int(rng.integers(0, 3)))
        originals.append(v)
    # the generator is itself checked: the oracle gets the values back
    for s, cap, v in zip(streams, caps, originals):
        st, exp, used = oracle.delta_bitpacked_decode(s, cap)
        assert st == 0 and used == len(s) and (exp == v).all()
    seen = check_against_oracle(ctx, oracle, streams, caps, pad_front=int(rng.integers(0, 16)))
    assert seen == {0: len(streams)}


def values_of_deltas(first, deltas):
    """The i64 values whose differences are `deltas` (wrapping, as the decoder's additions do)."""
    out, v = [first], first
    for d in deltas:
        v = (v + d) & M64
        out.append(v - (1 << 64) if v >> 63 else v)
    return out


@pytest.mark.parametrize("num_mini_blocks", [1, 2, 4])
def test_blocks_of_128_values(ctx, oracle, num_mini_blocks):
    """Blocks of 128 values in 1, 2 or 4 mini-blocks have a walk and an expansion of their own in delta_kernel:
    min_delta varints of every length (those of 9 and 10 bytes leave that path), widths on both sides of 24 (up to
    which a block's sums are taken in 32 bits), of 32 and of 64, streams of several chunks, last blocks of every
    length, junk in the width bytes of mini-blocks that hold no value, and streams that end within 12 bytes of their
    last block's start."""
    rng = np.random.default_rng(40 + num_mini_blocks)
    per_mini = 128 // num_mini_blocks
    streams, caps = [], []
    for k in range(160):
        n_deltas = int(rng.choice([1, 2, per_mini - 1, per_mini, per_mini + 1, 127, 128, 129, 1000, 128 * 70 + int(rng.integers(0, 128)),
                                   int(rng.integers(1, 6000))]))
        deltas = []
        for at in range(0, n_deltas, 128):
            length = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10]))  # bytes of min_delta's varint
            magnitude = 0 if length == 1 and k % 2 else int(rng.integers(1 << max(7 * length - 8, 0), 1 << min(7 * length - 1, 63)))
            low = -magnitude if rng.integers(0, 2) else magnitude - 1 if magnitude else 0
            block = []
            for m in range(num_mini_blocks):
                w = int(rng.choice([0, 1, 7, 12, 23, 24, 25, 31, 32, 33, 48, 63, 64]))
                if k % 3 == 0:
                    w = min(w, 24)  # whole streams of narrow blocks, and mixed ones
                rel = [int.from_bytes(rng.bytes(8), "little") >> (64 - w) if w else 0 for _ in range(per_mini)]
                if w:
                    rel[int(rng.integers(0, per_mini))] = (1 << w) - 1  # the width is reached
                block += [low + r for r in rel]
            block[int(rng.integers(0, 128))] = low  # min_delta is the block's least delta
            deltas += block
        deltas = deltas[:n_deltas]
        values = values_of_deltas(int(rng.integers(-(1 << 63), 1 << 63)), deltas)
        s = bytearray(encode_delta(values, 128, num_mini_blocks))
        tail = n_deltas % 128
        if k % 4 == 1 and tail and (tail - 1) // per_mini + 1 < num_mini_blocks:
            # the widths of the last block's unused mini-blocks: whatever the writer left there
            used_minis = (tail - 1) // per_mini + 1
            # the last block begins where the stream of the values before it ends
            head = encode_delta(values[:n_deltas + 1 - tail], 128, num_mini_blocks)
            at = len(head) + len(uleb(n_deltas + 1)) - len(uleb(n_deltas + 1 - tail))
            while s[at] & 0x80:
                at += 1
            at += 1
            for m in range(used_minis, num_mini_blocks):
                s[at + m] = int(rng.integers(0, 256))
        streams.append(bytes(s))
        caps.append(n_deltas + 1 + int(rng.integers(0, 2)))
    seen = check_against_oracle(ctx, oracle, streams, caps, pad_front=int(rng.integers(0, 16)))
    assert seen.get(0, 0) >= 150, seen  # (junk in unused width bytes may be refused: then by both)


def test_every_prefix_of_a_stream(ctx, oracle):
    rng = np.random.default_rng(5)
    v = np.cumsum(rng.integers(-1000, 1000, size=300, dtype=np.int64))
    whole = encode_delta(v, 128, 4)
    streams = [whole[:n] for n in range(len(whole) + 1)] + [whole + b"\x01\x02\x03"]
    seen = check_against_oracle(ctx, oracle, streams, [300] * len(streams))
    assert seen[0] >= 2 and seen[1] > 10 and seen[5] >= 1  # the empty stream: a division by zero mini-blocks


def test_headers_the_reference_refuses(ctx, oracle):
    ok = encode_delta(range(10))
    cases = {
        "block size not a multiple of 128": uleb(64) + uleb(4) + uleb(10) + zigzag(0) + ok[5:],
        "zero mini-blocks": uleb(128) + uleb(0) + uleb(10) + zigzag(0),
        "values per mini-block not a multiple of 8": uleb(128) + uleb(32) + uleb(10) + zigzag(0) + bytes(64),
        "a varint that overflows": b"\x80" * 9 + b"\x02",
        "a ten-byte varint that fits": b"\x80" * 9 + b"\x01" + uleb(4) + uleb(1) + zigzag(3),
        "a width above 64": uleb(128) + uleb(4) + uleb(10) + zigzag(0) + zigzag(0) + bytes([65, 0, 0, 0]) + bytes(600),
        "a width above 64 without its bytes": uleb(128) + uleb(4) + uleb(10) + zigzag(0) + zigzag(0) + bytes([65, 0, 0, 0]),
        "a width above 64 behind the last value": uleb(128) + uleb(4) + uleb(10) + zigzag(0) + zigzag(0) + bytes([1, 200, 0, 0]) + bytes(4),
        "a block size of zero": uleb(0) + uleb(4) + uleb(10) + zigzag(0) + zigzag(0) + bytes(4) + zigzag(0) + bytes(4),
        "a block size of zero, cut": uleb(0) + uleb(4) + uleb(10) + zigzag(0) + zigzag(0) + bytes(4) + zigzag(0) + bytes(3),
        "a block size of zero, wide": uleb(0) + uleb(1) + uleb(10) + zigzag(0) + zigzag(0) + bytes([200]),
        "a block size of zero, one value": uleb(0) + uleb(4) + uleb(1) + zigzag(-5),
        "more mini-blocks than bytes": uleb(128 << 20) + uleb(1 << 20) + uleb(10) + zigzag(0) + zigzag(0) + bytes(100),
        "absurd block size": uleb(1 << 50) + uleb(1) + uleb(10) + zigzag(0) + zigzag(0) + bytes([1]) + bytes(100),
        "no values": uleb(128) + uleb(4) + uleb(0) + zigzag(9) + b"junk",
        "one value": uleb(128) + uleb(4) + uleb(1) + zigzag(-9) + b"junk",
        "more values than room": encode_delta(range(400)),
        "min delta overflows": uleb(128) + uleb(4) + uleb(10) + zigzag(0) + b"\xff" * 9 + b"\x7f",
    }
    streams = list(cases.values())
    seen = check_against_oracle(ctx, oracle, streams, [300] * len(streams), pad_front=3)
    assert seen.get(0, 0) >= 3 and seen.get(1, 0) >= 3 and seen.get(5, 0) >= 6 and seen.get(2, 0) == 1
    status, counts, _, _, _ = decode_batch(ctx, [cases["more values than room"]], [300])
    assert status[0] == 2 and counts[0] == 400  # the caller learns how much room the stream needs


def test_corrupted_streams(ctx, oracle):
    rng = np.random.default_rng(11)
    streams, caps = [], []
    for k in range(400):
        n = int(rng.integers(2, 700))
        shape = SHAPES[k % 4]
        s = bytearray(encode_delta(np.cumsum(rng.integers(-50, 50, size=n, dtype=np.int64)), *shape))
        for _ in range(int(rng.integers(1, 4))):  # a few bytes anywhere: widths, varints, payload
            s[int(rng.integers(0, len(s)))] = int(rng.integers(0, 256))
        if k % 3 == 0:
            del s[int(rng.integers(4, len(s))):]
        streams.append(bytes(s))
        caps.append(n + 64)
    seen = check_against_oracle(ctx, oracle, streams, caps, pad_front=5)
    assert seen.get(0, 0) > 20 and seen.get(1, 0) > 20


def test_many_streams_in_one_batch(ctx, oracle):
    rng = np.random.default_rng(13)
    streams, caps = [], []
    for k in range(3000):
        n = int(rng.integers(0, 900))
        streams.append(encode_delta(np.cumsum(rng.integers(-(1 << (k % 40)), 1 << (k % 40), size=n, dtype=np.int64))))
        caps.append(n)
    seen = check_against_oracle(ctx, oracle, streams, caps)
    assert seen == {0: 3000}


def test_host_entry_point_errors(ctx):
    with pytest.raises(OutOfSpec):
        ctx.delta_bitpacked_decode_host(encode_delta(range(300))[:-1], 300)
    with pytest.raises(ReferencePanic):
        ctx.delta_bitpacked_decode_host(b"", 300)
    with pytest.raises(InvalidParameter):
        ctx.delta_bitpacked_decode_host(encode_delta(range(300)), 299)
    values, consumed = ctx.delta_bitpacked_decode_host(encode_delta(range(300)) + b"tail", 300)
    assert values.tolist() == list(range(300)) and consumed == len(encode_delta(range(300)))


def test_descriptor_bounds(ctx):
    import torch

    from parquet2_amd import PAGE_DESC_DTYPE

    d_in = torch.zeros(64, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(16, dtype=torch.int64, device="cuda")
    descs = np.zeros(1, dtype=PAGE_DESC_DTYPE)
    descs[0] = (60, 8, 0, 4, 0)  # ends behind the input
    with pytest.raises(InvalidParameter):
        ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), 64, descs, d_out.data_ptr(), 16)
    descs[0] = (0, 8, 0, 4, 14)  # ends behind the output
    with pytest.raises(InvalidParameter):
        ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), 64, descs, d_out.data_ptr(), 16)
    descs[0] = (0, 8, 0, 4, 0)
    with pytest.raises(InvalidParameter):  # int64 values need an 8-byte aligned output
        ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), 64, descs, d_out.data_ptr() + 4, 8)
