"""Parity of the device bitpacked::Decoder<u64> (pq2_bitunpack_u64_host/_device) against the reference's own
vectors (src/encoding/bitpacked/decode.rs:137-147 test_decode_u64, :199-211 test_errors, unpack.rs:97-137 all-ones
property for every width of unpack64) and the CPU oracle."""
import numpy as np
import pytest

from parquet2_amd.error import OutOfSpec, ReferencePanic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from parquet2_amd import Context

    c = Context(0)
    yield c
    c.close()


def test_reference_u64_kats(ctx, kats):
    seen = 0
    for c in kats["bitpacked_decode"]:
        if c["type"] != "u64":
            continue
        got = ctx.bitunpack_u64_host(bytes(c["packed"]), c["num_bits"], c["length"])
        assert got.tolist() == c["expected"], c["src"]
        seen += 1
    assert seen >= 1
    # the other decode vectors hold for every T: the fields of the stream do not depend on the block size
    for c in kats["bitpacked_decode"]:
        assert ctx.bitunpack_u64_host(bytes(c["packed"]), c["num_bits"], c["length"]).tolist() == c["expected"]
    for c in kats["bitpacked_errors"]["cases"]:  # decode.rs:199-211, run there for u64 as well
        if c["ok"]:
            ctx.bitunpack_u64_host(bytes(c["packed"]), c["num_bits"], c["length"])
        else:
            with pytest.raises(OutOfSpec):
                ctx.bitunpack_u64_host(bytes(c["packed"]), c["num_bits"], c["length"])


def test_all_ones_every_width(ctx):
    data = bytes([0xFF]) * 8192
    for w in range(1, 65):
        got = ctx.bitunpack_u64_host(data, w, 1024)
        assert (got == np.uint64((1 << w) - 1)).all(), w  # unpack.rs:128-136


@pytest.mark.parametrize("w", list(range(1, 65)))
def test_every_width_against_oracle(ctx, oracle, w):
    rng = np.random.default_rng(7000 + w)
    for length in (1, 63, 64, 65, 1000, 40000 + w):
        packed = rng.integers(0, 256, size=(length * w + 7) // 8 + (w % 5), dtype=np.uint8)
        st, exp = oracle.bitunpack_u64(packed, w, length)
        assert st == 0
        got = ctx.bitunpack_u64_host(packed, w, length)
        assert got.dtype == np.uint64 and (got == exp).all(), (w, length)


def test_device_entry_point_alignment_and_errors(ctx, oracle):
    import torch

    rng = np.random.default_rng(5)
    w, length = 37, 1 << 20
    packed = rng.integers(0, 256, size=length * w // 8 + 3, dtype=np.uint8)
    st, exp = oracle.bitunpack_u64(packed, w, length)
    buf = torch.zeros(packed.size + 16, dtype=torch.uint8, device="cuda")
    for shift in (0, 1, 5):  # the stream may start at any byte
        buf[shift:shift + packed.size] = torch.from_numpy(packed).cuda()
        d_out = torch.zeros(length, dtype=torch.int64, device="cuda")
        ctx.bitunpack_u64_device(buf.data_ptr() + shift, packed.size, w, length, d_out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
        assert (d_out.cpu().numpy().view(np.uint64) == exp).all()
    with pytest.raises(ReferencePanic):  # unpack.rs:87: no such width
        ctx.bitunpack_u64_host(packed, 65, 8)
    with pytest.raises(OutOfSpec):  # decode.rs:33-35
        ctx.bitunpack_u64_host(packed, 0, 8)
    assert ctx.bitunpack_u64_host(b"", 3, 0).size == 0  # decode.rs:46-50
