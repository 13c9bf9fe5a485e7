"""Pins the CPU oracle against every known-answer vector the reference's own
unit tests hold for the hybrid-RLE / bit-unpack path (tests/golden/reference_kats.json).

Parity unpinned (no reference test covers these; the oracle follows the cited
code only): mixed streams with true RLE runs of length > 1 decoded to values,
RLE values with bits above num_bits set, the dropped header at the end of the
stream, the min(bytes, len) clamp beyond the `small` case, hybrid streams at
widths other than 1/2/3/10, num_values larger than the stream for w > 1, and
every panic path.  tests/test_oracle_properties.py exercises those against a
second model written from the same reference lines, and pins the conforming
streams among them with pages written by pyarrow (tests/golden/pyarrow_pages.json).
"""
import numpy as np

from _oracle import BitmapRun, OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC


def test_uleb128(oracle, kats):
    for c in kats["uleb128_decode"]:
        st, v, n = oracle.uleb128_decode(bytes(c["data"]))
        assert (st, v, n) == (OK, c["value"], c["consumed"]), c["src"]
    for c in kats["uleb128_round_trip"]:
        enc = oracle.uleb128_encode(c["value"])
        st, v, n = oracle.uleb128_decode(enc + bytes(10 - len(enc)))
        assert (st, v, n) == (OK, c["value"], len(enc)), c["src"]


def test_get_bit_width(oracle, kats):
    for level, width in kats["get_bit_width"]["table"]:
        assert oracle.get_bit_width(level) == width


def test_ceil8_get_length(oracle):
    # src/encoding/mod.rs:14-27
    assert [oracle.ceil8(v) for v in (0, 1, 7, 8, 9, 16, 17)] == [0, 1, 1, 1, 2, 2, 3]
    assert oracle.get_length(bytes([2, 0, 0, 0, 9])) == 2
    assert oracle.get_length(bytes([0x78, 0x56, 0x34, 0x12])) == 0x12345678
    assert oracle.get_length(bytes([1, 2, 3])) is None


def test_bitpacked_case1(oracle, kats):
    c = kats["bitpacked_case1"]
    st, out = oracle.bitunpack_u32(bytes(c["packed"]), c["num_bits"], len(c["unpacked"]))
    assert st == OK and out.tolist() == c["unpacked"]
    packed = oracle.bitpacked_encode_u32(c["unpacked"], c["num_bits"], 4 * 32)
    assert packed[:15].tolist() == c["packed"]
    c = kats["bitpacked_test_encode"]
    packed = oracle.bitpacked_encode_u32(c["unpacked"], c["num_bits"], 4 * 32)
    assert packed[:3].tolist() == c["packed_prefix"]


def test_bitpacked_decode(oracle, kats):
    for c in kats["bitpacked_decode"]:
        fn = oracle.bitunpack_u32 if c["type"] == "u32" else oracle.bitunpack_u64
        st, out = fn(bytes(c["packed"]), c["num_bits"], c["length"])
        assert st == OK, c["src"]
        assert out.tolist() == c["expected"], c["src"]


def test_bitpacked_errors(oracle, kats):
    for c in kats["bitpacked_errors"]["cases"]:
        st, _ = oracle.bitunpack_u64(bytes(c["packed"]), c["num_bits"], c["length"])
        assert (st == OK) == c["ok"], c
        assert st in (OK, OUT_OF_SPEC)
        st32, _ = oracle.bitunpack_u32(bytes(c["packed"]), c["num_bits"], c["length"])
        assert (st32 == OK) == c["ok"], c


def test_unpack_all_ones(oracle, kats):
    data = bytes([0xFF]) * 4096
    for bits in (8, 16, 32, 64):
        for w in range(bits + 1):
            st, out = oracle.unpack(bits, data, w)
            assert st == OK
            for v in out.tolist():
                trailing = (~v & (v + 1)).bit_length() - 1  # trailing_ones
                assert trailing == w, (bits, w, v)
        st, _ = oracle.unpack(bits, data, bits + 1)
        assert st == REFERENCE_WOULD_PANIC


def test_pack_round_trip(oracle, kats):
    for bits, key in ((16, "u16"), (32, "u32")):
        c = kats["pack_round_trip"][key]
        for w in c["widths"]:
            st, packed = oracle.pack(bits, c["input"], w)
            assert st == OK
            st, out = oracle.unpack(bits, packed, w)
            assert st == OK and out.tolist() == c["input"], (bits, w)


def test_run_decoder(oracle, kats):
    for c in kats["run_decoder"]:
        stream = bytes(c["stream"])
        st, runs = oracle.runs(stream, c["bit_width"])
        assert st == OK and len(runs) >= 1, c["src"]
        run = runs[0]
        assert run[0] == c["kind"]
        assert list(stream[run[1]:run[1] + run[2]]) == c["slice"]
        if c["kind"] == "rle":
            assert run[3] == c["run_length"]
        else:
            st, vals = oracle.bitunpack_u32(stream[run[1]:run[1] + run[2]], c["bit_width"], c["length"])
            assert st == OK and vals.tolist() == c["values"]


def test_hybrid_rle_decode(oracle, kats):
    for c in kats["hybrid_rle_decode"]:
        st, out = oracle.hybrid_rle_decode(bytes(c["data"]), c["num_bits"], c["num_values"])
        assert st == OK, c["src"]
        expected = list(range(c["expected_range"])) if "expected_range" in c else c["expected"]
        assert out.tolist() == expected, c["src"]


def test_hybrid_rle_round_trip(oracle, kats):
    c = kats["hybrid_rle_round_trip"]
    data = list(range(c["range"]))
    buf = oracle.encode_u32(data, c["num_bits"])
    st, out = oracle.hybrid_rle_decode(buf, c["num_bits"], len(data))
    assert st == OK and out.tolist() == data


def test_encoders(oracle, kats):
    for c in kats["encoder"]:
        if c["kind"] == "u32":
            got = oracle.encode_u32(c["values"], c["num_bits"])
        elif c["kind"] == "bool":
            got = oracle.encode_bool(c["values"])
        else:  # bools drawn from BitmapIter::new(bitmap, offset, len)
            bm = c["bitmap"]
            bools = [(bm[(c["offset"] + i) // 8] >> ((c["offset"] + i) % 8)) & 1 for i in range(c["len"])]
            got = oracle.encode_bool(bools)
        assert list(got) == c["expected"], c["src"]


def _mk_runs(spec):
    """spec rows: ["bitmap", bytes, len] | ["repeated", bool, len] -> (BitmapRun list, stream bytes)."""
    stream = bytearray()
    runs = []
    for row in spec:
        r = BitmapRun()
        if row[0] == "bitmap":
            r.kind, r.offset, r.bytes, r.length = 0, len(stream), len(row[1]), row[2]
            stream += bytes(row[1])
        else:
            r.kind, r.is_set, r.length = 1, int(row[1]), row[2]
        runs.append(r)
    return runs, bytes(stream)


def test_filtered_runs(oracle, kats):
    for c in kats["filtered_runs"]:
        runs, stream = _mk_runs(c["runs"])
        st, out = oracle.filtered_runs(runs, stream, c["intervals"])
        assert st == OK, c["src"]
        got = []
        for o in out:
            if o.kind == 0:
                got.append(["bitmap", list(stream[o.values_off:o.values_off + o.values_len]), o.offset, o.length])
            elif o.kind == 1:
                got.append(["repeated", bool(o.is_set), o.length])
            else:
                got.append(["skipped", o.set_count])
        assert got == c["expected"], c["src"]


def test_bitmap_iter_via_boolean_iter(oracle):
    # src/encoding/hybrid_rle/decoder.rs:97-121 (basics_2 bitmap, pyarrow-validated) read through
    # HybridRleBooleanIter: only the first run is ever decoded (src/deserialize/hybrid_rle.rs:165-175)
    stream = bytes([0b00000101, 0b11101011, 0b00000010])
    st, out = oracle.boolean_iter(stream, 10)
    assert st == OK and out.tolist() == [1, 1, 0, 1, 0, 1, 1, 1, 0, 1]
    two_runs = bytes([0b00010000, 1, 0b00010000, 0])  # Rle(1, 8), Rle(0, 8)
    st, out = oracle.boolean_iter(two_runs, 16)
    assert st == OK and out.tolist() == [1] * 8


def test_bitmap_runs(oracle):
    # src/deserialize/hybrid_rle.rs:90-115
    stream = bytes([0b00000101, 0b11101011, 0b00000010, 0b00010000, 1])
    st, runs, n = oracle.bitmap_runs(stream, 20)
    assert st == OK and n == 2
    assert (runs[0].kind, runs[0].offset, runs[0].bytes, runs[0].length) == (0, 1, 2, 16)
    assert (runs[1].kind, runs[1].is_set, runs[1].length) == (1, 1, 4)
    st, runs, n = oracle.bitmap_runs(stream, 10)  # stops once `length` items are consumed
    assert n == 1 and runs[0].length == 10
    st, runs, n = oracle.bitmap_runs(stream, 100)  # no zero padding in this adapter
    assert n == 2 and runs[1].length == 8


def test_split_buffer_and_dict_locate(oracle):
    # src/page/mod.rs:350-411
    buf = bytes([2, 0, 0, 0, 0xAA, 0xBB, 1, 0, 0, 0, 0xCC, 0xDD, 0xEE])
    st, s = oracle.split_buffer_v1(buf, True, True)
    assert st == OK and (s.rep_off, s.rep_len, s.def_off, s.def_len, s.val_off, s.val_len) == (4, 2, 10, 1, 11, 2)
    st, s = oracle.split_buffer_v1(buf, False, True)
    assert st == OK and (s.rep_len, s.def_off, s.def_len, s.val_off, s.val_len) == (0, 4, 2, 6, 7)
    st, s = oracle.split_buffer_v1(buf, False, False)
    assert st == OK and (s.val_off, s.val_len) == (0, len(buf))
    assert oracle.split_buffer_v1(bytes([9, 0, 0, 0, 1]), True, False)[0] == OUT_OF_SPEC
    assert oracle.split_buffer_v1(bytes([9, 0, 0]), False, True)[0] == OUT_OF_SPEC
    st, s = oracle.split_buffer_v2(10, 3, 4)
    assert st == OK and (s.rep_off, s.rep_len, s.def_off, s.def_len, s.val_off, s.val_len) == (0, 3, 3, 4, 7, 3)
    assert oracle.split_buffer_v2(5, 3, 4)[0] == REFERENCE_WOULD_PANIC
    # src/deserialize/utils.rs:13-27
    assert oracle.dict_indices_locate(bytes([12, 3, 2])) == (OK, 12, 1)
    assert oracle.dict_indices_locate(bytes([33, 3, 2]))[0] == OUT_OF_SPEC
    assert oracle.dict_indices_locate(b"")[0] == REFERENCE_WOULD_PANIC


def test_zigzag_and_delta_bitpacked(oracle, kats):
    for byte, expected in kats["zigzag_leb128_decode"]["cases"]:
        assert oracle.zigzag_leb128_decode(bytes([byte])) == (0, expected, 1)
    for c in kats["delta_bitpacked_decode"]:
        st, values, consumed = oracle.delta_bitpacked_decode(bytes(c["data"]), len(c["expected"]))
        assert st == 0 and values.tolist() == c["expected"] and consumed == c["consumed"], c["src"]


def test_delta_bitpacked_encoder(oracle, kats):
    for byte, value in kats["zigzag_leb128_encode"]["cases"]:
        assert oracle.zigzag_leb128_encode(value) == bytes([byte])
    for c in kats["delta_bitpacked_encode"]:
        assert oracle.delta_bitpacked_encode(c["data"]) == bytes(c["expected"]), c["src"]
    for c in kats["delta_bitpacked_round_trip"]:
        buf = oracle.delta_bitpacked_encode(c["data"])
        st, values, consumed = oracle.delta_bitpacked_decode(buf, len(c["data"]))
        assert st == 0 and values.tolist() == c["data"], c["src"]
        if c.get("consumed_is_len"):
            assert consumed == len(buf)
