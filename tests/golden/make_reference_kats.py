#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the literal known-answer vectors
(inputs and expected outputs) that the reference's own unit tests hold for the
hybrid-RLE / bit-unpack path.  Each entry names the reference test it comes
from (file:line, relative to the reference root).

Small vectors are transcribed below as data.  The one long vector (the
pyarrow-produced stream of `pyarrow_integration`, ~1.25 KB) is read out of the
reference's test source as numbers when /root/reference is present; otherwise
the previously generated JSON is kept.  Only numbers are extracted - no
reference source text is stored.

Run:  python tests/golden/make_reference_kats.py
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_kats.json")
REF = os.environ.get("PQ2_REFERENCE_ROOT", "/root/reference")


def pyarrow_stream():
    path = os.path.join(REF, "src/encoding/hybrid_rle/mod.rs")
    if not os.path.exists(path):
        return None
    text = open(path).read()
    start = text.index("fn pyarrow_integration")
    m = re.search(r"let data = vec!\[(.*?)\];", text[start:], re.S)
    nums = [int(x) for x in re.findall(r"\d+", m.group(1))]
    return nums


def main():
    kats = {}

    # ---- src/encoding/uleb128.rs:52-96 --------------------------------
    kats["uleb128_decode"] = [
        {"src": "src/encoding/uleb128.rs:52-58 decode_1", "data": [0xE5, 0x8E, 0x26, 0xDE, 0xAD, 0xBE, 0xEF],
         "value": 624485, "consumed": 3},
        {"src": "src/encoding/uleb128.rs:60-66 decode_2", "data": [0b00010000, 0b00000001, 0b00000011, 0b00000011],
         "value": 16, "consumed": 1},
    ]
    kats["uleb128_round_trip"] = [
        {"src": "src/encoding/uleb128.rs:68-76 round_trip", "value": 123124234},
        {"src": "src/encoding/uleb128.rs:78-86 min_value", "value": 0},
        {"src": "src/encoding/uleb128.rs:88-96 max_value", "value": 2**64 - 1},
    ]

    # ---- src/read/levels.rs:12-26 --------------------------------------
    kats["get_bit_width"] = {
        "src": "src/read/levels.rs:12-26",
        "table": [[0, 0], [1, 1], [2, 2], [3, 2], [4, 3], [5, 3], [6, 3], [7, 3], [8, 4], [15, 4], [255, 8], [256, 9]],
    }

    # ---- src/encoding/bitpacked/mod.rs:172-219 -------------------------
    b3 = [0b10001000, 0b11000110, 0b11111010]
    kats["bitpacked_case1"] = {
        "src": "src/encoding/bitpacked/mod.rs:172-196 case1 (decode.rs:116-123 decode_large, mod.rs:199-205 encode_large)",
        "num_bits": 3, "packed": b3 * 5, "unpacked": list(range(8)) * 5,
    }
    kats["bitpacked_test_encode"] = {
        "src": "src/encoding/bitpacked/mod.rs:207-219 test_encode",
        "num_bits": 3, "unpacked": list(range(8)), "packed_prefix": b3,
    }

    # ---- src/encoding/bitpacked/decode.rs:94-211 -----------------------
    even = [0b10001000, 0b11000110, 0b00011010]
    kats["bitpacked_decode"] = [
        {"src": "src/encoding/bitpacked/decode.rs:94-113 test_decode_rle", "type": "u32", "num_bits": 3, "length": 8,
         "packed": b3, "expected": list(range(8))},
        {"src": "src/encoding/bitpacked/decode.rs:125-135 test_decode_bool", "type": "u32", "num_bits": 1, "length": 8,
         "packed": [0b10101010], "expected": [0, 1, 0, 1, 0, 1, 0, 1]},
        {"src": "src/encoding/bitpacked/decode.rs:137-147 test_decode_u64", "type": "u64", "num_bits": 1, "length": 8,
         "packed": [0b10101010], "expected": [0, 1, 0, 1, 0, 1, 0, 1]},
        {"src": "src/encoding/bitpacked/decode.rs:149-171 even_case", "type": "u32", "num_bits": 3, "length": 8 * 99,
         "packed": even * 99, "expected": [0, 1, 2, 3, 4, 5, 6, 0] * 99},
        {"src": "src/encoding/bitpacked/decode.rs:173-197 odd_case", "type": "u32", "num_bits": 3, "length": 33,
         "packed": even * 4 + [0b00000010], "expected": [0, 1, 2, 3, 4, 5, 6, 0] * 4 + [2]},
    ]
    kats["bitpacked_errors"] = {
        "src": "src/encoding/bitpacked/decode.rs:199-211 test_errors (Decoder::<u64>)",
        "cases": [
            {"packed": [], "num_bits": 1, "length": 0, "ok": True},
            {"packed": [], "num_bits": 1, "length": 1, "ok": False},
            {"packed": [1], "num_bits": 1, "length": 8, "ok": True},
            {"packed": [1, 1], "num_bits": 2, "length": 8, "ok": True},
            {"packed": [1], "num_bits": 1, "length": 9, "ok": False},
            {"packed": [1], "num_bits": 0, "length": 1, "ok": False},
        ],
    }
    kats["unpack_all_ones"] = {
        "src": "src/encoding/bitpacked/unpack.rs:101-136 test_basic",
        "note": "input = 0xFF bytes; every output of unpack{8,16,32,64} at width w has exactly w trailing ones",
        "widths": {"8": 8, "16": 16, "32": 32, "64": 64},
    }
    kats["pack_round_trip"] = {
        "src": "src/encoding/bitpacked/pack.rs:82-107 test_basic / test_u32",
        "u16": {"input": list(range(16)), "widths": list(range(4, 16))},
        "u32": {"input": list(range(16)) * 2, "widths": list(range(4, 32))},
    }

    # ---- src/encoding/hybrid_rle/decoder.rs:73-143 ---------------------
    kats["run_decoder"] = [
        {"src": "src/encoding/hybrid_rle/decoder.rs:73-95 basics_1", "bit_width": 1,
         "stream": [0b00000011, 0b00001011], "kind": "bitpacked", "slice": [0b00001011],
         "length": 5, "values": [1, 1, 0, 1, 0]},
        {"src": "src/encoding/hybrid_rle/decoder.rs:97-121 basics_2", "bit_width": 1,
         "stream": [0b00000101, 0b11101011, 0b00000010], "kind": "bitpacked", "slice": [0b11101011, 0b00000010],
         "length": 10, "values": [1, 1, 0, 1, 0, 1, 1, 1, 0, 1]},
        {"src": "src/encoding/hybrid_rle/decoder.rs:123-143 basics_3", "bit_width": 1,
         "stream": [0b00010000, 0b00000001], "kind": "rle", "slice": [0b00000001], "run_length": 8},
    ]

    # ---- src/encoding/hybrid_rle/mod.rs:122-263 ------------------------
    stream = pyarrow_stream()
    if stream is None and os.path.exists(OUT):
        old = json.load(open(OUT))
        stream = [c for c in old["hybrid_rle_decode"] if "pyarrow_integration" in c["src"]][0]["data"]
    if stream is None:
        sys.exit("reference not found and no previous JSON to take the pyarrow vector from")
    kats["hybrid_rle_decode"] = [
        {"src": "src/encoding/hybrid_rle/mod.rs:139-221 pyarrow_integration", "data": stream, "num_bits": 10,
         "num_values": 1000, "expected_range": 1000},
        {"src": "src/encoding/hybrid_rle/mod.rs:223-235 small", "data": [3, 2], "num_bits": 3, "num_values": 1,
         "expected": [2]},
        {"src": "src/encoding/hybrid_rle/mod.rs:237-249 zero_bit_width", "data": [3], "num_bits": 0, "num_values": 2,
         "expected": [0, 0]},
        {"src": "src/encoding/hybrid_rle/mod.rs:251-263 empty_values", "data": [], "num_bits": 1, "num_values": 100,
         "expected": [0] * 100},
    ]
    kats["hybrid_rle_round_trip"] = {
        "src": "src/encoding/hybrid_rle/mod.rs:122-137 roundtrip", "num_bits": 10, "range": 1000,
    }

    # ---- src/encoding/hybrid_rle/encoder.rs:100-167 --------------------
    kats["encoder"] = [
        {"src": "src/encoding/hybrid_rle/encoder.rs:100-111 bool_basics_1", "kind": "bool_bitmap",
         "bitmap": [0b10011101, 0b10011101], "offset": 0, "len": 14,
         "expected": [(2 << 1) | 1, 0b10011101, 0b00011101]},
        {"src": "src/encoding/hybrid_rle/encoder.rs:113-124 bool_from_iter", "kind": "bool",
         "values": [1] * 8, "expected": [(1 << 1) | 1, 0b11111111]},
        {"src": "src/encoding/hybrid_rle/encoder.rs:126-137 test_encode_u32", "kind": "u32", "num_bits": 2,
         "values": [0, 1, 2, 1, 2, 1, 1, 0, 3], "expected": [(2 << 1) | 1, 0b01100100, 0b00010110, 0b00000011]},
        {"src": "src/encoding/hybrid_rle/encoder.rs:139-155 test_encode_u32_large", "kind": "u32", "num_bits": 2,
         "values": [x % 4 for x in range(128)], "expected": [((128 // 8) << 1) | 1] + [0b11100100] * 32},
        {"src": "src/encoding/hybrid_rle/encoder.rs:157-167 test_u32_other", "kind": "u32", "num_bits": 2,
         "values": [3, 3, 0, 3, 2, 3, 3, 3, 3, 1, 3, 3, 3, 0, 3], "expected": [5, 207, 254, 247, 51]},
    ]

    # ---- src/deserialize/utils.rs:156-176 ------------------------------
    kats["slice_filtered_iter"] = {
        "src": "src/deserialize/utils.rs:156-176 basic", "iter_range_inclusive": [0, 100],
        "intervals": [[0, 2], [20, 11], [31, 1]],
    }

    # ---- tests/it/read/deserialize.rs:4-312 ----------------------------
    B, R, S = "bitmap", "repeated", "skipped"
    kats["filtered_runs"] = [
        {"src": "tests/it/read/deserialize.rs:4-23 bitmap_incomplete",
         "runs": [[B, [0b01000011], 7]], "intervals": [[1, 2]],
         "expected": [[S, 1], [B, [0b01000011], 1, 2]]},
        {"src": "tests/it/read/deserialize.rs:25-41 bitmap_complete",
         "runs": [[B, [0b01000011], 8]], "intervals": [[0, 8]],
         "expected": [[B, [0b01000011], 0, 8]]},
        {"src": "tests/it/read/deserialize.rs:43-70 bitmap_interval_incomplete",
         "runs": [[B, [0b01000011], 8], [B, [0b11111111], 8]], "intervals": [[0, 10]],
         "expected": [[B, [0b01000011], 0, 8], [B, [0b11111111], 0, 2]]},
        {"src": "tests/it/read/deserialize.rs:72-105 bitmap_interval_run_incomplete",
         "runs": [[B, [0b01100011], 8], [B, [0b11111111], 8]], "intervals": [[0, 5], [7, 4]],
         "expected": [[B, [0b01100011], 0, 5], [S, 2], [B, [0b01100011], 7, 1], [B, [0b11111111], 0, 3]]},
        {"src": "tests/it/read/deserialize.rs:107-131 bitmap_interval_run_skipped",
         "runs": [[B, [0b01100011], 8], [B, [0b11111111], 8]], "intervals": [[9, 2]],
         "expected": [[S, 4], [S, 1], [B, [0b11111111], 1, 2]]},
        {"src": "tests/it/read/deserialize.rs:133-162 bitmap_interval_run_offset_skipped",
         "runs": [[B, [0b01100011], 8], [B, [0b11111111], 8]], "intervals": [[0, 1], [9, 2]],
         "expected": [[B, [0b01100011], 0, 1], [S, 3], [S, 1], [B, [0b11111111], 1, 2]]},
        {"src": "tests/it/read/deserialize.rs:164-182 repeated_incomplete",
         "runs": [[R, True, 7]], "intervals": [[1, 2]],
         "expected": [[S, 1], [R, True, 2]]},
        {"src": "tests/it/read/deserialize.rs:184-199 repeated_complete",
         "runs": [[R, True, 8]], "intervals": [[0, 8]],
         "expected": [[R, True, 8]]},
        {"src": "tests/it/read/deserialize.rs:201-226 repeated_interval_incomplete",
         "runs": [[R, True, 8], [R, False, 8]], "intervals": [[0, 10]],
         "expected": [[R, True, 8], [R, False, 2]]},
        {"src": "tests/it/read/deserialize.rs:228-258 repeated_interval_run_incomplete",
         "runs": [[R, True, 8], [R, False, 8]], "intervals": [[0, 5], [7, 4]],
         "expected": [[R, True, 5], [S, 2], [R, True, 1], [R, False, 3]]},
        {"src": "tests/it/read/deserialize.rs:260-283 repeated_interval_run_skipped",
         "runs": [[R, True, 8], [R, False, 8]], "intervals": [[9, 2]],
         "expected": [[S, 8], [S, 0], [R, False, 2]]},
        {"src": "tests/it/read/deserialize.rs:285-312 repeated_interval_run_offset_skipped",
         "runs": [[R, True, 8], [R, False, 8]], "intervals": [[0, 1], [9, 2]],
         "expected": [[R, True, 1], [S, 7], [S, 0], [R, False, 2]]},
    ]

    # ---- zigzag_leb128 and delta_bitpacked (the u64 unpack's caller) --------------------------------------
    kats["zigzag_leb128_decode"] = {
        "src": "src/encoding/zigzag_leb128.rs:23-43 test_decode",
        "cases": [[0, 0], [1, -1], [2, 1], [3, -2], [4, 2], [5, -3], [6, 3], [7, -4], [8, 4], [9, -5]],
    }
    kats["delta_bitpacked_decode"] = [
        {"src": "src/encoding/delta_bitpacked/decoder.rs:250-266 single_value",
         "data": [128, 1, 4, 1, 2], "expected": [1], "consumed": 5},
        {"src": "src/encoding/delta_bitpacked/decoder.rs:268-288 test_from_spec",
         "data": [128, 1, 4, 5, 2, 2, 0, 0, 0, 0], "expected": [1, 2, 3, 4, 5], "consumed": 10},
        {"src": "src/encoding/delta_bitpacked/decoder.rs:290-318 case2",
         "data": [128, 1, 4, 6, 2, 7, 3, 0, 0, 0, 0b01101101, 0b00001011, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3],
         "expected": [1, 2, 3, 4, 5, 1], "consumed": 22},
        {"src": "src/encoding/delta_bitpacked/decoder.rs:320-364 multiple_miniblocks",
         "data": [128, 1, 4, 65, 100, 7, 3, 4, 255, 0] + [0] * 12 + [0x88] * 16 + [1, 2, 3],
         "expected": [50] + [46 - 4 * i for i in range(32)] + [-74 + 4 * i for i in range(32)],
         "consumed": 38},
    ]

    kats["zigzag_leb128_encode"] = {
        "src": "src/encoding/zigzag_leb128.rs:45-64 test_encode",
        "cases": [[0, 0], [1, -1], [2, 1], [3, -2], [4, 2], [5, -3], [6, 3], [7, -4], [8, 4], [9, -5]],
    }
    kats["delta_bitpacked_encode"] = [
        {"src": "src/encoding/delta_bitpacked/encoder.rs:84-100 constant_delta",
         "data": [1, 2, 3, 4, 5], "expected": [128, 1, 1, 5, 2, 2, 0]},
        {"src": "src/encoding/delta_bitpacked/encoder.rs:102-124 negative_min_delta",
         "data": [1, 2, 3, 4, 5, 1],
         "expected": [128, 1, 1, 6, 2, 7, 3, 0b01101101, 0b00001011] + [0] * (128 * 3 // 8 - 2)},
    ]
    # encode -> Decoder gives the data back (src/encoding/delta_bitpacked/mod.rs:13-90); test_another also
    # checks consumed_bytes() == buffer.len()
    kats["delta_bitpacked_round_trip"] = [
        {"src": "mod.rs:13-24 basic", "data": [1, 3, 1, 2, 3]},
        {"src": "mod.rs:26-37 negative_value", "data": [1, 3, -1, 2, 3]},
        {"src": "mod.rs:39-58 some",
         "data": [-2147483648, -1777158217, -984917788, -1533539476, -731221386, -1322398478, 906736096]},
        {"src": "mod.rs:60-74 more_than_one_block", "data": [1, 3, -1, 2, 3, 10, 1] + [x - 10 for x in range(128)]},
        {"src": "mod.rs:76-90 test_another", "data": [2, 3, 1, 2, 1], "consumed_is_len": True},
    ]

    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
        f.write("\n")
    print("wrote", OUT, "with", len(kats), "groups;", "pyarrow stream bytes:", len(stream))


if __name__ == "__main__":
    main()
