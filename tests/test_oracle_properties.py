"""The CPU oracle beyond the reference's own known-answer vectors: against a second, independent model of the
hybrid-RLE decoder written from the cited reference lines (pure Python, value at a time), and against pages
written by pyarrow's C++ Parquet writer (tests/golden/pyarrow_pages.json, made by tests/golden/make_pyarrow_pages.py)
whose expected levels and values come from the rows that were written, not from any decoder of this repository.

The pyarrow pages pin the streams a conforming writer produces (RLE_DICTIONARY index streams with their width
byte, definition and repetition levels of V1 and V2 pages).  The reference-specific behaviour on streams no writer
produces (unmasked RLE values, the dropped header at the end, panics) stays unpinned: there the two models agree
with each other and with the cited code only."""
import base64

import numpy as np
import pytest

from _golden_pages import bit_width, leaves_of, load_pages
from _oracle import OK, REFERENCE_WOULD_PANIC


# ---- an independent model --------------------------------------------------------------------------------------

class Panic(Exception):
    pass


def model_decode(stream, w, n):
    """HybridRleDecoder::try_new(stream, w, n).collect() (src/encoding/hybrid_rle/mod.rs:35-116 over
    decoder.rs:8-65 and bitpacked/decode.rs): values one at a time.

    Runs: uleb128 header (a header that ends the stream is dropped: decoder.rs:34-37); bit 0 set -> bit-packed,
    min(header >> 1 groups of w bytes, what is left) (decoder.rs:41-45); else RLE of header >> 1 copies of the next
    ceil8(w) bytes, read as a little-endian integer and not masked (mod.rs:66-76; fewer bytes left: slice panic,
    decoder.rs:48-52).  Runs that yield nothing are skipped (mod.rs:55-83); a bit-packed run yields
    min(bytes * 8 / w, values still wanted) values, whole packs of 32 with the last one zero-extended
    (bitpacked/decode.rs:32-63); behind the last run the values are zeros (mod.rs:100-108)."""
    rle_bytes = (w + 7) // 8
    if w > 32:
        raise Panic
    pos = 0

    def read_next(remaining):
        """One run as a list of values, or None when the stream has ended (mod.rs:41-64)."""
        nonlocal pos
        if w == 0 or pos >= len(stream):
            return None
        header = shift = 0
        while pos < len(stream):                      # uleb128::decode ends with the slice (uleb128.rs:3-24)
            b = stream[pos]
            pos += 1
            if shift == 63 and b > 1:
                raise Panic                           # uleb128.rs:11-13
            header |= (b & 0x7F) << shift
            if not b & 0x80:
                break
            shift += 7
        if pos >= len(stream):
            return None                               # the header was the last byte: no run
        if header & 1:
            nbytes = min((header >> 1) * w, len(stream) - pos)
            run = stream[pos:pos + nbytes]
            pos += nbytes
            acc = int.from_bytes(run, "little")
            return [(acc >> (i * w)) & ((1 << w) - 1) for i in range(min(nbytes * 8 // w, remaining))]
        if len(stream) - pos < rle_bytes:
            raise Panic
        value = int.from_bytes(stream[pos:pos + rle_bytes], "little") & 0xFFFFFFFF
        pos += rle_bytes
        return [value] * min(header >> 1, remaining)

    out = []
    state = read_next(n)                              # try_new reads the first run whatever n is (mod.rs:68-78)
    while len(out) < n:
        if state is None:
            out.extend([0] * (n - len(out)))
        elif state:
            out.extend(state)
            state = []
        else:
            state = read_next(n - len(out))
    return out


def uleb(v):
    out = bytearray()
    while True:
        b, v = v & 0x7F, v >> 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def random_stream(rng, w, conforming):
    parts, values = [], []
    for _ in range(int(rng.integers(1, 30))):
        kind = rng.random()
        if kind < 0.45:
            groups = int(rng.integers(0 if not conforming else 1, 6))
            vals = [int(x) for x in rng.integers(0, 1 << w, groups * 8, dtype=np.uint64)]
            acc = sum(v << (i * w) for i, v in enumerate(vals))
            parts.append(uleb((groups << 1) | 1) + acc.to_bytes(groups * w, "little"))
            values += vals
        else:
            count = int(rng.integers(0 if not conforming else 1, 70))
            top = 8 * ((w + 7) // 8) if not conforming else w
            v = int(rng.integers(0, 1 << top, dtype=np.uint64))
            parts.append(uleb(count << 1) + v.to_bytes((w + 7) // 8, "little"))
            values += [v] * count
    return b"".join(parts), values


@pytest.mark.parametrize("w", list(range(1, 33)))
def test_oracle_against_the_model(oracle, w):
    rng = np.random.default_rng(1000 + w)
    for case in range(60):
        conforming = case % 2 == 0
        stream, values = random_stream(rng, w, conforming)
        cuts = [len(stream)]
        if not conforming:
            cuts += [int(c) for c in rng.integers(0, len(stream) + 1, 3)]   # truncated anywhere
        for cut in cuts:
            s = stream[:cut]
            for n in {len(values), max(0, len(values) - 5), len(values) + 40, 1, 0}:
                try:
                    want = model_decode(s, w, n)
                except Panic:
                    want = None
                st, got = oracle.hybrid_rle_decode(np.frombuffer(s, dtype=np.uint8), w, n)
                if want is None:
                    assert st == REFERENCE_WOULD_PANIC, (w, case, cut, n)
                else:
                    assert st == OK, (w, case, cut, n, st)
                    assert got.tolist() == want, (w, case, cut, n)
                    if conforming and cut == len(stream):
                        assert want[:min(n, len(values))] == values[:min(n, len(values))]


def test_model_on_the_reference_vectors(kats):
    """The second model is itself pinned by the reference's vectors (mod.rs tests)."""
    for c in kats["hybrid_rle_decode"]:
        expected = list(range(c["expected_range"])) if "expected_range" in c else c["expected"]
        assert model_decode(bytes(c["data"]), c["num_bits"], c["num_values"]) == expected, c["src"]


# ---- pages written by pyarrow ----------------------------------------------------------------------------------

def test_fixture_covers_what_it_should():
    pages = load_pages()
    assert {p["encoding"] for p in pages} == {"Plain", "RleDictionary"}
    assert {p["v2"] for p in pages} == {False, True}
    assert {p["max_def"] for p in pages} == {0, 1, 2, 3} and any(p["max_rep"] == 1 for p in pages)


def test_oracle_on_pyarrow_pages(oracle):
    for i, page in enumerate(load_pages()):
        buf = np.frombuffer(base64.b64decode(page["buffer"]), dtype=np.uint8)
        n, max_def, max_rep = page["num_values"], page["max_def"], page["max_rep"]
        if page["v2"]:
            st, s = oracle.split_buffer_v2(buf.size, page["rep_len"], page["def_len"])
        else:
            st, s = oracle.split_buffer_v1(buf, max_rep > 0, max_def > 0)
        assert st == OK, i
        if max_rep:
            st, rep = oracle.hybrid_rle_decode(buf[s.rep_off:s.rep_off + s.rep_len], bit_width(max_rep), n)
            assert st == OK and "".join(map(str, rep.tolist())) == page["rep_levels"], (i, page["column"])
        n_leaves = n
        if max_def:
            assert oracle.get_bit_width(max_def) == bit_width(max_def)
            st, d = oracle.hybrid_rle_decode(buf[s.def_off:s.def_off + s.def_len], bit_width(max_def), n)
            assert st == OK and "".join(map(str, d.tolist())) == page["def_levels"], (i, page["column"])
            n_leaves = int((d == max_def).sum())
        values = buf[s.val_off:s.val_off + s.val_len]
        if page["encoding"] == "RleDictionary":
            st, bits, off = oracle.dict_indices_locate(values)
            assert st == OK and off == 1 and bits == values[0]
            st, idx = oracle.hybrid_rle_decode(values[off:], bits, n_leaves)
            assert st == OK
            assert leaves_of(page, decoded_indices=idx.tolist()) == page["leaves"], (i, page["column"])
            # the reference asks for num_values indices (utils.rs:13-27): the same ones, then zeros
            st, idx_n = oracle.hybrid_rle_decode(values[off:], bits, n)
            assert st == OK and (idx_n[:n_leaves] == idx).all()
        else:
            assert leaves_of(page, plain=values.tobytes()) == page["leaves"], (i, page["column"])
