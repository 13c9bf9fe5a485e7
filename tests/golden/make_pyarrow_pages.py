"""Writes tests/golden/pyarrow_pages.json: decompressed data pages cut out of Parquet files written by pyarrow
(its C++ writer is an independent implementation of the RLE / bit-packing hybrid), with the rows pyarrow itself
reads back from the same files as the expected result.

    python tests/golden/make_pyarrow_pages.py

Each page entry holds the page buffer exactly as a reader sees it after decompression (the files are written
uncompressed), the header fields a reader needs (kind V1/V2, num_values, encoding, level byte lengths), the column's
maximum levels, and - as the expectation - the definition and repetition levels computed from the written rows by
the Dremel rules, plus the non-null leaf values in order.  Dictionary pages are stored as their PLAIN values.

The only Parquet parsing done here is the Thrift compact protocol of the page headers (to find where each page
starts and ends); no level or index stream is decoded by this script.
"""
import base64
import io
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- Thrift compact protocol, as far as page headers need it ---------------------------------------------------

class Compact:
    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos

    def byte(self):
        b = self.buf[self.pos]
        self.pos += 1
        return b

    def varint(self):
        shift = value = 0
        while True:
            b = self.byte()
            value |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return value

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def value(self, typ):
        if typ in (1, 2):          # bool true / false, carried by the field header
            return typ == 1
        if typ == 3:
            return self.byte()
        if typ in (4, 5, 6):       # i16, i32, i64
            return self.zigzag()
        if typ == 7:
            self.pos += 8
            return None
        if typ == 8:               # binary
            n = self.varint()
            self.pos += n
            return None
        if typ in (9, 10):         # list, set
            head = self.byte()
            n, et = head >> 4, head & 0xF
            if n == 15:
                n = self.varint()
            return [self.value(et) if et not in (1, 2) else bool(self.byte() == 1) for _ in range(n)]
        if typ == 12:
            return self.struct()
        raise ValueError("thrift compact type %d" % typ)

    def struct(self):
        fields, last = {}, 0
        while True:
            head = self.byte()
            if head == 0:
                return fields
            delta, typ = head >> 4, head & 0xF
            fid = last + delta if delta else self.zigzag()
            last = fid
            fields[fid] = self.value(typ)


ENCODINGS = {0: "Plain", 2: "PlainDictionary", 3: "Rle", 8: "RleDictionary"}


def pages_of_chunk(buf, col):
    """Yields (header fields, payload bytes) of every page of a column chunk (format: parquet.thrift PageHeader)."""
    pos = col.dictionary_page_offset if col.has_dictionary_page else col.data_page_offset
    end = pos + col.total_compressed_size
    while pos < end:
        c = Compact(buf, pos)
        h = c.struct()
        size = h[3]
        assert h[2] == size, "pages must be uncompressed"
        yield h, bytes(buf[c.pos:c.pos + size])
        pos = c.pos + size


# ---- Dremel levels of the written rows -------------------------------------------------------------------------

def flat_levels(rows):
    """optional leaf: def 1 where set."""
    return [0 if r is None else 1 for r in rows], None, [r for r in rows if r is not None]


def list_levels(rows):
    """optional list<optional leaf> (max_def 3, max_rep 1): null list 0, empty list 1, null item 2, item 3; the
    first entry of a row has rep 0, the others rep 1."""
    d, r, leaves = [], [], []
    for row in rows:
        if row is None:
            d.append(0), r.append(0)
        elif len(row) == 0:
            d.append(1), r.append(0)
        else:
            for k, item in enumerate(row):
                r.append(0 if k == 0 else 1)
                d.append(2 if item is None else 3)
                if item is not None:
                    leaves.append(item)
    return d, r, leaves


def struct_levels(rows):
    """optional struct<optional leaf> (max_def 2): null struct 0, null leaf 1, leaf 2."""
    d, leaves = [], []
    for row in rows:
        if row is None:
            d.append(0)
        elif row["x"] is None:
            d.append(1)
        else:
            d.append(2)
            leaves.append(row["x"])
    return d, None, leaves


# ---- the columns -----------------------------------------------------------------------------------------------

def columns(rng):
    n = 3000
    # long runs of nulls and of repeated keys, so that both run kinds appear in levels and indices
    key = np.repeat(rng.integers(0, 700, n), rng.integers(1, 16, n))[:n]
    null_runs = np.repeat(rng.random(n) < 0.2, rng.integers(1, 32, n))[:n]
    ints = [None if null_runs[i] else int(key[i]) * 1001 for i in range(n)]
    few = [None if rng.random() < 0.05 else int(rng.integers(0, 3)) for _ in range(n)]
    required = [int(x) for x in rng.integers(0, 5000, n)]
    lists, structs = [], []
    for i in range(n // 3):
        u = rng.random()
        if u < 0.1:
            lists.append(None)
        elif u < 0.2:
            lists.append([])
        else:
            lists.append([None if rng.random() < 0.15 else int(rng.integers(0, 40)) for _ in
                          range(int(rng.integers(1, 9)))])
        u = rng.random()
        structs.append(None if u < 0.1 else {"x": None if u < 0.3 else int(rng.integers(0, 90))})
    strings = [None if null_runs[i] else "k%03d" % (key[i] % 50) for i in range(n)]
    return [
        ("opt_i64", pa.array(ints, pa.int64()), flat_levels(ints), 1, 0, "i8"),
        ("opt_i32_few", pa.array(few, pa.int32()), flat_levels(few), 1, 0, "i4"),
        ("req_i32", pa.array(required, pa.int32()), ([], None, required), 0, 0, "i4"),
        ("list_i32", pa.array(lists, pa.list_(pa.int32())), list_levels(lists), 3, 1, "i4"),
        ("struct_i32", pa.array(structs, pa.struct([("x", pa.int32())])), struct_levels(structs), 2, 0, "i4"),
        ("opt_str", pa.array(strings, pa.string()), flat_levels(strings), 1, 0, "str"),
    ]


def main():
    rng = np.random.default_rng(20240611)
    out = {"pyarrow": pa.__version__, "pages": []}
    for name, arr, (defs, reps, leaves), max_def, max_rep, kind in columns(rng):
        for version in ("1.0", "2.0"):
            for use_dict in (True, False):
                if not use_dict and name in ("opt_str", "opt_i32_few"):
                    continue       # these two are kept for their index streams
                schema = pa.schema([pa.field(name, arr.type, nullable=max_def > 0)])
                sink = io.BytesIO()
                pq.write_table(pa.Table.from_arrays([arr], schema=schema), sink, compression="NONE",
                               use_dictionary=use_dict, data_page_version=version, data_page_size=2048,
                               write_statistics=False, row_group_size=1 << 20)
                buf = sink.getvalue()
                md = pq.ParquetFile(io.BytesIO(buf)).metadata
                assert md.num_row_groups == 1
                col = md.row_group(0).column(0)
                # what pyarrow reads back is what was written
                back = pq.read_table(io.BytesIO(buf)).column(0).to_pylist()
                assert back == arr.to_pylist()
                dictionary, at_level, at_leaf = None, 0, 0
                for h, payload in pages_of_chunk(buf, col):
                    if 7 in h:     # dictionary page: PLAIN values
                        assert ENCODINGS[h[7][2]] in ("Plain", "PlainDictionary")
                        dictionary = base64.b64encode(payload).decode()
                        continue
                    v2 = 8 in h
                    ph = h[8] if v2 else h[5]
                    num_values = ph[1]
                    entry = {
                        "column": name, "writer": {"data_page_version": version, "use_dictionary": use_dict},
                        "kind": kind, "v2": v2, "num_values": num_values,
                        "encoding": ENCODINGS[ph[4] if v2 else ph[2]],
                        "max_def": max_def, "max_rep": max_rep,
                        "buffer": base64.b64encode(payload).decode(), "dictionary": dictionary,
                    }
                    if v2:
                        entry["def_len"], entry["rep_len"] = ph[5], ph[6]
                    n_levels = num_values
                    if max_def > 0:
                        page_defs = defs[at_level:at_level + n_levels]
                        entry["def_levels"] = "".join(map(str, page_defs))      # one digit per level
                        n_leaves = sum(1 for x in page_defs if x == max_def)
                    else:
                        n_leaves = num_values
                    if max_rep > 0:
                        entry["rep_levels"] = "".join(map(str, reps[at_level:at_level + n_levels]))
                    entry["leaves"] = leaves[at_leaf:at_leaf + n_leaves]
                    at_level += n_levels
                    at_leaf += n_leaves
                    out["pages"].append(entry)
                assert at_leaf == len(leaves) and (max_def == 0 or at_level == len(defs)), (name, at_level, at_leaf)
    path = os.path.join(HERE, "pyarrow_pages.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    enc = {}
    for p in out["pages"]:
        enc[p["encoding"]] = enc.get(p["encoding"], 0) + 1
    print(path, os.path.getsize(path), "bytes,", len(out["pages"]), "pages", enc)


if __name__ == "__main__":
    main()
