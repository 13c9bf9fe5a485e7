"""Readers of tests/golden/pyarrow_pages.json (pages written by pyarrow; made by tests/golden/make_pyarrow_pages.py)."""
import base64
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_pages():
    with open(os.path.join(HERE, "golden", "pyarrow_pages.json")) as f:
        return json.load(f)["pages"]


def leaves_of(page, decoded_indices=None, plain=None):
    """The leaf values of a page as Python objects, from dictionary + indices or from PLAIN values."""
    kind = page["kind"]
    if decoded_indices is not None:
        raw = base64.b64decode(page["dictionary"])
        if kind == "str":
            table, pos = [], 0
            while pos < len(raw):
                n = int.from_bytes(raw[pos:pos + 4], "little")
                table.append(raw[pos + 4:pos + 4 + n].decode())
                pos += 4 + n
        else:
            table = np.frombuffer(raw, dtype="<" + kind).tolist()
        return [table[i] for i in decoded_indices]
    return np.frombuffer(plain, dtype="<" + kind).tolist()


def bit_width(max_level):
    return int(max_level).bit_length()
