"""Parity of pq2_decode_raw_pages_v1_device (split_buffer_v1 + dict_indices_decoder on the GPU, then the decode of the
page's level and index streams) against the host functions of the same library -- themselves checked against the
reference's behaviour in tests/test_host_logic.py -- and the CPU oracle's values."""
import struct

import numpy as np
import pytest

from parquet2_amd import RAW_DICT_INDICES, RAW_PAGE_DTYPE, synth
from parquet2_amd._lib import INVALID_PARAMETER, OK, OUT_OF_SPEC, REFERENCE_WOULD_PANIC
from parquet2_amd.encoding.hybrid_rle import encode_u32
from parquet2_amd.read.levels import get_bit_width

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["per_tile", "per_page", "ring", "scout"])
def ctx(request):
    import os

    from parquet2_amd import Context

    os.environ["PQ2_MODE"] = {"per_tile": "tile", "per_page": "page", "ring": "ring", "scout": "scout"}[request.param]
    try:
        c = Context(0)
    finally:
        del os.environ["PQ2_MODE"]
    yield c
    c.close()


def v1_page(rep_stream, def_stream, values):
    """The buffer of a V1 data page: length-prefixed level streams, then the values (page/mod.rs:350-398)."""
    out = b""
    for s in (rep_stream, def_stream):
        if s is not None:
            out += struct.pack("<I", len(s)) + bytes(s)
    return out + bytes(values)


def run(ctx, buffers, rows, max_index_bits=0, pad=3):
    """buffers: [(bytes, num_values, max_rep, max_def, flags)]."""
    import torch

    chunks, at, out_off = [], 0, 0
    pages = np.zeros(len(buffers), dtype=RAW_PAGE_DTYPE)
    for i, (buf, n, max_rep, max_def, flags) in enumerate(buffers):
        chunks.append(np.frombuffer(buf, dtype=np.uint8))
        pages[i] = (at, len(buf), n, max_rep, max_def, flags, out_off)
        at += len(buf)
        chunks.append(np.zeros(pad, dtype=np.uint8))  # page buffers at unrelated alignments
        at += pad
        out_off += n + (i % 2)
    data = np.concatenate(chunks + [np.zeros(16, dtype=np.uint8)])
    d_buf = torch.from_numpy(data).cuda()
    outs = [torch.full((max(out_off, 1),), 0x5A5A5A5A, dtype=torch.int32, device="cuda") for _ in range(3)]
    status = ctx.decode_raw_pages_v1_device(d_buf.data_ptr(), d_buf.numel(), pages, max_index_bits,
                                            outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), out_off,
                                            torch.cuda.current_stream().cuda_stream)
    return [o.cpu().numpy().view(np.uint32) for o in outs], status, pages


def test_pages_of_every_shape(ctx, oracle):
    rng = np.random.default_rng(21)
    buffers, expect = [], []
    for p in range(70):  # enough streams for the per-page schedule, with the per-tile one forced as well
        n = int(rng.integers(1, 30000))
        max_rep = int(rng.choice([0, 0, 1, 3]))
        max_def = int(rng.choice([0, 1, 1, 2, 5]))
        dict_page = p % 5 != 4
        rep = dfl = None
        rep_v = def_v = idx_v = None
        if max_rep:
            rep_v = np.repeat(rng.integers(0, max_rep + 1, size=n // 5 + 1), 5)[:n].astype(np.uint32)
            rep = encode_u32(rep_v, get_bit_width(max_rep))
        if max_def == 1:
            s, def_v = synth.def_levels_page(synth.seed_for(3, 200 + p), n)
            dfl = s.tobytes()
        elif max_def:
            def_v = np.repeat(rng.integers(0, max_def + 1, size=n // 9 + 1), 9)[:n].astype(np.uint32)
            dfl = encode_u32(def_v, get_bit_width(max_def))
        if dict_page:
            w = int(rng.integers(0, 13))
            if w:
                s, idx_v = synth.dict_page(synth.seed_for(2, 200 + p), n, w, 9.0, 30.0)
                values = bytes([w]) + s.tobytes()
            else:
                idx_v, values = np.zeros(n, dtype=np.uint32), bytes([0])
        else:
            values = rng.integers(0, 256, size=int(rng.integers(0, 64)), dtype=np.uint8).tobytes()  # PLAIN: not ours
        buffers.append((v1_page(rep, dfl, values), n, max_rep, max_def, RAW_DICT_INDICES if dict_page else 0))
        expect.append((rep_v, def_v, idx_v))
    outs, status, pages = run(ctx, buffers, None, max_index_bits=12)
    assert (status == OK).all()
    written = [np.zeros(len(outs[0]), dtype=bool) for _ in range(3)]
    for pg, exp in zip(pages, expect):
        o, n = int(pg["out_off"]), int(pg["num_values"])
        for k in range(3):
            if exp[k] is not None:
                assert (outs[k][o:o + n] == exp[k]).all(), (k, pg)
                written[k][o:o + n] = True
    for k in range(3):
        assert (outs[k][~written[k]] == 0x5A5A5A5A).all()  # absent streams write nothing


def test_split_errors_and_widths(ctx):
    n = 100
    levels = encode_u32(np.ones(n, dtype=np.uint32), 1)
    idx = bytes([3]) + bytes(encode_u32(np.arange(n, dtype=np.uint32) % 8, 3))
    good = v1_page(None, levels, idx)
    cases = [
        (good, OK),
        (struct.pack("<I", len(levels) + 1000) + bytes(levels) + idx, OUT_OF_SPEC),   # def length beyond the page
        (b"\x01\x02", OUT_OF_SPEC),                                                    # no room for the length prefix
        (v1_page(None, levels, b""), REFERENCE_WOULD_PANIC),                          # dictionary page without values
        (v1_page(None, levels, bytes([33]) + idx[1:]), OUT_OF_SPEC),                  # index width above 32
        (v1_page(None, levels, bytes([9]) + idx[1:]), INVALID_PARAMETER),             # above the promised bound
        (good, OK),
    ]
    outs, status, pages = run(ctx, [(b, n, 0, 1, RAW_DICT_INDICES) for b, _ in cases], None, max_index_bits=8)
    assert list(status) == [c for _, c in cases]
    for i in (0, 6):
        o = int(pages[i]["out_off"])
        assert (outs[1][o:o + n] == 1).all() and (outs[2][o:o + n] == np.arange(n) % 8).all()
