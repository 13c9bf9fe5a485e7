"""pq2_decode_batch_device_noalloc (the batch decode that allocates nothing: the device memory its kernel needs is a
fixed array of the library's code object) through the C ABI: against the oracle and against pq2_decode_batch_device
on the same batches, bit for bit.  Two calls in a row on the same GPU, the second with a smaller batch; the call
recorded in a HIP graph on a side stream and replayed (an allocation inside it would end the capture with an
error); the first call of a process that has never created a context; the limits."""
import os
import subprocess
import sys

import numpy as np
import pytest

from parquet2_amd import synth
from parquet2_amd._lib import (INVALID_PARAMETER, OK, REFERENCE_WOULD_PANIC, decode_batch_device_noalloc)

pytestmark = pytest.mark.gpu


def make_batch(n_pages, seed, widths=(1, 3, 7, 12, 16, 17, 31, 32)):
    rng = np.random.default_rng(seed)
    batch = synth.PageBatch(align=1)
    pages = []
    for p in range(n_pages):
        w = int(widths[p % len(widths)])
        n = int(rng.integers(1, 60000))
        if p % 11 == 3:
            stream, _ = synth.adversarial(seed * 1000 + p, n, w)
        else:
            stream, _ = synth.dict_page(seed * 1000 + p, n, w, 25.0, 70.0)
        # some pages ask for more values than the stream holds (zero padding), some for fewer
        n_ask = n + int(rng.integers(0, 500)) if p % 3 == 0 else max(1, n - int(rng.integers(0, 300)))
        batch.add(stream, w, n_ask, out_align=int(rng.integers(1, 5)))
        pages.append((bytes(stream), w, n_ask))
    for stream, w, n_ask in ((bytes([0b00010000, 1]), 16, 40),  # an RLE run without its second value byte: a panic
                             (bytes(3), 0, 1000),               # bit width 0: zeros
                             (b"", 9, 777)):                     # empty stream: zeros
        batch.add(np.frombuffer(stream, np.uint8), w, n_ask)
        pages.append((stream, w, n_ask))
    return batch.finish() + (pages,)


def check_against_oracle(oracle, pages, descs, got, statuses):
    """Every page of the batch: status and values as the CPU restatement of the reference gives them."""
    for (stream, w, n_ask), d, st in zip(pages, descs, statuses):
        want_st, want = oracle.hybrid_rle_decode(stream, w, n_ask)
        assert st == want_st, (w, n_ask, st, want_st)
        if want_st == OK:
            o = int(d["out_off"])
            assert np.array_equal(got[o:o + n_ask], want), (w, n_ask)


def noalloc(data, descs, out_values, max_bits, stream=None):
    import torch

    d_in = torch.from_numpy(data).cuda()
    got = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).cuda()
    d_status = torch.full((len(descs),), -1, dtype=torch.int32, device="cuda")
    decode_batch_device_noalloc(0, d_in.data_ptr(), d_in.numel() - 16, d_descs.data_ptr(), len(descs), max_bits,
                                got.data_ptr(), out_values, d_status.data_ptr(),
                                torch.cuda.current_stream().cuda_stream if stream is None else stream)
    torch.cuda.synchronize()
    return got, d_status.cpu().numpy()


def test_noalloc_matches_the_oracle_and_the_allocating_entry_point(oracle):
    import torch

    from parquet2_amd import Context

    ctx = Context(0)
    try:
        # first call: 300 pages; second call, right behind it on the same GPU: 37 pages of other shapes
        for n_pages, seed, max_bits in ((300, 1, 0), (37, 2, 32), (70, 3, 16)):
            widths = (1, 3, 7, 12, 16) if max_bits == 16 else (1, 3, 7, 12, 16, 17, 31, 32)
            data, descs, out_values, pages = make_batch(n_pages, seed, widths)
            got, st_got = noalloc(data, descs, out_values, max_bits)
            check_against_oracle(oracle, pages, descs, got.cpu().numpy().view(np.uint32), st_got.tolist())
            assert REFERENCE_WOULD_PANIC in st_got.tolist() and (st_got == OK).sum() == len(descs) - 1
            d_in = torch.from_numpy(data).cuda()
            want = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            st_want = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel() - 16, descs, want.data_ptr(), out_values,
                                              torch.cuda.current_stream().cuda_stream)
            assert st_want.tolist() == st_got.tolist()
            assert torch.equal(want, got), "the two entry points differ on a batch of %d pages" % n_pages
    finally:
        ctx.close()


def test_noalloc_inside_a_graph_capture(oracle):
    """The call recorded on a side stream under capture, then replayed twice: the second replay over outputs that
    were overwritten in between.  hipMalloc and its like are not permitted while a stream captures."""
    import torch

    data, descs, out_values, pages = make_batch(90, 5, (1, 3, 7, 12, 16))
    d_in = torch.from_numpy(data).cuda()
    got = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).cuda()
    d_status = torch.full((len(descs),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        decode_batch_device_noalloc(0, d_in.data_ptr(), d_in.numel() - 16, d_descs.data_ptr(), len(descs), 16,
                                    got.data_ptr(), out_values, d_status.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_status[0]) == -1, "recording a graph must not run it"
    for _ in range(2):
        got.fill_(0x77777777)
        d_status.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        check_against_oracle(oracle, pages, descs, got.cpu().numpy().view(np.uint32), d_status.cpu().tolist())


FIRST_CALL = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from parquet2_amd import synth
from parquet2_amd._lib import decode_batch_device_noalloc
from _oracle import oracle as get_oracle
stream, _ = synth.dict_page(11, 30000, 12)
batch = synth.PageBatch()
for _ in range(3):
    batch.add(stream, 12, 30000)
data, descs, out_values = batch.finish()
d_in = torch.from_numpy(data).cuda()
d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).cuda()
d_out = torch.zeros(out_values, dtype=torch.int32, device="cuda")
d_status = torch.full((3,), -1, dtype=torch.int32, device="cuda")
# no Context has been created in this process: the entry point finds its array itself
decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, d_descs.data_ptr(), 3, 12, d_out.data_ptr(), out_values,
                            d_status.data_ptr())
torch.cuda.synchronize()
st, want = get_oracle().hybrid_rle_decode(bytes(stream), 12, 30000)
got = d_out.cpu().numpy().view(np.uint32)
assert d_status.cpu().tolist() == [0, 0, 0] and st == 0
for d in descs:
    o = int(d["out_off"])
    assert np.array_equal(got[o:o + 30000], want)
print("first call ok")
"""


def test_noalloc_first_call_of_a_process_without_a_context():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", FIRST_CALL, root], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "first call ok" in r.stdout, r.stdout + r.stderr


def test_noalloc_descriptor_errors_and_limits():
    import torch

    from parquet2_amd.error import InvalidParameter, WouldOverAllocate

    stream, vals = synth.dict_page(7, 5000, 12)
    batch = synth.PageBatch()
    for _ in range(4):
        batch.add(stream, 12, 5000)
    data, descs, out_values = batch.finish()
    descs = descs.copy()
    descs[1]["in_off"] = data.size  # stream outside the input buffer: this page fails, the others decode
    descs[2]["num_bits"] = 20       # wider than the bound given with the call
    d_in = torch.from_numpy(data).cuda()
    d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).cuda()
    d_out = torch.zeros(out_values, dtype=torch.int32, device="cuda")
    d_status = torch.full((4,), -1, dtype=torch.int32, device="cuda")
    decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, d_descs.data_ptr(), 4, 12, d_out.data_ptr(),
                                out_values, d_status.data_ptr())
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [OK, INVALID_PARAMETER, INVALID_PARAMETER, OK]
    got = d_out.cpu().numpy().view(np.uint32)
    for i in (0, 3):
        o = int(descs[i]["out_off"])
        assert (got[o:o + 5000] == vals).all()
    # beyond the fixed array: the call says so and enqueues nothing
    with pytest.raises(WouldOverAllocate):
        decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, d_descs.data_ptr(), 4097, 12, d_out.data_ptr(),
                                    out_values, d_status.data_ptr())
    with pytest.raises(InvalidParameter):
        decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, 0, 4, 12, d_out.data_ptr(), out_values,
                                    d_status.data_ptr())


def test_noalloc_pages_of_a_million_values(oracle):
    """Nothing the batch needs grows with its pages: 40 pages of 2^20 values (more header-list entries than the
    entry point's first version had room for)."""
    import torch

    n = 1 << 20
    distinct = [synth.dict_page(synth.seed_for(2, p), n, 12) for p in range(4)]
    batch = synth.PageBatch()
    for p in range(40):
        batch.add(distinct[p % 4][0], 12, n)
    data, descs, out_values = batch.finish()
    got, st = noalloc(data, descs, out_values, 12)
    assert st.tolist() == [OK] * 40
    got = got.cpu().numpy().view(np.uint32)
    wants = [oracle.hybrid_rle_decode(bytes(s), 12, n)[1] for s, _ in distinct]
    for p, d in enumerate(descs):
        o = int(d["out_off"])
        assert np.array_equal(got[o:o + n], wants[p % 4])
