"""pq2_decode_batch_device_noalloc (the batch decode that allocates nothing: its working memory is a fixed array
of the library's code object) against pq2_decode_batch_device on the same batches, bit for bit, through the C ABI.
Two calls in a row on the same GPU, the second with a smaller batch, then the limits."""
import numpy as np
import pytest

from parquet2_amd import PAGE_DESC_DTYPE, synth
from parquet2_amd._lib import (INVALID_PARAMETER, OK, REFERENCE_WOULD_PANIC, WOULD_OVER_ALLOCATE,
                               decode_batch_device_noalloc)

pytestmark = pytest.mark.gpu


def make_batch(n_pages, seed, widths=(1, 3, 7, 12, 16, 17, 31, 32)):
    rng = np.random.default_rng(seed)
    batch = synth.PageBatch(align=1)
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
    batch.add(bytes([0b00010000, 1]), 16, 40)  # an RLE run without its second value byte: the reference panics
    batch.add(np.zeros(3, np.uint8), 0, 1000)  # bit width 0: zeros
    batch.add(np.zeros(0, np.uint8), 9, 777)   # empty stream: zeros
    return batch.finish()


def run_both(ctx, data, descs, out_values, max_bits):
    import torch

    d_in = torch.from_numpy(data).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    st_want = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel() - 16, descs, want.data_ptr(), out_values, stream)
    got = torch.full((out_values,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).cuda()
    d_status = torch.full((len(descs),), -1, dtype=torch.int32, device="cuda")
    decode_batch_device_noalloc(0, d_in.data_ptr(), d_in.numel() - 16, d_descs.data_ptr(), len(descs), max_bits,
                                got.data_ptr(), out_values, d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    return want, st_want, got, d_status.cpu().numpy()


def test_noalloc_matches_the_allocating_entry_point():
    import torch

    from parquet2_amd import Context

    ctx = Context(0)
    try:
        # first call: 300 pages; second call, right behind it on the same GPU: 37 pages of other shapes
        for n_pages, seed, max_bits in ((300, 1, 0), (37, 2, 32), (70, 3, 16)):
            widths = (1, 3, 7, 12, 16) if max_bits == 16 else (1, 3, 7, 12, 16, 17, 31, 32)
            data, descs, out_values = make_batch(n_pages, seed, widths)
            want, st_want, got, st_got = run_both(ctx, data, descs, out_values, max_bits)
            assert st_want.tolist() == st_got.tolist()
            assert REFERENCE_WOULD_PANIC in st_got.tolist() and (st_got == OK).sum() == len(descs) - 1
            assert torch.equal(want, got), "the two entry points differ on a batch of %d pages" % n_pages
    finally:
        ctx.close()


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
    with pytest.raises(WouldOverAllocate):
        decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, d_descs.data_ptr(), 4, 12, d_out.data_ptr(),
                                    (1 << 26) + 1, d_status.data_ptr())
    with pytest.raises(InvalidParameter):
        decode_batch_device_noalloc(0, d_in.data_ptr(), data.size - 16, 0, 4, 12, d_out.data_ptr(), out_values,
                                    d_status.data_ptr())
    # header-position lists beyond the array: only the device knows; every page says so, nothing is written
    big = np.zeros(40, dtype=PAGE_DESC_DTYPE)
    big["in_len"] = 1 << 20
    big["num_bits"] = 12
    big["num_values"] = 1 << 20
    big["out_off"] = np.arange(40, dtype=np.uint64) << 20
    d_big_in = torch.zeros((1 << 20) + 16, dtype=torch.uint8, device="cuda")
    d_big_out = torch.full((40 << 20,), 0x1234, dtype=torch.int32, device="cuda")
    d_big_descs = torch.from_numpy(big.view(np.uint8).reshape(-1).copy()).cuda()
    d_big_status = torch.full((40,), -1, dtype=torch.int32, device="cuda")
    decode_batch_device_noalloc(0, d_big_in.data_ptr(), 1 << 20, d_big_descs.data_ptr(), 40, 12, d_big_out.data_ptr(),
                                40 << 20, d_big_status.data_ptr())
    torch.cuda.synchronize()
    assert d_big_status.cpu().tolist() == [WOULD_OVER_ALLOCATE] * 40
    assert bool((d_big_out == 0x1234).all())
