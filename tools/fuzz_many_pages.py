#!/usr/bin/env python3
"""Batches of 1100 to 1600 random pages (those of tools/fuzz_schedules.py; every other batch with at most 6000 values
a page) through the default schedule on the device path, against the CPU oracle: what more than 1024 pages take
(parse_kernel + expand_kernel for small pages, ring_kernel otherwise).  Needs a GPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_schedules import random_page
from _oracle import oracle as get_oracle
from parquet2_amd import Context, synth
from parquet2_amd._lib import OK
import torch
oracle = get_oracle()
ctx = Context(0)
rng = np.random.default_rng(77)
for b in range(4):
    n_pages = int(rng.integers(1100, 1600))
    small = b % 2 == 0  # every other batch: pages of a few thousand values only
    batch = synth.PageBatch(align=1); pages = []
    while len(pages) < n_pages:
        stream, w, n = random_page(rng)
        if small: n = min(n, 6000)
        batch.add(stream, w, n, out_align=1); pages.append((stream, w, n))
    data, descs, out_values = batch.finish()
    d_in = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
    d_out = torch.zeros(max(out_values, 1), dtype=torch.int32, device="cuda")
    status = ctx.decode_batch_device(d_in.data_ptr(), data.size, descs, d_out.data_ptr(), out_values, 0)
    got = d_out.cpu().numpy().view(np.uint32)
    for k, (stream, w, n) in enumerate(pages):
        st_o, exp = oracle.hybrid_rle_decode(stream, w, n)
        assert status[k] == st_o, (b, k, status[k], st_o, w, n, stream.size)
        if st_o == OK:
            o = int(descs[k]["out_off"])
            assert (got[o:o + n] == exp).all(), (b, k, w, n)
    print("batch %d: %d pages, %d values, schedule %s: all as the oracle" % (b, n_pages, out_values, ctx.last_schedule()), flush=True)
