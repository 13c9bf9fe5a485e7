"""Rounds the resolve step of the scout kernels takes per page, for the bench's few-page workloads."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PQ2_MODE"] = "scout"
from parquet2_amd import Context, synth  # noqa: E402

ctx = Context(0)
cases = [("adversarial w=3", lambda p: synth.adversarial(synth.seed_for(4, p), 1 << 20, 3), 3, 16),
         ("adversarial w=17", lambda p: synth.adversarial(synth.seed_for(4, p), 1 << 20, 17), 17, 16),
         ("adversarial w=31", lambda p: synth.adversarial(synth.seed_for(4, p), 1 << 20, 31), 31, 16),
         ("dict w=12", lambda p: synth.dict_page(synth.seed_for(2, p), 1 << 20, 12), 12, 32),
         ("dict w=20", lambda p: synth.dict_page(synth.seed_for(2, p), 1 << 20, 20), 20, 8),
         ("def levels", lambda p: synth.def_levels_page(synth.seed_for(3, p), 1 << 20), 1, 16),
         ("one bit-packed run w=12", lambda p: (synth.encode_hybrid(synth.uniform_values(p, 1 << 20, 12), 12, 1 << 30),
                                                None), 12, 4)]
for name, make, w, pages in cases:
    batch = synth.PageBatch()
    for p in range(pages):
        stream, _ = make(p)
        batch.add(stream, w, 1 << 20)
    data, descs, out_values = batch.finish()
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.zeros(out_values, dtype=torch.int32, device="cuda")
    st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), out_values,
                                 torch.cuda.current_stream().cuda_stream)
    r = ctx.scout_rounds(pages)
    tiles = (descs["in_len"].astype(np.int64) + 2047) // 2048
    print("%-26s tiles/page %6d  rounds min %3d  median %3d  max %3d  status ok %s" %
          (name, int(np.median(tiles)), r.min(), int(np.median(r)), r.max(), bool((st == 0).all())), flush=True)
