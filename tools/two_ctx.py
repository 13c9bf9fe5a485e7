#!/usr/bin/env python3
"""Experiment: two contexts on two streams, each decoding its own batch of dict pages, issued back to back.
Shows how well parse_kernel of one batch hides behind expand_kernel of the other."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet2_amd import PAGE_DESC_DTYPE, Context, synth

pages_n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 20
n = 1 << 20
dev = torch.device("cuda", 0)
pages = [synth.dict_page(synth.seed_for(2, p), n, 12) for p in range(16)]
stride = (max(len(s) for s, _ in pages) + 16 + 255) // 256 * 256
h_in = np.zeros((16, stride), dtype=np.uint8)
for i, (s, _) in enumerate(pages):
    h_in[i, :len(s)] = s
d_distinct = torch.from_numpy(h_in).to(dev)
jobs = []
for c in range(nctx):
    idx = torch.arange(pages_n, device=dev) % 16
    d_in = d_distinct[idx].contiguous()
    d_out = torch.empty(pages_n * n, dtype=torch.int32, device=dev)
    descs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
    descs["in_off"] = np.arange(pages_n, dtype=np.uint64) * stride
    descs["in_len"] = [len(pages[i % 16][0]) for i in range(pages_n)]
    descs["num_bits"] = 12
    descs["num_values"] = n
    descs["out_off"] = np.arange(pages_n, dtype=np.uint64) * n
    jobs.append((Context(0), torch.cuda.Stream(), d_in, d_out, descs))

def step():
    for ctx, st, d_in, d_out, descs in jobs:
        ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, st.cuda_stream, want_status=False)

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
exp = torch.from_numpy(pages[3][1].view(np.int32)).to(dev)
for ctx, st, d_in, d_out, descs in jobs:
    assert torch.equal(d_out.view(pages_n, n)[3], exp)
print(json.dumps({"contexts": nctx, "pages_per_ctx": pages_n, "ms_per_step": round(el * 1e3, 4),
                  "GBps": round(4 * n * pages_n * nctx / el / 1e9, 1)}))
