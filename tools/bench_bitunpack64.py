#!/usr/bin/env python3
"""Rate of pq2_bitunpack_u64_device (bitpacked::Decoder<u64>) for a few widths: 2^25 values per call, device-resident."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet2_amd import Context

ctx = Context(0)
n = 1 << 25
res = {}
for w in (1, 7, 13, 20, 32, 33, 48, 64):
    packed = torch.randint(0, 256, (n * w // 8 + 16,), dtype=torch.uint8, device="cuda")
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        ctx.bitunpack_u64_device(packed.data_ptr(), n * w // 8, w, n, out.data_ptr(), s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ctx.bitunpack_u64_device(packed.data_ptr(), n * w // 8, w, n, out.data_ptr(), s)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    # spot check against numpy on the first 4096 values
    raw = packed[:4096 * w // 8 + 16].cpu().numpy()
    bits = np.unpackbits(raw, bitorder="little")[:4096 * w].reshape(4096, w)
    want = (bits.astype(np.uint64) << np.arange(w, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
    assert (out[:4096].cpu().numpy().view(np.uint64) == want).all(), w
    res[w] = {"ms": round(ms, 4), "out_gbs": round(8 * n / ms / 1e6, 1), "in_plus_out_gbs": round((8 + w / 8) * n / ms / 1e6, 1)}
print(json.dumps({"kernel": "bitunpack64_kernel", "values_per_call": n, "by_width": res}))
