#!/usr/bin/env python3
"""Diagnostic: one ring_kernel batch of config-2 pages; prints the status histogram (a PQ2_RING_DEBUG build codes
the wait that ran out into the status) and the time of a few calls."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PQ2_MODE", "ring")
from parquet2_amd import PAGE_DESC_DTYPE, Context, synth
pages_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = 1 << 20
kind = sys.argv[3] if len(sys.argv) > 3 else "dict"   # dict (width 12) or def (definition levels, width 1)
bits = 12 if kind == "dict" else 1
distinct = [synth.dict_page(synth.seed_for(2, p), n, 12) if kind == "dict" else synth.def_levels_page(synth.seed_for(3, p), n)
            for p in range(16)]
stride = (max(len(s) for s, _ in distinct) + 16 + 255) // 256 * 256
h = np.zeros((16, stride), np.uint8)
for i, (s, _) in enumerate(distinct): h[i, :len(s)] = s
d_in = torch.from_numpy(h).cuda()[torch.arange(pages_n, device="cuda") % 16].contiguous()
d_out = torch.zeros(pages_n * n, dtype=torch.int32, device="cuda")
descs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
descs["in_off"] = np.arange(pages_n, dtype=np.uint64) * stride
descs["in_len"] = [len(distinct[i % 16][0]) for i in range(pages_n)]
descs["num_bits"] = bits; descs["num_values"] = n; descs["out_off"] = np.arange(pages_n, dtype=np.uint64) * n
ctx = Context(0)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
times = []
for it in range(calls):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    vals, cnt = np.unique(st, return_counts=True)
    times.append(dt * 1e3)
    if it < 2 or len(vals) > 1 or vals[0] != 0 or dt > 0.02:
        print("call %d: %.3f ms, statuses %s" % (it, dt * 1e3, dict(zip(vals.tolist(), cnt.tolist()))), flush=True)
print("%d pages, %d calls: median %.3f ms, min %.3f, max %.3f" % (pages_n, calls, np.median(times), min(times), max(times)))

import ctypes
from parquet2_amd import _lib
h = _lib.lib()
if hasattr(h, "pq2_ring_clocks"):  # a PQ2_RING_TIMING build
    buf = (ctypes.c_ulonglong * 32)()
    h.pq2_ring_clocks(buf)  # clear
    ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, 0)
    torch.cuda.synchronize()
    h.pq2_ring_clocks(buf)
    c = [float(x) for x in buf]
    print("walk:   total %.0f clocks/page; room %.1f%%, fills %.1f%%, chase %.1f%%, tail %.1f%%, place %.1f%%, "
          "ensure at the top %.1f%%, more bytes inside a pass %.1f%% (fills are inside the last two)" %
          (c[0] / pages_n, 100 * c[1] / c[0], 100 * c[2] / c[0], 100 * c[3] / c[0], 100 * c[4] / c[0],
           100 * c[5] / c[0], 100 * c[6] / c[0], 100 * c[7] / c[0]))
    print("expand: total %.0f clocks/wave; waits: block start %.1f%%, runs %.1f%%, bytes %.1f%%; phase 1 %.1f%% (its waits "
          "included), index %.1f%%, phase 2 %.1f%%, run starts %.1f%%" %
          (c[8] / pages_n / 3, 100 * c[9] / c[8], 100 * c[10] / c[8], 100 * c[11] / c[8], 100 * c[12] / c[8],
           100 * c[13] / c[8], 100 * c[14] / c[8], 100 * c[15] / c[8]))
    k = c[16:28]
    print("walk counts per page: chase entries %.0f, fills waited for %.0f, looks at the limits %.0f (in vain %.0f), "
          "bytes needed with nothing in flight %.0f, passes %.0f" % tuple(x / pages_n for x in k[:6]))
    if k[5]:
        print("at the top of a pass, on average: lowest need %.0f bytes behind the walk, %.0f bytes landed ahead, "
              "%.0f in flight, %.0f after topping up" % (k[6] / k[5], k[7] / k[5], k[8] / k[5], k[9] / k[5]))
