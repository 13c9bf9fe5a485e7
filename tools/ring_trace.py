#!/usr/bin/env python3
"""Diagnostic for a PQ2_RING_TRACE build (PQ2_LIB=.../libpq2hip_trace.so): when the workgroup of every page of one
ring_kernel batch began and ended, and how busy the GPU was over the launch.

    python tools/ring_trace.py [pages = 2048] [dict | def]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PQ2_MODE", "ring")
from parquet2_amd import PAGE_DESC_DTYPE, Context, _lib, synth  # noqa: E402

pages_n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kind = sys.argv[2] if len(sys.argv) > 2 else "dict"
n = 1 << 20
bits = 12 if kind == "dict" else 1
D = 64
distinct = [synth.dict_page(synth.seed_for(2, p), n, 12) if kind == "dict" else synth.def_levels_page(synth.seed_for(3, p), n)
            for p in range(D)]
stride = (max(len(s) for s, _ in distinct) + 16 + 255) // 256 * 256
hbuf = np.zeros((D, stride), np.uint8)
for i, (s, _) in enumerate(distinct):
    hbuf[i, :len(s)] = s
d_in = torch.from_numpy(hbuf).cuda()[torch.arange(pages_n, device="cuda") % D].contiguous()
d_out = torch.zeros(pages_n * n, dtype=torch.int32, device="cuda")
descs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
descs["in_off"] = np.arange(pages_n, dtype=np.uint64) * stride
descs["in_len"] = [len(distinct[i % D][0]) for i in range(pages_n)]
descs["num_bits"] = bits
descs["num_values"] = n
descs["out_off"] = np.arange(pages_n, dtype=np.uint64) * n
ctx = Context(0)
h = _lib.lib()
if not hasattr(h, "pq2_ring_trace"):
    sys.exit("not a PQ2_RING_TRACE build")
for _ in range(3):
    ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, 0)
torch.cuda.synchronize()
m = min(pages_n, 8192)
buf = (ctypes.c_ulonglong * (3 * m))()
h.pq2_ring_trace(buf, m)
t = np.frombuffer(buf, dtype=np.uint64).reshape(m, 3).astype(np.int64)
t0 = t[:, 0].min()
begin = (t[:, 0] - t0) / 100.0  # microseconds (100 MHz)
end = (t[:, 1] - t0) / 100.0
where = t[:, 2]
total = end.max()
print("%d pages (%s): first begin to last end %.1f us" % (pages_n, kind, total))
q = [0, 1, 5, 25, 50, 75, 95, 99, 100]
print("begin percentiles %s us: %s" % (q, np.percentile(begin, q).round(1).tolist()))
print("end   percentiles %s us: %s" % (q, np.percentile(end, q).round(1).tolist()))
print("page time percentiles us: %s" % np.percentile(end - begin, q).round(1).tolist())
# pages in progress over time
edges = np.linspace(0, total, 41)
running = [(int(((begin <= x) & (end > x)).sum())) for x in edges]
print("workgroups running at 40 points of the launch: %s" % running)
first = begin < np.percentile(begin, 50) if pages_n > 1024 else np.ones(m, bool)
print("first wave of pages: page time mean %.1f us; the others: %.1f us" %
      ((end - begin)[first].mean(), (end - begin)[~first].mean() if (~first).any() else 0.0))
xcc = where >> 8
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    print("XCC %d: %d pages, mean page time %.1f us, last end %.1f us" % (x, sel.sum(), (end - begin)[sel].mean(), end[sel].max()))
# what the page times go with: the page's content (page number mod D), the compute unit, the position in the launch
dur = end - begin
def spread(keys, name):
    ks = sorted(set(keys.tolist()))
    means = np.array([dur[keys == k].mean() for k in ks])
    within = np.sqrt(np.mean([dur[keys == k].var() for k in ks]))
    print("by %s: %d groups, spread of the group means %.1f us (min %.1f, max %.1f), spread inside a group %.1f us" %
          (name, len(ks), means.std(), means.min(), means.max(), within))
print("all pages: mean %.1f us, spread %.1f us" % (dur.mean(), dur.std()))
spread(np.arange(m) % D, "content")
spread(where, "compute unit")
spread(where & 0xff, "compute unit number inside its XCC")
# slots: the pages of one compute unit in order of their begin; pairs (first page, the page that began when it ended)
cu_sum = {}
for k in sorted(set(where.tolist())):
    sel = where == k
    cu_sum[k] = (dur[sel].sum(), end[sel].max(), sel.sum())
sums = np.array([v[0] for v in cu_sum.values()])
ends = np.array([v[1] for v in cu_sum.values()])
cnt = np.array([v[2] for v in cu_sum.values()])
print("per compute unit: pages %s; sum of page times mean %.0f us spread %.0f; last end mean %.0f spread %.0f min %.0f max %.0f" %
      (sorted(set(cnt.tolist())), sums.mean(), sums.std(), ends.mean(), ends.std(), ends.min(), ends.max()))
print("sizes of the streams: min %d max %d bytes" % (descs["in_len"].min(), descs["in_len"].max()))
lens = descs["in_len"][:m].astype(np.float64)
print("correlation of page time with stream bytes: %.2f" % np.corrcoef(lens, dur)[0, 1])
for k in sorted(set(where.tolist()))[:6]:
    sel = np.nonzero(where == k)[0]
    sel = sel[np.argsort(begin[sel])]
    print("XCC %d compute unit %3d: %s" % (k >> 8, k & 0xff, "  ".join("%.0f-%.0f" % (begin[i], end[i]) for i in sel)))
