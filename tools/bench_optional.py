#!/usr/bin/env python3
"""Times pq2_decode_optional_dict_batch_device: `pages` nullable dictionary pages of `rows` rows (definition levels
as BASELINE configs[2]: bit width 1, long runs, 5 % nulls; indices of bit width 12, mixed runs; 4-byte values)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet2_amd import DICT_REF_DTYPE, PAGE_DESC_DTYPE, Context, synth

pages_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
value_size = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda", 0)
distinct = [(synth.def_levels_page(synth.seed_for(3, p), rows), synth.dict_page(synth.seed_for(2, p), rows, 12))
            for p in range(8)]
chunks, at = [], 0
place = []
for (ds, _), (xs, _) in distinct:
    place.append((at, len(ds), at + len(ds), len(xs)))
    chunks += [ds, xs]
    at += len(ds) + len(xs)
    pad = (-at) % 16
    chunks.append(np.zeros(pad, dtype=np.uint8))
    at += pad
one = np.concatenate(chunks + [np.zeros(16, dtype=np.uint8)])
reps = (pages_n + 7) // 8
d_in = torch.from_numpy(one).to(dev).repeat(reps)
stride = one.size
defs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
idxs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
for i in range(pages_n):
    o = place[i % 8]
    base = (i // 8) * stride
    defs[i] = (base + o[0], o[1], 1, rows, i * rows)
    idxs[i] = (base + o[2], o[3], 12, rows, 0)
words = value_size // 4
table = torch.randint(-2**31, 2**31 - 1, (4096 * words,), dtype=torch.int32, device=dev)
refs = np.zeros(pages_n, dtype=DICT_REF_DTYPE)
refs["dict_len"] = 4096
d_out = torch.empty(pages_n * rows * words, dtype=torch.int32, device=dev)
d_val = torch.empty(pages_n * rows, dtype=torch.uint8, device=dev)
ctx = Context(0)
stream = torch.cuda.current_stream().cuda_stream

def step(want=False):
    return ctx.decode_optional_dict_batch_device(d_in.data_ptr(), d_in.numel(), defs, idxs, np.ones(pages_n, np.uint32),
                                                 table.data_ptr(), table.numel() * 4, refs, value_size, d_out.data_ptr(),
                                                 d_val.data_ptr(), pages_n * rows, stream, want_status=want)

assert (step(True) == 0).all()
lv, ix = distinct[3][0][1], distinct[3][1][1]
valid = lv == 1
exp = torch.zeros(rows, words, dtype=torch.int32)
exp[torch.from_numpy(valid)] = table.view(4096, words).cpu()[torch.from_numpy(ix[:int(valid.sum())].astype(np.int64))]
assert os.environ.get("PQ2_BENCH_NO_CHECK") or torch.equal(d_out.view(pages_n, rows, words)[3].cpu(), exp)  # (diagnostic builds write nothing)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for _ in range(steps):
    h0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
if os.environ.get("PQ2_BENCH_HOST"):
    print("host ms per call:", [round(h * 1e3, 2) for h in host], file=sys.stderr)
print(json.dumps({"pages": pages_n, "rows_per_page": rows, "value_size": value_size, "ms_per_step": round(el * 1e3, 3),
                  "rows_per_second_G": round(pages_n * rows / el / 1e9, 2),
                  "value_plus_validity_GBps": round(pages_n * rows * (value_size + 1) / el / 1e9, 1)}))
