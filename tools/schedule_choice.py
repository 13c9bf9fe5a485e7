#!/usr/bin/env python3
"""What the default schedule costs beside the best forced one: batches of 64 / 256 / 1024 pages of 2^20 values --
pages of short runs (BASELINE configs[3], width 17), dictionary-index pages (configs[1]) and definition levels --
through ring_kernel, through the scout kernels and through "auto" on one context.

    python tools/schedule_choice.py [steps = 10]        (needs a GPU; one line per batch)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet2_amd import PAGE_DESC_DTYPE, Context, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1 << 20
D = 16
pools = {
    "short runs, w = 17": [synth.adversarial(synth.seed_for(4, p), n, 17) + (17,) for p in range(D)],
    "index pages, w = 12": [synth.dict_page(synth.seed_for(2, p), n, 12) + (12,) for p in range(D)],
    "definition levels": [synth.def_levels_page(synth.seed_for(3, p), n) + (1,) for p in range(D)],
}
ctx = Context(0)
print("%-22s %6s | %9s %9s %9s | %-42s %s" % ("pages", "count", "ring ms", "scout ms", "auto ms", "auto took", "auto / best"))
worst = 0.0
for name, pool in pools.items():
    stride = (max(len(s) for s, _, _ in pool) + 16 + 255) // 256 * 256
    h = np.zeros((D, stride), np.uint8)
    for i, (s, _, _) in enumerate(pool):
        h[i, :len(s)] = s
    d_pool = torch.from_numpy(h).cuda()
    for pages_n in (64, 256, 1024):
        d_in = d_pool[torch.arange(pages_n, device="cuda") % D].contiguous()
        d_out = torch.empty(pages_n * n, dtype=torch.int32, device="cuda")
        descs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
        descs["in_off"] = np.arange(pages_n, dtype=np.uint64) * stride
        descs["in_len"] = [len(pool[i % D][0]) for i in range(pages_n)]
        descs["num_bits"] = pool[0][2]
        descs["num_values"] = n
        descs["out_off"] = np.arange(pages_n, dtype=np.uint64) * n
        ms, took = {}, ""
        for schedule in ("ring", "scout", "auto"):
            ctx.set_schedule(schedule)
            few = 2 if schedule == "ring" and name.startswith("short") else steps  # (tens of milliseconds a step)
            st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, 0)
            assert (st == 0).all()
            expect = torch.from_numpy(pool[0][1].view(np.int32)).cuda()
            assert torch.equal(d_out[:n], expect), "%s: %s differs" % (name, schedule)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(few):
                ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), pages_n * n, 0,
                                        want_status=False)
            torch.cuda.synchronize()
            ms[schedule] = 1e3 * (time.perf_counter() - t0) / few
            if schedule == "auto":
                from parquet2_amd import _lib
                took = "%s (%d headers expected)" % (ctx.last_schedule(), _lib.lib().pq2_ctx_last_density(ctx._h))
        ratio = ms["auto"] / min(ms["ring"], ms["scout"])
        worst = max(worst, ratio)
        print("%-22s %6d | %9.3f %9.3f %9.3f | %-42s %.2f" % (name, pages_n, ms["ring"], ms["scout"], ms["auto"], took, ratio),
              flush=True)
        del d_in, d_out
print("worst auto / best: %.2f" % worst)
ctx.close()
