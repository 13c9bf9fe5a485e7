"""Times pq2_decode_selected_batch_device against the full decode of the same pages: one interval of `--rows` rows
in every `--every`-th tile of 4096 values.  python tools/bench_selected.py [--pages 1024] [--values 1048576]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=1024)
    ap.add_argument("--values", type=int, default=1 << 20)
    ap.add_argument("--bits", type=int, default=12)
    ap.add_argument("--every", type=int, default=16)
    ap.add_argument("--rows", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch

    from parquet2_amd import INTERVAL_DTYPE, Context, synth

    distinct = [synth.dict_page(synth.seed_for(2, p), a.values, a.bits)[0] for p in range(4)]
    batch = synth.PageBatch()
    for p in range(a.pages):
        batch.add(distinct[p % 4], a.bits, a.values)
    data, descs, _ = batch.finish()
    tiles = (a.values + 4095) // 4096
    starts = np.arange(0, tiles, a.every, dtype=np.uint64) * 4096 + 11
    starts = starts[starts + a.rows <= a.values]
    one = np.zeros(len(starts), dtype=INTERVAL_DTYPE)
    one["start"], one["length"] = starts, a.rows
    intervals = np.tile(one, a.pages)
    offsets = np.arange(a.pages + 1, dtype=np.uint64) * len(one)
    sel_descs = descs.copy()
    sel_descs["out_off"] = np.arange(a.pages, dtype=np.uint64) * (len(one) * a.rows)
    d_in = torch.from_numpy(data).cuda()
    d_full = torch.zeros(a.pages * a.values, dtype=torch.int32, device="cuda")
    d_sel = torch.zeros(a.pages * len(one) * a.rows, dtype=torch.int32, device="cuda")
    ctx = Context(0)
    best_full = best_sel = 1e9
    for r in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_full.data_ptr(), d_full.numel())
        t1 = time.perf_counter()
        st2 = ctx.decode_selected_batch_device(d_in.data_ptr(), d_in.numel(), sel_descs, intervals, offsets,
                                               d_sel.data_ptr(), d_sel.numel())
        t2 = time.perf_counter()
        if r:
            best_full, best_sel = min(best_full, t1 - t0), min(best_sel, t2 - t1)
    assert (st == 0).all() and (st2 == 0).all()
    full = d_full.view(a.pages, a.values)
    sel = d_sel.view(a.pages, len(one), a.rows)
    for p in (0, a.pages - 1):
        for k in (0, len(one) - 1):
            s0 = int(starts[k])
            assert torch.equal(sel[p, k], full[p, s0:s0 + a.rows])
    print(f"pages={a.pages} values/page={a.values} bits={a.bits}: full decode {best_full * 1e3:.3f} ms; "
          f"{a.rows} rows of every {a.every}th tile ({len(one) * a.rows} rows/page): {best_sel * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
