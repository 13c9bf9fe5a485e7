"""Times pq2_delta_bitpacked_decode_batch_device on a batch of equal streams (distinct copies in HBM) and the CPU
oracle on one of them.  python tools/bench_delta.py [--streams 4096] [--values 65537] [--bits 10] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--values", type=int, default=65537)
    ap.add_argument("--bits", type=int, default=10)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--minis", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch

    from parquet2_amd import PAGE_DESC_DTYPE, Context
    from test_delta_bitpacked_gpu import encode_delta

    rng = np.random.default_rng(1)
    v = np.cumsum(rng.integers(0, 1 << a.bits, size=a.values, dtype=np.int64))
    one = np.frombuffer(encode_delta(v, a.block, a.minis), dtype=np.uint8)
    stride = (one.size + 15) & ~15
    blob = np.zeros((a.streams, stride), dtype=np.uint8)
    blob[:, :one.size] = one
    descs = np.zeros(a.streams, dtype=PAGE_DESC_DTYPE)
    descs["in_off"] = np.arange(a.streams, dtype=np.uint64) * stride
    descs["in_len"] = one.size
    descs["num_values"] = a.values
    descs["out_off"] = np.arange(a.streams, dtype=np.uint64) * a.values
    d_in = torch.from_numpy(blob.reshape(-1)).cuda()
    d_out = torch.zeros(a.streams * a.values, dtype=torch.int64, device="cuda")
    ctx = Context(0)
    best = 1e9
    for r in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status, counts, consumed = ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), blob.size, descs,
                                                                          d_out.data_ptr(), d_out.numel())
        dt = time.perf_counter() - t0
        if r:
            best = min(best, dt)
    assert (status == 0).all() and (counts == a.values).all() and (consumed == one.size).all()
    got = d_out.view(a.streams, a.values)
    assert (got[0].cpu().numpy() == v).all() and (got[-1].cpu().numpy() == v).all()
    total_bytes = a.streams * (one.size + 8 * a.values)
    from _oracle import oracle as get

    o = get()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        st, exp, used = o.delta_bitpacked_decode(one, a.values)
        n += 1
    cpu = (time.perf_counter() - t0) / n
    print(f"streams={a.streams} values/stream={a.values} bits={a.bits} block={a.block}/{a.minis} "
          f"in={one.size} B/stream: {best * 1e3:.3f} ms (call, results fetched) = "
          f"{a.streams * a.values / best / 1e9:.2f} G values/s, {total_bytes / best / 1e9:.1f} GB/s in+out; "
          f"oracle, one core: {a.values / cpu / 1e9:.3f} G values/s")


if __name__ == "__main__":
    main()
