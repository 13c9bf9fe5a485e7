#!/usr/bin/env python3
"""Longer random comparison of every schedule with the CPU oracle than the test suite runs: pages built from random
runs (short and long, RLE and bit-packed, headers of one to four bytes), cut and padded at random, plus pages of random
bytes, in batches of random size.  Statuses and values must match bit for bit.

usage: fuzz_schedules.py [batches = 40] [seed = 1] [edges]      (needs a GPU; prints one line per schedule)
"edges": half of the longer streams are cut so that they end within a few bytes of a 2 KiB tile boundary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import oracle as get_oracle  # noqa: E402

from parquet2_amd import Context, synth  # noqa: E402
from parquet2_amd._lib import OK  # noqa: E402


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def random_page(rng):
    w = int(rng.integers(1, 33))
    kind = rng.random()
    if kind < 0.15:  # random bytes
        length = int(rng.integers(0, 6000))
        data = rng.integers(0, 256, length, dtype=np.uint8)
        if length and rng.random() < 0.5:
            data[rng.integers(0, length, length // 3)] &= 0x1F
        return data, w, int(rng.integers(0, 40000))
    rb = (w + 7) // 8
    short = rng.random() < 0.5  # pages of short runs / of long ones
    parts, total = [], 0
    for _ in range(int(rng.integers(1, 600 if short else 60))):
        if rng.random() < 0.5:
            g = int(rng.integers(0, 3 if short else 400))
            if rng.random() < 0.02:
                g = int(rng.integers(8192, 20000))  # a header of three bytes
            parts.append(varint((g << 1) | 1) + rng.integers(0, 256, g * w, dtype=np.uint8).tobytes())
            total += 8 * g
        else:
            c = int(rng.integers(0, 9 if short else 3000))
            if rng.random() < 0.02:
                c = int(rng.integers(8192, 1 << 21))
            parts.append(varint(c << 1) + rng.integers(0, 256, rb, dtype=np.uint8).tobytes())
            total += c
    stream = np.frombuffer(b"".join(parts), dtype=np.uint8)
    how = rng.random()
    if how < 0.2 and stream.size:
        stream = stream[:int(rng.integers(0, stream.size))]  # cut anywhere
    n = total
    if how > 0.6:
        n = int(rng.integers(0, total + 2000))
    return stream, w, min(n, 1 << 22)


def fuzz(ctx, oracle, schedule, batches, seed, edges=False, max_pages=200):
    """`batches` random batches through `schedule` on `ctx`; returns (pages, values, pages with a status other than
    ok); raises AssertionError at the first page that differs from the oracle."""
    ctx.set_schedule(schedule)
    rng = np.random.default_rng(seed)
    pages_done = values_done = not_ok = 0
    for b in range(batches):
        n_pages = int(rng.integers(1, max_pages))
        batch = synth.PageBatch(align=1)
        pages = []
        for _ in range(n_pages):
            stream, w, n = random_page(rng)
            if edges and stream.size > 2048 and rng.random() < 0.5:
                # cut so that the stream ends at, or a byte or two either side of, a multiple of 2 KiB counted
                # from the 16-byte line it begins in (where the scout kernels cut their tiles)
                delta = batch.in_bytes & 15
                want = int(rng.choice([0, 0, 1, 2, 3, 2045, 2046, 2047]))
                size = stream.size - (delta + stream.size - want) % 2048
                if size > 0:
                    stream = stream[:size]
            batch.add(stream, w, n, out_align=1)
            pages.append((stream, w, n))
        data, descs, out_values = batch.finish()
        got, status = ctx.decode_batch_host(data, descs, out_values)
        for k, (stream, w, n) in enumerate(pages):
            st_o, exp = oracle.hybrid_rle_decode(stream, w, n)
            assert status[k] == st_o, "%s: batch %d page %d: status %d, oracle %d (w=%d n=%d len=%d)" % (
                schedule, b, k, status[k], st_o, w, n, stream.size)
            if st_o == OK:
                o = int(descs[k]["out_off"])
                bad = np.nonzero(got[o:o + n] != exp)[0]
                assert not bad.size, "%s: batch %d page %d (w=%d n=%d len=%d): first mismatch at %d: %d, oracle %d" % (
                    schedule, b, k, w, n, stream.size, bad[0], got[o + bad[0]], exp[bad[0]])
            else:
                not_ok += 1
            values_done += n
        pages_done += n_pages
    return pages_done, values_done, not_ok


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    edges = len(sys.argv) > 3 and sys.argv[3] == "edges"
    oracle = get_oracle()
    ctx = Context(0)
    for schedule in ("scout", "ring", "tile", "page", "auto"):
        try:
            pages_done, values_done, not_ok = fuzz(ctx, oracle, schedule, batches, seed, edges)
        except AssertionError as e:
            raise SystemExit(str(e))
        print("%-6s %d batches, %d pages, %d values, %d pages with a status other than ok: all as the oracle" %
              (schedule, batches, pages_done, values_done, not_ok), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
