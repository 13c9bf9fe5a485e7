"""DELTA_BINARY_PACKED streams against the CPU oracle, seed after seed: generated streams of the block shapes that
delta_kernel treats differently (128 values in 1, 2 or 4 mini-blocks; other shapes of the chunked path; the general
path), whole, cut and with a few bytes overwritten.  python tools/fuzz_delta.py [first_seed] [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from _oracle import oracle as get
    from parquet2_amd import Context
    from test_delta_bitpacked_gpu import check_against_oracle, encode_delta, values_of_deltas

    ctx, oracle = Context(0), get()
    shapes = [(128, 4), (128, 2), (128, 1), (128, 4), (256, 8), (512, 16), (128, 16), (1024, 4)]
    total = {}
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        streams, caps = [], []
        for k in range(300):
            block, minis = shapes[int(rng.integers(0, len(shapes)))]
            n = int(rng.choice([int(rng.integers(0, 400)), int(rng.integers(0, 5000)), int(rng.integers(5000, 40000))], p=[0.5, 0.4, 0.1]))
            top = int(rng.integers(0, 64))
            deltas, at = [], 0
            while at < n:  # stretches of deltas of one magnitude, so that widths and min_delta change along the stream
                m = int(rng.integers(1, 700))
                bits = int(rng.integers(0, top + 1))
                low = int(rng.integers(-(1 << bits), 1 << bits)) if rng.integers(0, 2) else 0
                deltas += [low + (int.from_bytes(rng.bytes(8), "little") >> (64 - bits) if bits else 0) for _ in range(min(m, n - at))]
                at += m
            values = values_of_deltas(int(rng.integers(-(1 << 63), 1 << 63)), deltas) if n or rng.integers(0, 2) else []
            s = bytearray(encode_delta(values, block, minis, width_floor=int(rng.integers(0, 2))))
            kind = int(rng.integers(0, 4))
            if kind == 1 and len(s) > 6:
                for _ in range(int(rng.integers(1, 4))):
                    s[int(rng.integers(0, len(s)))] = int(rng.integers(0, 256))
            elif kind == 2 and len(s) > 6:
                del s[int(rng.integers(4, len(s))):]
            elif kind == 3:
                s += rng.bytes(int(rng.integers(1, 40)))
            streams.append(bytes(s))
            caps.append(len(values) + int(rng.integers(0, 3)) if rng.integers(0, 8) else int(rng.integers(0, len(values) + 1)))
        seen = check_against_oracle(ctx, oracle, streams, caps, pad_front=int(rng.integers(0, 16)))
        for st, c in seen.items():
            total[st] = total.get(st, 0) + c
        print(f"seed {seed}: {dict(sorted(seen.items()))}", flush=True)
    print(f"{seeds} seeds from {first}: every stream as the oracle has it; statuses {dict(sorted(total.items()))}")


if __name__ == "__main__":
    main()
