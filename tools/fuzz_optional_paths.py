#!/usr/bin/env python3
"""More seeds of tests/test_optional_paths_fuzz_gpu.py (the three ways through a batch of flat nullable pages must
agree on every byte).    python tools/fuzz_optional_paths.py [first seed = 10] [seeds = 30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_optional_paths_fuzz_gpu import test_the_three_paths_agree
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for seed in range(first, first + count):
    for value_size in (4, 8, 12):
        try:
            test_the_three_paths_agree(seed, value_size)
        except AssertionError as e:
            if "sum() >" in str(e) or not str(e).strip():
                raise
            print("seed", seed, "value_size", value_size, "FAILED:", e)
            sys.exit(1)
    print("seed", seed, "ok", flush=True)
