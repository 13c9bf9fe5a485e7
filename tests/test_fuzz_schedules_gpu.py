"""tools/fuzz_schedules.py once per test run: random pages (runs short and long, headers of one to four bytes,
streams cut and padded, random bytes) in random batches through every schedule, every page's status and values
against the oracle.  Two fixed seeds; the second cuts half of the longer streams within a few bytes of a 2 KiB
boundary, where the scout kernels cut their tiles (the case that found the one bug of these kernels a fuzzer
has found so far)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed,edges", [(101, False), (202, True)])
def test_random_batches_in_every_schedule(oracle, seed, edges):
    import fuzz_schedules

    from parquet2_amd import Context

    ctx = Context(0)
    try:
        total = 0
        for schedule in ("scout", "ring", "tile", "page", "auto"):
            pages, values, not_ok = fuzz_schedules.fuzz(ctx, oracle, schedule, 3, seed, edges, max_pages=120)
            assert pages > 0 and values > 0
            total += pages
        assert total >= 100
    finally:
        ctx.close()
