"""The N > 1 control path on CPU: two ranks (gloo) shard one page list with pq2_partition_pages, every rank
owns a contiguous, disjoint range, the ranges cover the batch, and the per-rank byte counts reduce to the
whole-job total exactly as bench.py computes it.  No data-path collective exists to test: pages share no state."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from parquet2_amd import PAGE_DESC_DTYPE, page_costs, partition_pages

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(7)                      # every rank builds the same global page list
    descs = np.zeros(301, dtype=PAGE_DESC_DTYPE)
    descs["num_values"] = rng.integers(1, 1 << 20, len(descs))
    descs["num_values"][::7] = 0                         # empty pages take part too
    descs["in_len"] = descs["num_values"] * rng.integers(1, 33, len(descs)) // 8
    bounds = partition_pages(descs, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    cost = page_costs(descs)
    loads = [None] * world
    dist.all_gather_object(loads, int(cost[lo:hi].sum()))
    assert sum(loads) == int(cost.sum()) and max(loads) <= cost.sum() / world + cost.max()
    mine = int(descs["num_values"][lo:hi].sum()) * 4
    tot = torch.tensor([float(mine)], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    assert tot.item() == float(descs["num_values"].sum() * 4), (tot.item(), descs["num_values"].sum() * 4)
    ranges = [None] * world
    dist.all_gather_object(ranges, (lo, hi))
    assert ranges[0][0] == 0 and ranges[-1][1] == len(descs)
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    share = descs["num_values"].sum() * 4 / world
    assert mine <= share + int(descs["num_values"].max()) * 4
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)   # the max-over-ranks timing reduction
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.001 * world) < 1e-12
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["PQ2_TEST_OUT"], "rank%%d.ok" %% rank), "w").write("%%d %%d" %% (lo, hi))
""") % ROOT


def test_two_ranks_shard_pages_with_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", PQ2_TEST_OUT=str(tmp_path))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    ranges = [tuple(int(v) for v in (tmp_path / ("rank%d.ok" % r)).read_text().split()) for r in range(2)]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 301
