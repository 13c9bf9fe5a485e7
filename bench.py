#!/usr/bin/env python3
"""Benchmark of the device-resident hybrid-RLE / bit-packed decode.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the decoder (pq2_decode_batch_device: decode_kernel, or parse + expand kernels) over one batch of
synthetic RLE_DICTIONARY index pages already resident in HBM: `--pages` pages per GPU of 2^20 values, bit
width 12, mixed RLE / bit-packed runs (BASELINE.json configs[1]; the default of 2048 pages per GPU is the per-GPU
share of configs[4]).  With N > 1 the script runs under
torch.distributed.run, one rank per GPU; every rank decodes its own shard of the global page list
(pq2_partition_pages), there is no data-path collective, and the reported value is the whole-job rate
(all ranks' output bytes / slowest rank's time).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from parquet2_amd import PAGE_DESC_DTYPE, Context, partition_pages, synth  # noqa: E402

METRIC = "device-resident hybrid-RLE/bitpack decode GB/s (output bytes), 1–8 GPUs"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=2048, help="pages per GPU in one batch")
    ap.add_argument("--values", type=int, default=1 << 20, help="values per page")
    ap.add_argument("--bits", type=int, default=12)
    ap.add_argument("--distinct", type=int, default=16, help="distinct generated pages (replicated on the device)")
    ap.add_argument("--workload", default="dict", choices=["dict", "def", "adversarial", "bitpacked"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-path", action="store_true",
                    help="time pq2_decode_batch_host instead (page-locked host buffers in, host buffers out: the "
                         "copies over PCIe are inside the timed region); prints its own JSON line")
    ap.add_argument("--gather-bytes", type=int, default=0, choices=[0, 4, 8, 12],
                    help="time pq2_decode_dict_batch_device instead (indices -> dictionary values of this many bytes, "
                         "a dictionary of 2^bits entries); prints its own JSON line")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the parity gate (only for timing experiments with builds that write no results)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def make_pages(args):
    """`distinct` different pages: list of (stream ndarray, expected values ndarray)."""
    pages = []
    for p in range(args.distinct):
        if args.workload == "dict":
            pages.append(synth.dict_page(synth.seed_for(2, p), args.values, args.bits))
        elif args.workload == "def":
            pages.append(synth.def_levels_page(synth.seed_for(3, p), args.values))
        elif args.workload == "adversarial":
            pages.append(synth.adversarial(synth.seed_for(4, p), args.values, args.bits))
        else:  # one bit-packed run, as the reference's own encoder writes pages
            from parquet2_amd.encoding.hybrid_rle import encode_u32
            v = synth.uniform_values(synth.seed_for(1, p), args.values, args.bits)
            pages.append((np.frombuffer(encode_u32(v, args.bits), dtype=np.uint8), v))
    return pages


def cpu_baseline(pages, bits, seconds, workload):
    """The CPU oracle (plain-C port of the reference decoder) timed on this host: one thread, and all the
    threads this box gives us, each decoding whole pages.  Also the parity checker of the sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle

    tmp = tempfile.mkdtemp(prefix="pq2oracle_")
    try:
        path = _oracle.build(native=True, out_dir=tmp)  # -O3 -march=native for the host we are on
        flags = "gcc -O3 -march=native"
    except Exception:
        path = None
        flags = "gcc -O3"
    orc = _oracle.Oracle(path)
    n_values = sum(len(v) for _, v in pages)
    # parity of the oracle itself on the sample (the GPU output is compared with `values` by the caller)
    for stream, vals in pages[:2]:
        st, out = orc.hybrid_rle_decode(stream, bits, len(vals))
        assert st == 0 and (out == vals).all(), "oracle disagrees with the generator"
    # single thread
    t_budget = max(1.0, seconds / 2)
    best = None
    t_start = time.perf_counter()
    reps_done = 0
    while time.perf_counter() - t_start < t_budget:
        t = sum(orc.time_hybrid_rle_decode(s, bits, len(v), 1) for s, v in pages)
        best = t if best is None else min(best, t)
        reps_done += 1
    one = 4.0 * n_values / best / 1e9
    # all threads of our CPU share (16 on a one-GPU box), pages sharded over threads; ctypes drops the GIL
    n_thr = max(1, min(16, len(os.sched_getaffinity(0))))
    results = [0.0] * n_thr

    def work(i):
        mine = pages[i % len(pages)]
        reps = max(1, reps_done // 2)
        results[i] = orc.time_hybrid_rle_decode(mine[0], bits, len(mine[1]), reps) / reps

    thr = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    t0 = time.perf_counter()
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    wall = time.perf_counter() - t0
    reps = max(1, reps_done // 2)
    many = 4.0 * sum(len(pages[i % len(pages)][1]) for i in range(n_thr)) * reps / wall / 1e9
    return {
        "value": round(one, 3), "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "%d %s pages x %d values, bit width %d, C port of HybridRleDecoder drained per value (%s), "
                  "best of %d passes" % (len(pages), workload, len(pages[0][1]), bits, flags, reps_done),
        "all_threads": {"value": round(many, 3), "cores": n_thr},
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    # PQ2_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs)
    backend = os.environ.get("PQ2_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif args.gpus != 1:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`" %
              (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pages = make_pages(args)
    n = args.values
    # global page list: world * pages pages, page g is a copy of distinct page g % distinct; rank r takes
    # the contiguous shard pq2_partition_pages gives it
    total_pages = world * args.pages
    g_descs = np.zeros(total_pages, dtype=PAGE_DESC_DTYPE)
    g_descs["num_values"] = n
    bounds = partition_pages(g_descs, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    my_pages = hi - lo

    # device-resident input: the distinct streams once, then one replica per page at its own address
    stride = max(len(s) for s, _ in pages)
    stride = (stride + 16 + 255) // 256 * 256
    h_in = np.zeros((len(pages), stride), dtype=np.uint8)
    for i, (s, _) in enumerate(pages):
        h_in[i, :len(s)] = s
    d_distinct = torch.from_numpy(h_in).to(dev)
    idx = torch.arange(lo, hi, device=dev) % len(pages)
    d_in = d_distinct[idx].contiguous()  # [my_pages, stride]
    d_out = torch.empty(my_pages * n, dtype=torch.int32, device=dev)
    descs = np.zeros(my_pages, dtype=PAGE_DESC_DTYPE)
    descs["in_off"] = np.arange(my_pages, dtype=np.uint64) * stride
    descs["in_len"] = [len(pages[(lo + i) % len(pages)][0]) for i in range(my_pages)]
    descs["num_bits"] = args.bits if args.workload != "def" else 1
    descs["num_values"] = n
    descs["out_off"] = np.arange(my_pages, dtype=np.uint64) * n
    bits = int(descs["num_bits"][0])
    in_bytes_alg = int(descs["in_len"].sum())
    out_bytes = 4 * my_pages * n

    ctx = Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    if args.host_path:  # end to end, host memory to host memory
        h_in_t = torch.empty(d_in.numel(), dtype=torch.uint8, pin_memory=True)
        h_in_t.copy_(d_in.reshape(-1).cpu())
        h_out_t = torch.zeros(my_pages * n, dtype=torch.int32, pin_memory=True)
        h_in_np, h_out_np = h_in_t.numpy(), h_out_t.numpy().view(np.uint32)
        ctx.decode_batch_host(h_in_np, descs, my_pages * n, out=h_out_np)
        for i in range(min(my_pages, len(pages))):
            assert np.array_equal(h_out_np[i * n:(i + 1) * n], pages[(lo + i) % len(pages)][1]), "host path differs"
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, status = ctx.decode_batch_host(h_in_np, descs, my_pages * n, out=h_out_np)
        elapsed = time.perf_counter() - t0
        assert (status == 0).all()
        print(json.dumps({
            "metric": "host-to-host hybrid-RLE/bitpack decode GB/s (output bytes), copies included",
            "value": round(out_bytes * args.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": 1,
            "steps": args.steps, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "bytes_over_pcie_per_step": int(d_in.numel()) + out_bytes,
            "pcie_gbs": round((d_in.numel() + out_bytes) * args.steps / elapsed / 1e9, 2),
            "config": {"workload": "%s: %d pages x %d values, bit width %d, page-locked host buffers" %
                       (args.workload, my_pages, n, bits)}}), flush=True)
        ctx.close()
        return

    if args.gather_bytes:  # decode + dictionary gather, device-resident
        from parquet2_amd import DICT_REF_DTYPE
        words = args.gather_bytes // 4
        table = torch.randint(-2**31, 2**31 - 1, ((1 << bits) * words,), dtype=torch.int32, device=dev)
        refs = np.zeros(my_pages, dtype=DICT_REF_DTYPE)
        refs["dict_len"] = 1 << bits
        d_val = torch.empty(my_pages * n * words, dtype=torch.int32, device=dev)

        def gstep(want_status=False):
            return ctx.decode_dict_batch_device(d_in.data_ptr(), d_in.numel(), descs, table.data_ptr(),
                                                table.numel() * 4, refs, args.gather_bytes, d_val.data_ptr(),
                                                my_pages * n, stream, want_status=want_status)

        assert (gstep(True) == 0).all()
        t2 = table.view(1 << bits, words)
        for i in range(min(my_pages, len(pages))):
            idx = torch.from_numpy(pages[(lo + i) % len(pages)][1].astype(np.int64)).to(dev)
            assert torch.equal(d_val.view(my_pages, n, words)[i], t2[idx]), "gathered values differ on page %d" % i
        for _ in range(args.warmup):
            gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gstep()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        vbytes = my_pages * n * args.gather_bytes
        print(json.dumps({
            "metric": "device-resident dictionary decode + gather GB/s (value bytes out)",
            "value": round(vbytes * args.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "values_per_second": round(my_pages * n * args.steps / elapsed / 1e9, 3),
            "config": {"workload": "%s: %d pages x %d values, bit width %d, %d-byte dictionary values" %
                       (args.workload, my_pages, n, bits, args.gather_bytes)}}), flush=True)
        ctx.close()
        return

    def step(want_status=False):
        return ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), my_pages * n, stream,
                                       want_status=want_status)

    # ---- parity gate: every page of the batch must equal the generator's values ------------------------
    if not args.no_parity:  # kernel-timing experiments with builds that do not write results
        status = step(want_status=True)
        assert (status == 0).all(), "non-OK page status"
        out2d = d_out.view(my_pages, n)
        for i in range(min(my_pages, len(pages))):
            expect = torch.from_numpy(pages[(lo + i) % len(pages)][1].view(np.int32)).to(dev)
            assert torch.equal(out2d[i], expect), "GPU output differs from the expected values on page %d" % i
        for i in range(len(pages), my_pages):
            if i % 61 == 0 or i == my_pages - 1:
                assert torch.equal(out2d[i], out2d[i % len(pages)]), "replica %d differs" % i
        expect = None

    for _ in range(args.warmup):
        step()
    ctx.enable_profiling(True)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    parse_ms, expand_ms, batches, expand_launches = ctx.kernel_times()
    ctx.enable_profiling(False)

    if distributed:
        rdev = dev if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(out_bytes)], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_out_bytes = float(tot.item())
    else:
        total_out_bytes = float(out_bytes)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_out_bytes * args.steps / elapsed / 1e9
        # a batch is decoded in `rounds` launches of each kernel, every launch covering 1/rounds of every page
        rounds = max(expand_launches // max(batches, 1), 1)
        expand_avg_ms = expand_ms / max(expand_launches, 1)
        # batches of many pages go through decode_kernel alone (one workgroup per page walks and expands);
        # the context times that launch in the place of expand_kernel's
        kernel_name = "decode_kernel" if parse_ms == 0.0 else "expand_kernel"
        bytes_per_launch = (in_bytes_alg + out_bytes) / rounds
        achieved = bytes_per_launch / (expand_avg_ms * 1e-3) / 1e9
        # HBM bytes of that kernel from the PMC passes recorded for this workload (FETCH_SIZE / WRITE_SIZE cannot be
        # collected from inside the process)
        traffic = None
        try:
            t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")))
            if t["workload"] == {"workload": args.workload, "pages_per_gpu": args.pages, "values_per_page": n,
                                 "num_bits": bits}:
                e = t[kernel_name]
                traffic = (e["fetch_bytes_per_step"] + e["write_bytes_per_step"]) // rounds
        except (OSError, KeyError, ValueError):
            pass
        result = {
            "metric": METRIC, "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": "RLE_DICTIONARY index pages (BASELINE configs[1]): %d pages/GPU x %d values, "
                            "bit width %d, mixed RLE/bit-packed runs" % (args.pages, n, bits)
                if args.workload == "dict" else "%s: %d pages/GPU x %d values, bit width %d" %
                (args.workload, args.pages, n, bits),
                "pages_per_gpu": args.pages, "values_per_page": n, "num_bits": bits,
                "encoded_bytes_per_gpu": in_bytes_alg, "decoded_bytes_per_gpu": out_bytes,
                "sharding": "pages partitioned across ranks, no collective",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "avg_launch_ms": round(expand_avg_ms, 4), "launches_per_step": rounds,
                "parse_kernel_ms_per_step": round(parse_ms / max(batches, 1), 4),
                "%s_ms_per_step" % kernel_name: round(expand_ms / max(batches, 1), 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pages, bits, args.cpu_seconds, args.workload)
        print(json.dumps(result), flush=True)
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
