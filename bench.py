#!/usr/bin/env python3
"""Benchmark of the device-resident hybrid-RLE / bit-packed decode.

    python bench.py --gpus N --steps K --warmup W [--workload ...]

One step = one pass of the decoder (pq2_decode_batch_device) over one batch of synthetic pages already resident
in HBM.  The default workload is BASELINE.json configs[1]: RLE_DICTIONARY index pages of 2^20 values, bit width
12, mixed RLE / bit-packed runs, 2048 pages per GPU (the per-GPU share of configs[4]), 256 distinct pages
replicated to distinct addresses.  Other workloads (parity-test shapes of the other configs; each prints its own
JSON line with the same fields):

    --workload def          configs[2]: definition levels, bit width 1, long RLE runs, 5 % nulls
                            (use --pages 10000 --values 65536)
    --workload defbits      the same pages decoded as DefLevelsDecoder::Bitmap has them: one validity bit per row
                            (pq2_decode_validity_bitmap_batch_device), with a roofline of its own
    --workload adversarial  configs[3]: alternating 1-8-value RLE runs and one-group bit-packed runs
                            (use --pages 16 --bits 3|17|31)
    --workload bitpacked    pages as the reference's own encoder writes them: one bit-packed run
    --workload mixed5       configs[4]'s page mix: 75 % config-2 pages, 25 % definition-level pages of the same size
    --workload bitunpack    configs[0]: bitpacked::Decoder<u32> over 2^20 values at every width 1..32 (and over
                            2^26 values, where the launch no longer dominates), next to the CPU port per width

With no --workload the line is the default workload's, and it carries the other single-GPU BASELINE configs under
"other_configs" (definition levels 10 000 x 65 536; adversarial pages at widths 3 / 17 / 31; 64 index pages, the batch
size SURVEY quotes config 2 on; the 75 / 25 page mix of configs[4]; bit-unpack at every width), each with its step
time, rate, dominant kernel and roofline fraction, and under "next_rows" one measurement each of the entry points
either side of the main decode (DELTA_BINARY_PACKED streams, flat nullable dictionary pages, a batch of few
dictionary pages with their values).  --no-other-configs leaves both out (A/B runs, counter passes).

The K timed steps run with nothing but the decoder's own work in the stream: `value` and `ms_per_step` are theirs.
The duration of the dominant kernel (`roofline`) is taken with HIP events over a second run of the same K steps
directly behind them -- every event holds the stream up for some 5 microseconds, 15 a step, which is a tenth of a
batch of few pages; the line says what a step took with the events in.

Before anything is timed, every distinct page of a batch is decoded by the CPU oracle (oracle/: the plain-C
restatement of the reference decoder) and the GPU's output of every page of the batch has to be identical to it.

With N > 1 the script runs under torch.distributed.run, one rank per GPU; every rank decodes its own shard of the
global page list (pq2_partition_pages), there is no data-path collective, and the reported value is the whole-job
rate (all ranks' output bytes / slowest rank's time).  Rank 0 prints one JSON line (the default workload, with the
configs[4] page mix under "other_configs").
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
    ap.add_argument("--distinct", type=int, default=256,
                    help="distinct generated pages of each kind (replicated on the device); every page has its own "
                         "run lengths, so the header walks of a batch are not in lock-step")
    ap.add_argument("--workload", default=None,
                    choices=["dict", "def", "defbits", "adversarial", "bitpacked", "mixed5", "bitunpack"],
                    help="one workload only (default: dict, with the other configs' results under other_configs)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--level-bits", type=int, default=1,
                    help="bit width of the level pages of --workload def (1: BASELINE configs[2]; 2, 3: columns "
                         "that are optional inside optional or repeated ones)")
    ap.add_argument("--page-latency", action="store_true",
                    help="one page per call, host memory to host memory (what a per-page HybridRleDecoder::try_new "
                         "costs), beside the CPU port on the same page")
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpus():
    """(CPUs of the machine, CPUs this process may run on)."""
    return os.cpu_count() or 1, len(os.sched_getaffinity(0))


LEVEL_BITS = 1


def make_pool(kind, count, values, bits):
    """`count` different pages of one kind: list of (stream ndarray, expected values ndarray, bit width)."""
    pool = []
    for p in range(count):
        if kind == "dict":
            s, v = synth.dict_page(synth.seed_for(2, p), values, bits)
        elif kind in ("def", "defbits"):
            s, v = synth.def_levels_page(synth.seed_for(3, p), values)
            bits = 1
            if LEVEL_BITS > 1:  # --level-bits: the same runs with the valid rows at the maximum level of that width
                v = v * np.uint32((1 << LEVEL_BITS) - 1)
                s, bits = synth.encode_hybrid(v, LEVEL_BITS), LEVEL_BITS
        elif kind == "adversarial":
            s, v = synth.adversarial(synth.seed_for(4, p), values, bits)
        else:  # one bit-packed run, as the reference's own encoder writes pages
            from parquet2_amd.encoding.hybrid_rle import encode_u32
            v = synth.uniform_values(synth.seed_for(1, p), values, bits)
            s = np.frombuffer(encode_u32(v, bits), dtype=np.uint8)
        pool.append((np.ascontiguousarray(s, dtype=np.uint8), v, bits))
    return pool


_ORACLE = None


def load_oracle():
    """(the oracle, the flags it was built with); built once per process."""
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = _build_oracle()
    return _ORACLE


def _build_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle

    tmp = tempfile.mkdtemp(prefix="pq2oracle_")
    try:
        path = _oracle.build(native=True, out_dir=tmp)  # -O3 -march=native for the host we are on
        flags = "gcc -O3 -march=native"
    except Exception:
        path = None
        flags = "gcc -O3"
    return _oracle.Oracle(path), flags


def cpu_baseline(sample, seconds, what):
    """The CPU oracle (plain-C port of the reference decoder) timed on this host: one thread, and all the
    threads this box gives us, each decoding whole pages.  Also the parity checker of the sample.
    sample: list of (stream, values, bits)."""
    orc, flags = load_oracle()
    n_values = sum(len(v) for _, v, _ in sample)
    for stream, vals, bits in sample[:2]:
        st, out = orc.hybrid_rle_decode(stream, bits, len(vals))
        assert st == 0 and (out == vals).all(), "oracle disagrees with the generator"
    t_budget = max(1.0, seconds / 2)
    best = None
    t_start = time.perf_counter()
    reps_done = 0
    while time.perf_counter() - t_start < t_budget:
        t = sum(orc.time_hybrid_rle_decode(s, b, len(v), 1) for s, v, b in sample)
        best = t if best is None else min(best, t)
        reps_done += 1
    one = 4.0 * n_values / best / 1e9
    # all threads of our CPU share (16 on a one-GPU box), pages sharded over threads; ctypes drops the GIL
    n_thr = max(1, min(16, len(os.sched_getaffinity(0))))
    results = [0.0] * n_thr
    reps = max(1, reps_done // 2)

    def work(i):
        s, v, b = sample[i % len(sample)]
        results[i] = orc.time_hybrid_rle_decode(s, b, len(v), reps) / reps

    thr = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    t0 = time.perf_counter()
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    wall = time.perf_counter() - t0
    many = 4.0 * sum(len(sample[i % len(sample)][1]) for i in range(n_thr)) * reps / wall / 1e9
    return {
        "value": round(one, 3), "unit": "GB/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
        "sample": "%d pages of the workload (%s), C port of HybridRleDecoder drained per value (%s), best of %d passes"
                  % (len(sample), what, flags, reps_done),
        "all_threads": {"value": round(many, 3), "cores": n_thr},
        "host_cpus": host_cpus()[0], "cpus_this_process_may_use": host_cpus()[1],
    }


def traffic_from_profiles(workload_key, kernel_name):
    """HBM bytes of `kernel_name` per launch from the PMC passes recorded for exactly this workload under
    profiles/ (FETCH_SIZE / WRITE_SIZE cannot be collected from inside the process): (bytes or None, source)."""
    import glob

    # (the latest first: r2_step3_traffic.json before r1_traffic.json)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*traffic.json")), reverse=True):
        name = os.path.basename(path)
        if "gather" in name:
            continue
        try:
            t = json.load(open(path))
            if "+" in kernel_name and kernel_name not in t:
                kernel_name = "all_pq2_kernels"  # a schedule of several kernels: what they move together
            if t["workload"] == workload_key and kernel_name in t:
                e = t[kernel_name]
                per_launch = (e["fetch_bytes_per_step"] + e["write_bytes_per_step"]) // max(e["launches_per_step"], 1)
                return per_launch, "profiles/%s (%s: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of an earlier run " \
                                   "of this command; not measured in this run)" % (name, kernel_name)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def bench_page_latency(ctx):
    """HybridRleDecoder::try_new(data, w, n).collect() one page at a time through pq2_decode_batch_host (stream up,
    decode, values and status down) against the CPU port on the same page: where a per-page binding starts to pay."""
    orc, flags = load_oracle()
    rows = []
    for n in (10, 1000, 16384, 65536, 262144, 1 << 20, 1 << 22):
        stream_bytes, vals = synth.dict_page(synth.seed_for(2, 7), n, 12)
        out, st = ctx.decode_page_host(stream_bytes, 12, n)
        assert st == 0 and np.array_equal(out, vals)
        st, ref = orc.hybrid_rle_decode(stream_bytes, 12, n)
        assert st == 0 and np.array_equal(ref, vals)
        reps = 200 if n <= 65536 else 30
        gpu, cpu = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.decode_page_host(stream_bytes, 12, n)
            gpu.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            orc.hybrid_rle_decode(stream_bytes, 12, n)
            cpu.append(time.perf_counter() - t0)
        rows.append({"values": n, "stream_bytes": int(len(stream_bytes)), "gpu_call_us": round(1e6 * float(np.median(gpu)), 1),
                     "cpu_port_us": round(1e6 * float(np.median(cpu)), 1)})
    return {"metric": "one page per call, host memory to host memory, microseconds (median)", "config": {
        "workload": "RLE_DICTIONARY index pages, bit width 12, mixed runs; pageable host buffers, numpy wrapper included "
                    "on both sides (an output array is allocated per call)"},
        "cpu": "C port of HybridRleDecoder drained per value (%s), one thread" % flags, "pages": rows}


def bench_bitunpack(args, dev, ctx, stream):
    """configs[0]: bitpacked::Decoder<u32>::try_new(bytes, w, N) drained, w = 1..32 (benches/decode_bitpacking.rs)."""
    from parquet2_amd.encoding import bitpacked

    results = {"small": [], "large": []}
    cpu = []
    orc, flags = (None, None) if args.no_cpu_baseline else load_oracle()
    for w in range(1, 33):
        for tag, n in (("small", 1 << 20), ("large", 1 << 26)):
            vals = synth.uniform_values(synth.seed_for(1, w), 1 << 20, w)
            packed = bitpacked.encode(vals, w)
            if tag == "small":
                # the reference's bench pattern at this width as well: parity only
                pattern = synth.bench_pattern(n * w // 8)
                d_pat = torch.from_numpy(np.concatenate([pattern, np.zeros(16, np.uint8)])).to(dev)
                out = torch.zeros(n, dtype=torch.int32, device=dev)
                ctx.bitunpack_u32_device(d_pat.data_ptr(), pattern.size, w, n, out.data_ptr(), stream)
                if orc is not None:
                    st, exp = orc.bitunpack_u32(pattern, w, n)
                    assert st == 0 and (out.cpu().numpy().view(np.uint32) == exp).all(), "pattern differs at w=%d" % w
                    cpu.append(round(4.0 * n / (orc.time_bitunpack_u32(packed, w, n, 5) / 5) / 1e9, 3))
            reps = n >> 20
            d_packed = torch.from_numpy(np.concatenate([np.tile(packed[:(1 << 20) * w // 8], reps),
                                                        np.zeros(16, np.uint8)])).to(dev)
            out = torch.zeros(n, dtype=torch.int32, device=dev)
            ctx.bitunpack_u32_device(d_packed.data_ptr(), n * w // 8, w, n, out.data_ptr(), stream)
            got = out.view(reps, 1 << 20)
            expect = torch.from_numpy(vals.view(np.int32)).to(dev)
            assert torch.equal(got[0], expect) and torch.equal(got[reps - 1], expect), "bit-unpack differs at w=%d" % w
            for _ in range(args.warmup):
                ctx.bitunpack_u32_device(d_packed.data_ptr(), n * w // 8, w, n, out.data_ptr(), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                ctx.bitunpack_u32_device(d_packed.data_ptr(), n * w // 8, w, n, out.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            results[tag].append((w, ms, 4.0 * n / ms / 1e6, (4.0 + w / 8.0) * n / ms / 1e6))
            del d_packed, out
    large = results["large"]
    worst = min(large, key=lambda r: r[3])
    total_out = sum(4.0 * (1 << 26) for _ in large)
    total_ms = sum(r[1] for r in large)
    line = {
        "metric": "device-resident bit-unpack GB/s (output bytes), bitpacked::Decoder<u32>, widths 1..32",
        "value": round(total_out / total_ms / 1e6, 2), "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(total_ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "bitunpack (BASELINE configs[0]): uniform values, every width 1..32; a step is one "
                               "call per width over 2^26 values (value, roofline) -- and over 2^20 values, the "
                               "reference bench's size, where one call is a few microseconds of launch",
                   "gpu_2p26_values_gbs_out_by_width": [round(r[2], 1) for r in large],
                   "gpu_2p26_values_gbs_in_plus_out_by_width": [round(r[3], 1) for r in large],
                   "gpu_2p20_values_us_per_call_by_width": [round(r[1] * 1e3, 2) for r in results["small"]],
                   "gpu_2p20_values_gbs_out_by_width": [round(r[2], 1) for r in results["small"]]},
        "roofline": {"bound": "hbm", "kernel": "bitunpack_kernel", "achieved": round(worst[3], 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(worst[3] / HBM_PEAK_GBS, 4), "traffic": None,
                     "traffic_source": None,
                     "note": "the width furthest from the roofline (w=%d); algorithmic bytes = (4 + w/8) per value; "
                             "launch duration between events on the launch stream" % worst[0]},
    }
    if orc is not None:
        line["cpu_baseline"] = {"value": round(sum(cpu) / len(cpu), 3), "unit": "GB/s", "cores": 1, "kind": "port",
                                "cpu_model": cpu_model(),
                                "sample": "2^20 values per width, C port of bitpacked::Decoder<u32> drained per "
                                          "value (%s), mean over widths 1..32 of 5 passes each" % flags,
                                "gbs_out_by_width": cpu}
    return line


NAMES = {"dict": "RLE_DICTIONARY index pages (BASELINE configs[1]): %d pages/GPU x %d values, bit width %s, "
                 "mixed RLE/bit-packed runs",
         "def": "definition levels (BASELINE configs[2]): %d pages/GPU x %d values, bit width %s, long RLE "
                "runs, 5 %% nulls in bursts",
         "defbits": "definition levels (BASELINE configs[2]) as validity bitmaps, one bit per row "
                    "(DefLevelsDecoder::Bitmap): %d pages/GPU x %d rows, bit width %s",
         "adversarial": "adversarial short runs (BASELINE configs[3]): %d pages/GPU x %d values, bit width %s, "
                        "alternating 1-8-value RLE runs and one-group bit-packed runs",
         "bitpacked": "pages of one bit-packed run (the reference's encoder): %d pages/GPU x %d values, bit "
                      "width %s",
         "mixed5": "page mix of BASELINE configs[4]: %d pages/GPU x %d values, 75 %% dictionary-index pages "
                   "and 25 %% definition-level pages (bit widths %s)"}


def run_pages(a, env):
    """One workload of pages (`a`: workload, pages, values, bits, distinct, steps, warmup and the switches of
    parse_args) on this rank's GPU; rank 0 gets the result line, the others None."""
    ctx, dev, stream = env["ctx"], env["dev"], env["stream"]
    rank, world, dist, backend = env["rank"], env["world"], env["dist"], env["backend"]
    distributed = world > 1
    n = a.values
    total_pages = world * a.pages
    distinct = max(1, min(a.distinct, total_pages))
    # pools of distinct pages; `kind_of[g]` = pool of global page g, `slot_of[g]` = its entry of that pool
    if a.workload == "mixed5":  # every fourth page holds definition levels, the others dictionary indices
        pools = [make_pool("dict", distinct, n, a.bits), make_pool("def", max(1, distinct // 4), n, 1)]
        kind_of = np.where(np.arange(total_pages) % 4 == 3, 1, 0)
    else:
        pools = [make_pool(a.workload, distinct, n, a.bits)]
        kind_of = np.zeros(total_pages, dtype=np.int64)
    slot_of = np.zeros(total_pages, dtype=np.int64)
    for k, pool in enumerate(pools):
        sel = np.nonzero(kind_of == k)[0]
        slot_of[sel] = np.arange(len(sel)) % len(pool)
    pool_base = np.cumsum([0] + [len(p) for p in pools])[:-1]
    flat = [e for pool in pools for e in pool]
    flat_of = pool_base[kind_of] + slot_of

    # global descriptor list; rank r takes the contiguous shard pq2_partition_pages gives it
    g_descs = np.zeros(total_pages, dtype=PAGE_DESC_DTYPE)
    g_descs["num_values"] = n
    g_descs["in_len"] = [len(flat[f][0]) for f in flat_of]
    g_descs["num_bits"] = [flat[f][2] for f in flat_of]
    bounds = partition_pages(g_descs, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    my_pages = hi - lo

    # device-resident input: the distinct streams once, then one replica per page at its own address
    stride = (max(len(s) for s, _, _ in flat) + 16 + 255) // 256 * 256
    h_in = np.zeros((len(flat), stride), dtype=np.uint8)
    for i, (s, _, _) in enumerate(flat):
        h_in[i, :len(s)] = s
    d_flat = torch.from_numpy(h_in).to(dev)
    my_flat = torch.from_numpy(flat_of[lo:hi]).to(dev)
    d_in = d_flat[my_flat].contiguous()  # [my_pages, stride]
    del d_flat
    as_bits = a.workload == "defbits"  # one validity bit per row instead of one u32 per value
    assert not as_bits or n % 32 == 0
    d_out = torch.empty(my_pages * n // 32 if as_bits else my_pages * n, dtype=torch.int32, device=dev)
    descs = g_descs[lo:hi].copy()
    descs["in_off"] = np.arange(my_pages, dtype=np.uint64) * stride
    descs["out_off"] = np.arange(my_pages, dtype=np.uint64) * (n // 8 if as_bits else n)  # bytes | values
    in_bytes_alg = int(descs["in_len"].sum())
    out_bytes = my_pages * n // 8 if as_bits else 4 * my_pages * n
    bits_label = "/".join(str(b) for b in sorted(set(int(b) for b in descs["num_bits"]), reverse=True))

    if a.host_path:  # end to end, host memory to host memory
        h_in_t = torch.empty(d_in.numel(), dtype=torch.uint8, pin_memory=True)
        h_in_t.copy_(d_in.reshape(-1).cpu())
        h_out_t = torch.zeros(my_pages * n, dtype=torch.int32, pin_memory=True)
        h_in_np, h_out_np = h_in_t.numpy(), h_out_t.numpy().view(np.uint32)
        ctx.decode_batch_host(h_in_np, descs, my_pages * n, out=h_out_np)
        for i in range(min(my_pages, 2 * len(flat))):
            assert np.array_equal(h_out_np[i * n:(i + 1) * n], flat[flat_of[lo + i]][1]), "host path differs"
        t0 = time.perf_counter()
        for _ in range(a.steps):
            _, status = ctx.decode_batch_host(h_in_np, descs, my_pages * n, out=h_out_np)
        elapsed = time.perf_counter() - t0
        assert (status == 0).all()
        return {
            "metric": "host-to-host hybrid-RLE/bitpack decode GB/s (output bytes), copies included",
            "value": round(out_bytes * a.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": 1,
            "steps": a.steps, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "bytes_over_pcie_per_step": int(d_in.numel()) + out_bytes,
            "pcie_gbs": round((d_in.numel() + out_bytes) * a.steps / elapsed / 1e9, 2),
            "config": {"workload": "%s: %d pages x %d values, bit width %s, page-locked host buffers" %
                       (a.workload, my_pages, n, bits_label)}}

    if a.gather_bytes:  # decode + dictionary gather, device-resident
        from parquet2_amd import DICT_REF_DTYPE
        bits = int(descs["num_bits"][0])
        words = a.gather_bytes // 4
        table = torch.randint(-2**31, 2**31 - 1, ((1 << bits) * words,), dtype=torch.int32, device=dev)
        refs = np.zeros(my_pages, dtype=DICT_REF_DTYPE)
        refs["dict_len"] = 1 << bits
        d_val = torch.empty(my_pages * n * words, dtype=torch.int32, device=dev)

        def gstep(want_status=False):
            return ctx.decode_dict_batch_device(d_in.data_ptr(), d_in.numel(), descs, table.data_ptr(),
                                                table.numel() * 4, refs, a.gather_bytes, d_val.data_ptr(),
                                                my_pages * n, stream, want_status=want_status)

        assert (gstep(True) == 0).all()
        t2 = table.view(1 << bits, words)
        for i in range(min(my_pages, 16)):
            idx = torch.from_numpy(flat[flat_of[lo + i]][1].astype(np.int64)).to(dev)
            assert torch.equal(d_val.view(my_pages, n, words)[i], t2[idx]), "gathered values differ on page %d" % i
        for _ in range(a.warmup):
            gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gstep()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        vbytes = my_pages * n * a.gather_bytes
        return {
            "metric": "device-resident dictionary decode + gather GB/s (value bytes out)",
            "value": round(vbytes * a.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": a.steps,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "values_per_second": round(my_pages * n * a.steps / elapsed / 1e9, 3),
            "config": {"workload": "%s: %d pages x %d values, bit width %d, %d-byte dictionary values" %
                       (a.workload, my_pages, n, bits, a.gather_bytes)}}

    def step(want_status=False):
        if as_bits:
            r = ctx.decode_validity_bitmap_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(),
                                                        out_bytes, stream, want_counts=want_status)
            return r[2] if want_status else None
        return ctx.decode_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_out.data_ptr(), my_pages * n, stream,
                                       want_status=want_status)

    # ---- parity gate: every distinct page of this rank's batch decoded by the CPU oracle; the GPU's output of
    # every page of the batch has to be identical to it --------------------------------------------------------
    gate = None
    if not a.no_parity:  # (skipped only by kernel-timing experiments with builds that do not write results)
        orc, _ = load_oracle()
        status = step(want_status=True)
        assert (status == 0).all(), "non-OK page status"
        out2d = d_out.view(my_pages, n // 32 if as_bits else n)
        first_of = {}
        for i in range(my_pages):
            f = int(flat_of[lo + i])
            if f not in first_of:  # the first replica of a distinct page against the oracle's values ...
                first_of[f] = i
                stream_f, vals_f, bits_f = flat[f]
                st_o, want = orc.hybrid_rle_decode(stream_f, bits_f, n)
                assert st_o == 0 and np.array_equal(want, vals_f), "oracle and generator differ on distinct page %d" % f
                if as_bits:  # the levels of a flat nullable column are its validity: row r at bit r % 8 of byte r / 8
                    want = np.packbits(want.astype(np.uint8), bitorder="little").view(np.uint32)
                expect = torch.from_numpy(want.view(np.int32)).to(dev)
                assert torch.equal(out2d[i], expect), "GPU output differs from the oracle's on page %d" % i
            else:  # ... every other replica against the first
                assert torch.equal(out2d[i], out2d[first_of[f]]), "replica %d differs" % i
        expect = None
        gate = "%d pages: %d distinct ones byte-identical to the CPU oracle, every replica to its first" % (
            my_pages, len(first_of))

    for _ in range(a.warmup):
        step()
    # The timed region: K steps, nothing else in the stream.  The kernels' own durations are taken with HIP events over
    # a second run of the same K steps right behind it: events, whether recorded in the stream or handed to a kernel's
    # launch, hold the stream up for some 5 microseconds each -- 15 a step, a tenth of a batch of few pages -- and
    # `value` is the rate of the work, not of the work and its measurement.
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.enable_profiling(True)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    elapsed_with_events = time.perf_counter() - t1
    parse_ms, expand_ms, batches, expand_launches = ctx.kernel_times()
    ctx.enable_profiling(False)
    schedule = ctx.last_schedule()
    kernel_name = schedule.split("+")[-1]  # the kernel that writes the values ("scout+expand_kernel": the scout
    #                                        kernels walk the headers in front of it; their time is parse_kernel_ms)

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
    del d_in, d_out
    if rank != 0:
        return None

    ms_per_step = 1e3 * elapsed / a.steps
    value = total_out_bytes * a.steps / elapsed / 1e9
    # a batch is decoded in `rounds` launches of the kernel that writes the values (one, except for batches of
    # few large pages, which go through parse_kernel + expand_kernel in rounds)
    rounds = max(expand_launches // max(batches, 1), 1)
    expand_avg_ms = expand_ms / max(expand_launches, 1)
    bytes_per_launch = (in_bytes_alg + out_bytes) / rounds
    scouted = "+" in schedule
    if scouted:
        # the header walk (seven small launches) and the expansion are the hot path together: priced as one unit
        kernel_name = schedule
        expand_avg_ms = (parse_ms + expand_ms) / max(batches, 1)
    achieved = bytes_per_launch / (expand_avg_ms * 1e-3) / 1e9
    workload_key = {"workload": a.workload, "pages_per_gpu": a.pages, "values_per_page": n,
                    "num_bits": a.bits if a.workload not in ("def", "defbits") else LEVEL_BITS}
    traffic, traffic_source = traffic_from_profiles(workload_key, kernel_name)
    result = {
        "metric": "device-resident validity-bitmap decode, 10^9 rows/s (DefLevelsDecoder::Bitmap)" if as_bits else METRIC,
        "value": round(value * 8 if as_bits else value, 2), "unit": "Grows/s" if as_bits else "GB/s", "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u1" if as_bits else "u32", "data": "synthetic",
        "config": {
            "workload": NAMES[a.workload] % (a.pages, n, bits_label),
            "pages_per_gpu": a.pages, "values_per_page": n, "num_bits": bits_label,
            "distinct_pages": len(flat),
            "encoded_bytes_per_gpu": in_bytes_alg, "decoded_bytes_per_gpu": out_bytes,
            "sharding": "pages partitioned across ranks, no collective",
            "parity_gate": gate,
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel_name, "schedule": schedule, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "avg_launch_ms": round(expand_avg_ms, 4), "launches_per_step": rounds,
            "kernel_times_from": "HIP events over a second run of the same %d steps right behind the timed ones "
                                 "(%.4f ms per step with the events in the stream)" % (a.steps, 1e3 * elapsed_with_events / a.steps),
            "parse_kernel_ms_per_step": round(parse_ms / max(batches, 1), 4),
            "%s_ms_per_step" % schedule.split("+")[-1]: round(expand_ms / max(batches, 1), 4),
        },
    }
    if world == 1 and not a.no_cpu_baseline and not as_bits:  # (the port that is timed expands levels to u32)
        # a bounded sample of the same pages: about cpu_seconds of decoding
        per_pool = max(1, 16 // len(pools))
        sample = [e for pool in pools for e in pool[:per_pool]]
        while len(sample) > 1 and sum(len(v) for _, v, _ in sample) > (16 << 20):
            sample = sample[:len(sample) // 2]
        result["cpu_baseline"] = cpu_baseline(sample, a.cpu_seconds, a.workload)
    return result


def bench_next_rows(ctx, dev, stream):
    """The entry points either side of the main decode, one measurement each (one GPU, inputs resident, what the
    tools under tools/ measure at length): DELTA_BINARY_PACKED streams into i64, flat nullable dictionary pages into
    values + validity, and a batch of few dictionary pages with the lookup inside expand_kernel."""
    from parquet2_amd import DICT_REF_DTYPE, PAGE_DESC_DTYPE
    from parquet2_amd.encoding import delta_bitpacked

    rows_out = []
    # -- delta_bitpacked::Decoder: 4096 streams of 65 537 values, deltas of 10 bits (the library's encoder: blocks of
    #    128 values in one mini-block, the reference's layout, src/encoding/delta_bitpacked/encoder.rs:11-78)
    rng = np.random.default_rng(1)
    n_streams, n_values = 4096, 65537
    v = np.cumsum(rng.integers(0, 1 << 10, size=n_values, dtype=np.int64))
    one = np.frombuffer(delta_bitpacked.encode(v), dtype=np.uint8)
    stride = (one.size + 15) & ~15
    blob = np.zeros((n_streams, stride), dtype=np.uint8)
    blob[:, :one.size] = one
    descs = np.zeros(n_streams, dtype=PAGE_DESC_DTYPE)
    descs["in_off"] = np.arange(n_streams, dtype=np.uint64) * stride
    descs["in_len"] = one.size
    descs["num_values"] = n_values
    descs["out_off"] = np.arange(n_streams, dtype=np.uint64) * n_values
    d_in = torch.from_numpy(blob.reshape(-1)).to(dev)
    d_out = torch.zeros(n_streams * n_values, dtype=torch.int64, device=dev)
    best = 1e9
    for r in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status, counts, consumed = ctx.delta_bitpacked_decode_batch_device(d_in.data_ptr(), blob.size, descs,
                                                                          d_out.data_ptr(), d_out.numel())
        dt = time.perf_counter() - t0
        if r:
            best = min(best, dt)
    got = d_out.view(n_streams, n_values)
    ok = bool((status == 0).all() and (counts == n_values).all() and (consumed == one.size).all() and
              (got[0].cpu().numpy() == v).all() and (got[-1].cpu().numpy() == v).all())
    rows_out.append({"entry_point": "pq2_delta_bitpacked_decode_batch_device",
                     "workload": "%d DELTA_BINARY_PACKED streams x %d values, 10-bit deltas, blocks of 128 / 1 mini-block" % (n_streams, n_values),
                     "ms_per_call": round(best * 1e3, 4), "includes": "the download of the per-stream results",
                     "values_per_second_G": round(n_streams * n_values / best / 1e9, 1),
                     "stream_plus_value_GBps": round(n_streams * (one.size + 8 * n_values) / best / 1e9, 1),
                     "parity_gate": "generator values, first and last stream: " + ("equal" if ok else "DIFFERENT")})
    del d_in, d_out, got
    torch.cuda.empty_cache()

    # -- flat nullable dictionary pages: 4096 pages x 2^18 rows, levels as configs[2], 12-bit indices, 4-byte values
    pages_n, rows, value_size = 4096, 1 << 18, 4
    distinct = [(synth.def_levels_page(synth.seed_for(3, p), rows), synth.dict_page(synth.seed_for(2, p), rows, 12))
                for p in range(8)]
    chunks, at, place = [], 0, []
    for (ds, _), (xs, _) in distinct:
        place.append((at, len(ds), at + len(ds), len(xs)))
        chunks += [ds, xs]
        at += len(ds) + len(xs)
        chunks.append(np.zeros((-at) % 16, dtype=np.uint8))
        at += (-at) % 16
    one = np.concatenate(chunks + [np.zeros(16, dtype=np.uint8)])
    d_in = torch.from_numpy(one).to(dev).repeat((pages_n + 7) // 8)
    defs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
    idxs = np.zeros(pages_n, dtype=PAGE_DESC_DTYPE)
    for i in range(pages_n):
        o, base = place[i % 8], (i // 8) * one.size
        defs[i] = (base + o[0], o[1], 1, rows, i * rows)
        idxs[i] = (base + o[2], o[3], 12, rows, 0)
    table = torch.randint(-2**31, 2**31 - 1, (4096,), dtype=torch.int32, device=dev)
    refs = np.zeros(pages_n, dtype=DICT_REF_DTYPE)
    refs["dict_len"] = 4096
    d_out = torch.empty(pages_n * rows, dtype=torch.int32, device=dev)
    d_val = torch.empty(pages_n * rows, dtype=torch.uint8, device=dev)
    levels_max = np.ones(pages_n, np.uint32)

    def step(want=False):
        return ctx.decode_optional_dict_batch_device(d_in.data_ptr(), d_in.numel(), defs, idxs, levels_max,
                                                     table.data_ptr(), table.numel() * 4, refs, value_size,
                                                     d_out.data_ptr(), d_val.data_ptr(), pages_n * rows, stream,
                                                     want_status=want)

    ok = bool((step(True) == 0).all())
    lv, ix = distinct[3][0][1], distinct[3][1][1]
    valid = lv == 1
    exp = torch.zeros(rows, dtype=torch.int32)
    exp[torch.from_numpy(valid)] = table.cpu()[torch.from_numpy(ix[:int(valid.sum())].astype(np.int64))]
    ok = ok and torch.equal(d_out.view(pages_n, rows)[3].cpu(), exp) and \
        torch.equal(d_val.view(pages_n, rows)[3].cpu(), torch.from_numpy(valid.astype(np.uint8)))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    rows_out.append({"entry_point": "pq2_decode_optional_dict_batch_device",
                     "workload": "%d flat nullable dictionary pages x %d rows, one-bit levels with 5 %% nulls in bursts, 12-bit indices, %d-byte values" % (pages_n, rows, value_size),
                     "ms_per_step": round(el * 1e3, 4), "schedule": ctx.last_schedule(),
                     "rows_per_second_G": round(pages_n * rows / el / 1e9, 1),
                     "value_plus_validity_GBps": round(pages_n * rows * (value_size + 1) / el / 1e9, 1),
                     "parity_gate": "generator values and validity, one page: " + ("equal" if ok else "DIFFERENT")})
    del d_in, d_out, d_val
    torch.cuda.empty_cache()
    return rows_out


def brief(line):
    """What other_configs keeps of a workload's line."""
    r = line["roofline"]
    out = {"workload": line["config"]["workload"], "ms_per_step": line["ms_per_step"], "value": line["value"],
           "unit": line["unit"],
           "roofline": {k: r[k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_source") if k in r}}
    for k in ("avg_launch_ms", "launches_per_step", "algorithmic_bytes_per_launch", "note"):
        if k in r:
            out["roofline"][k] = r[k]
    if "parity_gate" in line["config"]:
        out["parity_gate"] = line["config"]["parity_gate"]
    return out


def main():
    args = parse_args()
    global LEVEL_BITS
    LEVEL_BITS = args.level_bits
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    dist = None
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
    ctx = Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    env = {"ctx": ctx, "dev": dev, "stream": stream, "rank": rank, "world": world, "dist": dist, "backend": backend}

    if args.page_latency:
        print(json.dumps(bench_page_latency(ctx)), flush=True)
        ctx.close()
        return
    if args.workload == "bitunpack":
        print(json.dumps(bench_bitunpack(args, dev, ctx, stream)), flush=True)
        ctx.close()
        return

    others = args.workload is None and not args.no_other_configs and not args.host_path and not args.gather_bytes
    if args.workload is None:
        args.workload = "dict"
    result = run_pages(args, env)
    if others:
        def sub(**kw):
            a = argparse.Namespace(**vars(args))
            a.no_cpu_baseline = True
            for k, v in kw.items():
                setattr(a, k, v)
            return a

        # the other BASELINE configs (one GPU: all of them; several ranks: the page mix configs[4] names)
        cases = [sub(workload="mixed5")]
        if world == 1:
            cases = [sub(workload="def", pages=10000, values=65536, bits=1),
                     sub(workload="defbits", pages=10000, values=65536, bits=1),
                     sub(workload="adversarial", pages=16, bits=3), sub(workload="adversarial", pages=16, bits=17),
                     sub(workload="adversarial", pages=16, bits=31), sub(workload="dict", pages=64)] + cases
        lines = []
        for a in cases:
            torch.cuda.empty_cache()
            line = run_pages(a, env)
            if line is not None:
                lines.append(brief(line))
        if world == 1:
            torch.cuda.empty_cache()
            b = argparse.Namespace(**vars(args))
            b.no_cpu_baseline = True
            b.steps = min(args.steps, 10)
            lines.append(brief(bench_bitunpack(b, dev, ctx, stream)))
        if result is not None:
            result["other_configs"] = lines
        if world == 1 and result is not None:
            # beside the BASELINE configs: the entry points either side of the main decode
            torch.cuda.empty_cache()
            next_rows = bench_next_rows(ctx, dev, stream)
            torch.cuda.empty_cache()
            line = run_pages(sub(workload="dict", pages=64, gather_bytes=4), env)
            if line is not None:
                next_rows.append({"entry_point": "pq2_decode_dict_batch_device", "workload": line["config"]["workload"],
                                  "ms_per_step": line["ms_per_step"], "values_per_second_G": line.get("values_per_second"),
                                  "schedule": ctx.last_schedule()})
            result["next_rows"] = next_rows
    if result is not None:
        print(json.dumps(result), flush=True)
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
