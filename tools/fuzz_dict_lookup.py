"""pq2_decode_dict_batch_device under every schedule against the CPU oracle's indices looked up with numpy, seed after
seed: batches of 1-300 dictionary pages (widths 0-17, long, mixed and short runs, streams that are cut, pages that ask
for more values than their stream holds), dictionaries that miss indices, 4-, 8- and 12-byte values at every
alignment the value size allows.  Statuses and every value word must match; nothing outside the pages is written.
python tools/fuzz_dict_lookup.py [first_seed] [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    import torch
    from _oracle import oracle as get
    from parquet2_amd import DICT_REF_DTYPE, Context, synth
    from parquet2_amd._lib import OK, OUT_OF_SPEC

    oracle = get()
    contexts = {}
    for mode, env in (("auto", None), ("scout", "scout"), ("tile", "tile"), ("page", "page"), ("ring", "ring")):
        if env:
            os.environ["PQ2_MODE"] = env
        contexts[mode] = Context(0)
        os.environ.pop("PQ2_MODE", None)
    schedules = {}
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        value_size = int(rng.choice([4, 8, 12]))
        words = value_size // 4
        n_pages = int(rng.choice([1, 2, 5, 17, 64, 120, 300]))
        top = int(rng.choice([8, 12, 16, 17]))
        batch = synth.PageBatch(align=1)
        pages, tables, refs = [], [], []
        dict_off = 0
        for p in range(n_pages):
            w = int(rng.integers(0, top + 1))
            n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20000)), int(rng.integers(20000, 150000))],
                               p=[0.2, 0.6, 0.2]))
            kind = int(rng.integers(0, 4))
            if w == 0:
                stream = np.zeros(0, dtype=np.uint8)
            elif kind == 0:
                stream = synth.adversarial(int(rng.integers(1, 1 << 30)), n, w)[0]
            else:
                stream = synth.dict_page(int(rng.integers(1, 1 << 30)), n, w, float(rng.choice([1.5, 8.0, 64.0, 1000.0])),
                                         float(rng.choice([2.0, 40.0, 256.0])))[0]
            if kind == 3 and len(stream) > 4:
                stream = stream[:int(rng.integers(1, len(stream)))]
            want = n + int(rng.integers(0, 60)) if rng.integers(0, 3) == 0 else max(1, n - int(rng.integers(0, min(n, 50))))
            batch.chunks.append(np.zeros(int(rng.integers(0, 4)), np.uint8))
            batch.in_bytes += len(batch.chunks[-1])
            batch.add(stream, w, want, out_align=int(rng.choice([1, 2, 4])))
            dict_len = min(max(1 << w, 1), 49152 // value_size)
            if rng.integers(0, 6) == 0:
                dict_len = int(rng.integers(0, dict_len + 1))
            table = rng.integers(0, 1 << 32, size=(dict_len, words), dtype=np.uint64).astype(np.uint32)
            dict_off += (-dict_off) % 8
            refs.append((dict_off, dict_len, 0))
            tables.append((dict_off, table))
            dict_off += table.nbytes
            pages.append((stream, w, want, table))
        data, descs, out_values = batch.finish()
        dict_bytes = np.zeros(dict_off + 8, dtype=np.uint8)
        for off, table in tables:
            dict_bytes[off:off + table.nbytes] = table.view(np.uint8).reshape(-1)
        # what every schedule has to produce
        want_status = np.zeros(n_pages, dtype=np.int32)
        want_out = np.full((max(out_values, 1), words), 0x5A5A5A5A, dtype=np.uint32)
        defined = np.ones(max(out_values, 1), dtype=bool)  # (the values of a page whose stream fails are not specified)
        for k, (d, (stream, w, n, table)) in enumerate(zip(descs, pages)):
            st, idx = oracle.hybrid_rle_decode(stream, w, n)
            o = int(d["out_off"])
            if st != OK:
                want_status[k] = st
                defined[o:o + n] = False
                continue
            ok = idx < len(table)
            want_status[k] = OK if ok.all() else OUT_OF_SPEC
            vals = np.zeros((n, words), dtype=np.uint32)
            vals[ok] = table[idx[ok]]
            want_out[o:o + n] = vals
        d_in = torch.from_numpy(data).cuda()
        d_dict = torch.from_numpy(dict_bytes).cuda()
        refs_np = np.array(refs, dtype=DICT_REF_DTYPE)
        for mode, ctx in contexts.items():
            d_out = torch.full((max(out_values, 1) * words,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            status = ctx.decode_dict_batch_device(d_in.data_ptr(), d_in.numel(), descs, d_dict.data_ptr(), d_dict.numel(),
                                                  refs_np, value_size, d_out.data_ptr(), out_values,
                                                  torch.cuda.current_stream().cuda_stream)
            got = d_out.cpu().numpy().view(np.uint32).reshape(-1, words)
            assert (status == want_status).all(), (seed, mode, np.nonzero(status != want_status)[0][:5])
            bad = np.nonzero(((got != want_out).any(axis=1)) & defined)[0]
            assert bad.size == 0, (seed, mode, ctx.last_schedule(), bad[:5])
            key = (mode, ctx.last_schedule())
            schedules[key] = schedules.get(key, 0) + 1
        print(f"seed {seed}: {n_pages} pages, {out_values} values of {value_size} bytes, widths up to {top}: "
              f"{int((want_status != 0).sum())} pages with a status, every schedule as the oracle", flush=True)
    for (mode, name), count in sorted(schedules.items()):
        print(f"  {mode:5s} {name:40s} {count} batches")


if __name__ == "__main__":
    main()
