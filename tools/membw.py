#!/usr/bin/env python3
"""Achievable HBM rates on this GPU with plain torch kernels: fill (write only), sum (read only), copy."""
import torch

dev = torch.device("cuda:0")
n = 1 << 30  # 4 GiB of int32
a = torch.empty(n, dtype=torch.int32, device=dev)
b = torch.empty(n, dtype=torch.int32, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: a.fill_(7))
print("fill  4 GiB: %.3f ms  %.0f GB/s written" % (ms, 4 * n / ms / 1e6))
ms = timed(lambda: a.sum())
print("sum   4 GiB: %.3f ms  %.0f GB/s read" % (ms, 4 * n / ms / 1e6))
ms = timed(lambda: b.copy_(a))
print("copy  4 GiB: %.3f ms  %.0f GB/s read + written" % (ms, 8 * n / ms / 1e6))
