#!/usr/bin/env python3
"""Raw copy rates between page-locked host memory and the GPU: each direction alone and both at once."""
import time, torch
n = 1 << 30
h_a = torch.empty(n, dtype=torch.uint8, pin_memory=True); h_b = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda"); d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def h2d():
    with torch.cuda.stream(s1): d_a.copy_(h_a, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_b.copy_(d_b, non_blocking=True)
def both(): h2d(); d2h()
print("H2D 1 GiB: %.1f GB/s" % (n / timed(h2d) / 1e9))
print("D2H 1 GiB: %.1f GB/s" % (n / timed(d2h) / 1e9))
t = timed(both); print("both at once: %.1f GB/s each way, %.1f GB/s in all" % (n / t / 1e9, 2 * n / t / 1e9))
for mb in (16, 64):
    m = mb << 20
    def chunks():
        with torch.cuda.stream(s2):
            for o in range(0, n, m): h_b[o:o + m].copy_(d_b[o:o + m], non_blocking=True)
    print("D2H in %d MiB pieces: %.1f GB/s" % (mb, n / timed(chunks) / 1e9))
