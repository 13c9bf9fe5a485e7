// ring_kernel for plain u32 output: the launcher (the kernel itself: pq2_ring.h).
#include <hip/hip_ext.h>

#include "pq2_ring.h"

#ifdef PQ2_RING_TIMING
// copies the clock sums to `out` (16 values) and clears them
extern "C" int pq2_ring_clocks(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pq2::g_ring_clocks), sizeof(pq2::g_ring_clocks)) != hipSuccess) return 1;
    unsigned long long zero[32] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(pq2::g_ring_clocks), zero, sizeof(zero)) != hipSuccess;
}
#endif

#ifdef PQ2_RING_TRACE
// copies the per-page trace (3 words per page: begin, end, place) of the first `n_pages` pages to `out`
extern "C" int pq2_ring_trace(unsigned long long *out, uint32_t n_pages) {
    if (n_pages > pq2::kRingTracePages) n_pages = pq2::kRingTracePages;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pq2::g_ring_trace), (size_t)n_pages * 24u) != hipSuccess;
}
#endif

extern "C" uint32_t pq2_ring_ticket_bytes() { return pq2::kRingCuTickets * 4u; }

// One workgroup per page, the stream staged once through an LDS ring of 32 KiB, whatever the width: a ring of
// 64 KiB for pages wider than 16 bits (so that the ring holds the bytes of the three tiles in hand, up to 48 KiB at
// width 32) leaves room for two workgroups per compute unit instead of four and is 13-17 % slower at every such
// width.  `max_bits`: the largest width of a page of the batch.
// `epoch`: a count of the launches that use `cu_tickets` (see ring_pick_walker).
// `go`: null, or a word of device memory that says, by the time the kernel runs, whether it is to run at all.
// `begin`, `end`: null, or events that take the kernel's own start and end (pq2_ctx_enable_profiling: events recorded
// around a launch also see the few microseconds between the launch's neighbours in the stream and the kernel).
extern "C" hipError_t pq2_launch_ring(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets,
                                      uint32_t epoch, uint32_t max_bits, const uint32_t *tile_mask, const uint32_t *go,
                                      hipEvent_t begin, hipEvent_t end, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    uint32_t ring_bytes = 32768u;
    if (const char *v = getenv("PQ2_RING_BYTES")) {  // tuning knob: 16384, 32768 or 65536
        const uint32_t b = (uint32_t)atoi(v);
        if (b == 16384u || b == 32768u || b == 65536u) ring_bytes = b;
    }
    const uint32_t lds_bytes = pq2::ring_lds_bytes(ring_bytes);
    // a batch of nothing but levels: whole tiles as spans (see kRingWholeTileBits)
    const uint32_t flags = (max_bits != 0u && max_bits <= pq2::kRingWholeTileBits ? pq2::kRingFlagWholeTileSpans : 0u) |
                           (n_pages >= pq2::kRingTurnsMinPages && max_bits > pq2::kRingWholeTileBits && max_bits <= pq2::kRingTurnsMaxBits
                                ? pq2::kRingFlagTurns
                                : 0u);
    // (the launch with events only where they are asked for: a plain launch can be captured into a graph)
#define PQ2_RING_LAUNCH(WIDE)                                                                                              \
    if (begin || end)                                                                                                      \
        hipExtLaunchKernelGGL((pq2::ring_kernel<WIDE, 0>), dim3(n_pages), dim3(pq2::kRingThreads), lds_bytes, stream,      \
                              begin, end, 0, (const pq2::PageDev *)pages, n_pages, status, cu_tickets, ring_bytes,         \
                              tile_mask, (const pq2::RingDict *)nullptr, 0u, flags, epoch, (uint32_t *)nullptr, go,        \
                              (const pq2::RingNullable *)nullptr);                                                         \
    else                                                                                                                   \
        hipLaunchKernelGGL((pq2::ring_kernel<WIDE, 0>), dim3(n_pages), dim3(pq2::kRingThreads), lds_bytes, stream,         \
                           (const pq2::PageDev *)pages, n_pages, status, cu_tickets, ring_bytes, tile_mask,                \
                           (const pq2::RingDict *)nullptr, 0u, flags, epoch, (uint32_t *)nullptr, go,                      \
                           (const pq2::RingNullable *)nullptr)
    if (max_bits > pq2::kRingNarrowBits) {  // ring_kernel<true, .>: every width, and a header walk that keeps four more bytes per run
        // more than 64 KiB of dynamic LDS has to be asked for (per device; the call is cheap)
        if (lds_bytes > 65536u) {
            hipError_t e = hipFuncSetAttribute((const void *)pq2::ring_kernel<true, 0>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        PQ2_RING_LAUNCH(true);
    } else {
        PQ2_RING_LAUNCH(false);
    }
#undef PQ2_RING_LAUNCH
    return hipGetLastError();
}

// ring_kernel for pages of one-bit definition levels that come out as validity bitmaps (ring_expand_span_bits).
// `bit_meta`: two words per page, rows and set bits; the set bits must have been cleared on `stream`.
extern "C" hipError_t pq2_launch_ring_bits(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets,
                                           uint32_t epoch, uint32_t *bit_meta, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    // Level streams are a few bytes per hundred rows and nothing is written per value: what a page takes is its walk,
    // so the more walks a compute unit holds, the better.  A ring of 16 KiB lets six workgroups share its LDS.
    uint32_t ring_bytes = 16384u;
    if (const char *v = getenv("PQ2_RING_BYTES")) {  // tuning knob: 16384 or 32768
        const uint32_t b = (uint32_t)atoi(v);
        if (b == 16384u || b == 32768u) ring_bytes = b;
    }
    hipLaunchKernelGGL((pq2::ring_kernel<false, -1>), dim3(n_pages), dim3(pq2::kRingThreads), pq2::ring_lds_bytes(ring_bytes),
                       stream, (const pq2::PageDev *)pages, n_pages, status, cu_tickets, ring_bytes,
                       (const uint32_t *)nullptr, (const pq2::RingDict *)nullptr, 0u, 0u, epoch, bit_meta, (const uint32_t *)nullptr, (const pq2::RingNullable *)nullptr);
    return hipGetLastError();
}
