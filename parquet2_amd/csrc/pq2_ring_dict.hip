// ring_kernel with the dictionary lookup inside: the launcher (the kernel itself: pq2_ring.h).
//
// The expanding waves of ring_kernel hold every decoded value in a register just before they store it.  For a
// dictionary-encoded page that value is an index, and what the reader wants is the dictionary's value at that
// index (tests/it/read/primitive.rs:99-108): with the page's dictionary in LDS the lookup happens there, between
// the extraction and the store, and the indices never exist in device memory.
#include "pq2_ring.h"

extern "C" uint32_t pq2_ring_dict_max_bits() { return pq2::kRingDictMaxBits; }
extern "C" uint32_t pq2_ring_nullable_max_bits() { return pq2::kRingNullableMaxBits; }
// the largest dictionary (bytes) a workgroup keeps in LDS beside its ring
extern "C" uint32_t pq2_ring_dict_max_bytes() { return 49152u; }

// `dicts`: one pq2::RingDict per page (device); `value_words`: 1, 2 or 3; `max_dict_bytes`: of the largest
// dictionary of the batch (at most pq2_ring_dict_max_bytes()); every page is at most pq2_ring_dict_max_bits() wide.
// The statuses must have been cleared on `stream`.  `go`: as pq2_launch_ring's.
// `nulls`: null, or one pq2::RingNullable per page: the pages are the index streams of flat nullable pages, and the
// values go to the pages' rows (ring_kernel<., ., true>; dictionaries of at most pq2_ring_nullable_max_dict_bytes()).
extern "C" uint32_t pq2_ring_nullable_max_dict_bytes() { return 16384u; }  // (beyond: one workgroup per compute unit)
extern "C" hipError_t pq2_launch_ring_dict(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets,
                                           uint32_t epoch, const uint32_t *tile_mask, const void *dicts, uint32_t value_words,
                                           uint32_t max_dict_bytes, const void *nulls, const uint32_t *go, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    const uint32_t dict_bytes = (max_dict_bytes + 15u) & ~15u;
    // (nullable pages: every expanding wave keeps a span's indices in LDS, two bytes each, and 4096 bits of the bitmap)
    const uint32_t stage_bytes = nulls ? pq2::kRingExpanders * pq2::kRingStageBytes : 0u;
    const uint32_t extra = dict_bytes + stage_bytes;
    // The ring of 32 KiB where three workgroups still share a compute unit's LDS (dictionaries up to 13 KiB); up to
    // 16 KiB the ring of 16 KiB keeps four workgroups there; beyond that two workgroups fit either way (one from
    // 40 KiB on: still ahead of a second pass over the indices).
    // (Nullable pages: the ring of 32 KiB whatever it leaves room for.  With 16 KiB of dictionary two workgroups
    // share a compute unit either way, or three with a ring of 16 KiB -- which take 3.4 ms where two with the larger
    // ring take 2.6: the fewer walks a compute unit holds, the more each one's standing still for room costs.)
    uint32_t ring_bytes = nulls                                                        ? 32768u
                          : 3u * (pq2::ring_lds_bytes(32768u) + extra) <= 160u * 1024u ? 32768u
                          : extra <= 16384u                                            ? 16384u
                                                                                       : 32768u;
    if (const char *v = getenv("PQ2_RING_BYTES")) {  // tuning knob
        const uint32_t b = (uint32_t)atoi(v);
        if (b == 16384u || b == 32768u) ring_bytes = b;
    }
    const uint32_t dict_at = pq2::ring_lds_bytes(ring_bytes);
    static_assert(pq2::ring_lds_bytes(16384u) % 16u == 0u, "the dictionary begins at a 16-byte boundary");
    const uint32_t lds_bytes = dict_at + extra;
    const uint32_t flags = (n_pages >= pq2::kRingTurnsMinPages ? pq2::kRingFlagTurns : 0u) |  // (indices of at most 16 bits)
                           ((dict_at + dict_bytes) << pq2::kRingFlagStageShift);
#define PQ2_LAUNCH_AS(WORDS, NULLABLE)                                                                                \
    if (lds_bytes > 65536u) { /* more than 64 KiB of dynamic LDS has to be asked for */                               \
        hipError_t e = hipFuncSetAttribute((const void *)pq2::ring_kernel<false, WORDS, NULLABLE>,                    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);              \
        if (e != hipSuccess) return e;                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((pq2::ring_kernel<false, WORDS, NULLABLE>), dim3(n_pages), dim3(pq2::kRingThreads), lds_bytes, \
                       stream, (const pq2::PageDev *)pages, n_pages, status, cu_tickets, ring_bytes, tile_mask,       \
                       (const pq2::RingDict *)dicts, dict_at, flags, epoch, (uint32_t *)nullptr,                      \
                       go, (const pq2::RingNullable *)nulls)
#define PQ2_LAUNCH(WORDS)                \
    if (nulls) {                         \
        PQ2_LAUNCH_AS(WORDS, true);      \
    } else {                             \
        PQ2_LAUNCH_AS(WORDS, false);     \
    }
    if (value_words == 1u) {
        PQ2_LAUNCH(1)
    } else if (value_words == 2u) {
        PQ2_LAUNCH(2)
    } else if (value_words == 3u) {
        PQ2_LAUNCH(3)
    } else {
        return hipErrorInvalidValue;
    }
#undef PQ2_LAUNCH
#undef PQ2_LAUNCH_AS
    return hipGetLastError();
}
