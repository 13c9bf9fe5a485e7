// ring_kernel: one workgroup per page, the stream read from memory exactly once (gfx950 / MI355X).
//
// decode_kernel (pq2_kernels.hip) has the wave that walks a page's run headers and the three waves that expand its
// tiles meet through global memory: the walk loads the stream, lists header positions and block starts in device
// arrays, and the expanding waves load those lists and the stream a second time.  Here everything the four waves
// hand to each other stays in the workgroup's LDS:
//
//   ring   the page's stream bytes, a-offset `a` at ring[a mod R] (R = 32 KiB, or 64 KiB for pages wider than 16
//          bits).  The walking wave fills it in slots of 4 KiB straight from memory (buffer loads with an LDS
//          destination: no registers, no second copy), chases the headers inside it, and the expanding waves take
//          headers and bit-packed fields out of it.  64 bytes behind its end mirror its first 64, so that a read of
//          a few dwords never has to wrap.
//   hq     the header-position queue: a-offset of the header of every run that yields values, run i at hq[i mod 512].
//   rq     the block-start queue: the Resume entry of 1024-value block b at rq[b mod 64].
//   ctl    the hand-off words: what the walk has published (values covered, runs listed, bytes landed), and for
//          every expanding wave the tile it holds and the lowest stream byte it still needs.
//
// Flow control replaces the tile lead of decode_kernel: the walk overwrites ring bytes, hq and rq entries only
// once no expanding wave needs them (the lowest needed byte / the lowest held tile), and it says so when it has to
// wait for room, so that the expanding waves go on with whatever is published instead of waiting for a whole span.
// Every wait is bounded; a wait that runs out marks the page PQ2_DEVICE_ERROR and ends all four waves.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "pq2_device.h"

namespace pq2 {

constexpr uint32_t kSlot = 4096;                // bytes per ring fill
constexpr uint32_t kMirror = 64;                // bytes behind the ring that repeat its first bytes
#ifndef PQ2_RING_EXPANDERS
#define PQ2_RING_EXPANDERS 3
#endif
constexpr uint32_t kRingExpanders = PQ2_RING_EXPANDERS;  // 3 or 4 expanding waves beside the walking one
constexpr uint32_t kRingThreads = 64u * (kRingExpanders + 1u);
// the queues share the compute unit's LDS with the ring: four workgroups of 40 KiB each
constexpr uint32_t kHq = kRingExpanders == 3 ? 512 : 256;  // entries of the header-position queue
constexpr uint32_t kRq = kRingExpanders == 3 ? 64 : 32;    // entries of the block-start queue
#ifndef PQ2_RING_SLOTS_IN_FLIGHT
#define PQ2_RING_SLOTS_IN_FLIGHT 4
#endif
constexpr uint32_t kSlotsInFlight = PQ2_RING_SLOTS_IN_FLIGHT;  // ring fills the walk keeps in flight
constexpr uint32_t kPassRuns = 64;  // runs the walk parks before it places and publishes them (one lane each)
#ifndef PQ2_RING_PASS_BYTES
#define PQ2_RING_PASS_BYTES 0xffffffffu
#endif
// ... or once the parked runs span this many stream bytes: what is parked is not published, so nobody can take
// it out of the ring yet
constexpr uint32_t kPassBytes = PQ2_RING_PASS_BYTES;
#ifndef PQ2_RING_WALK_PRIO
#define PQ2_RING_WALK_PRIO 2
#endif
#ifndef PQ2_RING_PRIO_FIRST
#define PQ2_RING_PRIO_FIRST 0x0123u   // hex digit r: the priority of the r-th workgroup of those that begin together
#endif
#ifndef PQ2_RING_PRIO_SECOND
#define PQ2_RING_PRIO_SECOND 0x3210u  // ... of the r-th of their successors
#endif
#ifndef PQ2_RING_PRIO_TURNS
#define PQ2_RING_PRIO_TURNS 1  // 0: the walking wave above the others, and nothing else (A/B builds)
#endif
#ifndef PQ2_RING_FILL_POLICY
#define PQ2_RING_FILL_POLICY ""  // cache policy bits of the ring fills (" nt": the stream is read once)
#endif
constexpr uint32_t kNotYet = 0xfffffffeu;       // header position of a run the walk has not published yet
constexpr uint32_t kTileFresh = 0xffffffffu;    // "lowest byte needed" of a wave that has not looked at its tile yet
constexpr uint32_t kNoNeed = 0xffffffc0u;       // ... of a wave that needs no stream byte any more
constexpr uint32_t kRingMaxPolls = 1u << 22;    // of >= 0.1 us each: far beyond any wait of a healthy page

#ifndef PQ2_RING_TILE_BLOCKS
#define PQ2_RING_TILE_BLOCKS 4
#endif
// What an expanding wave takes at a time: a tile of this many 1024-value blocks (smaller than the tiles of the
// other kernels if that pays: the bytes of the tiles in progress are what keeps the ring occupied).
constexpr uint32_t kRingTileBlocks = PQ2_RING_TILE_BLOCKS;  // 2 or 4
constexpr uint32_t kRingTileValues = kRingTileBlocks * kBlockValues;
constexpr uint32_t kRingTableRuns = 64;                      // run-table entries per expanding wave
// What an expanding wave builds one run table and one chunk index for: a span of 2 blocks, with two bytes per
// chunk-index entry -- or, for pages of at most kRingWholeTileBits bits (definition and repetition levels: long
// RLE runs, little of the ring in use), the whole tile of 4 blocks with one byte per entry: half as many spans to
// set up, and the same LDS.  (Measured: config 3 gains 8 %; config-2 pages lose 12 % with whole-tile spans, because
// a wave then holds on to the ring bytes of a whole tile.  Width 1 only: at width 2, where the values go through
// ring_expand_span and not through their bits, level pages of 2^20 rows reach 0.64 of peak with spans of 2 blocks
// and 0.53 with whole tiles, index-shaped pages 0.59 against 0.48; only pages of 65 536 levels gain, 2 %.)
constexpr uint32_t kRingSpanBlocks = 2;
#ifndef PQ2_RING_WHOLE_TILE_BITS
#define PQ2_RING_WHOLE_TILE_BITS 1
#endif
constexpr uint32_t kRingWholeTileBits = PQ2_RING_WHOLE_TILE_BITS;  // 0, 1 or 2
static_assert(kRingTileBlocks == 4 || kRingTileBlocks == 2, "tiles of 2 or 4 blocks");
template <uint32_t kBlocks>
struct RingSpan {
    static_assert(kBlocks == 2 || kBlocks == 4, "spans of 2 or 4 blocks");
    static constexpr uint32_t kValues = kBlocks * kBlockValues;
    static constexpr uint32_t kChunks = kValues / 4u;  // 16-byte output chunks per span
    // + 2 chunks when the output is not 16-byte aligned, rounded up to 16 bytes of entries
    static constexpr uint32_t kSlots = kChunks + (kBlocks == 4 ? 16u : 8u);
    typedef typename std::conditional<kBlocks == 4, uint8_t, uint16_t>::type cidx_t;  // (a table has at most 64 runs)
    static constexpr uint32_t kBytes = kSlots * (uint32_t)sizeof(cidx_t);
};
static_assert(RingSpan<2>::kBytes == RingSpan<4>::kBytes, "either chunk index fits the same LDS");
constexpr uint32_t kRingNoRun = 0xffffffffu;
// Emission steps whose LDS reads are in flight together: all nine of a span in the kernel for pages of at most 16
// bits, where the dwords of nine chunks fit the registers of a wave (four waves per SIMD: 128 each); five in the
// kernel that also takes wider pages (nine there make the compiler keep the kernel's long-lived values in scratch
// memory, and an expanding wave that waits for a scratch load waits for its stores).
#ifndef PQ2_RING_STEP_GROUP
#define PQ2_RING_STEP_GROUP 9
#endif
#ifndef PQ2_RING_STEP_GROUP_WIDE
#define PQ2_RING_STEP_GROUP_WIDE 5
#endif
constexpr uint32_t kRingNarrowBits = 16;  // widest page of the kernel for narrow pages

struct RingCtl {
    uint32_t covered;    // values covered by the published entries (kAllParsed: the page's entries are complete)
    uint32_t runs;       // hq entries published
    uint32_t floor;      // every run published from now on has its header at or behind this a-offset
    uint32_t landed;     // the ring holds the stream up to this a-offset
    uint32_t final_hi;   // the ring will never hold more than this
    uint32_t blocked;    // the walk waits for room
    uint32_t abort;      // a wait ran out: everybody leaves
    uint32_t next_tile;  // the next tile to hand out
    uint64_t expander[4];  // per expanding wave: tile it holds << 32 | lowest a-offset it still needs
    uint32_t pick[8];      // pick_walker's scratch
};
constexpr uint32_t kRingAllParsed = 0xffffffffu;

#ifdef PQ2_RING_TRACE  // diagnostic build: when every page's workgroup began and ended (100 MHz clock) and where it ran
constexpr uint32_t kRingTracePages = 8192;
static __device__ unsigned long long g_ring_trace[3 * kRingTracePages];  // per page: begin, end, XCC << 8 | compute unit
#define PQ2_TRACE_END(PAGE)                                                                                       \
    do {                                                                                                          \
        if (lane == 0 && (PAGE) < kRingTracePages) atomicMax(&g_ring_trace[3u * (PAGE) + 1u], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
    } while (0)
#else
#define PQ2_TRACE_END(PAGE) do { } while (0)
#endif
#ifdef PQ2_RING_TIMING  // diagnostic build: shader clocks spent per role and wait, summed over all waves
// 0 walk total, 1 walk waiting for room, 2 walk waiting for fills, 3 walk chase, 4 walk after its last publish,
// 8 expand total, 9 waiting for a block start, 10 waiting for runs, 11 waiting for bytes
static __device__ unsigned long long g_ring_clocks[32];  // 16..: event counts of the walk (see PQ2_COUNT uses)
#define PQ2_CLOCK() __builtin_amdgcn_s_memtime()
#define PQ2_CLOCK_ADD(SLOT, T0) clocks[(SLOT) & 7] += __builtin_amdgcn_s_memtime() - (T0)
#define PQ2_COUNT(SLOT, V) counts[SLOT] += (V)
#define PQ2_CLOCK_FLUSH(BASE)                                                                            \
    do {                                                                                                 \
        if (lane == 0)                                                                                   \
            for (int i_ = 0; i_ < 8; ++i_)                                                               \
                if (clocks[i_]) atomicAdd(&g_ring_clocks[(BASE) + i_], (unsigned long long)clocks[i_]);  \
    } while (0)
#else
#define PQ2_CLOCK() 0ull
#define PQ2_CLOCK_ADD(SLOT, T0) do { (void)(T0); } while (0)
#define PQ2_CLOCK_FLUSH(BASE) do { } while (0)
#define PQ2_COUNT(SLOT, V) do { } while (0)
#endif

// LDS bytes of one workgroup for a ring of `ring_bytes`
__host__ __device__ constexpr uint32_t ring_lds_bytes(uint32_t ring_bytes) {
    return ring_bytes + kMirror + kHq * 4u + kRq * 16u + (uint32_t)sizeof(RingCtl) +
           kRingExpanders * ((kRingTableRuns + 2u) * 8u + RingSpan<2>::kBytes);
}

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// What this wave has written to LDS is in place before what it writes next (LDS operations of a wave execute in
// order; the wait also covers the hand-off words themselves).
__device__ __forceinline__ void lds_release() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Keeps the compiler from reading LDS ahead of a hand-off word it has just looked at.
__device__ __forceinline__ void lds_acquire() { asm volatile("" ::: "memory"); }

// ---------------------------------------------------------------------------
// The walking wave
// ---------------------------------------------------------------------------
struct RingShared {
    uint8_t *ring;
    uint32_t ring_bytes;  // R, a power of two
    uint32_t *hq;
    Resume *rq;
    RingCtl *ctl;
    uint32_t *rows_out;   // validity bitmaps: where the walk says how many rows the page's stream holds (else null)
};

// kMerge: the page's status is shared with the expanding waves (a dictionary index out of range): the larger
// status stays.
template <bool kWide, bool kMerge>
__device__ __forceinline__ void ring_walk(const PageDev &pg, const uint32_t page, const RingShared sh,
                                          int32_t *__restrict__ status, const uint32_t lane) {
    auto set_status = [&](int32_t st) {
        if (lane == 0) {
            if (kMerge)
                atomicMax(&status[page], st);
            else
                status[page] = st;
        }
    };
    RingCtl *const ctl = sh.ctl;
#ifdef PQ2_RING_TIMING
    uint64_t clocks[8] = {};
    uint64_t counts[12] = {};
#endif
    const uint64_t clock_begin = PQ2_CLOCK();
    const uint32_t n = pg.num_values;  // > 0
    const uint32_t w = pg.num_bits;
    const uint32_t n_blocks = (n + kBlockValues - 1u) / kBlockValues;
    const uint32_t n_tiles = (n + kRingTileValues - 1u) / kRingTileValues;
    const uint32_t R = sh.ring_bytes, rmask = R - 1u;
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;  // a-offset one past the last stream byte
    const uint32_t rb = ceil8(w);
    const uint32_t rb_mask = rb < 4 ? (1u << (rb * 8u)) - 1u : ~0u;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)sh.ring;

    uint32_t n_runs = 0;       // hq entries written
    uint32_t produced = 0;     // values the listed runs yield
    uint32_t blocks_done = 0;  // Resume entries of blocks [0, blocks_done) are written
    uint32_t issued_hi = 0, landed_hi = 0;  // ring fills requested / arrived, in a-offsets (multiples of kSlot)
    uint32_t final_hi = 0;     // nothing behind this a-offset is ever loaded
    bool aborted = false;
#ifdef PQ2_RING_DEBUG
    uint32_t dbg_site = 0;
#define PQ2_SITE(N) dbg_site = N
#else
#define PQ2_SITE(N)
#endif
    uint32_t keep = delta;     // header of the first run that is not placed yet: the bytes from there on stay in the ring

    // ---- what the expanding waves still need ----------------------------------------------------------------
    // low_byte: no stream byte below it is needed any more; low_tile: no block below 4 * low_tile is.
    auto limits = [&](uint32_t &low_byte, uint32_t &low_tile) {
        low_byte = 0xffffffffu;
        low_tile = 0xffffffffu;
#pragma unroll
        for (uint32_t e = 0; e < kRingExpanders; ++e) {
            const uint64_t x = __hip_atomic_load(&ctl->expander[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t tile = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
            uint32_t byte = __builtin_amdgcn_readfirstlane((uint32_t)x);
            if (tile >= n_tiles) continue;  // that wave is through
            low_tile = min(low_tile, tile);
            if (byte == kTileFresh) {
                // it has not read its tile's first Resume entry yet: the entry says what it will ask for
                if (tile * kRingTileBlocks >= blocks_done) continue;  // not even written: nothing below the walk
                const Resume r = sh.rq[(tile * kRingTileBlocks) & (kRq - 1u)];
                const uint32_t rem_kind = __builtin_amdgcn_readfirstlane(r.rem_kind);
                const uint32_t payload = __builtin_amdgcn_readfirstlane(r.payload);
                const uint32_t skip = __builtin_amdgcn_readfirstlane(r.skip);
                byte = (rem_kind >> 31) == kBitpacked ? payload + (skip >> 3) * w : skip;
            }
            low_byte = min(low_byte, byte);
        }
    };
    // What limits() said last, lowered wherever the truth may since have dropped below it: the needs of the expanding
    // waves only rise, except that a wave whose tile had no entry yet (no need) gets one when the walk publishes
    // the tile -- never below the first byte published then (`keep` at that time).  The checks on the walk's way
    // look at these and ask again only when they fail.
    uint32_t known_byte = 0, known_tile = 0;
    auto refresh = [&]() { limits(known_byte, known_tile); };
    auto abort_all = [&]() {
#ifdef PQ2_RING_DEBUG
        if (lane == 0 && lds_load(&ctl->abort) == 0u) status[page] = 100 + (int32_t)dbg_site;
#else
        set_status(6);  // PQ2_DEVICE_ERROR
#endif
        lds_store(&ctl->abort, 1u);
        aborted = true;
    };
    // ---- ring fills ------------------------------------------------------------------------------------------
    // The four loads of a fill are written in assembly so that the compiler does not know of them: it would wait for
    // every load in flight in front of each LDS store of the walk (it cannot tell that they go elsewhere), and the
    // fills are meant to stay in flight while the walk goes on.  land() waits for them by count.
    // M0: LDS address of the destination; lane l's 16 bytes land at M0 + 16 l.
    const u32x4 rsrc_words = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pg.in_addr - delta)),
                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((pg.in_addr - delta) >> 32) & 0xffffu),
                              (uint32_t)__builtin_amdgcn_readfirstlane((end + 3u) & ~3u), 0x00020000u};
    auto issue_slot = [&]() {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (issued_hi & rmask));
        const uint32_t off0 = issued_hi + lane * 16u, off1 = off0 + 1024u, off2 = off0 + 2048u, off3 = off0 + 3072u;
        uint32_t m0_was;
        asm volatile(
            "s_mov_b32 %[m0_was], m0\n\t"
            "s_mov_b32 m0, %[dst]\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[off0], %[rsrc], 0 offen" PQ2_RING_FILL_POLICY " lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[off1], %[rsrc], 0 offen" PQ2_RING_FILL_POLICY " lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[off2], %[rsrc], 0 offen" PQ2_RING_FILL_POLICY " lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[off3], %[rsrc], 0 offen" PQ2_RING_FILL_POLICY " lds\n\t"
            "s_mov_b32 m0, %[m0_was]"
            : [m0_was] "=&s"(m0_was)
            : [dst] "s"(dst), [rsrc] "s"(rsrc_words), [off0] "v"(off0), [off1] "v"(off1), [off2] "v"(off2),
              [off3] "v"(off3)
            : "memory", "scc");
        issued_hi += kSlot;
    };
    // may the slot at issued_hi be filled, if nothing below `low` is needed?  (it replaces bytes [issued_hi - R, ...))
    auto room_for_slot = [&](uint32_t low) { return (uint64_t)issued_hi + kSlot <= (uint64_t)(low & ~63u) + R; };
    // Requests ring fills as far as there is room and the in-flight limit allows.  `ask`: when the room known so far
    // does not do, look again at what the expanding waves need (a handful of LDS round trips: only where a fill is
    // needed now, not for the ones requested ahead).
    auto top_up = [&](uint32_t keep_from, bool ask) {
        while (issued_hi < final_hi && issued_hi - landed_hi < kSlotsInFlight * kSlot) {
            if (!room_for_slot(min(known_byte, keep_from))) {
                if (!ask) break;
                refresh();
                ask = false;
                PQ2_COUNT(2, 1);  // looks at the limits because a fill was needed
                if (!room_for_slot(min(known_byte, keep_from))) {
                    PQ2_COUNT(3, 1);  // ... in vain
                    break;
                }
            }
            issue_slot();
        }
    };
    // Waits for the oldest fill in flight (the memory operations of a wave complete in order, so that is "all but
    // the instructions of the fills requested after it") and says that its bytes are there.
    auto land = [&]() {
        PQ2_COUNT(1, 1);  // fills waited for
        const uint64_t clock_land = PQ2_CLOCK();
        static_assert(kSlot / 1024u == 4u && kSlotsInFlight <= 4u, "the waits below spell out four loads per fill");
        const uint32_t behind = (issued_hi - landed_hi) / kSlot - 1u;  // fills requested after the oldest one
        if (behind == 0u)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (behind == 1u)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (behind == 2u)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        PQ2_CLOCK_ADD(2, clock_land);
        if ((landed_hi & rmask) == 0u) {  // the ring's first bytes once more behind its end
            if (lane < kMirror / 4u) ((uint32_t *)(sh.ring + R))[lane] = ((const uint32_t *)sh.ring)[lane];
            lds_release();
        }
        landed_hi += kSlot;
        lds_store(&ctl->landed, landed_hi);
    };
    // Waits until `ready(low_byte, low_tile)`; tells the expanding waves that the walk stands still meanwhile.
    auto wait_room = [&](auto ready) {
        if (ready(known_byte, known_tile)) return;
        bool told = false;
        const uint64_t clock_wait = PQ2_CLOCK();
        for (uint32_t polls = 0;; ++polls) {
            refresh();
            if (ready(known_byte, known_tile)) break;
            if (!told) {
                lds_store(&ctl->blocked, 1u);
                told = true;
            }
            if (polls == kRingMaxPolls || lds_load(&ctl->abort) != 0u) {
                abort_all();
                break;
            }
            // The ring is kept as full as the room allows meanwhile: an expanding wave may be waiting for the
            // bytes of a run that is published already, and it is that wave's progress that makes the room.
            top_up(keep, false);  // (refreshed above)
            if (issued_hi != landed_hi) land();
            __builtin_amdgcn_s_sleep(2);
        }
        if (told) lds_store(&ctl->blocked, 0u);
        PQ2_CLOCK_ADD(1, clock_wait);
    };

    // Makes stream bytes [at, at + bytes) resident as far as that takes no waiting for the expanding waves
    // (`at` and everything behind it stay in the ring); false: there is no room yet.
    auto try_ensure = [&](uint32_t at, uint32_t bytes) -> bool {
        const uint32_t want = min(at + bytes, final_hi);
        while (landed_hi < want) {
            if (issued_hi == landed_hi) {
                PQ2_COUNT(4, 1);  // bytes needed with no fill in flight
                top_up(keep, true);
                if (issued_hi == landed_hi) return false;
            }
            land();
            top_up(keep, false);  // the next fills are on their way while the walk works through these
        }
        return true;
    };
    // the same, waiting for room; only when everything the walk knows has been published
    auto ensure = [&](uint32_t at, uint32_t bytes) {
        PQ2_SITE(3);
        // (the wait itself keeps the ring fed: what it is waiting for may have landed before there is room for
        // yet another slot)
        while (!aborted && !try_ensure(at, bytes))
            wait_room([&](uint32_t low_byte, uint32_t) {
                return landed_hi >= min(at + bytes, final_hi) || room_for_slot(min(low_byte, keep));
            });
    };

    // ---- publishing ------------------------------------------------------------------------------------------
    // runs first, floor second: a reader that takes the floor first and the run count second knows that every
    // run from that count on starts at or behind that floor
    auto publish = [&](uint32_t covered, uint32_t floor_pos) {
        known_byte = min(known_byte, keep);  // what becomes visible now lies behind `keep`, and so will its takers' needs
        lds_release();
        lds_store(&ctl->runs, n_runs);
        lds_store(&ctl->floor, floor_pos);
        lds_store(&ctl->covered, covered);
    };
    // hq entries [n_runs, n_runs + count) may be written once the entries they replace are not needed: those lie
    // below every byte an expanding wave still needs
    auto hq_room = [&](uint32_t count) {
        if (count == 0u || n_runs + count <= kHq) return;
        const uint32_t newest_replaced = __builtin_amdgcn_readfirstlane(sh.hq[(n_runs + count - 1u) & (kHq - 1u)]);
        PQ2_SITE(1);
        wait_room([&](uint32_t low_byte, uint32_t) { return newest_replaced < low_byte; });
    };
    // Writes the Resume entries of blocks [from, to) as make(b) gives them, as far ahead of the lowest held tile as
    // the queue allows, publishing as it goes.
    auto emit_blocks = [&](uint32_t from, uint32_t to, uint32_t floor_pos, auto make) {
        uint32_t b = from;
        while (b < to && !aborted) {
            uint32_t allowed = b;
            PQ2_SITE(2);
            wait_room([&](uint32_t, uint32_t low_tile) {
                allowed = low_tile == 0xffffffffu ? to : min(to, low_tile * kRingTileBlocks + kRq);
                return allowed > b;
            });
            if (aborted) break;
            for (uint32_t bb = b + lane; bb < allowed; bb += 64u) sh.rq[bb & (kRq - 1u)] = make(bb);
            b = allowed;
            blocks_done = b;
            if (b < to) publish(min(b * kBlockValues, n), floor_pos);
        }
    };
    // the page's entries are complete: closes the header list, the blocks behind the last value the stream holds
    // read zeros (mod.rs:95)
    auto finish = [&](int32_t st, uint32_t floor_pos) {
        hq_room(1u);
        if (!aborted) {
            if (lane == 0) sh.hq[n_runs & (kHq - 1u)] = kEndPos;
            ++n_runs;
            const uint32_t listed = n_runs - 1u;
            emit_blocks((produced + kBlockValues - 1u) / kBlockValues, n_blocks, floor_pos, [&](uint32_t) {
                Resume e;
                e.run_index = listed;
                e.rem_kind = kRemMask | (kRle << 31);
                e.payload = 0;
                e.skip = kEndPos;
                return e;
            });
        }
        if (aborted || lds_load(&ctl->abort) != 0u) return;
        set_status(st);
        // HybridRleIter ends with the stream (src/deserialize/hybrid_rle.rs:90-94): its rows are those of the runs
        if (sh.rows_out != nullptr && lane == 0) *sh.rows_out = min(produced, n);
        // nothing behind the walk's position is needed: the fills end there
        final_hi = min(final_hi, (floor_pos + 64u + kSlot - 1u) & ~(kSlot - 1u));
        lds_store(&ctl->final_hi, final_hi);
        publish(kRingAllParsed, floor_pos);
    };

    if (w == 0 || pg.in_len == 0 || w > 32) {  // decoder.rs:31-37: nothing to iterate
        finish(0, 0u);
        return;
    }
    final_hi = (end + 64u + kSlot - 1u) & ~(kSlot - 1u);  // reads of a few dwords behind the last byte stay inside
    lds_store(&ctl->final_hi, final_hi);

    uint32_t pos = delta;  // a-offset of the next header
    for (;;) {
        keep = pos;  // everything before is placed and published: the waves that expand it say what they need
        const uint64_t clock_top = PQ2_CLOCK();
        refresh();  // once per pass
        PQ2_COUNT(5, 1);                                   // passes
        PQ2_COUNT(6, keep - min(known_byte, keep));        // bytes between the lowest need and the first unplaced run
        PQ2_COUNT(7, landed_hi - min(landed_hi, keep));    // bytes landed ahead of the walk
        PQ2_COUNT(8, issued_hi - landed_hi);               // bytes in flight
        top_up(keep, false);
        PQ2_COUNT(9, issued_hi - landed_hi);               // ... after topping up
        ensure(pos, 32u);
        PQ2_CLOCK_ADD(6, clock_top);
        if (aborted) return;
        // ---- chase: park up to 64 runs, record i in lane i (see PQ2_CHASE) ---------------------------------
        uint32_t r_pos = 0, r_lo = 0, r_hi = 0, r_info = 0;
        uint32_t m = 0;     // parked records
        uint32_t term = 0;  // kEnd / kPanic once the chase cannot go on
        for (;;) {
            uint32_t lo, hi, groups, t, bytes, nm1, chk, addr;
            uint64_t exec_all;
            uint32_t m0_kept;
            const uint32_t voff = ring_lds - (pos & ~rmask);  // this turn of the ring, as if it were the whole stream
            // stops at the end of the stream, of the bytes that have landed, and of this turn of the ring (the
            // mirror holds the 8 bytes of a header that begins in the ring's last bytes)
            // (and, where passes are limited in bytes, behind the first run that reaches that far)
            const uint32_t chase_limit = __builtin_amdgcn_readfirstlane(
                min(min(min(end - 1u, landed_hi - 9u), pos | rmask), kPassBytes == 0xffffffffu ? 0xffffffffu : keep + kPassBytes));
            PQ2_COUNT(0, 1);  // entries into the chase loop
            const uint64_t clock_chase = PQ2_CLOCK();
            if constexpr (kWide) {
                PQ2_CHASE("v_readfirstlane_b32 %[hi], v3\n\t", "v_writelane_b32 %[r_hi], %[hi], m0\n\t")
            } else {
                PQ2_CHASE("", "")
            }
            PQ2_CLOCK_ADD(3, clock_chase);
            if ((~lo & kExplicit) == 0u || nm1 > end - 1u) {
                // three or more header bytes, a run cut by the end of the stream, or the end itself: the record
                // parked last does not count
                --m;
                pos = __builtin_amdgcn_readlane(r_pos, m);
                if (!try_ensure(pos, 32u)) break;  // place what is parked, then wait for room
                auto rd = [&](uint32_t a) -> uint32_t {
                    return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&sh.ring[a & rmask]);
                };
                const RunInfo run = decode_run(rd, pos, end, w, 0xffffffffu);
                const uint32_t e_lo = kExplicit | (run.kind << 16) | (((run.next - pos) & 0xffu) << 20);
                const uint32_t e_hi = __builtin_amdgcn_readfirstlane(run.count);
                const uint32_t e_info = __builtin_amdgcn_readfirstlane(run.kind == kRle ? run.value : run.payload);
                uint32_t m0_was;
                asm volatile(
                    "s_mov_b32 %[m0_was], m0\n\t"
                    "s_mov_b32 m0, %[m]\n\t"
                    "v_writelane_b32 %[r_pos], %[pos], m0\n\t"
                    "v_writelane_b32 %[r_lo], %[e_lo], m0\n\t"
                    "v_writelane_b32 %[r_hi], %[e_hi], m0\n\t"
                    "v_writelane_b32 %[r_info], %[e_info], m0\n\t"
                    "s_mov_b32 m0, %[m0_was]"
                    : [r_pos] "+v"(r_pos), [r_lo] "+v"(r_lo), [r_hi] "+v"(r_hi), [r_info] "+v"(r_info),
                      [m0_was] "=&s"(m0_was)
                    : [m] "s"(m), [pos] "s"(pos), [e_lo] "s"(e_lo), [e_hi] "s"(e_hi), [e_info] "s"(e_info));
                ++m;
                if (run.kind >= kEnd) {
                    term = run.kind;
                    break;
                }
                pos = run.next;
            }
            if (m == kPassRuns || pos - keep >= kPassBytes) break;
            const uint64_t clock_more = PQ2_CLOCK();
            const bool more = try_ensure(pos, 9u);
            PQ2_CLOCK_ADD(7, clock_more);
            if (!more) break;
        }
        const uint64_t clock_place = PQ2_CLOCK();

        // ---- place: lane i decodes record i; a prefix sum over the run lengths places the runs -------------
        const ParkedRun parked = decode_parked(r_pos, r_lo, r_hi, r_info, rb, rb_mask);
        const uint32_t kind = parked.kind, info = parked.info, after = parked.after;
        const uint32_t left = n - produced;  // > 0
        const uint32_t cnt = (lane < m && kind < kEnd) ? min(parked.count, left) : 0u;
        const uint32_t incl = wave_inclusive_sum_saturating(cnt, left);
        const uint32_t before = __builtin_amdgcn_update_dpp(0u, incl, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const uint32_t eff = min(cnt, left - before);  // the page ends inside or before this run: mod.rs:76-78
        const bool yields = eff > 0u;
        const uint64_t ymask = __ballot(yields);
        const uint32_t n_yield = (uint32_t)__popcll(ymask);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        hq_room(n_yield);
        if (aborted) return;
        const uint32_t idx =
            n_runs + __builtin_amdgcn_mbcnt_hi((uint32_t)(ymask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ymask, 0u));
        if (yields) sh.hq[idx & (kHq - 1u)] = r_pos;
        n_runs += n_yield;
        // the run holds values [first, first + eff) of the page: it starts blocks b_first..b_last (if any)
        const uint32_t first = produced + before;
        const uint32_t b_first = (first + kBlockValues - 1u) / kBlockValues;
        const uint32_t b_last = (first + eff - 1u) / kBlockValues;  // meaningful where `yields`
        const bool starts = yields && b_last >= b_first;
        if (total != 0u) {
            const uint32_t pass_last = (produced + total - 1u) / kBlockValues;  // the last block these runs start
            // in rounds, as far ahead of the lowest held tile as the block-start queue allows (one round nearly always)
            while (blocks_done <= pass_last && !aborted) {
                uint32_t allowed = blocks_done;
                PQ2_SITE(4);
                wait_room([&](uint32_t, uint32_t low_tile) {
                    allowed = low_tile == 0xffffffffu ? pass_last + 1u : min(pass_last + 1u, low_tile * kRingTileBlocks + kRq);
                    return allowed > blocks_done;
                });
                if (aborted) return;
                const uint32_t r_first = max(b_first, blocks_done), r_last = min(b_last, allowed - 1u);
                const uint32_t nb = (starts && r_last >= r_first) ? r_last - r_first + 1u : 0u;
                auto write_entries = [&](uint32_t b_from, uint32_t b_step, uint32_t b_to, uint32_t run_index,
                                         uint32_t run_first, uint32_t run_eff, uint32_t run_kind, uint32_t run_info,
                                         uint32_t run_after) {
                    for (uint32_t b = b_from; b <= b_to; b += b_step) {
                        const uint32_t skip = b * kBlockValues - run_first;
                        Resume e;
                        e.run_index = run_index;
                        e.rem_kind = (run_eff - skip) | (run_kind << 31);
                        e.payload = run_info;
                        e.skip = run_kind == kRle ? run_after : skip;
                        sh.rq[b & (kRq - 1u)] = e;
                    }
                };
                if (nb != 0u && nb <= 4u) write_entries(r_first, 1u, r_last, idx, first, eff, kind, info, after);
                for (uint64_t big = __ballot(nb > 4u); big != 0; big &= big - 1) {  // long runs: the whole wave writes
                    const uint32_t l = (uint32_t)__builtin_ctzll(big);
                    write_entries(__builtin_amdgcn_readlane(r_first, l) + lane, 64u, __builtin_amdgcn_readlane(r_last, l),
                                  __builtin_amdgcn_readlane(idx, l), __builtin_amdgcn_readlane(first, l),
                                  __builtin_amdgcn_readlane(eff, l), __builtin_amdgcn_readlane(kind, l),
                                  __builtin_amdgcn_readlane(info, l), __builtin_amdgcn_readlane(after, l));
                }
                blocks_done = allowed;
                if (blocks_done <= pass_last) {
                    publish(blocks_done * kBlockValues, pos);  // the blocks written so far
                    // what is published is kept in the ring by the waves that expand it; the walk keeps the rest:
                    // the bytes from the first block without an entry on (one run of this pass holds its first value)
                    const uint64_t holder = __ballot(starts && b_first <= blocks_done && blocks_done <= b_last);
                    if (holder != 0) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(holder);
                        const uint32_t h_first = __builtin_amdgcn_readlane(first, l);
                        const uint32_t h_info = __builtin_amdgcn_readlane(info, l);
                        keep = __builtin_amdgcn_readlane(kind, l) == kBitpacked
                                   ? h_info + ((blocks_done * kBlockValues - h_first) >> 3) * w
                                   : __builtin_amdgcn_readlane(after, l);
                    }
                }
            }
        }
        produced += total;
        publish(min(produced, n), pos);
        keep = pos;
        PQ2_CLOCK_ADD(5, clock_place);
        if (produced >= n) {
            finish(0, pos);
            break;
        }
        if (term != 0u) {
            finish(term == kPanic ? 5 : 0, pos);  // PQ2_REFERENCE_WOULD_PANIC | the stream is exhausted
            break;
        }
    }
    // ---- the rest of the stream the expanding waves ask for -----------------------------------------------------
    const uint64_t clock_tail = PQ2_CLOCK();
    while (!aborted && issued_hi < final_hi) {
        uint32_t low_tile = 0;
        PQ2_SITE(5);
        wait_room([&](uint32_t lb, uint32_t lt) {
            low_tile = lt;
            return lt == 0xffffffffu || issued_hi >= final_hi || room_for_slot(lb);
        });
        if (aborted || low_tile == 0xffffffffu) break;  // every tile is done
        top_up(0xffffffffu, true);
        while (issued_hi != landed_hi) land();
    }
    while (issued_hi != landed_hi) land();  // fills that were still on their way when the walk ended
    PQ2_CLOCK_ADD(4, clock_tail);
    PQ2_CLOCK_ADD(0, clock_begin);
    PQ2_CLOCK_FLUSH(0);
#ifdef PQ2_RING_TIMING
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_ring_clocks[16 + i], (unsigned long long)counts[i]);
#endif
}

// ---------------------------------------------------------------------------
// The expanding waves
// ---------------------------------------------------------------------------
struct RingExpander {
    RingShared sh;
    uint64_t *slot;    // this wave's word of ctl->expander
    uint32_t tile;     // the tile it holds
    uint32_t need;     // the lowest a-offset it has said it still needs (a multiple of 64, kTileFresh, kNoNeed)
    uint32_t cap;      // bytes behind `need` the walk can always provide
    int32_t *status;
    uint32_t page;
    bool aborted;
    bool merge;  // see ring_walk's kMerge
#ifdef PQ2_RING_TIMING
    uint64_t clocks[8];
#endif

    __device__ __forceinline__ void say(uint32_t lane) {
        if (lane == 0)
            __hip_atomic_store(slot, ((uint64_t)tile << 32) | need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void abort_all(uint32_t lane, uint32_t site = 0) {
#ifdef PQ2_RING_DEBUG
        if (lane == 0 && lds_load(&sh.ctl->abort) == 0u) status[page] = 200 + (int32_t)site;
#else
        if (lane == 0) {  // PQ2_DEVICE_ERROR
            if (merge)
                atomicMax(&status[page], 6);
            else
                status[page] = 6;
        }
#endif
        lds_store(&sh.ctl->abort, 1u);
        aborted = true;
    }
    // nothing below `from` is needed any more
    __device__ __forceinline__ void raise_need(uint32_t from, uint32_t lane) {
        const uint32_t v = from == kEndPos ? kNoNeed : (from & ~63u);
        if (need == kTileFresh || v > need) {
            need = v;
            say(lane);
        }
    }
    // Waits until the ring holds the stream up to `to` (or all it will ever hold); `to` lies within `cap` of `need`.
    __device__ __forceinline__ void await_bytes(uint32_t to, uint32_t lane) {
        const uint64_t clock_wait = PQ2_CLOCK();
        for (uint32_t polls = 0;; ++polls) {
            const uint32_t landed = lds_load(&sh.ctl->landed);
            const uint32_t final_hi = lds_load(&sh.ctl->final_hi);
            if (landed >= min(to, final_hi)) break;
            if (polls == kRingMaxPolls || lds_load(&sh.ctl->abort) != 0u) {
                abort_all(lane, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        PQ2_CLOCK_ADD(11, clock_wait);
        lds_acquire();
    }
    // Waits until the walk has covered value `first` (the Resume entry of the block that starts there is written).
    __device__ __forceinline__ void await_start(uint32_t first, uint32_t lane) {
        const uint64_t clock_wait = PQ2_CLOCK();
        for (uint32_t polls = 0;; ++polls) {
            if (lds_load(&sh.ctl->covered) > first) break;
            if (polls == kRingMaxPolls || lds_load(&sh.ctl->abort) != 0u) {
                abort_all(lane, 2);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        PQ2_CLOCK_ADD(9, clock_wait);
        lds_acquire();
    }
    // Number of published runs once a pass from run `ridx` on has what it can get: the walk has covered the span
    // (values up to `span_end` of the page), or listed 63 runs behind run ridx, or stands still waiting for room
    // (then whatever is listed has to do).  `must`: run ridx + 1 itself is needed (nothing else is left to do):
    // then the wave needs no byte below the walk's floor any more and says so.
    __device__ __forceinline__ uint32_t await_runs(uint32_t ridx, uint32_t span_end, bool must, uint32_t lane) {
        uint32_t runs = 0;
        const uint64_t clock_wait = PQ2_CLOCK();
        for (uint32_t polls = 0;; ++polls) {
            const uint32_t floor = lds_load(&sh.ctl->floor);  // before the run count: see ring_walk's publish
            lds_acquire();
            runs = lds_load(&sh.ctl->runs);
            const uint32_t covered = lds_load(&sh.ctl->covered);
            if (covered >= span_end || runs >= ridx + 1u + (kRingTableRuns - 1u)) break;
            if (runs > ridx + 1u) {
                if (lds_load(&sh.ctl->blocked) != 0u) break;
            } else if (must) {
                raise_need(floor, lane);
            } else if (lds_load(&sh.ctl->blocked) != 0u) {
                break;
            }
            if (polls == kRingMaxPolls || lds_load(&sh.ctl->abort) != 0u) {
                abort_all(lane, 3);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        PQ2_CLOCK_ADD(10, clock_wait);
        lds_acquire();
        return runs;
    }
};

// One wave decodes the span [span_start, span_start + block_n) of a page whose width is W (block_n <=
// kRingSpanValues) out of the ring.  expand_block of pq2_kernels.hip with the stream window, the header list and
// the waits replaced: see there for the run table, the chunk index (two bytes per chunk here) and the steps.
//
// Where the values go is the sink's business.  IndexSink stores them as they are (u32, the decode of
// HybridRleDecoder); DictSink<kWords> takes them as indices into a dictionary that lies in LDS and stores the
// dictionary's values of kWords dwords each (the consumer loop of a dictionary-encoded page,
// tests/it/read/primitive.rs:99-108 with PrimitivePageDict::value, tests/it/read/dictionary/primitive.rs:20-27).
// Positions are those of the span; put4 writes the four positions from p on, which lie in one run and begin at a
// 16-byte boundary of the output, put1 writes one.
struct IndexSink {
    out_u32 *out;  // the span's position 0
    __device__ __forceinline__ uint32_t mis() const { return (uint32_t)(((uintptr_t)out >> 2) & 3u); }
    __device__ __forceinline__ void put4(const u32x4 v, const int32_t p) { store4(v, out + p); }
    __device__ __forceinline__ void put1(const uint32_t v, const int32_t p) { out[p] = v; }
};
template <int kWords>
struct DictSink {
    out_u32 *out;          // the first dword of the value at the span's position 0
    const uint32_t *dict;  // LDS: dict_len values of kWords dwords
    uint32_t dict_len;
    bool bad;              // an index that is not below dict_len was seen (its value is written as zeros)
    // positions between the 16-byte grid of the output and position 0 (the output is aligned to a whole value)
    __device__ __forceinline__ uint32_t mis() const {
        const uint32_t a = (uint32_t)((uintptr_t)out >> 2);
        return kWords == 1 ? a & 3u : kWords == 2 ? (a >> 1) & 1u : (3u * a) & 3u;
    }
    __device__ __forceinline__ void put4(const u32x4 j, const int32_t p) {
        const uint32_t jj[4] = {j.x, j.y, j.z, j.w};
        uint32_t v[4 * kWords];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = jj[k] < dict_len;
            bad |= !ok;
#pragma unroll
            for (int c = 0; c < kWords; ++c) v[k * kWords + c] = ok ? dict[jj[k] * kWords + c] : 0u;
        }
#pragma unroll
        for (int c = 0; c < kWords; ++c) {
            const u32x4 o = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
            store4(o, out + (ptrdiff_t)p * kWords + 4 * c);
        }
    }
    __device__ __forceinline__ void put1(const uint32_t j, const int32_t p) {
        const bool ok = j < dict_len;
        bad |= !ok;
#pragma unroll
        for (int c = 0; c < kWords; ++c) out[(ptrdiff_t)p * kWords + c] = ok ? dict[j * kWords + c] : 0u;
    }
};

// StageSink keeps the span's values -- dictionary indices of at most 16 bits -- in LDS, position p at stage[p]:
// for pages whose rows are not their values' positions (flat nullable columns, ring_place_span below).
struct StageSink {
    uint16_t *stage;  // LDS: kRowDirStep entries
    __device__ __forceinline__ uint32_t mis() const { return 0u; }
    __device__ __forceinline__ void put4(const u32x4 j, const int32_t p) {
        *(uint2 *)&stage[p] = make_uint2((j.x & 0xffffu) | (j.y << 16), (j.z & 0xffffu) | (j.w << 16));
    }
    __device__ __forceinline__ void put1(const uint32_t j, const int32_t p) { stage[p] = (uint16_t)j; }
};

// The rows [row_lo, row_hi) of a flat nullable dictionary page, which hold the page's valid rows number
// first .. first + span_n - 1 (their indices lie in `stage`) and no others that have an index: row r is valid where
// its bit is set, the k-th valid row takes the dictionary value of the k-th index, a valid row behind the last index
// is null (deserialize_levels, tests/it/read/utils.rs:52-66, over an OptionalDictionary page,
// tests/it/read/primitive.rs:104-110).  The wave takes 256 rows a step, four consecutive rows per lane at a multiple
// of four (whole quads are stored with one instruction where the page's values and validity bytes are aligned); a
// prefix sum over the lanes' valid rows says which index a row takes.  Reads go through buffer descriptors: none is
// under a branch, and the bitmap's words come through LDS (`wbuf`: kRingPlaceWords words, which the caller has
// filled from the word of row_lo on if `preloaded` -- asked for before the span was expanded, so that nobody waits).
constexpr uint32_t kRingPlaceWords = 96;   // the rows of 3072 bits (with these, three workgroups fit a compute unit's LDS beside dictionaries of 16 KiB)
constexpr uint32_t kRingStageBytes = kRowDirStep * 2u + kRingPlaceWords * 4u;  // LDS per expanding wave (kNullable)
template <int kWords>
__device__ __forceinline__ void ring_place_span(const RingNullable &nl, const uint16_t *stage, uint32_t *wbuf, bool preloaded,
                                                const uint32_t *dict, const uint32_t dict_len, const uint32_t span_n,
                                                const uint32_t row_lo, const uint32_t row_hi, bool &bad, const uint32_t lane) {
    if (row_lo >= row_hi) return;
    const __amdgpu_buffer_rsrc_t bits = make_rsrc(nl.bitmap_addr, ((nl.rows + 31u) >> 5) * 4u);
    out_u32 *const out = (out_u32 *)nl.out_addr;
    __attribute__((address_space(1))) uint8_t *const validity = (__attribute__((address_space(1))) uint8_t *)nl.validity_addr;
    const bool aligned = ((nl.out_addr & 15u) | (nl.validity_addr & 3u)) == 0u;
    uint32_t used = 0;                // valid rows before this step
    uint32_t chunk_w = row_lo >> 5;   // wbuf holds the bitmap's words chunk_w .. chunk_w + kRingPlaceWords - 1
    for (uint32_t r_base = row_lo & ~3u; r_base < row_hi; r_base += 256u) {
        if (!preloaded || ((r_base + 255u) >> 5) >= chunk_w + kRingPlaceWords) {
            // (a span whose valid rows are spread over more than 3000 rows: the next words, one trip to memory)
            chunk_w = r_base >> 5;
            __builtin_amdgcn_wave_barrier();
            const uint32_t x0 = __builtin_amdgcn_raw_buffer_load_b32(bits, (chunk_w + lane) * 4u, 0, 0);
            const uint32_t x1 = __builtin_amdgcn_raw_buffer_load_b32(bits, (chunk_w + 64u + lane) * 4u, 0, 0);
            wbuf[lane] = x0;
            if (lane + 64u < kRingPlaceWords) wbuf[lane + 64u] = x1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            preloaded = true;
        }
        const uint32_t r0 = r_base + 4u * lane;
        // the validity bits of rows r0 .. r0 + 3, those outside [row_lo, row_hi) cleared
        // (rows are fewer than 2^31: the differences are taken as signed numbers and clamped to 0 .. 4)
        const uint32_t below = (uint32_t)max(0, min((int32_t)(row_lo - r0), 4));
        const uint32_t upto = (uint32_t)max(0, min((int32_t)(row_hi - r0), 4));
        const uint32_t inside = (15u << below) & ((1u << upto) - 1u);  // bit k: row r0 + k is one of [row_lo, row_hi)
        const uint32_t nib = (wbuf[min((r0 >> 5) - chunk_w, kRingPlaceWords - 1u)] >> (r0 & 31u)) & inside;
        const uint32_t c = (uint32_t)__builtin_popcount(nib);
        const uint32_t incl = wave_inclusive_sum(c);
        uint32_t rl = used + incl - c;
        used += __builtin_amdgcn_readlane(incl, 63);
        // the four rows' indices, then their values: every read is asked for before the first is waited for
        uint32_t rank[4], j[4], at[4], pres = 0;
        rank[0] = rl;
#pragma unroll
        for (uint32_t k = 1; k < 4u; ++k) rank[k] = rank[k - 1u] + ((nib >> (k - 1u)) & 1u);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) j[k] = stage[rank[k] & (kRowDirStep - 1u)];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const bool present = ((nib >> k) & 1u) != 0u && rank[k] < span_n;
            const bool ok = j[k] < dict_len;
            bad |= present && !ok;
            pres |= (present ? 1u : 0u) << k;
            at[k] = present && ok ? j[k] * (uint32_t)kWords : 0xffffffffu;
        }
        uint32_t val[4][kWords];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
            for (int c2 = 0; c2 < kWords; ++c2) val[k][c2] = dict[at[k] == 0xffffffffu ? 0u : at[k] + c2];  // (entry 0: any LDS word)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
            for (int c2 = 0; c2 < kWords; ++c2) val[k][c2] = at[k] == 0xffffffffu ? 0u : val[k][c2];
        if (aligned && inside == 15u) {
            uint32_t flat[4 * kWords];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
                for (int c2 = 0; c2 < kWords; ++c2) flat[k * kWords + c2] = val[k][c2];
#pragma unroll
            for (int c2 = 0; c2 < kWords; ++c2) {
                const u32x4 o = {flat[4 * c2], flat[4 * c2 + 1], flat[4 * c2 + 2], flat[4 * c2 + 3]};
                __builtin_nontemporal_store(o, (out_u32x4 *)out + (size_t)(r0 >> 2) * kWords + c2);
            }
            // bits 0..3 -> bytes 0..3
            *(__attribute__((address_space(1))) uint32_t *)(validity + r0) = (pres * 0x00204081u) & 0x01010101u;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t r = r0 + k;
                if ((inside >> k) & 1u) {
#pragma unroll
                    for (int c2 = 0; c2 < kWords; ++c2) out[(size_t)r * kWords + c2] = val[k][c2];
                    validity[r] = (uint8_t)((pres >> k) & 1u);
                }
            }
        }
    }
}

// `count` positions from 0 on, all of them `value` (a page of width 0, or one without a stream).
template <typename Sink>
__device__ __forceinline__ void ring_fill_span(Sink &sink, const uint32_t value, const uint32_t count, const uint32_t lane) {
    const uint32_t mis = sink.mis();
    const u32x4 v = {value, value, value, value};
    const uint32_t q_end = (count + mis + 3u) >> 2;
    for (uint32_t q = lane; q < q_end; q += 64u) {
        const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
        if (p >= 0 && p + 4 <= (int32_t)count) {
            sink.put4(v, p);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (p + k >= 0 && p + k < (int32_t)count) sink.put1(value, p + k);
        }
    }
}

template <int W, uint32_t kSpanBlocks, uint32_t kRingStepGroup, typename Sink>
__device__ __forceinline__ void ring_expand_span(RingExpander &io, uint2 *tbl, void *cidx_lds, const Resume rs,
                                                 const uint32_t end, const uint32_t n, const uint32_t span_start,
                                                 const uint32_t block_n, Sink &sink, const uint32_t lane) {
    typedef typename RingSpan<kSpanBlocks>::cidx_t ring_cidx_t;
    constexpr uint32_t kRingSpanValues = RingSpan<kSpanBlocks>::kValues;
    constexpr uint32_t kRingSpanChunks = RingSpan<kSpanBlocks>::kChunks;
    constexpr uint32_t kRingChunkSlots = RingSpan<kSpanBlocks>::kSlots;
    ring_cidx_t *const cidx = (ring_cidx_t *)cidx_lds;
    constexpr uint32_t w = (uint32_t)W;
    constexpr uint32_t kLaneRuns = kRingTableRuns - 1;
    const uint8_t *const ring = io.sh.ring;
    const uint32_t rmask = io.sh.ring_bytes - 1u;
    const uint32_t bit_mask = io.sh.ring_bytes * 8u - 1u;
    const uint32_t max_vals = ((io.cap - 256u) / w) * 8u;  // values of one run the ring surely holds at once
    const uint32_t span_end = span_start + block_n;

    uint32_t kind = rs.rem_kind >> 31;
    uint32_t rem = rs.rem_kind & kRemMask;
    uint32_t payload = rs.payload;  // RLE: the value
    uint32_t skip = rs.skip;        // bit-packed only
    uint32_t ridx = rs.run_index;

    const uint32_t mis = sink.mis();
    uint32_t lo = 0;

    do {
        // ---- phase 1: the runs that cover [lo, hi) into the table ------------------------------------
#ifdef PQ2_RING_TIMING
        uint64_t *const clocks = io.clocks;
#endif
        const uint64_t clock_p1 = PQ2_CLOCK();
        uint32_t take0 = min(rem, block_n - lo);
        uint32_t y0 = payload;
        bool alone = take0 == block_n - lo;
        uint32_t from = kEndPos;  // the first stream byte this pass reads
        if (kind == kBitpacked) {
            const uint32_t b_lo = payload + (skip >> 3) * w;
            if (take0 > max_vals) take0 = max_vals, alone = true;
            from = b_lo;
            y0 = b_lo * 8u + (skip & 7u) * w - lo * w;  // ring bit position of block position 0 (taken mod 8R)
        }
        uint32_t runs = 0;
        if (!alone) {
            runs = io.await_runs(ridx, span_end, false, lane);
            if (io.aborted) return;
        }
        uint32_t hi = lo + take0;

        if (alone && kind == kRle) {  // one value over the whole span: no table, no index, no stream byte
            const u32x4 v = {payload, payload, payload, payload};
            const uint32_t q_end = (hi + mis + 3u) >> 2;
            for (uint32_t q = ((lo + mis) >> 2) + lane; q < q_end; q += 64u) {
                const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                if (p >= (int32_t)lo && p + 4 <= (int32_t)hi) {
                    sink.put4(v, p);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (p + k >= (int32_t)lo && p + k < (int32_t)hi) sink.put1(payload, p + k);
                }
            }
        } else {
            // header positions: lane r <- run ridx + 1 + r, as far as they are published
            uint32_t hp = kNotYet;
            if (!alone && lane < kLaneRuns && ridx + 1u + lane < runs) hp = io.sh.hq[(ridx + 1u + lane) & (kHq - 1u)];
            const uint32_t hp0 = __builtin_amdgcn_readfirstlane(hp);
            if (kind == kRle && hp0 < kNotYet) from = hp0;
            // what the pass reads lies behind `from`: say so, and wait for the bytes of the first run (the headers of
            // published runs have landed: the walk has read them; a bit-packed run further on whose bytes have not
            // ends the pass and becomes the first run of the next one)
            uint32_t avail = lds_load(&io.sh.ctl->landed);
            if (from != kEndPos) {
                io.raise_need(from, lane);
                const uint32_t first_end = payload + ((skip + take0 + 7u) >> 3) * w + 8u;
                if (kind == kBitpacked && avail < min(first_end, lds_load(&io.sh.ctl->final_hi))) {
                    io.await_bytes(first_end, lane);
                    if (io.aborted) return;
                    avail = lds_load(&io.sh.ctl->landed);
                }
            }
            const uint32_t reach = io.need >= kNoNeed ? avail : min(avail, io.need + io.cap);  // bytes this pass may read

#pragma unroll
            for (uint32_t i = 0; i < kRingSpanChunks * (uint32_t)sizeof(ring_cidx_t) / 1024u; ++i)
                *(u32x4 *)((uint8_t *)cidx + 16u * (lane + 64u * i)) = u32x4{0u, 0u, 0u, 0u};
            if (lane == 0) *(u32x4 *)&cidx[kRingSpanChunks] = u32x4{0u, 0u, 0u, 0u};
            uint32_t key = kRingNoRun, y = 0;
            uint32_t start = 0;
            uint32_t g = 0;
            if (!alone) {
                const bool valid = hp < kNotYet;  // a published header: the walk has read it, so it has landed
                LaneRun run;
                run.kind = kRle, run.count = 0, run.info = 0;
                if (valid) run = decode_lane<W>(ring, hp & rmask, hp, end);
                const uint32_t cnt = min(run.count, 2u * kRingSpanValues);
                start = hi + wave_inclusive_sum(cnt) - cnt;
                bool ok = valid;
                uint32_t value = run.info;
                if (run.kind == kBitpacked) {
                    const uint32_t t = start < block_n ? min(run.count, block_n - start) : 0u;
                    ok = ok && run.info + ((t + 7u) >> 3) * w + 8u <= reach;
                    value = run.info * 8u - start * w;
                }
                const uint64_t bad = __ballot(!ok);  // lane 63 is never valid
                g = (uint32_t)__builtin_ctzll(bad);
                const uint32_t start_g = __builtin_amdgcn_readlane(start, g);
                const uint32_t hp_g = __builtin_amdgcn_readlane(hp, g);
                const bool ends = g < kLaneRuns && hp_g == kEndPos;  // no more runs: zeros follow (mod.rs:95)
                if (lane < g && start < block_n) {
                    key = (start << 1) | run.kind;
                    y = value;
                }
                if (ends && lane == g && start < block_n) key = (start << 1) | kRle;  // y stays 0
                hi = (start_g >= block_n || ends) ? block_n : start_g;
            }
            tbl[(lane + 1u) & 63u] = lane == 63u ? make_uint2((lo << 1) | kind, y0) : make_uint2(key, y);
            // the table entry of the run before this lane's (lane 0: the pass's first run) and the first position
            // of the run behind it, for the chunks in which runs start
            const uint32_t prev_key = __builtin_amdgcn_update_dpp((lo << 1) | kind, key, 0x138, 0xf, 0xf, false);  // wave_shr:1
            const uint32_t prev_y = __builtin_amdgcn_update_dpp(y0, y, 0x138, 0xf, 0xf, false);
            const uint32_t next_start = __builtin_amdgcn_update_dpp(
                0xffffffffu, key != kRingNoRun ? start : 0xffffffffu, 0x130, 0xf, 0xf, false);  // wave_shl:1
            PQ2_CLOCK_ADD(4, clock_p1);
            const uint64_t clock_ix = PQ2_CLOCK();
            // ---- index: the run that holds the first position of every chunk ---------------------------
            // The runs' first positions ascend with the lane, so do the chunks they mark: of the lanes that mark
            // one chunk the last one's number is the one to keep, and it is the only one that writes.
            {
                const uint32_t c = key != kRingNoRun ? (start + mis + 3u) >> 2 : 0xffffffffu;
                const uint32_t c_next = __builtin_amdgcn_update_dpp(0xffffffffu, c, 0x130, 0xf, 0xf, false);  // wave_shl:1
                if (key != kRingNoRun && c != c_next) cidx[c] = (ring_cidx_t)(lane + 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!alone) {
                if constexpr (kSpanBlocks == 4) {
                // lane l owns the 16 consecutive chunks from 16 l on: four per dword
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                const u32x4 mm = *(const u32x4 *)&cidx[16u * lane];
                uint32_t d[4] = {mm.x, mm.y, mm.z, mm.w};
                // the largest of the lane's sixteen (even and odd bytes as pairs of 16-bit numbers)
                u16x2 most = {0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    most = __builtin_elementwise_max(most, __builtin_bit_cast(u16x2, d[i] & 0x00ff00ffu));
                    most = __builtin_elementwise_max(most, __builtin_bit_cast(u16x2, (d[i] >> 8) & 0x00ff00ffu));
                }
                const uint32_t run_max = max((uint32_t)most.x, (uint32_t)most.y);
                const uint32_t upto = wave_inclusive_max(run_max);
                uint32_t run = __builtin_amdgcn_update_dpp(0u, upto, 0x138, 0xf, 0xf, false);  // wave_shr:1
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t o = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        run = max(run, (d[i] >> (8 * k)) & 0xffu);
                        o |= run << (8 * k);
                    }
                    d[i] = o;
                }
                *(u32x4 *)&cidx[16u * lane] = u32x4{d[0], d[1], d[2], d[3]};
                if (lane == 63u) {  // two more chunks exist when the output is not 16-byte aligned
                    const uint32_t c0 = max(upto, (uint32_t)cidx[kRingSpanChunks]);
                    cidx[kRingSpanChunks] = (ring_cidx_t)c0;
                    cidx[kRingSpanChunks + 1u] = (ring_cidx_t)max(c0, (uint32_t)cidx[kRingSpanChunks + 1u]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                // lane l owns the 8 consecutive chunks from 8 l on: two per dword
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u32x4 mm = *(const u32x4 *)&cidx[8u * lane];
                uint32_t d[4] = {mm.x, mm.y, mm.z, mm.w};
                uint32_t run_max = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t a = max(d[i] & 0xffffu, run_max);
                    const uint32_t b = max(d[i] >> 16, a);
                    run_max = b;
                    d[i] = a | (b << 16);
                }
                const uint32_t upto = wave_inclusive_max(run_max);
                const uint32_t before = __builtin_amdgcn_update_dpp(0u, upto, 0x138, 0xf, 0xf, false);  // wave_shr:1
                const u16x2 bb = {(unsigned short)before, (unsigned short)before};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, d[i]), bb));
                *(u32x4 *)&cidx[8u * lane] = u32x4{d[0], d[1], d[2], d[3]};
                if (lane == 63u) {  // two more chunks exist when the output is not 16-byte aligned
                    const uint32_t c0 = max(upto, (uint32_t)cidx[kRingSpanChunks]);
                    cidx[kRingSpanChunks] = (ring_cidx_t)c0;
                    cidx[kRingSpanChunks + 1u] = (ring_cidx_t)max(c0, (uint32_t)cidx[kRingSpanChunks + 1u]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            }
            PQ2_CLOCK_ADD(5, clock_ix);
            const uint64_t clock_p2 = PQ2_CLOCK();
            // ---- phase 2: one chunk per lane and step ---------------------------------------------------
            const uint32_t q_end = (hi + mis + 3u) >> 2;
            const uint32_t q_first = ((lo + mis) >> 2) + lane;
            // A chunk q is stored by these steps if it lies inside the pass, q_lo <= q < q_hi (one unsigned
            // comparison of q - q_lo), and the run behind the one that holds its first position starts behind its
            // last one: 2 (p + 3) + 1 < that run's key, with p = 4 q - mis the chunk's first position.
            const uint32_t q_lo = (lo + mis + 3u) >> 2, q_hi = (hi + mis) >> 2;
            const uint32_t q_span = q_hi > q_lo ? q_hi - q_lo : 0u;
            const uint32_t q_rel = q_first - q_lo;
            const uint32_t key_first = 8u * q_first + 7u - 2u * mis;
            // the first and the last chunk of a pass that does not begin or end with a chunk: value by value, below
            const bool ragged = ((lo + mis) & 3u) != 0u || ((hi + mis) & 3u) != 0u;
            constexpr uint32_t kSteps = 4u * kSpanBlocks + 1u;
#pragma unroll
            for (uint32_t g0 = 0; g0 < kSteps; g0 += kRingStepGroup) {
                if (((lo + mis) >> 2) + 64u * g0 >= q_end) break;  // a pass of few short runs ends early
                uint32_t j[kRingStepGroup];
#pragma unroll
                for (uint32_t i = 0; i < kRingStepGroup; ++i)
                    if (g0 + i < kSteps) j[i] = cidx[min(q_first + 64u * (g0 + i), kRingChunkSlots - 1u)];
                // (the pins go from the value asked for last to the one asked for first: LDS answers in order, so the
                // first wait is the only one)
#pragma unroll
                for (uint32_t i = kRingStepGroup; i-- > 0;)
                    if (g0 + i < kSteps) pin(j[i]);
                uint2 e[kRingStepGroup];
                uint32_t nxt[kRingStepGroup];
#pragma unroll
                for (uint32_t i = 0; i < kRingStepGroup; ++i)
                    if (g0 + i < kSteps) e[i] = tbl[j[i]], nxt[i] = tbl[j[i] + 1u].x;
#pragma unroll
                for (uint32_t i = kRingStepGroup; i-- > 0;)
                    if (g0 + i < kSteps) pin(nxt[i]), pin(e[i].y), pin(e[i].x);
                FieldWords<W> x[kRingStepGroup];
#pragma unroll
                for (uint32_t i = 0; i < kRingStepGroup; ++i)
                    if (g0 + i < kSteps) {
                        const int32_t p = (int32_t)((q_first + 64u * (g0 + i)) * 4u) - (int32_t)mis;
                        x[i] = load_fields<W>(ring, (e[i].y + (uint32_t)p * w) & bit_mask);
                    }
#pragma unroll
                for (uint32_t i = kRingStepGroup; i-- > 0;)
                    if (g0 + i < kSteps) x[i].pin_all();
#pragma unroll
                for (uint32_t i = 0; i < kRingStepGroup; ++i)
                    if (g0 + i < kSteps) {
                        const uint32_t q = q_first + 64u * (g0 + i);
                        const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                        const bool single = key_first + 512u * (g0 + i) < nxt[i];
                        const bool full = q_rel + 64u * (g0 + i) < q_span;
                        const u32x4 f = x[i].fields4();
                        const bool rle = (e[i].x & 1u) != 0u;
                        const uint32_t r = e[i].y;
                        const u32x4 v = {rle ? r : f.x, rle ? r : f.y, rle ? r : f.z, rle ? r : f.w};
                        if (single & full) sink.put4(v, p);
                    }
            }
            PQ2_CLOCK_ADD(6, clock_p2);
            const uint64_t clock_st = PQ2_CLOCK();
            if (__builtin_expect(ragged, 0)) {
#pragma unroll
                for (uint32_t side = 0; side < 2u; ++side) {
                    const uint32_t q = side == 0 ? (lo + mis) >> 2 : q_end - 1u;
                    const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                    const int32_t at = p + (int32_t)(lane & 3u);
                    const uint32_t j = cidx[q];
                    const uint2 e = tbl[j];
                    const uint32_t nxt = tbl[j + 1u].x;
                    const bool single = ((((uint32_t)(p + 3)) << 1) | 1u) < nxt;
                    const bool full = p >= (int32_t)lo && p + 4 <= (int32_t)hi;
                    const uint32_t f = field_at<W>(ring, (e.y + (uint32_t)at * w) & bit_mask);
                    if (lane < 4u && single && !full && at >= (int32_t)lo && at < (int32_t)hi)
                        sink.put1((e.x & 1u) ? e.y : f, at);
                }
            }
            // The chunks in which a run starts are left to the lanes of those runs: of the lanes whose runs start in
            // one chunk the lowest writes it, whole (one 16-byte store where the chunk lies inside the pass: the
            // steps above left a hole of 16 bytes there, and a hole filled by one store is one write for the memory
            // side of the cache instead of four).  A run that starts with its chunk, with no other start behind it
            // in the chunk, has been written by the steps above.
            const uint32_t q_run = key != kRingNoRun ? (start + mis) >> 2 : 0xffffffffu;
            const uint32_t q_below = __builtin_amdgcn_update_dpp(0xffffffffu, q_run, 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (key != kRingNoRun && q_run != q_below) {
                const uint32_t q = q_run;
                const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;  // p <= start < p + 4
                const int32_t first_pos = max(p, (int32_t)lo), past = min(p + 4, (int32_t)hi);
                const bool whole = p >= (int32_t)lo && p + 4 <= (int32_t)hi;
                // the usual case: the run before holds everything in front of this run's first value (its table
                // entry came over from the lane below), and no further run starts inside the chunk
                const bool simple = (int32_t)(prev_key >> 1) <= first_pos && next_start >= (uint32_t)past;
                const bool written = (int32_t)start == p && next_start >= (uint32_t)(p + 4);
                uint32_t v[4];
                if (written) {
                } else if (simple) {
                    uint32_t f[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t pk = (uint32_t)max(p + k, (int32_t)lo);
                        f[k] = field_at<W>(ring, ((pk >= start ? y : prev_y) + pk * w) & bit_mask);
                    }
#pragma unroll
                    for (int k = 4; k-- > 0;) pin(f[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool own = p + k >= (int32_t)start;
                        const uint32_t ek = own ? key : prev_key, ey = own ? y : prev_y;
                        v[k] = (ek & 1u) ? ey : f[k];
                    }
                } else {
                    // the chunk's four positions lie in the run that holds its first position and the up to three
                    // behind it (runs in the table hold at least one value): their entries, then the four fields
                    const uint32_t j0 = cidx[q];
                    uint2 t[5];  // entries j0 .. j0 + 4 (behind the table's end: the sentinel)
#pragma unroll
                    for (uint32_t i = 0; i < 5u; ++i) t[i] = tbl[min(j0 + i, kRingTableRuns + 1u)];
#pragma unroll
                    for (uint32_t i = 5u; i-- > 0;) pin(t[i].y), pin(t[i].x);
                    uint2 cur[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t pk = (uint32_t)max(p + k, (int32_t)lo);
                        if (((pk << 1) | 1u) >= t[1].x) {  // one step forward is enough: the window moves up
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i) t[i] = t[i + 1u];
                        }
                        cur[k] = t[0];
                    }
                    uint32_t f[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t pk = (uint32_t)max(p + k, (int32_t)lo);
                        f[k] = field_at<W>(ring, (cur[k].y + pk * w) & bit_mask);
                    }
#pragma unroll
                    for (int k = 4; k-- > 0;) pin(f[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (cur[k].x & 1u) ? cur[k].y : f[k];
                }
                if (!written) {
                    if (whole) {
                        sink.put4(u32x4{v[0], v[1], v[2], v[3]}, p);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (p + k >= first_pos && p + k < past) sink.put1(v[k], p + k);
                    }
                }
            }
            PQ2_CLOCK_ADD(7, clock_st);
            // ---- next pass: run ridx + 1 + g becomes the first run -----------------------------------------
            if (hi < block_n && !alone) {
                ridx += 1u + g;
                // everything before that run is written: the wave needs that run's header and nothing below it
                const uint32_t have = io.await_runs(ridx - 1u, span_end, true, lane);
                if (io.aborted) return;
                (void)have;
                const uint32_t at = __builtin_amdgcn_readfirstlane(io.sh.hq[ridx & (kHq - 1u)]);
                io.raise_need(at, lane);
                if (at != kEndPos) {
                    io.await_bytes(at + 32u, lane);
                    if (io.aborted) return;
                }
                auto rd = [&](uint32_t a) -> uint32_t {
                    return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&ring[a & rmask]);
                };
                const RunInfo run = decode_run(rd, at, end, w, n - (span_start + hi));
                if (run.kind >= kEnd || run.count == 0) {  // the closing entry: the stream is exhausted, zeros follow
                    kind = kRle, rem = kRemMask, payload = 0, skip = 0;
                } else {
                    kind = run.kind, rem = run.count, skip = 0;
                    payload = run.kind == kRle ? run.value : run.payload;
                }
            }
        }
        if (hi < block_n && alone) {  // the first run continues
            rem -= take0;
            skip += take0;
        }
        lo = hi;
    } while (lo < block_n);
}

// 32 stream bits from bit position `bit` of the ring on.
__device__ __forceinline__ uint32_t ring_bits32(const uint8_t *ring, uint32_t bit) {
    const uint32_t *src = (const uint32_t *)(ring + ((bit >> 3) & ~3u));
    return __builtin_amdgcn_alignbit(src[1], src[0], bit & 31u);
}

constexpr uint32_t kRingBitsSpanValues = 4u * kBlockValues;        // a span of the bitmap mode: the whole tile
constexpr uint32_t kRingBitsSpanWords = kRingBitsSpanValues / 32u;  // ... 128 words of validity bits
static_assert(kRingBitsSpanWords + 16u <= RingSpan<2>::kBytes, "the word index fits the chunk index's LDS");

// One wave turns the rows [span_start, span_start + block_n) of a page of one-bit definition levels into validity
// bits, row r of the span at bit r % 32 of words[r / 32]: what a consumer of DefLevelsDecoder::Bitmap assembles from
// the runs HybridRleIter hands it (src/deserialize/hybrid_rle.rs:90-115: a bit-packed run is a slice of the bitmap
// as it stands, an RLE run repeats `value[0] == 1`).  ring_expand_span with words of 32 rows in place of chunks of
// 4 values: the run table of a pass is the same, the index says which run holds each word's first row, and a lane
// assembles one word from the runs that overlap it -- no value is ever expanded.  A pass that ends inside a word
// hands the word's bits on to the next pass (`carry`); spans begin and end with words.
__device__ __forceinline__ void ring_expand_span_bits(RingExpander &io, uint2 *tbl, void *cidx_lds, const Resume rs,
                                                      const uint32_t end, const uint32_t n, const uint32_t span_start,
                                                      const uint32_t block_n, uint32_t *__restrict__ words,
                                                      uint32_t &set_bits, const uint32_t lane) {
    constexpr uint32_t w = 1u;
    constexpr uint32_t kLaneRuns = kRingTableRuns - 1;
    uint8_t *const cidx = (uint8_t *)cidx_lds;
    const uint8_t *const ring = io.sh.ring;
    const uint32_t rmask = io.sh.ring_bytes - 1u;
    const uint32_t bit_mask = io.sh.ring_bytes * 8u - 1u;
    const uint32_t max_vals = ((io.cap - 256u) / w) * 8u;  // values of one run the ring surely holds at once
    const uint32_t span_end = span_start + block_n;

    uint32_t kind = rs.rem_kind >> 31;
    uint32_t rem = rs.rem_kind & kRemMask;
    uint32_t payload = rs.payload;  // RLE: the value
    uint32_t skip = rs.skip;        // bit-packed only
    uint32_t ridx = rs.run_index;
    uint32_t lo = 0;
    uint32_t carry = 0;  // the bits below row `lo` of the word that holds it

    do {
        // ---- the runs that cover [lo, hi) into the table (as ring_expand_span's phase 1) ---------------------
        uint32_t take0 = min(rem, block_n - lo);
        uint32_t y0 = payload;
        bool alone = take0 == block_n - lo;
        uint32_t from = kEndPos;  // the first stream byte this pass reads
        if (kind == kBitpacked) {
            const uint32_t b_lo = payload + (skip >> 3) * w;
            if (take0 > max_vals) take0 = max_vals, alone = true;
            from = b_lo;
            y0 = b_lo * 8u + (skip & 7u) * w - lo * w;  // ring bit position of the span's row 0 (taken mod 8R)
        }
        uint32_t runs = 0;
        if (!alone) {
            runs = io.await_runs(ridx, span_end, false, lane);
            if (io.aborted) return;
        }
        uint32_t hi = lo + take0;
        uint32_t hp = kNotYet;
        if (!alone && lane < kLaneRuns && ridx + 1u + lane < runs) hp = io.sh.hq[(ridx + 1u + lane) & (kHq - 1u)];
        const uint32_t hp0 = __builtin_amdgcn_readfirstlane(hp);
        if (kind == kRle && hp0 < kNotYet) from = hp0;
        uint32_t avail = lds_load(&io.sh.ctl->landed);
        if (from != kEndPos) {
            io.raise_need(from, lane);
            const uint32_t first_end = payload + ((skip + take0 + 7u) >> 3) * w + 8u;
            if (kind == kBitpacked && avail < min(first_end, lds_load(&io.sh.ctl->final_hi))) {
                io.await_bytes(first_end, lane);
                if (io.aborted) return;
                avail = lds_load(&io.sh.ctl->landed);
            }
        }
        const uint32_t reach = io.need >= kNoNeed ? avail : min(avail, io.need + io.cap);  // bytes this pass may read

        if (lane < (kRingBitsSpanWords + 16u) / 16u) *(u32x4 *)(cidx + 16u * lane) = u32x4{0u, 0u, 0u, 0u};
        uint32_t key = kRingNoRun, y = 0;
        uint32_t start = 0;
        uint32_t g = 0;
        if (!alone) {
            const bool valid = hp < kNotYet;  // a published header: the walk has read it, so it has landed
            LaneRun run;
            run.kind = kRle, run.count = 0, run.info = 0;
            if (valid) run = decode_lane<1>(ring, hp & rmask, hp, end);
            const uint32_t cnt = min(run.count, 2u * kRingBitsSpanValues);
            start = hi + wave_inclusive_sum(cnt) - cnt;
            bool ok = valid;
            uint32_t value = run.info;
            if (run.kind == kBitpacked) {
                const uint32_t t = start < block_n ? min(run.count, block_n - start) : 0u;
                ok = ok && run.info + ((t + 7u) >> 3) * w + 8u <= reach;
                value = run.info * 8u - start * w;
            }
            const uint64_t bad = __ballot(!ok);  // lane 63 is never valid
            g = (uint32_t)__builtin_ctzll(bad);
            const uint32_t start_g = __builtin_amdgcn_readlane(start, g);
            const uint32_t hp_g = __builtin_amdgcn_readlane(hp, g);
            const bool ends = g < kLaneRuns && hp_g == kEndPos;  // no more runs: the rows behind are not the stream's
            if (lane < g && start < block_n) {
                key = (start << 1) | run.kind;
                y = value;
            }
            if (ends && lane == g && start < block_n) key = (start << 1) | kRle;  // y stays 0: null rows
            hi = (start_g >= block_n || ends) ? block_n : start_g;
        }
        tbl[(lane + 1u) & 63u] = lane == 63u ? make_uint2((lo << 1) | kind, y0) : make_uint2(key, y);
        // ---- index: the run that holds the first row of every word -------------------------------------------
        {
            const uint32_t c = key != kRingNoRun ? (start + 31u) >> 5 : 0xffffffffu;
            const uint32_t c_next = __builtin_amdgcn_update_dpp(0xffffffffu, c, 0x130, 0xf, 0xf, false);  // wave_shl:1
            if (key != kRingNoRun && c != c_next) cidx[c] = (uint8_t)(lane + 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!alone) {  // lane l owns the words 2 l and 2 l + 1
            const uint32_t d = *(const unsigned short *)&cidx[2u * lane];
            uint32_t m0 = d & 0xffu, m1 = max(m0, d >> 8);
            const uint32_t upto = wave_inclusive_max(m1);
            const uint32_t before = __builtin_amdgcn_update_dpp(0u, upto, 0x138, 0xf, 0xf, false);  // wave_shr:1
            m0 = max(m0, before), m1 = max(m1, before);
            *(unsigned short *)&cidx[2u * lane] = (unsigned short)(m0 | (m1 << 8));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- one word per lane and step -------------------------------------------------------------------------
        const uint32_t wq0 = lo >> 5, wq_end = (hi + 31u) >> 5;
        const bool hands_on = (hi & 31u) != 0u && hi < block_n;  // the pass ends inside a word that the next one completes
        uint32_t carry_next = 0;
#pragma unroll
        for (uint32_t i = 0; i < 2u; ++i) {
            if (wq0 + 64u * i >= wq_end) break;
            const uint32_t q = wq0 + 64u * i + lane;
            const bool active = q < wq_end;
            uint32_t word = 0;
            if (active) {
                const uint32_t row0 = q * 32u;
                uint32_t j = cidx[min(q, kRingBitsSpanWords)];
                for (;;) {
                    const uint2 e = tbl[j];
                    const uint32_t nxt = tbl[j + 1u].x;
                    const uint32_t r_end = min(nxt >> 1, hi);
                    const uint32_t a = max(e.x >> 1, row0), b = min(r_end, row0 + 32u);
                    if (a < b) {
                        const uint32_t bits = (e.x & 1u) ? (e.y == 1u ? 0xffffffffu : 0u) : ring_bits32(ring, (e.y + row0) & bit_mask);
                        const uint32_t m = (b - a == 32u) ? 0xffffffffu : (((1u << (b - a)) - 1u) << (a - row0));
                        word |= bits & m;
                    }
                    if (r_end >= row0 + 32u || r_end >= hi) break;
                    ++j;
                }
                if (q == wq0) word |= carry;
            }
            const uint32_t last_q = wq_end - 1u;
            if (hands_on && last_q >= wq0 + 64u * i && last_q < wq0 + 64u * i + 64u)
                carry_next = __builtin_amdgcn_readlane(word, __builtin_amdgcn_readfirstlane(last_q - wq0 - 64u * i));
            if (active && !(hands_on && q == last_q)) {
                words[q] = word;
                set_bits += (uint32_t)__popc(word);
            }
        }
        carry = carry_next;
        // ---- next pass: run ridx + 1 + g becomes the first run -----------------------------------------------
        if (hi < block_n && !alone) {
            ridx += 1u + g;
            const uint32_t have = io.await_runs(ridx - 1u, span_end, true, lane);
            if (io.aborted) return;
            (void)have;
            const uint32_t at = __builtin_amdgcn_readfirstlane(io.sh.hq[ridx & (kHq - 1u)]);
            io.raise_need(at, lane);
            if (at != kEndPos) {
                io.await_bytes(at + 32u, lane);
                if (io.aborted) return;
            }
            auto rd = [&](uint32_t a) -> uint32_t {
                return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&ring[a & rmask]);
            };
            const RunInfo run = decode_run(rd, at, end, w, n - (span_start + hi));
            if (run.kind >= kEnd || run.count == 0) {  // the closing entry: the stream is exhausted
                kind = kRle, rem = kRemMask, payload = 0, skip = 0;
            } else {
                kind = run.kind, rem = run.count, skip = 0;
                payload = run.kind == kRle ? run.value : run.payload;
            }
        }
        if (hi < block_n && alone) {  // the first run continues
            rem -= take0;
            skip += take0;
        }
        lo = hi;
    } while (lo < block_n);
}

// One wave decodes the span [span_start, span_start + block_n) of a page of one-bit levels into u32 values: the
// decode of HybridRleDecoder at width 1 (definition and repetition levels of flat columns), by way of the span's bits.
// ring_expand_span spends some twenty-five instructions on every 256 values whatever the width; at width 1 a pass
// can assemble its rows 32 to a word first (as ring_expand_span_bits does: a lane ORs together the parts of the runs
// that overlap its word) and then spread the words' bits over the output, four values per lane and step -- a
// third of the instructions, and pages of levels are bound by them.  An RLE run whose value is not 0 or 1 (the value
// byte is not masked to the width, mod.rs:51-55) has no bit to stand for it: a pass that holds one writes its runs
// one after the other instead.
template <uint32_t kSpanBlocks>
__device__ __forceinline__ void ring_expand_span_levels(RingExpander &io, uint2 *tbl, void *cidx_lds, const Resume rs,
                                                        const uint32_t end, const uint32_t n, const uint32_t span_start,
                                                        const uint32_t block_n, IndexSink &sink, const uint32_t lane) {
    constexpr uint32_t kSpanValues = kSpanBlocks * kBlockValues;
    constexpr uint32_t kSpanWords = kSpanValues / 32u;
    static_assert(kSpanWords + 16u + 16u + (kSpanWords + 4u) * 4u <= RingSpan<2>::kBytes, "index and words fit the chunk index's LDS");
    uint32_t *const wbuf = (uint32_t *)((uint8_t *)cidx_lds + ((kSpanWords + 16u + 15u) & ~15u));  // kSpanWords + 4 words
    const uint32_t mis = sink.mis();
    constexpr uint32_t w = 1u;
    constexpr uint32_t kLaneRuns = kRingTableRuns - 1;
    uint8_t *const cidx = (uint8_t *)cidx_lds;
    const uint8_t *const ring = io.sh.ring;
    const uint32_t rmask = io.sh.ring_bytes - 1u;
    const uint32_t bit_mask = io.sh.ring_bytes * 8u - 1u;
    const uint32_t max_vals = ((io.cap - 256u) / w) * 8u;  // values of one run the ring surely holds at once
    const uint32_t span_end = span_start + block_n;

    uint32_t kind = rs.rem_kind >> 31;
    uint32_t rem = rs.rem_kind & kRemMask;
    uint32_t payload = rs.payload;  // RLE: the value
    uint32_t skip = rs.skip;        // bit-packed only
    uint32_t ridx = rs.run_index;
    uint32_t lo = 0;
    // the span's bits, 0 until a pass puts its rows in
    for (uint32_t i = lane; i < kSpanWords + 4u; i += 64u) wbuf[i] = 0u;

    do {
        // ---- the runs that cover [lo, hi) into the table (as ring_expand_span's phase 1) ---------------------
        uint32_t take0 = min(rem, block_n - lo);
        uint32_t y0 = payload;
        bool alone = take0 == block_n - lo;
        uint32_t from = kEndPos;  // the first stream byte this pass reads
        if (kind == kBitpacked) {
            const uint32_t b_lo = payload + (skip >> 3) * w;
            if (take0 > max_vals) take0 = max_vals, alone = true;
            from = b_lo;
            y0 = b_lo * 8u + (skip & 7u) * w - lo * w;  // ring bit position of the span's row 0 (taken mod 8R)
        }
        uint32_t runs = 0;
        if (!alone) {
            runs = io.await_runs(ridx, span_end, false, lane);
            if (io.aborted) return;
        }
        uint32_t hi = lo + take0;
        uint32_t hp = kNotYet;
        if (!alone && lane < kLaneRuns && ridx + 1u + lane < runs) hp = io.sh.hq[(ridx + 1u + lane) & (kHq - 1u)];
        const uint32_t hp0 = __builtin_amdgcn_readfirstlane(hp);
        if (kind == kRle && hp0 < kNotYet) from = hp0;
        uint32_t avail = lds_load(&io.sh.ctl->landed);
        if (from != kEndPos) {
            io.raise_need(from, lane);
            const uint32_t first_end = payload + ((skip + take0 + 7u) >> 3) * w + 8u;
            if (kind == kBitpacked && avail < min(first_end, lds_load(&io.sh.ctl->final_hi))) {
                io.await_bytes(first_end, lane);
                if (io.aborted) return;
                avail = lds_load(&io.sh.ctl->landed);
            }
        }
        const uint32_t reach = io.need >= kNoNeed ? avail : min(avail, io.need + io.cap);  // bytes this pass may read

        if (lane < (kSpanWords + 16u + 15u) / 16u) *(u32x4 *)(cidx + 16u * lane) = u32x4{0u, 0u, 0u, 0u};
        uint32_t key = kRingNoRun, y = 0;
        uint32_t start = 0;
        uint32_t g = 0;
        if (!alone) {
            const bool valid = hp < kNotYet;  // a published header: the walk has read it, so it has landed
            LaneRun run;
            run.kind = kRle, run.count = 0, run.info = 0;
            if (valid) run = decode_lane<1>(ring, hp & rmask, hp, end);
            const uint32_t cnt = min(run.count, 2u * kSpanValues);
            start = hi + wave_inclusive_sum(cnt) - cnt;
            bool ok = valid;
            uint32_t value = run.info;
            if (run.kind == kBitpacked) {
                const uint32_t t = start < block_n ? min(run.count, block_n - start) : 0u;
                ok = ok && run.info + ((t + 7u) >> 3) * w + 8u <= reach;
                value = run.info * 8u - start * w;
            }
            const uint64_t bad = __ballot(!ok);  // lane 63 is never valid
            g = (uint32_t)__builtin_ctzll(bad);
            const uint32_t start_g = __builtin_amdgcn_readlane(start, g);
            const uint32_t hp_g = __builtin_amdgcn_readlane(hp, g);
            const bool ends = g < kLaneRuns && hp_g == kEndPos;  // no more runs: the rows behind are not the stream's
            if (lane < g && start < block_n) {
                key = (start << 1) | run.kind;
                y = value;
            }
            if (ends && lane == g && start < block_n) key = (start << 1) | kRle;  // y stays 0: null rows
            hi = (start_g >= block_n || ends) ? block_n : start_g;
        }
        tbl[(lane + 1u) & 63u] = lane == 63u ? make_uint2((lo << 1) | kind, y0) : make_uint2(key, y);
        // ---- index: the run that holds the first row of every word -------------------------------------------
        {
            const uint32_t c = key != kRingNoRun ? (start + 31u) >> 5 : 0xffffffffu;
            const uint32_t c_next = __builtin_amdgcn_update_dpp(0xffffffffu, c, 0x130, 0xf, 0xf, false);  // wave_shl:1
            if (key != kRingNoRun && c != c_next) cidx[c] = (uint8_t)(lane + 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!alone) {
            if constexpr (kSpanWords == 128u) {  // lane l owns the words 2 l and 2 l + 1
                const uint32_t d = *(const unsigned short *)&cidx[2u * lane];
                uint32_t m0 = d & 0xffu, m1 = max(m0, d >> 8);
                const uint32_t upto = wave_inclusive_max(m1);
                const uint32_t before = __builtin_amdgcn_update_dpp(0u, upto, 0x138, 0xf, 0xf, false);  // wave_shr:1
                m0 = max(m0, before), m1 = max(m1, before);
                *(unsigned short *)&cidx[2u * lane] = (unsigned short)(m0 | (m1 << 8));
            } else {  // lane l owns word l
                static_assert(kSpanWords == 64u || kSpanWords == 128u, "spans of 2 or 4 blocks");
                cidx[lane] = (uint8_t)wave_inclusive_max((uint32_t)cidx[lane]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // an RLE run of this pass whose value no bit stands for?
        const bool wide = __ballot(key != kRingNoRun && (key & 1u) != 0u && y > 1u) != 0ull || (kind == kRle && y0 > 1u);
        if (__builtin_expect(!wide, 1)) {
            // ---- the pass's rows into the span's bits: one word per lane and step -----------------------------
            const uint32_t wq0 = lo >> 5, wq_end = (hi + 31u) >> 5;
#pragma unroll
            for (uint32_t i = 0; i < kSpanWords / 64u; ++i) {
                if (wq0 + 64u * i >= wq_end) break;
                const uint32_t q = wq0 + 64u * i + lane;
                if (q < wq_end) {
                    const uint32_t row0 = q * 32u;
                    uint32_t word = 0;
                    uint32_t j = cidx[min(q, kSpanWords)];
                    for (;;) {
                        const uint2 e = tbl[j];
                        const uint32_t nxt = tbl[j + 1u].x;
                        const uint32_t r_end = min(nxt >> 1, hi);
                        const uint32_t a = max(e.x >> 1, row0), b = min(r_end, row0 + 32u);
                        if (a < b) {
                            const uint32_t bits = (e.x & 1u) ? (e.y == 1u ? 0xffffffffu : 0u) : ring_bits32(ring, (e.y + row0) & bit_mask);
                            const uint32_t m = (b - a == 32u) ? 0xffffffffu : (((1u << (b - a)) - 1u) << (a - row0));
                            word |= bits & m;
                        }
                        if (r_end >= row0 + 32u || r_end >= hi) break;
                        ++j;
                    }
                    wbuf[q] |= word;  // (the word that holds row `lo` has the bits of the pass before)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- the bits of rows [lo, hi) as values: one 16-byte chunk per lane and step ---------------------
            const uint32_t q_end = (hi + mis + 3u) >> 2;
            for (uint32_t q = ((lo + mis) >> 2) + lane; q < q_end; q += 64u) {
                const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;  // the chunk's first row
                if (p >= (int32_t)lo && p + 4 <= (int32_t)hi) {
                    const uint32_t wi = (uint32_t)p >> 5;
                    const uint32_t nib = __builtin_amdgcn_alignbit(wbuf[wi + 1u], wbuf[wi], (uint32_t)p & 31u);
                    sink.put4(u32x4{nib & 1u, (nib >> 1) & 1u, (nib >> 2) & 1u, (nib >> 3) & 1u}, p);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int32_t r = p + k;
                        if (r >= (int32_t)lo && r < (int32_t)hi) sink.put1((wbuf[(uint32_t)r >> 5] >> ((uint32_t)r & 31u)) & 1u, r);
                    }
                }
            }
        } else {
            // ---- run after run, a value per lane and step (streams no writer produces) -----------------------------
            for (uint32_t j = 0;; ++j) {
                const uint2 e = tbl[j];
                const uint32_t nxt = tbl[j + 1u].x;
                const uint32_t r_end = min(nxt >> 1, hi);
                for (uint32_t r = (e.x >> 1) + lane; r < r_end; r += 64u) {
                    const uint32_t bit = (ring[(((e.y + r) & bit_mask) >> 3)] >> ((e.y + r) & 7u)) & 1u;
                    sink.put1((e.x & 1u) ? e.y : bit, (int32_t)r);
                }
                if (r_end >= hi || j + 1u > kRingTableRuns) break;
            }
        }
        // ---- next pass: run ridx + 1 + g becomes the first run -----------------------------------------------
        if (hi < block_n && !alone) {
            ridx += 1u + g;
            const uint32_t have = io.await_runs(ridx - 1u, span_end, true, lane);
            if (io.aborted) return;
            (void)have;
            const uint32_t at = __builtin_amdgcn_readfirstlane(io.sh.hq[ridx & (kHq - 1u)]);
            io.raise_need(at, lane);
            if (at != kEndPos) {
                io.await_bytes(at + 32u, lane);
                if (io.aborted) return;
            }
            auto rd = [&](uint32_t a) -> uint32_t {
                return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&ring[a & rmask]);
            };
            const RunInfo run = decode_run(rd, at, end, w, n - (span_start + hi));
            if (run.kind >= kEnd || run.count == 0) {  // the closing entry: the stream is exhausted
                kind = kRle, rem = kRemMask, payload = 0, skip = 0;
            } else {
                kind = run.kind, rem = run.count, skip = 0;
                payload = run.kind == kRle ? run.value : run.payload;
            }
        }
        if (hi < block_n && alone) {  // the first run continues
            rem -= take0;
            skip += take0;
        }
        lo = hi;
    } while (lo < block_n);
}

// Selected rows: see tile_wanted in pq2_kernels.hip.
__device__ __forceinline__ bool ring_tile_wanted(const PageDev &pg, const uint32_t *__restrict__ tile_mask, uint32_t t) {
    if (pg.reserved == 0u || tile_mask == nullptr) return true;
    return ((tile_mask[pg.reserved - 1u + (t >> 5)] >> (t & 31u)) & 1u) != 0u;
}

constexpr uint32_t kRingCuTickets = 4096;  // entries of the per-compute-unit ticket table (as decode_kernel's)

// Which wave walks: the one on SIMD `ticket % 4` of its compute unit, so that the walks of the workgroups that
// share a compute unit sit on different SIMDs (see pick_walker in pq2_kernels.hip).  Contains a workgroup barrier.
//
// The ticket says how many workgroups of this launch came to the compute unit before this one (`ticket`).  The
// table has two halves that take turns from launch to launch (`epoch`): a workgroup counts in this launch's half
// and clears its compute unit's word of the other one, which the next launch counts in.
__device__ __forceinline__ uint32_t ring_pick_walker(uint32_t *cu_tickets, uint32_t *scratch, const uint32_t wave,
                                                     const uint32_t lane, const uint32_t epoch, uint32_t &ticket) {
    const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const uint32_t xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
    if (lane == 0) scratch[wave] = (hw_id >> 4) & 3u;                   // SIMD_ID
    if (threadIdx.x == 0) {
        const uint32_t cu = (((hw_id >> 8) & 0xffu) | (xcc_id << 8)) & (kRingCuTickets / 2u - 1u);
        const uint32_t half = (epoch & 1u) * (kRingCuTickets / 2u);
        scratch[7] = atomicAdd(&cu_tickets[half + cu], 1u);
        cu_tickets[(kRingCuTickets / 2u - half) + cu] = 0u;
    }
    __syncthreads();
    ticket = __builtin_amdgcn_readfirstlane(scratch[7]);
    const uint32_t target = scratch[7] & 3u;
    uint32_t walker = 0;
    for (uint32_t k = kRingExpanders + 1u; k-- > 0;)
        if (scratch[k] == target) walker = k;
    return __builtin_amdgcn_readfirstlane(walker);
}

#ifndef PQ2_RING_MIN_WAVES
#define PQ2_RING_MIN_WAVES 4
#endif
// What ring_kernel<., kWords > 0> needs of a page beside its PageDev (the host's GatherPage).
struct RingDict {
    uint64_t idx_addr;   // not used here
    uint64_t out_addr;   // the page's values, kWords dwords each
    uint64_t dict_addr;  // the page's dictionary
    uint32_t dict_len;
    uint32_t num_values, first_tile, reserved;  // not used here
};
constexpr uint32_t kRingFlagWholeTileSpans = 1u;  // ring_kernel's `flags`: a batch of nothing but levels (see kRingWholeTileBits)
constexpr uint32_t kRingFlagTurns = 2u;           // ... the workgroups of a compute unit take turns in priority
constexpr uint32_t kRingFlagStageShift = 8;       // ... from this bit on: where in LDS the expanding waves' StageSinks begin (kNullable)
constexpr uint32_t kRingTurnsMaxBits = 20;        // the widest page of a batch that is given turns
constexpr uint32_t kRingTurnsMinPages = 1025;     // ... and its fewest pages: one more than 256 compute units hold workgroups
constexpr uint32_t kRingDictMaxBits = 16;      // widest index the kernels with a dictionary are built for
constexpr uint32_t kRingNullableMaxBits = 12;  // ... and those that write the rows of nullable pages

// kWords = 0: the page's values are the decoded u32 (IndexSink).  kWords = 1, 2, 3: they are indices into the page's
// dictionary of kWords-dword values, which the workgroup copies to LDS (from byte `dict_at` of its LDS on) before
// it starts; `dicts` has a record per page, the statuses must be 0 at the launch.  kWords = -1: the pages hold
// one-bit definition levels and come out as validity bitmaps (ring_expand_span_bits), page p's at the 4-byte aligned
// address PageDev::out_addr; bit_meta[2 p] receives the rows the page's stream holds and bit_meta[2 p + 1], which
// must be 0 at the launch, the number of set bits.
//
// kNullable (with kWords > 0): the pages are the index streams of flat nullable dictionary pages, and the values go
// to the pages' ROWS: `nulls` has a RingNullable per page.  An expanding wave keeps the indices of a span in LDS
// (StageSink; kRowDirStep two-byte entries per wave behind the dictionary), looks up in the page's row directory
// which rows the span's valid rows lie in, and writes those rows, nulls included (ring_place_span): neither the
// indices nor the levels ever lie in device memory, only the validity bitmap the levels were turned into before.
template <bool kWide, int kWords, bool kNullable = false>
__global__ __launch_bounds__(kRingThreads, PQ2_RING_MIN_WAVES) void ring_kernel(const PageDev *__restrict__ pages, uint32_t n_pages,
                                                       int32_t *__restrict__ status, uint32_t *__restrict__ cu_tickets,
                                                       uint32_t ring_bytes, const uint32_t *__restrict__ tile_mask,
                                                       const RingDict *__restrict__ dicts, uint32_t dict_at,
                                                       uint32_t flags, uint32_t epoch,
                                                       uint32_t *__restrict__ bit_meta, const uint32_t *__restrict__ go,
                                                       const RingNullable *__restrict__ nulls) {
    static_assert(!kNullable || kWords > 0, "nullable pages are dictionary pages");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // a launch that was enqueued before the batch's schedule was final (density_kernel, pq2_density.hip) and has
    // been told that the other schedule takes the batch: nothing is touched
    if (go != nullptr && *go == 0u) return;

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    const uint32_t n = pg.num_values;
    if (n == 0) {
        if constexpr (kNullable) {  // no index: every row is null
            const RingNullable nl = nulls[page];
            bool bad = false;
            if (wave == 0)
                ring_place_span<kWords>(nl, (const uint16_t *)(lds + 1024), (uint32_t *)lds, false, (const uint32_t *)lds, 0u, 0u, 0u,
                                        nl.rows, bad, lane);
        }
        if (threadIdx.x == 0 && kWords <= 0) status[page] = 0;
        if (threadIdx.x == 0 && kWords < 0) bit_meta[2u * page] = 0u;
        return;
    }
#ifdef PQ2_RING_TRACE
    if (threadIdx.x == 0 && page < kRingTracePages) {
        g_ring_trace[3u * page] = __builtin_amdgcn_s_memrealtime();
        g_ring_trace[3u * page + 2u] = ((uint64_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8) |
                                       ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xffu);
    }
#endif
    uint32_t *const dict_lds = (uint32_t *)(lds + dict_at);
    uint32_t dict_len = 0;
    uint64_t value_addr = 0;
    if constexpr (kWords > 0) {
        const RingDict dp = dicts[page];
        dict_len = dp.dict_len;
        value_addr = dp.out_addr;
        typedef __attribute__((address_space(1))) const uint32_t gu32;
        for (uint32_t i = threadIdx.x; i < dp.dict_len * (uint32_t)kWords; i += kRingThreads) dict_lds[i] = ((gu32 *)dp.dict_addr)[i];
    }  // (the barrier inside ring_pick_walker below is also the one behind this copy)
    RingShared sh;
    sh.ring = lds;
    sh.ring_bytes = ring_bytes;
    sh.hq = (uint32_t *)(lds + ring_bytes + kMirror);
    sh.rq = (Resume *)(lds + ring_bytes + kMirror + kHq * 4u);
    sh.ctl = (RingCtl *)(lds + ring_bytes + kMirror + kHq * 4u + kRq * 16u);
    sh.rows_out = kWords < 0 ? bit_meta + 2u * page : nullptr;
    uint8_t *const tables = (uint8_t *)sh.ctl + sizeof(RingCtl);
    const uint32_t n_tiles = (n + kRingTileValues - 1u) / kRingTileValues;
    if (threadIdx.x < sizeof(RingCtl) / 4u) ((uint32_t *)sh.ctl)[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x < kRingExpanders)  // the first tiles are handed out here: wave e holds tile e
        sh.ctl->expander[threadIdx.x] = ((uint64_t)threadIdx.x << 32) | kTileFresh;
    if (threadIdx.x == 0) sh.ctl->next_tile = kRingExpanders;
    uint32_t ticket;
    const uint32_t walker = ring_pick_walker(cu_tickets, sh.ctl->pick, wave, lane, epoch, ticket);  // with a workgroup barrier inside
    // Which workgroup of a compute unit goes first where they compete for issue slots.  The hardware prefers the
    // older wave, so of the workgroups that come to a compute unit together the first one finishes first, its
    // successor starts first, is the oldest of the successors and finishes first again: a launch of two pages per
    // slot ends over a long tail in which the compute units run one, two, three workgroups short.  Priorities by
    // arrival turn that round: the workgroup that came last goes first, so the slots that took their second
    // page late are the ones that get through it fast, and the slots of a compute unit end together.
    // In turns, because a launch goes on: the workgroups that begin together (tickets 0-3, 8-11, ...) are left in
    // the order the hardware would give them anyway, first come first; their successors (4-7, 12-15, ...), which
    // begin one after the other as those end, get the reverse.  The walking wave keeps at least the priority it
    // had before there were turns: a page's walk is its serial part.
    // Only where there is something to turn round (kRingFlagTurns: the launcher sets it for batches of more pages than
    // the GPU holds workgroups, and of widths up to 20 bits -- wider pages, whose expanding waves are short of
    // instruction issue rather than of memory, lose 1-4 % by it, a batch of nothing but level pages 2 %, and a batch
    // of one round up to 7 %).
#if PQ2_RING_PRIO_TURNS
    const uint32_t rank = ticket & 3u;
    uint32_t turn = (((ticket & 4u) ? PQ2_RING_PRIO_SECOND : PQ2_RING_PRIO_FIRST) >> (4u * rank)) & 3u;
    if ((flags & kRingFlagTurns) == 0u) turn = 0u;
    const uint32_t prio_walk = max(turn, (uint32_t)PQ2_RING_WALK_PRIO), prio_expand = turn;
#else
    const uint32_t prio_walk = PQ2_RING_WALK_PRIO, prio_expand = 0u;
#endif
    {
        const uint32_t prio = wave == walker ? prio_walk : prio_expand;
        if (prio == 1u) __builtin_amdgcn_s_setprio(1);
        if (prio == 2u) __builtin_amdgcn_s_setprio(2);
        if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    }

    if (wave == walker) {
        ring_walk<kWide, (kWords > 0)>(pg, page, sh, status, lane);
        PQ2_TRACE_END(page);
        return;
    }
    const uint32_t e = wave - (wave > walker ? 1u : 0u);
    uint2 *tbl = (uint2 *)tables + e * (kRingTableRuns + 2u);
    uint8_t *const cidx = tables + kRingExpanders * (kRingTableRuns + 2u) * 8u + e * RingSpan<2>::kBytes;
    if (lane < 2) tbl[kRingTableRuns + lane] = make_uint2(kRingNoRun, 0u);  // read by tbl[j + 1] only

    RingExpander io;
    io.sh = sh;
    io.slot = &sh.ctl->expander[e];
    io.cap = ring_bytes - 2u * kSlot;
    io.status = status;
    io.page = page;
    io.aborted = false;
    io.merge = kWords > 0;
    bool bad_index = false;
    uint32_t set_bits = 0;  // (bitmap mode) set bits in the words this lane has stored
#ifdef PQ2_RING_TIMING
    for (int i = 0; i < 8; ++i) io.clocks[i] = 0;
    uint64_t *const clocks = io.clocks;
#endif
    const uint32_t w = pg.num_bits;
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    constexpr uint32_t kGroup = kWide ? PQ2_RING_STEP_GROUP_WIDE : PQ2_RING_STEP_GROUP;

    const uint64_t clock_begin = PQ2_CLOCK();
    uint32_t dir_rows[3] = {0u, 0u, 0u};  // (kNullable) the row directory's entries of the tile in hand
    for (uint32_t t = e; t < n_tiles;) {
        io.tile = t;
        io.need = kTileFresh;
        io.say(lane);
        if (ring_tile_wanted(pg, tile_mask, t * kRingTileBlocks / kBlocksPerTile)) {
            const uint32_t tile_first = t * kRingTileValues;
            out_u32 *const out = (out_u32 *)pg.out_addr + tile_first;
            // pages of levels: the tile as one span
            // (kRingFlagWholeTileSpans: the launcher says so for batches of nothing but such pages; beside wider pages the
            // whole-tile spans cost more than they bring: configs[4]'s mix loses 2.6 % with them)
            const uint32_t span_blocks =
                kWords < 0 ? kRingTileBlocks
                : (kWords == 0 && kRingTileBlocks == 4 && (flags & kRingFlagWholeTileSpans) != 0u && w != 0u && w <= kRingWholeTileBits) ? 4u : kRingSpanBlocks;
            const uint32_t span_values = span_blocks * kBlockValues;
            for (uint32_t i = 0; i < kRingTileBlocks / span_blocks; ++i) {
                const uint32_t span_first = tile_first + i * span_values;
                if (span_first >= n) break;
                const uint32_t span_n = min(span_values, n - span_first);
                io.await_start(span_first, lane);
                if (io.aborted) return;
                const u32x4 r4 = *(const u32x4 *)&sh.rq[(t * kRingTileBlocks + i * span_blocks) & (kRq - 1u)];
                Resume rs;
                rs.run_index = __builtin_amdgcn_readfirstlane(r4.x);
                rs.rem_kind = __builtin_amdgcn_readfirstlane(r4.y);
                rs.payload = __builtin_amdgcn_readfirstlane(r4.z);
                rs.skip = __builtin_amdgcn_readfirstlane(r4.w);
                // the lowest byte the span (and everything behind it) can ask for: the group that holds a
                // bit-packed run's first value here, or the byte behind an RLE run
                io.raise_need((rs.rem_kind >> 31) == kBitpacked ? rs.payload + (rs.skip >> 3) * w : rs.skip, lane);
#define PQ2_W(N)                                                                                            \
    case N:                                                                                                 \
        ring_expand_span<N, kRingSpanBlocks, kGroup>(io, tbl, cidx, rs, end, n, span_first, span_n, sink, lane);    \
        break;
#define PQ2_W_BITS                                                                                               \
    case 1:                                                                                                          \
        if (span_blocks == 4u)                                                                                       \
            ring_expand_span_levels<kRingTileBlocks == 4 ? 4 : 2>(io, tbl, cidx, rs, end, n, span_first, span_n, sink, lane); \
        else                                                                                                         \
            ring_expand_span_levels<kRingSpanBlocks>(io, tbl, cidx, rs, end, n, span_first, span_n, sink, lane);      \
        break;
#define PQ2_W_LEVELS(N)                                                                                                     \
    case N:                                                                                                                 \
        if (span_blocks == 4u)                                                                                              \
            ring_expand_span<N, kRingTileBlocks == 4 ? 4 : 2, kGroup>(io, tbl, cidx, rs, end, n, span_first, span_n, sink, lane);   \
        else                                                                                                                \
            ring_expand_span<N, kRingSpanBlocks, kGroup>(io, tbl, cidx, rs, end, n, span_first, span_n, sink, lane);                \
        break;
                if constexpr (kWords < 0) {
                    static_assert(kRingTileBlocks == 4, "a span of the bitmap mode is a tile of four blocks");
                    uint32_t *const words = (uint32_t *)pg.out_addr + span_first / 32u;
                    if (w == 1u) {
                        ring_expand_span_bits(io, tbl, cidx, rs, end, n, span_first, span_n, words, set_bits, lane);
                    } else {  // (the launcher sends pages of one bit; a page without a width has no rows)
                        for (uint32_t q = lane; q < (span_n + 31u) / 32u; q += 64u) words[q] = 0u;
                    }
                } else if constexpr (kWords == 0) {
                    IndexSink sink = {out + i * span_values};
                    static_assert(kRingWholeTileBits <= 2, "the cases below");
                    switch (w) {
#ifdef PQ2_ONLY_W
                        PQ2_W(PQ2_ONLY_W)
#else
                        PQ2_W_BITS PQ2_W_LEVELS(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
                        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12) PQ2_W(13) PQ2_W(14) PQ2_W(15) PQ2_W(16)
#endif
                        default:
                            if constexpr (kWide) {
                                switch (w) {
#ifndef PQ2_ONLY_W
                                    PQ2_W(17) PQ2_W(18) PQ2_W(19) PQ2_W(20) PQ2_W(21) PQ2_W(22) PQ2_W(23) PQ2_W(24)
                                    PQ2_W(25) PQ2_W(26) PQ2_W(27) PQ2_W(28) PQ2_W(29) PQ2_W(30) PQ2_W(31) PQ2_W(32)
#endif
                                    default:  // width 0 (or beyond 32, whose page has no values): zeros
                                        ring_fill_span(sink, 0u, span_n, lane);
                                        break;
                                }
                            } else {
                                // width 0 (or beyond 32, whose page has no values): zeros; the launcher sends no
                                // page of 17 to 32 bits to this kernel
                                if (w > kRingNarrowBits && w <= 32u)
                                    io.abort_all(lane, 8);
                                else
                                    ring_fill_span(sink, 0u, span_n, lane);
                            }
                            break;
                    }
                } else if constexpr (kNullable) {
                    static_assert(kRingSpanBlocks * kBlockValues == kRowDirStep, "a span is a step of the row directory");
                    const RingNullable nl = nulls[page];
                    // The rows of the tile's spans: asked for at the tile's first span, so that only that one waits for
                    // the directory; the bitmap's words from the span's first row on are asked for now and are there by
                    // the time the span's indices are in LDS.
                    typedef __attribute__((address_space(1))) const uint32_t gu32;
                    if (i == 0) {
                        gu32 *const dir = (gu32 *)nl.rowdir_addr + tile_first / kRowDirStep;
                        dir_rows[0] = dir[0], dir_rows[1] = dir[1], dir_rows[2] = dir[2];
                    }
                    const bool last = span_first + span_n >= n;  // the rows behind the last index are the last span's too
                    // (clamped to the page: the bitmap of a page whose level stream failed holds whatever the memory held,
                    // and so does its directory -- such a page's rows are not defined, its neighbours' are)
                    const uint32_t page_rows = __builtin_amdgcn_readfirstlane(nl.rows);
                    const uint32_t row_hi = last ? page_rows : min(page_rows, __builtin_amdgcn_readfirstlane(i == 0 ? dir_rows[1] : dir_rows[2]));
                    const uint32_t row_lo = min(row_hi, __builtin_amdgcn_readfirstlane(i == 0 ? dir_rows[0] : dir_rows[1]));
                    const __amdgpu_buffer_rsrc_t bits = make_rsrc(nl.bitmap_addr, ((nl.rows + 31u) >> 5) * 4u);
                    uint32_t x0 = __builtin_amdgcn_raw_buffer_load_b32(bits, ((row_lo >> 5) + lane) * 4u, 0, 0);
                    uint32_t x1 = __builtin_amdgcn_raw_buffer_load_b32(bits, ((row_lo >> 5) + 64u + lane) * 4u, 0, 0);
                    uint8_t *const mine = lds + (flags >> kRingFlagStageShift) + e * kRingStageBytes;
                    uint16_t *const stage = (uint16_t *)mine;
                    uint32_t *const wbuf = (uint32_t *)(mine + kRowDirStep * 2u);
                    StageSink sink = {stage};
                    static_assert(kRingNullableMaxBits == 12, "the cases below");
                    switch (w) {
                        PQ2_W(1) PQ2_W(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
                        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12)
                        case 0:  // every index is 0
                            ring_fill_span(sink, 0u, span_n, lane);
                            break;
                        default:  // the host does not send wider pages here
                            io.abort_all(lane, 9);
                            break;
                    }
                    if (io.aborted) return;
                    wbuf[lane] = x0;
                    if (lane + 64u < kRingPlaceWords) wbuf[lane + 64u] = x1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    ring_place_span<kWords>(nl, stage, wbuf, true, dict_lds, dict_len, span_n, row_lo, row_hi, bad_index, lane);
                    __builtin_amdgcn_wave_barrier();  // (the next span's indices go where these lay)
                } else {
                    DictSink<kWords> sink = {(out_u32 *)value_addr + (size_t)span_first * kWords, dict_lds, dict_len, false};
                    static_assert(kRingDictMaxBits == 16, "the cases below");
                    switch (w) {
                        PQ2_W(1) PQ2_W(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
                        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12) PQ2_W(13) PQ2_W(14) PQ2_W(15) PQ2_W(16)
                        case 0:  // every index is 0
                            ring_fill_span(sink, 0u, span_n, lane);
                            break;
                        default:  // the host does not send wider pages here
                            io.abort_all(lane, 9);
                            break;
                    }
                    bad_index |= sink.bad;
                }
#undef PQ2_W
#undef PQ2_W_BITS
#undef PQ2_W_LEVELS
                if (io.aborted) return;
            }
        }
        // the next tile nobody holds
        uint32_t next = 0;
        if (lane == 0) next = atomicAdd(&sh.ctl->next_tile, 1u);
        t = __builtin_amdgcn_readfirstlane(next);
    }
    io.tile = 0xffffffffu;  // through: the walk no longer counts on this wave
    io.need = kNoNeed;
    io.say(lane);
    if (kWords > 0 && __ballot(bad_index) != 0 && lane == 0) atomicMax(&status[page], 1);  // PQ2_OUT_OF_SPEC
    if constexpr (kWords < 0) {  // this wave's share of the page's set bits
        const uint32_t total = wave_inclusive_sum(set_bits);
        if (lane == 63u && total != 0u) atomicAdd(&bit_meta[2u * page + 1u], total);
    }
    PQ2_CLOCK_ADD(8, clock_begin);
    PQ2_CLOCK_FLUSH(8);
    PQ2_TRACE_END(page);
}

}  // namespace pq2
