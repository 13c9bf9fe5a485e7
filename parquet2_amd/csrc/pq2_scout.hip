// scout kernels: the header walk of a page spread over the whole GPU (gfx950 / MI355X).
//
// Where a run header lies is known only once the header before it is decoded
// (src/encoding/hybrid_rle/decoder.rs:30-64), so one wave per page walks a chain (parse_kernel, decode_kernel,
// ring_kernel).  A batch of few pages leaves most of the GPU idle during that walk, and a page whose runs are
// short is nothing but walk.  Here the stream of every page is cut into tiles of 2 KiB (a-offsets
// [t * 2048, (t + 1) * 2048)) and the walk is done in steps that are parallel over all tiles of the batch:
//
//   table    for EVERY byte offset of a tile, as if a header began there: where the chain that starts there leaves
//            the tile.  Lane l works backwards through the 32 offsets of segment l (the bytes in registers, the
//            results in LDS); a pass over the segments from the last to the first then extends every entry to the
//            end of the tile.  Everything unusual (a varint of three or more bytes, a run that passes the end of
//            the stream or reaches further than 32 KiB from the tile) is an escape that names the header, which
//            is then decoded exactly by whoever gets there.  2 bytes per stream byte, written once (and 2 more for
//            the table before the pass over the segments, which the count step uses).
//            It also counts, per tile, at which offsets of the next tile most of its chains arrive (votes): the
//            four strongest are handed on as guesses of where the page's chain enters that tile.
//   resolve  one workgroup per page.  First the votes are carried forward from tile to tile through the tables
//            (the page's chain goes on and on, a chain that starts inside a payload soon ends: after a few tiles the
//            true entry has the most votes even where headers are hundreds of bytes apart).  Then every tile starts
//            with its strongest candidate, and each round looks up, for all tiles at once, where the chain from
//            that entry leaves the tile, marks the tiles that the chain from the page's first byte reaches (pointer
//            doubling over the tiles) and hands each reached tile's exit to the tile it lands in.  The page's first
//            tile is right from the start, so every round makes at least one more tile right; with good guesses
//            one to three rounds do.  The result is exact either way: it is the fixed point in which every
//            reached tile is entered where its predecessor leaves.
//   count    one wave per reached tile walks the tile's part of the chain (the chase loop of parse_kernel, out of
//            LDS), sums the values its runs hold and lists the runs that hold any (2 bytes per run, over the
//            tile's table, which nobody needs any more).
//   scan     one workgroup per page: prefix sums of the tiles' values and runs; the tile in which the page's last
//            value lies (or the chain ends) is the one that closes the page.
//   place    one wave per reached tile: lane r decodes listed run r, a prefix sum places it, and the wave writes
//            what parse_kernel writes -- the header-position list and the Resume entries -- for expand_kernel.
//
// Reference semantics are those of parse_page (pq2_kernels.hip): the same decode_run, the same clamps, the same
// statuses.
#include "pq2_device.h"

namespace pq2 {

constexpr uint32_t kScoutTile = 2048;                       // stream bytes per tile
constexpr uint32_t kScoutSeg = 32;                          // offsets per lane
constexpr uint32_t kScoutRow = 66;                          // table entries per row: 64 lanes + 2 (odd dword stride)
constexpr uint32_t kScoutTable = kScoutSeg * kScoutRow;     // entries (of 2 bytes) per tile
constexpr uint32_t kScoutEsc = 0x8000u;                     // entry: "decode the header at tile offset (entry & 0x7fff)"
constexpr uint32_t kScoutNone = 0xffffffffu;                // a tile the chain does not enter / a chain that has ended
constexpr uint32_t kScoutGaveUp = 0xfffffffeu;              // a chain whose exit was not worth finding out on a guess
constexpr uint32_t kScoutGuessEscapes = 3;                  // escapes the resolve step follows for an entry that is a guess
constexpr uint32_t kScoutLdsTiles = 2048;                   // pages of up to this many tiles are resolved in LDS
constexpr uint32_t kScoutWin = kScoutTile + 64u;            // bytes of a tile the chase may touch
constexpr uint32_t kScoutPanic = 0x80000000u, kScoutTerm = 0x40000000u, kScoutBad = 0x20000000u;
constexpr uint32_t kScoutRunsMask = 0x1fffffffu;
constexpr uint32_t kValueCap = 0x7fffffffu;
#ifndef PQ2_SCOUT_VOTERS
#define PQ2_SCOUT_VOTERS 64
#endif
constexpr uint32_t kScoutVoters = PQ2_SCOUT_VOTERS;         // the offsets of the tile's first so many segments vote
#ifndef PQ2_SCOUT_DENSE
#define PQ2_SCOUT_DENSE 40
#endif
constexpr uint32_t kScoutDense = PQ2_SCOUT_DENSE;           // headers per tile from which the count step works lane-parallel
constexpr uint32_t kScoutKeep = 4;                          // exits of a tile the table step hands on, most voted first
constexpr uint32_t kScoutResolveThreads = 1024;             // threads of the resolve step's one workgroup per page
constexpr uint32_t kScoutScanThreads = 1024;                // threads of the scan step's one workgroup per page
constexpr uint32_t kScoutResolveArrays = 12;                // arrays of kScoutLdsTiles + 8 words it keeps in LDS
constexpr uint32_t kScoutSure = 96;                         // votes (of 2048) from which the first exit is handed on alone
#ifndef PQ2_SCOUT_PUSHES
#define PQ2_SCOUT_PUSHES 12
#endif
constexpr uint32_t kScoutPushes = PQ2_SCOUT_PUSHES;         // tiles a vote is carried forward at most (resolve step)

struct ScoutPage {
    uint32_t tile_base;  // batch-wide number of the page's first stream tile
    uint32_t n_tiles;    // stream tiles of the page (0: nothing to walk)
};
struct ScoutFin {
    uint32_t tile;   // the page's tile that closes the page
    uint32_t flags;  // kScoutPanic | kScoutBad
};
// per stream tile (arrays of n_stiles words each, in this order, in one allocation)
struct ScoutTiles {
    uint32_t *page, *entry, *exit, *vals, *runs, *base_vals, *base_runs;
};
__device__ __forceinline__ ScoutTiles scout_tiles(uint32_t *base, uint32_t n_stiles) {
    ScoutTiles t;
    t.page = base;
    t.entry = base + (size_t)n_stiles;
    t.exit = base + 2 * (size_t)n_stiles;
    t.vals = base + 3 * (size_t)n_stiles;
    t.runs = base + 4 * (size_t)n_stiles;
    t.base_vals = base + 5 * (size_t)n_stiles;
    t.base_runs = base + 6 * (size_t)n_stiles;
    return t;
}
__device__ __forceinline__ uint32_t table_slot(uint32_t off) { return (off & 31u) * kScoutRow + (off >> 5); }
// probe_kernel (pq2_kernels.hip) has walked the whole page already: nothing is left for the scout kernels.  (Only
// asked about pages with stream tiles: those have values, and the probe has written their state.)
__device__ __forceinline__ bool scout_page_done(const ParseState *__restrict__ state, uint32_t page) {
    return __builtin_amdgcn_readfirstlane(state[page].done) != 0u;
}

// ---------------------------------------------------------------------------
// table
// ---------------------------------------------------------------------------
// The bytes of a lane's segment and the first four behind it (what a header in the segment's last byte may need).
struct SegBytes {
    uint32_t d[9];
};
__device__ __forceinline__ SegBytes scout_load_segment(const __amdgpu_buffer_rsrc_t rsrc, const uint32_t seg_a) {
    const u32x4 lo4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, seg_a, 0, 0);
    const u32x4 hi4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, seg_a + 16u, 0, 0);
    const uint32_t d8 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, seg_a + 32u, 0, 0);
    const SegBytes sb = {{lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w, d8}};
    return sb;
}
// The header that would begin at byte `i` (a constant) of the segment, read as one of one or two bytes
// (uleb128.rs:3-23): its bytes, the bytes of its run (decoder.rs:48-63) and the values the run holds (mod.rs:41-64).
struct SegHeader {
    uint32_t c, body, count;
    bool longer;  // the varint has three bytes or more: not this one
};
__device__ __forceinline__ SegHeader scout_header(const SegBytes &sb, const int i, const uint32_t w, const uint32_t rb) {
    const uint32_t b0 = (sb.d[i >> 2] >> ((i & 3) * 8)) & 0xffu;
    const uint32_t b1 = (sb.d[(i + 1) >> 2] >> (((i + 1) & 3) * 8)) & 0xffu;
    const bool two = (b0 & 0x80u) != 0u;
    const uint32_t ind = two ? ((b0 & 0x7fu) | (b1 << 7)) : b0;
    SegHeader h;
    h.c = two ? 2u : 1u;
    h.body = (ind & 1u) ? __umul24(ind >> 1, w) : rb;  // 14 bits by 6: the 24-bit multiply runs at full rate
    h.count = (ind & 1u) ? (ind >> 1) * 8u : (ind >> 1);
    h.longer = two && (b1 & 0x80u) != 0u;
    return h;
}

// Lane l: for every offset i of segment l of the tile at a-offset t0, where the chain of headers that starts at
// i leaves the segment -- a tile offset >= 32 (l + 1), or an escape -- into tbl[i * kScoutRow + l].
// kNearEnd: the stream may end within kScoutEsc bytes of the tile's start (elsewhere a run that lies within that
// reach, which is asked of it anyway, lies inside the stream, and the two comparisons with `end` are left out).
template <bool kNearEnd>
__device__ __forceinline__ void scout_segments(const SegBytes &sb, const uint32_t t0, const uint32_t end, const uint32_t w,
                                               uint16_t *tbl, const uint32_t lane) {
    const uint32_t rb = ceil8(w);
    const uint32_t seg_rel = lane * kScoutSeg;
    const uint32_t seg_a = t0 + seg_rel;
#pragma unroll
    for (int i = (int)kScoutSeg - 1; i >= 0; --i) {
        const SegHeader h = scout_header(sb, i, w, rb);
        const uint32_t q = seg_a + (uint32_t)i + h.c;      // the byte behind the header
        const uint32_t nxt = (uint32_t)i + h.c + h.body;   // the next header, from the segment's start
        const uint32_t far = seg_rel + nxt;                // ... from the tile's start
        // the short path of decode_run: a header of one or two bytes whose run lies inside the stream
        bool plain = !h.longer && far < kScoutEsc;
        if (kNearEnd) plain = plain && q < end && h.body <= end - q;
        const uint32_t inside = tbl[(nxt & 31u) * kScoutRow + lane];  // where the chain from nxt leaves (if nxt < 32)
        uint32_t v = nxt < kScoutSeg ? inside : far;
        if (!plain) v = kScoutEsc | (seg_rel + (uint32_t)i);
        tbl[(uint32_t)i * kScoutRow + lane] = (uint16_t)v;
    }
}

// One wave per tile, four tiles per workgroup.
__global__ __launch_bounds__(256) void scout_table_kernel(const PageDev *__restrict__ pages,
                                                          const ScoutPage *__restrict__ sp, uint32_t n_pages,
                                                          uint32_t *__restrict__ tiles, uint16_t *__restrict__ table,
                                                          uint32_t n_stiles, const ParseState *__restrict__ state,
                                                          uint32_t *__restrict__ hints) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4][kScoutTable];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x * 4u + wave;
    if (g >= n_stiles) return;
    const ScoutTiles t = scout_tiles(tiles, n_stiles);
    // the page this tile belongs to: the last one whose first tile is not behind g (a page without tiles has the
    // first tile of the page behind it, and comes before it); noted for the steps that follow
    // (up to 128 pages: every lane looks at two, one trip to memory; beyond: a binary search, one trip per step)
    uint32_t page = 0, page_base = 0;
    if (n_pages <= 128u) {
        uint32_t b0 = lane < n_pages ? sp[lane].tile_base : 0xffffffffu;
        uint32_t b1 = lane + 64u < n_pages ? sp[lane + 64u].tile_base : 0xffffffffu;
        page = (uint32_t)__popcll(__ballot(b0 <= g)) + (uint32_t)__popcll(__ballot(b1 <= g)) - 1u;  // page 0's is 0
        page_base = page < 64u ? __builtin_amdgcn_readlane(b0, page) : __builtin_amdgcn_readlane(b1, page - 64u);
    } else {
        for (uint32_t step = 1u << (31u - (uint32_t)__builtin_clz(n_pages)); step != 0u; step >>= 1) {
            const uint32_t p = page + step;
            if (p < n_pages && __builtin_amdgcn_readfirstlane(sp[p].tile_base) <= g) page = p;
        }
        page_base = __builtin_amdgcn_readfirstlane(sp[page].tile_base);
    }
    if (lane == 0) t.page[g] = page;
    const PageDev pg = pages[page];  // (asked for together with the page's state)
    if (scout_page_done(state, page)) return;
    const uint32_t t0 = (g - page_base) * kScoutTile;
    uint16_t *const tbl = lds[wave];
    {
        const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
        const uint32_t end = delta + pg.in_len;
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);
        const SegBytes sb = scout_load_segment(rsrc, t0 + lane * kScoutSeg);
        if (t0 + kScoutEsc + 8u <= end)
            scout_segments<false>(sb, t0, end, pg.num_bits, tbl, lane);
        else
            scout_segments<true>(sb, t0, end, pg.num_bits, tbl, lane);
    }
    wave_sync();
    // the table as it is now -- where every chain leaves its SEGMENT -- is what the count step walks a tile of many
    // headers with: a copy goes behind the tables proper (tile g's at table + (n_stiles + g) * kScoutTable)
    {
        u32x4 *const dst = (u32x4 *)(table + ((size_t)n_stiles + g) * kScoutTable);
        const u32x4 *const src = (const u32x4 *)tbl;
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            const uint32_t at = i * 64u + lane;
            if (at < kScoutTable * 2u / 16u) dst[at] = src[at];
        }
    }
    // An entry that leaves its segment inside the tile takes over the entry of the offset it lands on, once that
    // one is final.  In two levels, so that all lanes work: first inside every group of 8 segments, from each group's
    // last segment but one to its first (all 8 groups at a time: 256 entries a step), then group by group from the
    // last but one to the first (the 256 entries of a group a step).
    static_assert(kScoutTile == 8u * 8u * kScoutSeg && kScoutSeg == 32u, "8 groups of 8 segments of 32 offsets");
    for (int j = 6; j >= 0; --j) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t group = (lane >> 5) + 2u * k;
            const uint32_t at = (lane & 31u) * kScoutRow + 8u * group + (uint32_t)j;
            const uint32_t v = tbl[at];
            if (v < 256u * (group + 1u)) tbl[at] = tbl[table_slot(v)];  // lands in a later segment of its group
        }
        wave_sync();
    }
    for (int group = 6; group >= 0; --group) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t e = lane + 64u * k;  // entry e of the group: offset e & 31 of its segment e >> 5
            const uint32_t at = (e & 31u) * kScoutRow + 8u * (uint32_t)group + (e >> 5);
            const uint32_t v = tbl[at];
            if (v < kScoutTile) tbl[at] = tbl[table_slot(v)];  // lands in a later group
        }
        wave_sync();
    }
    // the table as it lies in LDS: kScoutTable * 2 = 4224 bytes = 264 pieces of 16
    {
        u32x4 *const dst = (u32x4 *)(table + (size_t)g * kScoutTable);
        const u32x4 *const src = (const u32x4 *)tbl;
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            const uint32_t at = i * 64u + lane;
            if (at < kScoutTable * 2u / 16u) dst[at] = src[at];
        }
    }
    // Guesses at where the page's chain enters the next tile: the offsets there at which most of this tile's chains
    // arrive.  Every header of the page's chain inside this tile is an offset that votes for the true one, while
    // the chains that start inside bit-packed payloads scatter -- but they gather too (a few dozen offsets of random
    // bytes may lead to one place), so where headers are sparse the true exit is often not the first but among the
    // first few: kScoutKeep of them are handed on with their votes.  (Only guesses: the resolve step decides.)
    // The votes are counted where the table lay: one 16-bit counter for each offset of the next tile.
    {
        static_assert(kScoutTable * 2u >= kScoutTile * 2u && kScoutSeg == 32u, "the counters fit the table's LDS");
        uint32_t ent[kScoutSeg / 2u];  // this lane's 32 entries (the offsets of segment `lane`), two to a register
#pragma unroll
        for (uint32_t i = 0; i < kScoutSeg / 2u; ++i)
            ent[i] = (uint32_t)tbl[2u * i * kScoutRow + lane] | ((uint32_t)tbl[(2u * i + 1u) * kScoutRow + lane] << 16);
        wave_sync();
        uint32_t *const vote = (uint32_t *)tbl;  // counter of offset x: half x & 1 of word x >> 1
#pragma unroll
        for (uint32_t i = 0; i < kScoutTile / 2u / 4u / 64u; ++i) *(u32x4 *)&vote[4u * (lane + 64u * i)] = u32x4{0u, 0u, 0u, 0u};
        wave_sync();
        // (the offsets of a segment that leave the tile at the same place -- in a stretch of short runs, all of them --
        // vote together)
        auto cast = [&](uint32_t v, uint32_t times) {
            const uint32_t rel = v - kScoutTile;
            if (rel < kScoutTile && lane < kScoutVoters) atomicAdd(&vote[rel >> 1], times << ((rel & 1u) * 16u));
        };
        uint32_t same = ent[0] & 0xffffu, times = 1u;
#pragma unroll
        for (uint32_t i = 1; i < kScoutSeg; ++i) {
            const uint32_t v = (ent[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
            if (v != same) {
                cast(same, times);
                same = v, times = 0u;
            }
            ++times;
        }
        cast(same, times);
        wave_sync();
        // The strongest: every lane finds the two strongest of its 32 counters (those of words lane, lane + 64, ...:
        // votes << 11 | offset), then the wave takes the strongest of the lanes' first four times over, the lane it
        // came from moving its second up.  (A lane with three of the four strongest loses one: these are guesses.)
        uint32_t m1 = 0, m2 = 0;
#pragma unroll
        for (uint32_t i = 0; i < kScoutTile / 2u / 64u; ++i) {
            const uint32_t at = lane + 64u * i;
            const uint32_t word = vote[at];
            const uint32_t k0 = ((word & 0xffffu) << 11) | (2u * at), k1 = ((word >> 16) << 11) | (2u * at + 1u);
            m2 = max(m2, min(m1, k0)), m1 = max(m1, k0);
            m2 = max(m2, min(m1, k1)), m1 = max(m1, k1);
        }
        uint32_t top[kScoutKeep];
#pragma unroll
        for (uint32_t r = 0; r < kScoutKeep; ++r) {
            top[r] = 0u;
            // (where headers are dense nearly all chains of the tile leave it together: no need for a second guess)
            if (r != 0u && (top[0] >> 11) >= kScoutSure) continue;
            top[r] = __builtin_amdgcn_readlane(wave_inclusive_max(m1), 63);
            if (m1 == top[r]) m1 = m2, m2 = 0u;
        }
        // candidates for the entry of the next tile: (offset in that tile) << 16 | votes, two votes at least
        if (lane < kScoutKeep) {
            const uint32_t mine = lane == 0u ? top[0] : lane == 1u ? top[1] : lane == 2u ? top[2] : top[3];
            static_assert(kScoutKeep == 4u, "lanes 0..3 write the candidates");
            hints[(size_t)g * kScoutKeep + lane] = (mine >> 11) >= 2u ? ((mine & (kScoutTile - 1u)) << 16) | (mine >> 11) : 0u;
        }
        if (lane == 0) {
            const uint32_t votes0 = top[0] >> 11;
            t.exit[g] = votes0 >= 2u ? t0 + kScoutTile + (top[0] & (kScoutTile - 1u)) : kScoutNone;
            t.vals[g] = votes0;  // about as many headers as the page's chain has in this tile (count step)
        }
    }
}

// ---------------------------------------------------------------------------
// resolve
// ---------------------------------------------------------------------------
// Where the chain that enters a tile at a-offset x leaves that tile: the a-offset of the first header in a later
// tile, or kScoutNone if the chain ends first (end of the stream, or a header on which the reference panics).
// Every escape costs two trips to memory, and a chain that starts at a wrong guess inside a bit-packed payload
// may go through many: after `escapes` of them the answer is kScoutGaveUp.
__device__ __forceinline__ uint32_t scout_leave(const PageDev &pg, const __amdgpu_buffer_rsrc_t rsrc, const uint32_t end,
                                                const uint16_t *__restrict__ page_table, uint32_t x, uint32_t escapes) {
    const uint32_t tile = x / kScoutTile;
    auto rd = [&](uint32_t a) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, a, 0, 0); };
    for (;;) {
        const uint32_t v = page_table[(size_t)tile * kScoutTable + table_slot(x % kScoutTile)];
        if ((v & kScoutEsc) == 0u) return tile * kScoutTile + v;
        if (escapes-- == 0u) return kScoutGaveUp;
        const RunInfo run = decode_run(rd, tile * kScoutTile + (v & 0x7fffu), end, pg.num_bits, 0xffffffffu);
        if (run.kind >= kEnd) return kScoutNone;
        if (run.next / kScoutTile != tile) return run.next;
        x = run.next;  // behind the header it came from: the loop ends
    }
}

// One workgroup per page.  Dynamic LDS: kScoutResolveArrays arrays of kScoutLdsTiles + 8 words (six for the rounds,
// twelve while votes are carried forward); pages with more tiles use `scratch` (as many arrays of n_tiles + 8 words,
// from word kScoutResolveArrays * (tile_base + 8 * page) on).
__global__ __launch_bounds__(kScoutResolveThreads) void scout_resolve_kernel(const PageDev *__restrict__ pages,
                                                            const ScoutPage *__restrict__ sp, uint32_t n_pages,
                                                            const uint16_t *__restrict__ table,
                                                            uint32_t *__restrict__ tiles, uint32_t n_stiles,
                                                            uint32_t *__restrict__ scratch,
                                                            uint32_t *__restrict__ rounds_out,
                                                            const ParseState *__restrict__ state,
                                                            const uint32_t *__restrict__ hints) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];
    __shared__ uint32_t changed;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const ScoutPage s = sp[page];
    const uint32_t n = s.n_tiles;
    if (n == 0u || scout_page_done(state, page)) {
        if (threadIdx.x == 0) rounds_out[page] = 0u;
        return;
    }
    const PageDev pg = pages[page];
    const ScoutTiles t = scout_tiles(tiles, n_stiles);
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);
    const uint16_t *const page_table = table + (size_t)s.tile_base * kScoutTable;

    const uint32_t stride = (n <= kScoutLdsTiles ? kScoutLdsTiles : n) + 8u;
    uint32_t *const arrays = n <= kScoutLdsTiles ? lds_words : scratch + kScoutResolveArrays * ((size_t)s.tile_base + 8u * page);
    uint32_t *const cand = arrays;                 // where tile k is entered, as far as is known
    uint32_t *const leave = arrays + stride;       // where the chain from cand[k] leaves tile k
    uint32_t *const flag = arrays + 2 * stride;    // bit 0: leave[k] is out of date; bit 1: tile k is reached;
    //                                                bit 2: tile k was reached once: its chain is followed to the end
    uint32_t *jump[2] = {arrays + 3 * stride, arrays + 4 * stride};  // the tile 2^r hops behind tile k (n: none)
    uint32_t *const offer = arrays + 5 * stride;   // who hands tile k its entry: reached << 31 | (tile + 1)
    const uint32_t tid = threadIdx.x;
    auto tile_of = [&](uint32_t x) { return x >= kScoutGaveUp ? n : min(x / kScoutTile, n); };

    // Votes carried forward.  The page's chain goes on from tile to tile while a chain that starts in a payload soon
    // ends in a header nobody follows, so of the exits that the chains of tile k - 1 vote for, the true one is the
    // one whose voters come from far back.  best[i][k]: candidates for the entry of tile k, (offset in the tile)
    // << 16 | votes; every pass adds to the votes of tile k - 1's own chains those of best[.][k - 1], carried through
    // tile k - 1 by its table, and keeps the kScoutKeep strongest: after d passes the votes of d + 1 tiles count.
    // It ends when no tile's strongest candidate changes.  (In LDS, over arrays that are not in use yet.)
    const bool carry_votes = kScoutPushes != 0u;
    if (carry_votes) {
        uint32_t *const best = arrays;                  // best[i * stride + k]
        uint32_t *const carried = arrays + 4 * stride;  // ... what arrives in tile k from the candidates of tile k - 1
        uint32_t *const own = arrays + 8 * stride;      // ... what the table step handed on
        static_assert(kScoutKeep == 4u, "candidates are loaded four at a time, and the arrays are laid out for four");
        if (tid == 0) changed = 0u;
        __syncthreads();
        for (uint32_t k = tid; k < n; k += kScoutResolveThreads) {
            u32x4 h = {(delta << 16) | 0xffffu, 0u, 0u, 0u};  // the page's first byte: known
            if (k != 0u) h = *(const u32x4 *)(hints + ((size_t)s.tile_base + k - 1u) * kScoutKeep);
            own[k] = h.x, own[stride + k] = h.y, own[2 * stride + k] = h.z, own[3 * stride + k] = h.w;
            best[k] = h.x, best[stride + k] = h.y, best[2 * stride + k] = h.z, best[3 * stride + k] = h.w;
            if (h.y != 0u) changed = 1u;  // a tile with a second candidate: there is something to decide
        }
        __syncthreads();
        // (a page of dense headers hands on one candidate per tile: no pass can change anything)
        const uint32_t passes = changed != 0u ? kScoutPushes : 0u;
        for (uint32_t pass = 0; pass < passes; ++pass) {
            __syncthreads();
            if (tid == 0) changed = 0u;
#pragma unroll 2
            for (uint32_t k = tid + 1u; k < n; k += kScoutResolveThreads) {
                uint32_t c[kScoutKeep], v[kScoutKeep];
#pragma unroll
                for (uint32_t i = 0; i < kScoutKeep; ++i) {
                    c[i] = best[i * stride + k - 1u];
                    v[i] = page_table[(size_t)(k - 1u) * kScoutTable + table_slot((c[i] >> 16) & (kScoutTile - 1u))];
                }
#pragma unroll
                for (uint32_t i = 0; i < kScoutKeep; ++i) {
                    const bool lands = (c[i] & 0xffffu) != 0u && v[i] >= kScoutTile && v[i] < 2u * kScoutTile;
                    carried[i * stride + k] = lands ? ((v[i] - kScoutTile) << 16) | (c[i] & 0xffffu) : 0u;
                }
            }
            __syncthreads();
#pragma unroll 1
            for (uint32_t k = tid + 1u; k < n; k += kScoutResolveThreads) {
                uint32_t rel[2u * kScoutKeep], cnt[2u * kScoutKeep];
#pragma unroll
                for (uint32_t i = 0; i < kScoutKeep; ++i) {
                    const uint32_t o = own[i * stride + k], a = carried[i * stride + k];
                    rel[i] = o >> 16, cnt[i] = o & 0xffffu;
                    rel[kScoutKeep + i] = a >> 16, cnt[kScoutKeep + i] = a & 0xffffu;
                }
                // votes for one offset count together (the table step's candidates are different offsets already)
#pragma unroll
                for (uint32_t i = kScoutKeep; i < 2u * kScoutKeep; ++i) {
#pragma unroll
                    for (uint32_t j = 0; j < i; ++j) {
                        if (cnt[i] != 0u && cnt[j] != 0u && rel[i] == rel[j]) {
                            cnt[j] = min(cnt[j] + cnt[i], 0xffffu);
                            cnt[i] = 0u;
                        }
                    }
                }
                uint32_t key[2u * kScoutKeep];  // votes << 16 | offset
#pragma unroll
                for (uint32_t i = 0; i < 2u * kScoutKeep; ++i) key[i] = cnt[i] != 0u ? (cnt[i] << 16) | rel[i] : 0u;
                const uint32_t was = best[k];
#pragma unroll
                for (uint32_t r = 0; r < kScoutKeep; ++r) {
                    uint32_t most = 0;
#pragma unroll
                    for (uint32_t i = 0; i < 2u * kScoutKeep; ++i) most = max(most, key[i]);
#pragma unroll
                    for (uint32_t i = 0; i < 2u * kScoutKeep; ++i) key[i] = key[i] == most ? 0u : key[i];
                    best[r * stride + k] = (most << 16) | (most >> 16);
                    if (r == 0u && (was >> 16) != (most & 0xffffu)) changed = 1u;
                }
            }
            __syncthreads();
            if (changed == 0u) break;
        }
        // (the strongest candidates move to where the rounds below do not write before they are read)
        for (uint32_t k = tid; k < n; k += kScoutResolveThreads) own[k] = best[k];
        __syncthreads();
    }

    // first guesses: the page's first byte; elsewhere the candidate with the most votes (without the passes above:
    // what the table step saw most chains of the tile before arrive at), or the tile's first byte
    for (uint32_t k = tid; k < n; k += kScoutResolveThreads) {
        uint32_t x = k == 0u ? delta : k * kScoutTile;
        bool guess = k != 0u;
        if (k != 0u) {
            if (carry_votes) {
                const uint32_t h = arrays[8 * stride + k];
                if ((h & 0xffffu) >= 2u) x = k * kScoutTile + (h >> 16), guess = false;
            } else {
                const uint32_t hint = t.exit[s.tile_base + k - 1u];
                if (hint != kScoutNone && hint / kScoutTile == k) x = hint, guess = false;
            }
        }
        cand[k] = x;
        // (the first byte of a tile is rarely a header: such a guess is not followed, the tile waits for an offer)
#ifdef PQ2_SCOUT_EVAL_GUESSES
        guess = false;
#endif
        leave[k] = kScoutGaveUp;
        flag[k] = guess ? 0u : 1u;
    }
    uint32_t rounds = 0;
    while ((1u << rounds) < n) ++rounds;
    uint32_t iter_count = 0;
    for (uint32_t iter = 0; iter <= 2u * n; ++iter) {
        ++iter_count;
        __syncthreads();
        if (tid == 0) changed = 0u;
        for (uint32_t k = tid; k < n; k += kScoutResolveThreads) {
            const uint32_t f = flag[k];
            if (f & 1u)
                leave[k] = scout_leave(pg, rsrc, end, page_table, cand[k], (k == 0u || (f & 4u)) ? 0xffffffffu : kScoutGuessEscapes);
            flag[k] = (k == 0u ? 2u : 0u) | (f & 4u);
            jump[0][k] = tile_of(leave[k]);
            offer[k] = 0u;
        }
        if (tid == 0) jump[0][n] = n, flag[n] = 0u;
        __syncthreads();
        // the tiles on the chain from tile 0: after round r those within 2^(r + 1) - 1 hops are marked
        for (uint32_t r = 0; r <= rounds; ++r) {
            const uint32_t *const from = jump[r & 1u];
            uint32_t *const to = jump[(r & 1u) ^ 1u];
            for (uint32_t k = tid; k <= n; k += kScoutResolveThreads) {
                const uint32_t j = from[k];
                if (flag[k] & 2u) flag[j] |= 2u;  // (tile n stands for "nowhere")
                to[k] = from[j];
            }
            __syncthreads();
        }
        // Every tile hands its exit to the tile it lands in.  A reached tile's offer counts before any other (the
        // reached tiles are one chain, so at most one of them lands in a tile); among the others, which only
        // speed things up where their own guess is right, the nearest one.
        for (uint32_t k = tid; k < n; k += kScoutResolveThreads) {
            const uint32_t m = tile_of(leave[k]);
            if (m < n) atomicMax(&offer[m], ((flag[k] & 2u) << 30) | (k + 1u));
            if ((flag[k] & 2u) && leave[k] == kScoutGaveUp) {  // reached after all: the whole chain next time
                flag[k] |= 5u;
                changed = 1u;
            }
        }
        __syncthreads();
        for (uint32_t m = tid; m < n; m += kScoutResolveThreads) {
            const uint32_t o = offer[m];
            if (o == 0u) continue;
            const uint32_t x = leave[(o & 0x7fffffffu) - 1u];
            if (cand[m] != x) {
                cand[m] = x;
                flag[m] |= 1u;
                if (o >> 31) changed = 1u;
            }
        }
        __syncthreads();
        if (changed == 0u) break;  // every reached tile is entered where its predecessor leaves
    }
    if (tid == 0) rounds_out[page] = iter_count;
    for (uint32_t k = tid; k < n; k += kScoutResolveThreads) {
        const bool reached = (flag[k] & 2u) != 0u;
        t.entry[s.tile_base + k] = reached ? cand[k] : kScoutNone;
        t.exit[s.tile_base + k] = leave[k];
    }
}

// ---------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------
// One wave per tile, four tiles per workgroup: the walk of parse_page (pq2_kernels.hip) from the tile's entry to
// the first header behind the tile.
template <bool kWide>
__global__ __launch_bounds__(256) void scout_count_kernel(const PageDev *__restrict__ pages,
                                                          const ScoutPage *__restrict__ sp, uint32_t *__restrict__ tiles,
                                                          uint16_t *__restrict__ table, uint32_t n_stiles,
                                                          const ParseState *__restrict__ state) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][kScoutWin];
    __shared__ __attribute__((aligned(16))) uint16_t seg_lds[4][kScoutTable];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x * 4u + wave;
    if (g >= n_stiles) return;
    const ScoutTiles tl = scout_tiles(tiles, n_stiles);
    // what is known about the tile, asked for in one go, then what is known about its page
    uint32_t page_v = tl.page[g], pos_v = tl.entry[g], votes_v = tl.vals[g];
    pin(page_v), pin(pos_v), pin(votes_v);
    const uint32_t page = __builtin_amdgcn_readfirstlane(page_v);
    const PageDev pg = pages[page];
    uint32_t base_v = sp[page].tile_base;
    const bool done = scout_page_done(state, page);
    pin(base_v);
    if (done) return;
    uint32_t pos = __builtin_amdgcn_readfirstlane(pos_v);
    if (pos == kScoutNone) {
        if (lane == 0) tl.vals[g] = 0u, tl.runs[g] = 0u;
        return;
    }
    const uint32_t t0 = (g - __builtin_amdgcn_readfirstlane(base_v)) * kScoutTile;
    const uint32_t tile_end = t0 + kScoutTile;
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    const uint32_t w = pg.num_bits;
    const uint32_t rb = ceil8(w);
    const uint32_t rb_mask = rb < 4 ? (1u << (rb * 8u)) - 1u : ~0u;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);
    uint16_t *const list = table + (size_t)g * kScoutTable;  // the tile's table is not needed any more

    uint32_t vals = 0, listed = 0, term = 0;
    // ---- a tile with many headers: all lanes at once --------------------------------------------------------
    // The table step has kept where the chains inside every segment lead; one walk from segment to
    // segment (at most 64 steps instead of one per header) then tells every segment where the page's chain
    // enters it, and each lane follows the chain through its own 32 bytes.  At the first header the table step
    // would leave to the exact decoder (the end of the stream is one) the walk below takes over.
    // (The table step's count of the chains that leave the tile together says how many headers to expect; it sees
    // none in the tile that holds the end of the stream.)
    // the tile's bytes and the 64 behind them, for both ways of walking it
    uint8_t *const win = lds[wave];
    {
        static_assert(kScoutWin == 2u * 1024u + 64u, "the window is loaded in two rounds and four pieces");
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, t0 + lane * 16u, 0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, t0 + 1024u + lane * 16u, 0, 0);
        u32x4 c = {0u, 0u, 0u, 0u};
        if (lane < 4u) c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, t0 + 2048u + lane * 16u, 0, 0);
        *(u32x4 *)&win[lane * 16u] = a;
        *(u32x4 *)&win[1024u + lane * 16u] = b;
        if (lane < 4u) *(u32x4 *)&win[2048u + lane * 16u] = c;
    }
    if (__builtin_amdgcn_readfirstlane(votes_v) >= kScoutDense || tile_end >= end) {
        uint16_t *const tbl = seg_lds[wave];
        {  // the per-segment table, as the table step left it
            const u32x4 *const src = (const u32x4 *)(table + ((size_t)n_stiles + g) * kScoutTable);
            u32x4 *const dst = (u32x4 *)tbl;
            u32x4 piece[5];
#pragma unroll
            for (uint32_t i = 0; i < 5; ++i) {
                const uint32_t at = i * 64u + lane;
                if (at < kScoutTable * 2u / 16u) piece[i] = src[at];
            }
#pragma unroll
            for (uint32_t i = 0; i < 5; ++i) {
                const uint32_t at = i * 64u + lane;
                if (at < kScoutTable * 2u / 16u) dst[at] = piece[i];
            }
        }
        wave_sync();
        uint32_t x = pos - t0, mine = kScoutSeg;  // mine: the offset at which the chain enters this lane's segment
        uint32_t stop = kScoutSeg;                // ... and the offset of its segment from which the lane keeps out
        // (one exit from the loop, one comparison a step -- an entry that leaves the tile and an escape are both at
        // least kScoutTile --: the scalar unit, which all waves of a compute unit share, sets this kernel's pace)
        static_assert(kScoutEsc >= kScoutTile, "an escape compares as an exit");
        if (x < kScoutTile) {
            for (;;) {
                const uint32_t seg = x >> 5, off = x & 31u;
                mine = lane == seg ? off : mine;
                x = __builtin_amdgcn_readfirstlane((uint32_t)tbl[off * kScoutRow + seg]);
                if (x >= kScoutTile) break;
            }
            if (x & kScoutEsc) {
                x &= 0x7fffu;  // in the same segment, at or behind the entry
                stop = lane == (x >> 5) ? (x & 31u) : stop;
            }
        }
        // from header to header through the lane's 32 bytes (all of them headers of one or two bytes whose runs lie
        // inside the stream: the table step has seen to that up to `stop`), as many steps as the segment with the
        // most headers has
        uint32_t at = mine, held = 0, sum = 0;  // held: bit i = a run that holds values begins at byte i
        const uint32_t *const win32 = (const uint32_t *)win;
        while (__ballot(at < stop) != 0) {
            if (at < stop) {
                const uint32_t a = lane * kScoutSeg + at;
                const uint32_t x0 = __builtin_amdgcn_alignbit(win32[(a >> 2) + 1u], win32[a >> 2], (a & 3u) * 8u);
                const bool two = (x0 & 0x80u) != 0u;
                const uint32_t ind = two ? (x0 & 0x7fu) | ((x0 >> 1) & 0x3f80u) : x0 & 0x7fu;
                const uint32_t count = (ind & 1u) ? (ind >> 1) * 8u : ind >> 1;
                held |= count != 0u ? 1u << at : 0u;
                sum += count;  // < 32 * 2^16
                at += (two ? 2u : 1u) + ((ind & 1u) ? __umul24(ind >> 1, w) : rb);
            }
        }
        const uint32_t n_held = (uint32_t)__popc(held);
        const uint32_t upto = wave_inclusive_sum(n_held);
        uint32_t slot = upto - n_held;
        for (uint32_t m = held; m != 0u; m &= m - 1u) list[slot++] = (uint16_t)(lane * kScoutSeg + (uint32_t)__builtin_ctz(m));
        vals = min(__builtin_amdgcn_readlane(wave_inclusive_sum(sum), 63), kValueCap);
        listed = __builtin_amdgcn_readlane(upto, 63);
        pos = __builtin_amdgcn_readfirstlane(t0 + x);  // behind the tile, or the header that is left to the walk
    }

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)win;
    wave_sync();
    auto rd = [&](uint32_t a) -> uint32_t { return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&win[a - t0]); };

    while (pos < tile_end && term == 0u) {
        // ---- chase: park up to 64 runs, record i in lane i (see PQ2_CHASE) ---------------------------------
        uint32_t r_pos = 0, r_lo = 0, r_hi = 0, r_info = 0;
        uint32_t m = 0;
        for (;;) {
            uint32_t lo, hi, groups, t, bytes, nm1, chk, addr;
            uint64_t exec_all;
            uint32_t m0_kept;
            const uint32_t voff = lds_base - t0;
            // goes on while the next header lies inside the tile and inside the stream
            const uint32_t chase_limit = min(end - 1u, tile_end - 2u);
            if constexpr (kWide) {
                PQ2_CHASE("v_readfirstlane_b32 %[hi], v3\n\t", "v_writelane_b32 %[r_hi], %[hi], m0\n\t")
            } else {
                PQ2_CHASE("", "")
            }
            if ((~lo & kExplicit) == 0u || nm1 > end - 1u) {
                // three or more header bytes, a run cut by the end of the stream, or the end itself: the record
                // parked last does not count
                --m;
                pos = __builtin_amdgcn_readlane(r_pos, m);
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
            if (m == 64u || pos >= tile_end) break;
        }
        // ---- lane i decodes record i: the values of the runs, and the runs that hold any ------------------
        const ParkedRun parked = decode_parked(r_pos, r_lo, r_hi, r_info, rb, rb_mask);
        const uint32_t cnt = (lane < m && parked.kind < kEnd) ? min(parked.count, kValueCap) : 0u;
        const uint32_t sum = __builtin_amdgcn_readlane(wave_inclusive_sum_saturating(cnt, kValueCap), 63);
        vals = min(vals + sum, kValueCap);
        const uint64_t holds = __ballot(cnt != 0u);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(holds >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)holds, 0u));
        if (cnt != 0u) list[listed + rank] = (uint16_t)(r_pos - t0);
        listed += (uint32_t)__popcll(holds);
    }
    if (lane == 0) {
        // The walk has to leave the tile where the table said it would.  "The chain ends here" has two spellings: a
        // header that lies inside the tile and holds nothing more (decode_run says so: kScoutNone, `term`), and a
        // position at or behind the last stream byte -- where the stream ends with the tile, the exact decode of a
        // last header without values (decoder.rs:43-45) says the first and the plain walk over it the second.
        const uint32_t said = tl.exit[g];
        const bool said_over = said == kScoutNone || (said != kScoutGaveUp && said >= end);
        const bool agrees = (term != 0u || pos >= end) ? said_over : said == pos;
        tl.vals[g] = vals;
        tl.runs[g] = listed | (term == kPanic ? kScoutPanic : 0u) | (term != 0u ? kScoutTerm : 0u) | (agrees ? 0u : kScoutBad);
    }
}

// ---------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------
// The Resume entry of block b of the page (one round: the tiles of a page are numbered from tile_base[0] on).
__device__ __forceinline__ Resume *scout_resume_entry(const PageDev &pg, Resume *resume, uint32_t b) {
    return resume + (size_t)pg.tile_number(b / kBlocksPerTile) * kBlocksPerTile + (b % kBlocksPerTile);
}
// Blocks [from, n_blocks) of the page lie behind the last value the stream holds: zeros (mod.rs:95); closes the
// header list after `n_runs` entries and gives the page its status.  All lanes of one wave.
__device__ __forceinline__ void scout_close_page(const PageDev &pg, const uint32_t page, Resume *resume, uint32_t *run_pos,
                                                 int32_t *status, const uint32_t produced,
                                                 const uint32_t n_runs, const int32_t st, const uint32_t lane) {
    const uint32_t n_blocks = (pg.num_values + kBlockValues - 1u) / kBlockValues;
    Resume e;
    e.run_index = n_runs;
    e.rem_kind = kRemMask | (kRle << 31);
    e.payload = 0;
    e.skip = kEndPos;
    for (uint32_t b = (produced + kBlockValues - 1u) / kBlockValues + lane; b < n_blocks; b += 64u)
        *scout_resume_entry(pg, resume, b) = e;
    if (lane == 0) {
        run_pos[pg.run_base + n_runs] = kEndPos;
        status[page] = st;  // (the page's ParseState stays as the probe left it: other waves are still asking it)
    }
}

// One workgroup (of kScoutScanThreads) per page.
__global__ __launch_bounds__(kScoutScanThreads) void scout_scan_kernel(const PageDev *__restrict__ pages,
                                                         const ScoutPage *__restrict__ sp, uint32_t n_pages,
                                                         uint32_t *__restrict__ tiles, uint32_t n_stiles,
                                                         ScoutFin *__restrict__ fin, Resume *__restrict__ resume,
                                                         uint32_t *__restrict__ tile_page, uint32_t *__restrict__ run_pos,
                                                         int32_t *__restrict__ status, const ParseState *__restrict__ state) {
    constexpr uint32_t kWaves = kScoutScanThreads / 64u;
    __shared__ uint32_t wave_vals[kWaves], wave_runs[kWaves], first_full, last_reached, any_flags;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t n = pg.num_values;
    if (n == 0u) {
        if (tid == 0) status[page] = 0;
        return;
    }
    // expand_kernel finds its page through tile_page[tile]
    for (uint32_t k = tid; k < (n + kTileValues - 1u) / kTileValues; k += kScoutScanThreads) tile_page[pg.tile_number(k)] = page;
    const ScoutPage s = sp[page];
    if (s.n_tiles == 0u) {  // decoder.rs:31-37: nothing to iterate
        if (wave == 0u) scout_close_page(pg, page, resume, run_pos, status, 0u, 0u, 0, lane);
        return;
    }
    if (scout_page_done(state, page)) return;
    const ScoutTiles t = scout_tiles(tiles, n_stiles);
    if (tid == 0) first_full = kScoutNone, last_reached = 0u, any_flags = 0u;
    uint32_t carry_vals = 0, carry_runs = 0;
    for (uint32_t k0 = 0; k0 < s.n_tiles; k0 += kScoutScanThreads) {
        __syncthreads();
        const uint32_t k = k0 + tid;
        const bool in = k < s.n_tiles;
        const uint32_t g = s.tile_base + k;
        const uint32_t v = in ? t.vals[g] : 0u;
        const uint32_t rf = in ? t.runs[g] : 0u;
        const bool reached = in && t.entry[g] != kScoutNone;
        const uint32_t r = rf & kScoutRunsMask;
        const uint32_t iv = wave_inclusive_sum_saturating(v, kValueCap);
        const uint32_t ir = wave_inclusive_sum(r);
        if (lane == 63u) wave_vals[wave] = iv, wave_runs[wave] = ir;
        __syncthreads();
        // the waves' totals, summed up by every wave for itself: lane i holds wave i's, one prefix sum over the
        // lanes gives the sum of the waves before this one and the sum of all (a chain of 2 x 16 LDS reads before)
        static_assert(kWaves <= 64u, "a lane for every wave");
        const uint32_t tv = wave_inclusive_sum_saturating(lane < kWaves ? wave_vals[lane] : 0u, kValueCap);
        const uint32_t tr = wave_inclusive_sum(lane < kWaves ? wave_runs[lane] : 0u);
        const uint32_t before = (wave_id + 63u) & 63u;  // (wave 0: lane 63, not used)
        const uint32_t bv = min(carry_vals + (wave_id != 0u ? (uint32_t)__builtin_amdgcn_readlane(tv, before) : 0u), kValueCap);
        const uint32_t br = carry_runs + (wave_id != 0u ? (uint32_t)__builtin_amdgcn_readlane(tr, before) : 0u);
        const uint32_t excl_v = min(bv + (uint32_t)__builtin_amdgcn_update_dpp(0u, iv, 0x138, 0xf, 0xf, false), kValueCap);  // wave_shr:1
        const uint32_t incl_v = min(bv + iv, kValueCap);
        if (in) {
            t.base_vals[g] = excl_v;
            t.base_runs[g] = br + ir - r;
        }
        if (reached) {
            atomicMax(&last_reached, k);
            if (incl_v >= n && excl_v < n) atomicMin(&first_full, k);
            if (rf & (kScoutPanic | kScoutBad)) atomicOr(&any_flags, rf & (kScoutPanic | kScoutBad));
        }
        carry_vals = min(carry_vals + (uint32_t)__builtin_amdgcn_readlane(tv, kWaves - 1u), kValueCap);
        carry_runs += (uint32_t)__builtin_amdgcn_readlane(tr, kWaves - 1u);
    }
    __syncthreads();
    if (tid == 0) {
        ScoutFin f;
        f.tile = first_full != kScoutNone ? first_full : last_reached;
        f.flags = any_flags;
        fin[page] = f;
    }
}

// ---------------------------------------------------------------------------
// place
// ---------------------------------------------------------------------------
// One wave per tile, four tiles per workgroup: the place step of parse_page over the runs the tile lists.
__global__ __launch_bounds__(256) void scout_place_kernel(const PageDev *__restrict__ pages,
                                                          const ScoutPage *__restrict__ sp, uint32_t *__restrict__ tiles,
                                                          const uint16_t *__restrict__ table, uint32_t n_stiles,
                                                          const ScoutFin *__restrict__ fin, Resume *__restrict__ resume,
                                                          uint32_t *__restrict__ run_pos, int32_t *__restrict__ status,
                                                          const ParseState *__restrict__ state) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x * 4u + wave;
    if (g >= n_stiles) return;
    const ScoutTiles t = scout_tiles(tiles, n_stiles);
    // what is known about the tile, asked for in one go, then what is known about its page
    uint32_t page_v = t.page[g], entry_v = t.entry[g], vals_v = t.base_vals[g], runs_v = t.runs[g], base_runs_v = t.base_runs[g];
    pin(page_v), pin(entry_v), pin(vals_v), pin(runs_v), pin(base_runs_v);
    const uint32_t page = __builtin_amdgcn_readfirstlane(page_v);
    const PageDev pg = pages[page];
    uint32_t base_v = sp[page].tile_base;
    const bool done = scout_page_done(state, page);
    pin(base_v);
    if (done) return;
    if (__builtin_amdgcn_readfirstlane(entry_v) == kScoutNone) return;
    const uint32_t n = pg.num_values;
    uint32_t produced = __builtin_amdgcn_readfirstlane(vals_v);
    if (n == 0u || produced >= n) return;  // the page's last value lies in an earlier tile
    const uint32_t k = g - __builtin_amdgcn_readfirstlane(base_v);
    const uint32_t t0 = k * kScoutTile;
    const uint32_t listed = __builtin_amdgcn_readfirstlane(runs_v) & kScoutRunsMask;
    uint32_t n_runs = __builtin_amdgcn_readfirstlane(base_runs_v);
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    const uint32_t w = pg.num_bits;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);
    auto rd = [&](uint32_t a) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, a, 0, 0); };
    const uint16_t *const list = table + (size_t)g * kScoutTable;
    uint32_t *const hdrs = run_pos + pg.run_base;

    for (uint32_t i0 = 0; i0 < listed && produced < n; i0 += 64u) {
        const uint32_t m = min(64u, listed - i0);
        uint32_t r_pos = 0, kind = kRle, count = 0, info = 0, after = 0;
        if (lane < m) {
            r_pos = t0 + list[i0 + lane];
            const RunInfo run = decode_run(rd, r_pos, end, w, 0xffffffffu);  // kBitpacked or kRle: it was listed
            kind = run.kind;
            count = run.count;
            info = run.kind == kRle ? run.value : run.payload;
            after = run.next;
        }
        // from here on as parse_page places the 64 runs of a pass
        const uint32_t left = n - produced;  // > 0
        const uint32_t cnt = lane < m ? min(count, left) : 0u;
        const uint32_t incl = wave_inclusive_sum_saturating(cnt, left);
        const uint32_t before = __builtin_amdgcn_update_dpp(0u, incl, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const uint32_t eff = min(cnt, left - before);  // the page ends inside or before this run: mod.rs:76-78
        const bool yields = eff > 0u;
        const uint64_t ymask = __ballot(yields);
        const uint32_t idx =
            n_runs + __builtin_amdgcn_mbcnt_hi((uint32_t)(ymask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ymask, 0u));
        if (yields) hdrs[idx] = r_pos;
        const uint32_t first = produced + before;
        const uint32_t b_first = (first + kBlockValues - 1u) / kBlockValues;
        const uint32_t b_last = (first + eff - 1u) / kBlockValues;  // meaningful where `yields`
        const uint32_t nb = (yields && b_last >= b_first) ? b_last - b_first + 1u : 0u;
        auto write_entries = [&](uint32_t b_from, uint32_t b_step, uint32_t b_to, uint32_t run_index, uint32_t run_first,
                                 uint32_t run_eff, uint32_t run_kind, uint32_t run_info, uint32_t run_after) {
            for (uint32_t b = b_from; b <= b_to; b += b_step) {
                const uint32_t skip = b * kBlockValues - run_first;
                Resume e;
                e.run_index = run_index;
                e.rem_kind = (run_eff - skip) | (run_kind << 31);
                e.payload = run_info;
                e.skip = run_kind == kRle ? run_after : skip;
                *scout_resume_entry(pg, resume, b) = e;
            }
        };
        if (nb != 0u && nb <= 4u) write_entries(b_first, 1u, b_last, idx, first, eff, kind, info, after);
        for (uint64_t big = __ballot(nb > 4u); big != 0; big &= big - 1) {  // long runs: the whole wave writes
            const uint32_t l = (uint32_t)__builtin_ctzll(big);
            write_entries(__builtin_amdgcn_readlane(b_first, l) + lane, 64u, __builtin_amdgcn_readlane(b_last, l),
                          __builtin_amdgcn_readlane(idx, l), __builtin_amdgcn_readlane(first, l),
                          __builtin_amdgcn_readlane(eff, l), __builtin_amdgcn_readlane(kind, l),
                          __builtin_amdgcn_readlane(info, l), __builtin_amdgcn_readlane(after, l));
        }
        n_runs += (uint32_t)__popcll(ymask);
        produced += __builtin_amdgcn_readlane(incl, 63);
    }
    const ScoutFin f = fin[page];
    if (__builtin_amdgcn_readfirstlane(f.tile) != k) return;
    // this tile holds the page's last value, or the end of its chain
    const uint32_t flags = __builtin_amdgcn_readfirstlane(f.flags);
    int32_t st = 0;
    if (produced < n && (flags & kScoutPanic)) st = 5;  // PQ2_REFERENCE_WOULD_PANIC
    if (flags & kScoutBad) st = 6;                      // PQ2_DEVICE_ERROR: table and walk disagree
    scout_close_page(pg, page, resume, run_pos, status, min(produced, n), n_runs, st, lane);
}

}  // namespace pq2

// ---- launcher (called from pq2_api.cpp) -------------------------------------------
// Bytes of device memory the scout kernels need for a batch with `n_stiles` stream tiles in `n_pages` pages:
// the tables, the per-tile words, the per-page closing records and the resolve step's arrays for large pages.
extern "C" uint32_t pq2_scout_tile_bytes() { return pq2::kScoutTile; }
// (two tables per tile: where the chains leave the tile, and where they leave their segments)
extern "C" size_t pq2_scout_table_bytes(uint64_t n_stiles) { return 2u * (size_t)n_stiles * pq2::kScoutTable * 2u; }
// where, in `words`, the resolve step leaves the number of rounds it took for every page (diagnostics)
extern "C" size_t pq2_scout_rounds_at(uint64_t n_stiles, uint64_t n_pages) { return (7u * (size_t)n_stiles + 2u * (size_t)n_pages) * 4u; }
extern "C" size_t pq2_scout_words_bytes(uint64_t n_stiles, uint64_t n_pages) {
    return (7u * (size_t)n_stiles + 3u * (size_t)n_pages + pq2::kScoutResolveArrays * ((size_t)n_stiles + 8u * (size_t)n_pages) +
            pq2::kScoutKeep * (size_t)n_stiles + 20u) * 4u;
}

// The header walk of a whole batch.  `spages`: n_pages ScoutPage records (device); `table`, `words`: scratch of
// the sizes above.  Leaves resume / tile_page / run_pos / status / state as parse_kernel's single round does.
extern "C" hipError_t pq2_launch_probe(const void *pages, uint32_t n_pages, void *resume, uint32_t *tile_page,
                                       uint32_t *run_pos, int32_t *status, void *state, uint32_t passes, uint32_t max_bits,
                                       hipStream_t stream);  // pq2_kernels.hip
#ifndef PQ2_SCOUT_PROBE_PASSES
#define PQ2_SCOUT_PROBE_PASSES 1
#endif
extern "C" hipError_t pq2_launch_scout(const void *pages, const void *spages, uint32_t n_pages, uint32_t n_stiles,
                                       void *table, void *words, void *resume, uint32_t *tile_page, uint32_t *run_pos,
                                       int32_t *status, void *state, uint32_t max_bits, hipStream_t stream) {
    using namespace pq2;
    if (n_pages == 0) return hipSuccess;
    // pages of few runs are done by a short ordinary walk (it also gives every page with values its ParseState)
    hipError_t e = pq2_launch_probe(pages, n_pages, resume, tile_page, run_pos, status, state, PQ2_SCOUT_PROBE_PASSES,
                                    max_bits, stream);
    if (e != hipSuccess) return e;
    const ParseState *done = (const ParseState *)state;
    const PageDev *pg = (const PageDev *)pages;
    const ScoutPage *sp = (const ScoutPage *)spages;
    uint32_t *tiles = (uint32_t *)words;
    ScoutFin *fin = (ScoutFin *)(tiles + 7u * (size_t)n_stiles);
    uint32_t *rounds = (uint32_t *)(fin + n_pages);  // per page: rounds of the resolve step (pq2_scout_rounds_at)
    uint32_t *scratch = rounds + n_pages;
    // kScoutKeep words per stream tile, at a multiple of 16 bytes
    uint32_t *hints = tiles + ((7u * (size_t)n_stiles + 3u * (size_t)n_pages +
                                kScoutResolveArrays * ((size_t)n_stiles + 8u * (size_t)n_pages) + 3u) & ~(size_t)3u);
    uint16_t *tbl = (uint16_t *)table;
    const uint32_t tile_groups = (n_stiles + 3u) / 4u;
    if (n_stiles != 0u) {
        hipLaunchKernelGGL(scout_table_kernel, dim3(tile_groups), dim3(256), 0, stream, pg, sp, n_pages, tiles, tbl, n_stiles,
                           done, hints);
        constexpr uint32_t kResolveLds = kScoutResolveArrays * (kScoutLdsTiles + 8u) * 4u;  // above the 64 KiB a kernel gets unasked
        e = hipFuncSetAttribute((const void *)scout_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResolveLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(scout_resolve_kernel, dim3(n_pages), dim3(kScoutResolveThreads), kResolveLds, stream, pg, sp,
                           n_pages, tbl, tiles, n_stiles, scratch, rounds, done, hints);
        if (max_bits > 16u)
            hipLaunchKernelGGL(scout_count_kernel<true>, dim3(tile_groups), dim3(256), 0, stream, pg, sp, tiles, tbl, n_stiles, done);
        else
            hipLaunchKernelGGL(scout_count_kernel<false>, dim3(tile_groups), dim3(256), 0, stream, pg, sp, tiles, tbl, n_stiles, done);
    }
    hipLaunchKernelGGL(scout_scan_kernel, dim3(n_pages), dim3(kScoutScanThreads), 0, stream, pg, sp, n_pages, tiles, n_stiles, fin,
                       (Resume *)resume, tile_page, run_pos, status, done);
    if (n_stiles != 0u)
        hipLaunchKernelGGL(scout_place_kernel, dim3(tile_groups), dim3(256), 0, stream, pg, sp, tiles, tbl, n_stiles, fin,
                           (Resume *)resume, run_pos, status, done);
    return hipGetLastError();
}
