// pq2hip kernels (gfx950 / MI355X).
//
// A batch of hybrid-RLE pages is decoded by two kernels:
//
//   parse_kernel   one wave per page.  Walks the run headers (a serial chain:
//                  the position of a header is only known once the previous
//                  one is decoded) with the stream staged through an LDS ring.
//                  It writes the position of every header (4 bytes per run)
//                  and, for every 1024-value block of the output, the run that
//                  covers the block's first value (Resume, 16 bytes per 4 KiB
//                  of output).  It reads the stream once and writes no values.
//   expand_kernel  one wave per 1024-value block, four blocks per workgroup.
//                  Stages the block's stream bytes in LDS with one coalesced
//                  fill; loads the positions of the headers that follow the
//                  block's first run, one per lane, and decodes those headers
//                  in parallel; a wave-wide prefix sum of the run lengths gives
//                  every run its output offset and fills a 64-entry run table.
//                  Then every lane produces 4 consecutive values per step: it
//                  finds its run by binary search in the table, takes RLE
//                  values as they are and bit-packed fields out of the LDS
//                  image, and writes one 16-byte-aligned dwordx4 -- 1 KiB of
//                  contiguous output per wave instruction, whatever the run
//                  boundaries are.  All value traffic of the decode is here.
//
// bitunpack_kernel is the header-less special case (bitpacked::Decoder<u32>).
#include "pq2_device.h"

namespace pq2 {

#ifndef PQ2_PARSE_HALF
#define PQ2_PARSE_HALF 4096
#endif
constexpr uint32_t kHalf = PQ2_PARSE_HALF;      // bytes the stream window of parse_kernel advances by
constexpr uint32_t kWin = 2 * kHalf;            // bytes of the window
constexpr int kHalfPieces = (int)(kHalf / 1024u);  // dwordx4 loads of the wave per half

// ---------------------------------------------------------------------------
// parse_kernel
//
// Finding the headers is a chain of dependent steps (LDS read of a header ->
// position of the next header -> LDS read ...), so one wave per page walks it
// with all lanes holding the same values, and since a single wave issues one
// instruction every four cycles the walk is written for instruction count:
//
//   chase   per run: one unaligned 8-byte LDS read (a header of 1..2 bytes and
//           the bytes behind it), the scalar arithmetic that yields the next
//           header's position, and three v_writelane that park (position, 8
//           bytes) in lane i of a register batch; 30 instructions, written in
//           assembly.  The read of the next header is issued before the checks
//           and the parking of the current one.  Longer varints, runs cut by
//           the end of the stream and the end itself leave the loop and go
//           through decode_run, which parks an explicit record.
//   place   once per 64 parked runs, lane-parallel: every lane decodes its
//           run, a wave-wide prefix sum of the run lengths gives each run its
//           output offset, runs that yield values get their slot in the
//           header-position list, and the runs that cover the first value of a
//           1024-value block write that block's Resume entry.
//
// The chase does not know how many values it has passed; the place step stops
// the page once num_values are covered, so at most one batch of headers behind
// the last needed one is looked at (and never reported: a header the reference
// would not reach cannot fail the page).
// ---------------------------------------------------------------------------
// The header walk of one page by one wave.  `win` (kWin bytes of LDS, private to the wave) holds stream byte a at
// win[a - win_lo] for a in [win_lo, win_lo + kWin).  `publish(v)` is called, by all lanes, with the number of
// values the listed runs cover once the entries that describe them (header positions, Resume) have been stored,
// and with kAllParsed when the page's entries are complete; whoever consumes the entries while the walk is
// still going waits on that number.
// kWide: some page of the batch is wider than 16 bits.
constexpr uint32_t kAllParsed = 0xffffffffu;
template <bool kWide, typename Publish>
__device__ __forceinline__ void parse_page(const PageDev &pg, const uint32_t page, uint8_t *win,
                                           Resume *__restrict__ resume, uint32_t *__restrict__ tile_page,
                                           uint32_t *__restrict__ run_pos, int32_t *__restrict__ status,
                                           ParseState *__restrict__ state, const uint32_t round,
                                           const uint32_t lane, Publish publish, uint32_t passes = 0xffffffffu) {
    // `passes`: the walk also stops after this many passes of up to 64 runs each (probe_kernel)
    const uint32_t n = pg.num_values;
    if (n == 0) {
        if (lane == 0 && round == 0) status[page] = 0;
        publish(kAllParsed);
        return;
    }
    const uint32_t w = pg.num_bits;
    const uint32_t n_blocks = (n + kBlockValues - 1) / kBlockValues;
    // the Resume entry of the page's block b
    auto entry = [&](uint32_t b) -> Resume * {
        return resume + (size_t)pg.tile_number(b / kBlocksPerTile) * kBlocksPerTile + (b % kBlocksPerTile);
    };
    // expand_kernel finds its page through tile_page[tile]
    if (round == 0 && tile_page != nullptr)
        for (uint32_t t = lane; t < (n_blocks + kBlocksPerTile - 1) / kBlocksPerTile; t += 64)
            tile_page[pg.tile_number(t)] = page;

    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    ParseState ps;
    if (round == 0) {
        ps.pos = delta, ps.produced = 0, ps.n_runs = 0, ps.done = 0;
    } else {
        ps = state[page];  // the same for every lane: into scalar registers
        ps.pos = __builtin_amdgcn_readfirstlane(ps.pos);
        ps.produced = __builtin_amdgcn_readfirstlane(ps.produced);
        ps.n_runs = __builtin_amdgcn_readfirstlane(ps.n_runs);
        ps.done = __builtin_amdgcn_readfirstlane(ps.done);
        if (ps.done) {
            publish(kAllParsed);
            return;
        }
    }
    // this round ends once the values of the page's tiles up to the round's last one are covered
    const uint64_t round_end = round + 1u < kRounds ? (uint64_t)pg.round_first(round + 1u) * kTileValues : n;
    const uint32_t target = round_end > n ? n : (uint32_t)round_end;

    uint32_t *const hdrs = run_pos + pg.run_base;  // header positions of the runs that yield values
    uint32_t n_runs = ps.n_runs;                   // entries of `hdrs` written so far
    uint32_t produced = ps.produced;               // values those runs yield

    // blocks [from, n_blocks): the stream is exhausted, zeros follow (mod.rs:95); closes the header list
    auto fill_end = [&](uint32_t from) {
        Resume e;
        e.run_index = n_runs;
        e.rem_kind = kRemMask | (kRle << 31);
        e.payload = 0;
        e.skip = kEndPos;
        for (uint32_t b = from + lane; b < n_blocks; b += 64) *entry(b) = e;
        if (lane == 0) hdrs[n_runs] = kEndPos;
    };
    auto finish = [&](int32_t st) {
        fill_end((produced + kBlockValues - 1u) / kBlockValues);
        if (lane == 0) {
            status[page] = st;
            state[page].done = 1u;
        }
        publish(kAllParsed);
    };

    if (w == 0 || pg.in_len == 0 || w > 32) {  // decoder.rs:31-37: nothing to iterate
        finish(0);
        return;
    }

    const uint64_t base16 = pg.in_addr - delta;
    const uint32_t end = delta + pg.in_len;  // a-offset one past the last stream byte
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base16, (end + 3u) & ~3u);
    const uint32_t rb = ceil8(w);                                    // bytes of an RLE value
    const uint32_t rb_mask = rb < 4 ? (1u << (rb * 8u)) - 1u : ~0u;  // the first min(4, rb) of them
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)win;

    u32x4 pre[kHalfPieces];                    // prefetched half [win_lo + kWin, win_lo + kWin + kHalf)
    uint32_t win_lo = ps.pos & ~(kHalf - 1u);  // a-offset (multiple of kHalf) of the first resident byte

    auto load_half = [&](uint32_t half_a, u32x4 *r) {
#pragma unroll
        for (int i = 0; i < kHalfPieces; ++i)
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, half_a + (uint32_t)i * 1024u + lane * 16u, 0, 0);
    };
    auto store_half = [&](uint32_t slot, const u32x4 *r) {
#pragma unroll
        for (int i = 0; i < kHalfPieces; ++i) *(u32x4 *)&win[slot + (uint32_t)i * 1024u + lane * 16u] = r[i];
    };
    // moves the window so that [at, at + 32) is resident; at >= win_lo + kHalf
    auto slide = [&](uint32_t at) {
        wave_sync();  // the window belongs to this wave alone, and its LDS operations execute in order
        if (at - win_lo + 32u <= kWin + kHalf) {
            // by one half: the upper half moves down, the prefetched half becomes the upper one
            u32x4 up[kHalfPieces];
#pragma unroll
            for (int i = 0; i < kHalfPieces; ++i) up[i] = *(const u32x4 *)&win[kHalf + (uint32_t)i * 1024u + lane * 16u];
            store_half(0, up);
            store_half(kHalf, pre);
            win_lo += kHalf;
        } else {
            // a run that jumped beyond the prefetched half
            win_lo = at & ~(kHalf - 1);
            u32x4 a[kHalfPieces], b[kHalfPieces];
            load_half(win_lo, a);
            load_half(win_lo + kHalf, b);
            store_half(0, a);
            store_half(kHalf, b);
        }
        load_half(win_lo + kWin, pre);  // lands while the chase works through the window
        wave_sync();
    };
    auto resident = [&](uint32_t at, uint32_t bytes) { return at - win_lo + bytes <= kWin; };  // at >= win_lo
    auto rd = [&](uint32_t a) -> uint32_t {
        return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&win[a - win_lo]);
    };

    {  // first window
        u32x4 a[kHalfPieces], b[kHalfPieces];
        load_half(win_lo, a);
        load_half(win_lo + kHalf, b);
        load_half(win_lo + kWin, pre);
        store_half(0, a);
        store_half(kHalf, b);
        wave_sync();
    }

    uint32_t pos = ps.pos;  // a-offset of the next header
    uint32_t announce = 0;  // values covered by the entries of the passes before the last one (0: nothing to announce)
    for (;;) {
        // ---- chase: park up to 64 runs, record i in lane i --------------------------------------------
        uint32_t r_pos = 0, r_lo = 0, r_hi = 0, r_info = 0;
        uint32_t m = 0;     // parked records
        uint32_t term = 0;  // kEnd / kPanic once the chase cannot go on
        for (;;) {
            // Runs whose header has one or two bytes (uleb128.rs:3-23) and which lie inside the stream
            // (decoder.rs:44-63): next = pos + header bytes + (bit-packed ? (header >> 1) * w : ceil8(w)).
            // The loop runs with lane 0 alone enabled: an LDS read that is not naturally aligned is replayed
            // per enabled lane, and v_readfirstlane / v_writelane do not need the other lanes.  It has one
            // branch: every reason to stop is folded into `chk` (a header of three or more bytes, a run that
            // passes the end of the stream, a next header whose 8 bytes are not all in the window, the 64th
            // record).  Each pass parks its record before it knows whether the record counts; `m` is kept
            // minus 64 inside so that the 64th increment sets the carry.
            uint32_t lo, hi, groups, t, bytes, nm1, chk, addr;  // nm1: the next header's position minus one
            uint64_t exec_all;
            uint32_t m0_kept;
            const uint32_t voff = lds_base - win_lo;
            const uint32_t chase_limit = min(end - 1u, win_lo + kWin - 9u);
            // the bytes behind the header's first four are only needed for RLE values of more than two bytes
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
                if (!resident(pos, 32u)) slide(pos);
                const RunInfo run = decode_run(rd, pos, end, w, 0xffffffffu);
                // bits 16..17: kind; bits 20..27: header + value bytes of an RLE run
                const uint32_t e_lo = kExplicit | (run.kind << 16) | (((run.next - pos) & 0xffu) << 20);
                const uint32_t e_hi = __builtin_amdgcn_readfirstlane(run.count);
                const uint32_t e_info = __builtin_amdgcn_readfirstlane(run.kind == kRle ? run.value : run.payload);
                // v_writelane_b32 ignores EXEC; it may read one scalar register only, so the lane goes through M0
                // (kept and put back: the compiler does not know that the statement writes it)
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
            if (!resident(pos, 8u)) slide(pos);
            if (m == 64u) break;
        }

        // what the previous pass stored has had this pass's chase to reach memory: announcing it now costs no wait
        if (announce != 0u) publish(announce);
        // ---- place: lane i decodes record i; a prefix sum over the run lengths places the runs -------------
        const ParkedRun parked = decode_parked(r_pos, r_lo, r_hi, r_info, rb, rb_mask);
        const uint32_t kind = parked.kind, count = parked.count, info = parked.info, after = parked.after;
        const uint32_t left = n - produced;  // > 0
        const uint32_t cnt = (lane < m && kind < kEnd) ? min(count, left) : 0u;
        // inclusive prefix sum, saturating at `left` (all terms <= left < 2^31, so no step overflows)
        const uint32_t incl = wave_inclusive_sum_saturating(cnt, left);
        const uint32_t before = __builtin_amdgcn_update_dpp(0u, incl, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const uint32_t eff = min(cnt, left - before);  // the page ends inside or before this run: mod.rs:76-78
        const bool yields = eff > 0u;
        const uint64_t ymask = __ballot(yields);
        const uint32_t idx =
            n_runs + __builtin_amdgcn_mbcnt_hi((uint32_t)(ymask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ymask, 0u));
        if (yields) hdrs[idx] = r_pos;
        // the run holds values [first, first + eff) of the page: it starts blocks b_first..b_last (if any)
        const uint32_t first = produced + before;
        const uint32_t b_first = (first + kBlockValues - 1u) / kBlockValues;
        const uint32_t b_last = (first + eff - 1u) / kBlockValues;  // meaningful where `yields`
        const uint32_t nb = (yields && b_last >= b_first) ? b_last - b_first + 1u : 0u;
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
                *entry(b) = e;
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
        if (produced >= n) {
            finish(0);
            return;
        }
        if (term != 0u) {
            finish(term == kPanic ? 5 : 0);  // PQ2_REFERENCE_WOULD_PANIC | the stream is exhausted
            return;
        }
        if (produced >= target) break;
        announce = produced;
        if (--passes == 0u) break;
    }
    if (lane == 0) {
        ParseState out;
        out.pos = pos, out.produced = produced, out.n_runs = n_runs, out.done = 0;
        state[page] = out;
    }
    publish(produced);
}

template <bool kWide>
__global__ __launch_bounds__(64) void parse_kernel(const PageDev *__restrict__ pages, uint32_t n_pages,
                                                   Resume *__restrict__ resume, uint32_t *__restrict__ tile_page,
                                                   uint32_t *__restrict__ run_pos, int32_t *__restrict__ status,
                                                   ParseState *__restrict__ state, uint32_t round) {
    __shared__ __attribute__((aligned(16))) uint8_t win[kWin];
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    parse_page<kWide>(pg, page, win, resume, tile_page, run_pos, status, state, round, threadIdx.x, [](uint32_t) {});
}

// The first `passes` passes of the walk of every page: a page of few runs (one bit-packed run, as the reference's
// own encoder writes it: src/encoding/hybrid_rle/encoder.rs:59-70) is complete after them (ParseState::done), and
// the scout kernels leave it alone.
template <bool kWide>
__global__ __launch_bounds__(64) void probe_kernel(const PageDev *__restrict__ pages, uint32_t n_pages,
                                                   Resume *__restrict__ resume, uint32_t *__restrict__ tile_page,
                                                   uint32_t *__restrict__ run_pos, int32_t *__restrict__ status,
                                                   ParseState *__restrict__ state, uint32_t passes) {
    __shared__ __attribute__((aligned(16))) uint8_t win[kWin];
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    parse_page<kWide>(pg, page, win, resume, tile_page, run_pos, status, state, 0u, threadIdx.x, [](uint32_t) {}, passes);
}

// ---------------------------------------------------------------------------
// expand_kernel
// ---------------------------------------------------------------------------
constexpr uint32_t kTableRuns = 64;       // run-table entries per wave (the last one stays a sentinel)
constexpr uint32_t kNoRun = 0xffffffffu;  // table key of an unused entry: compares greater than any position
#ifndef PQ2_SPAN_BLOCKS
#define PQ2_SPAN_BLOCKS 2
#endif
// expand_kernel works through a tile in spans of kSpanBlocks blocks: one run table, one chunk index and one set
// of emission steps per span (the Resume entry of a span's first block says where it starts; those of the
// other blocks are not needed).
constexpr uint32_t kSpanBlocks = PQ2_SPAN_BLOCKS;  // 1, 2 or 4
constexpr uint32_t kSpanValues = kSpanBlocks * kBlockValues;
constexpr uint32_t kSpanChunks = kSpanValues / 4u;     // chunks of 4 values per span ...
constexpr uint32_t kChunkSlots = kSpanChunks + 8u;     // ... + 2 when the output is not 16-byte aligned, rounded up
#ifndef PQ2_STEP_GROUP
#define PQ2_STEP_GROUP 5
#endif
constexpr uint32_t kStepGroup = PQ2_STEP_GROUP;  // emission steps whose LDS reads are in flight together
#ifndef PQ2_DIRECT_VALUES
#define PQ2_DIRECT_VALUES 8
#endif
constexpr uint32_t kDirectValues = PQ2_DIRECT_VALUES;  // a pass whose runs are all this short is written run by run (0: never)

// Where expand_block's values go.  ExpandIndexSink stores them as they are (u32, the decode of HybridRleDecoder);
// ExpandDictSink<kWords> takes them as indices into the page's dictionary, which lies in memory (a few KiB that
// every tile of the page reads: they stay in the caches), and stores the dictionary's values of kWords dwords each
// (the consumer loop of a dictionary-encoded page, tests/it/read/primitive.rs:99-108 with
// PrimitivePageDict::value, tests/it/read/dictionary/primitive.rs:20-27; an index that is not below the
// dictionary's length is an error of the page, its value is written as zeros).  put4 writes the four positions
// from p on, which begin at a 16-byte boundary of the output, put1 writes one; mis(): the positions between the
// 16-byte grid of the output and position 0.
struct ExpandIndexSink {
    static constexpr int kMaxBits = 32;
    out_u32 *out;  // position 0
    __device__ __forceinline__ uint32_t mis() const { return (uint32_t)(((uintptr_t)out >> 2) & 3u); }
    __device__ __forceinline__ void put4(const u32x4 v, const int32_t p) { store4(v, out + p); }
    __device__ __forceinline__ void put1(const uint32_t v, const int32_t p) { out[p] = v; }
    __device__ __forceinline__ ExpandIndexSink shifted(uint32_t positions) const { return ExpandIndexSink{out + positions}; }
    __device__ __forceinline__ void absorb(const ExpandIndexSink &) {}
};
template <int kWords>
struct ExpandDictSink {
    static constexpr int kMaxBits = 16;  // wider indices: through a buffer of indices and gather_kernel
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    out_u32 *out;  // the first dword of the value at position 0 (aligned to a whole value: 4, 8 and 4 bytes)
    gu32 *dict;    // dict_len values of kWords dwords
    uint32_t dict_len;
    bool bad;
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
            for (int c = 0; c < kWords; ++c) v[k * kWords + c] = ok ? dict[(size_t)jj[k] * kWords + c] : 0u;
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
        for (int c = 0; c < kWords; ++c) out[(ptrdiff_t)p * kWords + c] = ok ? dict[(size_t)j * kWords + c] : 0u;
    }
    __device__ __forceinline__ ExpandDictSink shifted(uint32_t positions) const {
        return ExpandDictSink{out + (size_t)positions * kWords, dict, dict_len, bad};
    }
    __device__ __forceinline__ void absorb(const ExpandDictSink &s) { bad |= s.bad; }
};

// One wave decodes the span [block_start, block_start + block_n) of a page
// whose width is W (block_n <= kSpanValues).  `tbl` (kTableRuns + 2 entries), `cidx` (kChunkSlots entries) and
// `win` are the wave's LDS; the window normally holds the block's stream bytes
// already and `hp_first` are the positions of the headers behind its first run.
//
// A pass covers block positions [lo, hi) with at most 64 runs:
//   phase 1  the first run comes from the Resume entry (or the previous pass);
//            lane r decodes the header of run r + 1, a wave prefix sum of the
//            run lengths gives every run its first position, and the runs go
//            into the table: key = first position << 1 | kind, y = the RLE
//            value, or for a bit-packed run the LDS bit position of its value
//            at block position 0 (so that position p sits at y + p * W).
//   index    output is produced in chunks of 4 values whose address is 16-byte
//            aligned; chunk q holds positions [4q - mis, 4q - mis + 4).  Every
//            run marks the first chunk that begins at or behind its first
//            position, and a prefix maximum over the chunks turns the marks
//            into "the run that holds the chunk's first position".
//   phase 2  every lane takes one chunk per step.  A chunk that lies inside
//            one run (nearly all do) takes its four fields out of the LDS
//            image with one funnel shift per dword, or is four copies of the
//            RLE value, and is written with one dwordx4.  The chunks that hold
//            the first position of a run are left to one last step in which
//            lane r works through the chunk that run r + 1 starts in, value
//            by value.
// kCoh: the header list is still being written by another wave (see decode_kernel): its entries are loaded past
// this compute unit's vector cache.
template <int W, bool kCoh, typename Sink>
__device__ __forceinline__ void expand_block(const __amdgpu_buffer_rsrc_t rsrc, Window &win, uint2 *tbl,
                                             uint32_t *cidx, const Resume rs, const uint32_t hp_first,
                                             const uint32_t *__restrict__ hdrs, const uint32_t end, const uint32_t n,
                                             const uint32_t block_start, const uint32_t block_n, Sink &sink,
                                             const uint32_t lane) {
    constexpr uint32_t w = (uint32_t)W;
    constexpr uint32_t kLaneRuns = kTableRuns - 1;               // runs decoded by lanes in one pass
    const uint32_t max_vals = ((win.cap - 64u) / w) * 8u;        // values of one span that surely fit the window
    const uint32_t bit_mask = win.bit_mask;

    // the first run of the pass (wave-uniform)
    uint32_t kind = rs.rem_kind >> 31;
    uint32_t rem = rs.rem_kind & kRemMask;
    uint32_t payload = rs.payload;  // RLE: the value
    uint32_t skip = rs.skip;        // bit-packed only
    uint32_t ridx = rs.run_index;
    bool first_pass = true;
    // header positions for the next pass, asked for as soon as the pass in hand knows where it ends
    uint32_t ahead_at = kEndPos, ahead_hp = kEndPos;

    const uint32_t mis = sink.mis();  // values between the 16-byte grid of the output and position 0
    uint32_t lo = 0;                                               // block positions [0, lo) are written

    do {  // block_n > 0
        // ---- phase 1: the runs that cover [lo, hi) into the table ------------------------------------
        const uint32_t est = (((block_n - lo) * w) >> 3) + 64u;  // bytes the rest of the block is expected to need
        uint32_t take0 = min(rem, block_n - lo);
        uint32_t y0 = payload;
        bool alone = take0 == block_n - lo;  // the first run covers the rest of the block
        if (kind == kBitpacked) {
            const uint32_t b_lo = payload + (skip >> 3) * w;  // first byte of the group holding value `skip`
            uint32_t b_hi = payload + ((skip + take0 + 7u) >> 3) * w + 8u;
            if (!win.holds(b_lo & ~3u, b_hi)) {
                if (take0 > max_vals) take0 = max_vals, alone = true;  // more than a window: this span only
                b_hi = payload + ((skip + take0 + 7u) >> 3) * w + 8u;
                win.fill(rsrc, b_lo, max(b_hi - b_lo + 16u, est), lane);
            }
            // LDS bit position of the run's value `skip`, which is block position lo
            y0 = (b_lo - win.lo) * 8u + (skip & 7u) * w - lo * w;
        }
        uint32_t hi = lo + take0;

        if (alone && kind == kRle) {  // one value over the whole span: no table, no index
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
            uint32_t key = kNoRun, y = 0;  // lane r: table entry r + 1
            uint32_t start = 0;            // block position of its first value
            uint32_t g = 0;                // lanes [0, g) hold usable runs
            LaneRun run;                   // lane r: run ridx + 1 + r
            run.kind = kRle, run.count = 0, run.info = 0;
            bool direct = false;           // a pass of short runs: see below
            bool next_known = false;
            uint32_t next_kind = kRle, next_count = 0, next_info = 0;
            if (!alone) {
                // lane r <- run ridx + 1 + r: position, header, length; a prefix sum places them
                uint32_t hp = kEndPos;
                hp = first_pass ? hp_first : ahead_hp;
                const uint32_t hp0 = __builtin_amdgcn_readfirstlane(hp);
                if (hp0 != kEndPos && !win.holds(hp0 & ~3u, (hp0 & ~3u) + 24u)) {
                    if (kind == kRle) win.fill(rsrc, hp0, est, lane);  // nothing of this pass lives in the window yet
                }
                const bool valid = hp != kEndPos && hp >= win.lo && (hp & ~3u) + 24u <= win.hi;
                if (valid) run = decode_lane<W>(win.lds, hp - win.lo, hp, end);
                const uint32_t cnt = min(run.count, 2u * kSpanValues);
                start = hi + wave_inclusive_sum(cnt) - cnt;
                bool ok = valid;
                uint32_t value = run.info;
                if (run.kind == kBitpacked) {
                    const uint32_t t = start < block_n ? min(run.count, block_n - start) : 0u;
                    ok = ok && run.info + ((t + 7u) >> 3) * w + 8u <= win.hi;
                    value = (run.info - win.lo) * 8u - start * w;
                }
                const uint64_t bad = __ballot(!ok) | (1ull << 63);  // (the table has no place for lane 63's run)
                g = (uint32_t)__builtin_ctzll(bad);
                // run ridx + 1 + g, the first run of the next pass, as far as its lane has decoded it
                next_known = __builtin_amdgcn_readlane((uint32_t)valid, g) != 0u;
                next_kind = __builtin_amdgcn_readlane(run.kind, g);
                next_count = __builtin_amdgcn_readlane(run.count, g);
                next_info = __builtin_amdgcn_readlane(run.info, g);
                const uint32_t start_g = __builtin_amdgcn_readlane(start, g);
                const uint32_t hp_g = __builtin_amdgcn_readlane(hp, g);
                const bool ends = g < kLaneRuns && hp_g == kEndPos;  // no more runs: zeros follow (mod.rs:95)
                if (lane < g && start < block_n) {
                    key = (start << 1) | run.kind;
                    y = value;
                }
                if (ends && lane == g && start < block_n) key = (start << 1) | kRle;  // y stays 0
                hi = (start_g >= block_n || ends) ? block_n : start_g;
                const bool lengthy = lane < g && start < block_n && min(run.count, block_n - start) > kDirectValues;
                direct = take0 <= 64u && !ends && __ballot(lengthy) == 0;
                if (hi < block_n) {  // run ridx + 1 + g will be the first run of the next pass
                    const uint32_t *const at = &hdrs[ridx + 1u + g];
                    ahead_at = kCoh ? load_coherent(at) : *at;
                    ahead_hp = kCoh ? load_coherent(at + 1u + lane) : at[1u + lane];
                }
            }
            if (direct) {
                // ---- a pass of nothing but short runs: every lane writes its own run ---------------------------
                // No run of the pass has more than kDirectValues values (pages of very short runs: one pass covers
                // a few hundred values, and the table, the chunk index and the steps over a whole span would cost
                // more than the values).  The first run's values go one to a lane.
                if (lane < take0) {
                    const uint32_t p0 = lo + lane;
                    sink.put1(kind == kRle ? payload : field_at<W>(win.lds, (y0 + p0 * w) & bit_mask), (int32_t)p0);
                }
                if (lane < g && start < block_n) {
                    const uint32_t mine = min(run.count, block_n - start);
                    const uint32_t bit0 = run.kind == kBitpacked ? (run.info - win.lo) * 8u : 0u;
                    uint32_t f[kDirectValues];
#pragma unroll
                    for (uint32_t k = 0; k < kDirectValues; ++k) f[k] = field_at<W>(win.lds, (bit0 + k * w) & bit_mask);
#pragma unroll
                    for (uint32_t k = 0; k < kDirectValues; ++k)
                        if (k < mine) sink.put1(run.kind == kRle ? run.info : f[k], (int32_t)(start + k));
                }
            } else {
#pragma unroll
                for (uint32_t i = 0; i < kSpanBlocks; ++i) *(u32x4 *)&cidx[4u * (lane + 64u * i)] = u32x4{0u, 0u, 0u, 0u};
                if (lane < kChunkSlots - kSpanChunks) cidx[kSpanChunks + lane] = 0u;
                tbl[(lane + 1u) & 63u] = lane == 63u ? make_uint2((lo << 1) | kind, y0) : make_uint2(key, y);
                // ---- index: the run that holds the first position of every chunk ---------------------------
                if (key != kNoRun) atomicMax(&cidx[(start + mis + 3u) >> 2], lane + 1u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!alone) {
                    // lane l owns the 4 * kSpanBlocks consecutive chunks from 4 * kSpanBlocks * l on
                    u32x4 m[kSpanBlocks];
                    uint32_t run_max = 0;
#pragma unroll
                    for (uint32_t i = 0; i < kSpanBlocks; ++i) {
                        m[i] = *(const u32x4 *)&cidx[4u * (kSpanBlocks * lane + i)];
                        m[i].x = max(m[i].x, run_max);
                        m[i].y = max(m[i].x, m[i].y);
                        m[i].z = max(m[i].y, m[i].z);
                        m[i].w = max(m[i].z, m[i].w);
                        run_max = m[i].w;
                    }
                    const uint32_t upto = wave_inclusive_max(run_max);
                    const uint32_t before = __builtin_amdgcn_update_dpp(0u, upto, 0x138, 0xf, 0xf, false);  // wave_shr:1
#pragma unroll
                    for (uint32_t i = 0; i < kSpanBlocks; ++i) {
                        m[i].x = max(m[i].x, before), m[i].y = max(m[i].y, before);
                        m[i].z = max(m[i].z, before), m[i].w = max(m[i].w, before);
                        *(u32x4 *)&cidx[4u * (kSpanBlocks * lane + i)] = m[i];
                    }
                    if (lane == 63u) {  // two more chunks exist when the output is not 16-byte aligned
                        const uint32_t c0 = max(upto, cidx[kSpanChunks]);
                        cidx[kSpanChunks] = c0;
                        cidx[kSpanChunks + 1u] = max(c0, cidx[kSpanChunks + 1u]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }

                // ---- phase 2: one chunk per lane and step ---------------------------------------------------
                // The steps do not depend on each other and there are at most 4 * kSpanBlocks + 1 of them (one more
                // than the chunks of a span when the output is not 16-byte aligned).  Every step is a chain of three
                // dependent LDS reads (chunk index -> table entry -> stream dwords); the steps are taken in groups
                // whose reads of one kind are issued together and waited for once (`pin` keeps the compiler from
                // sinking a read into the branch that uses it, which would serialise the chains).
                const uint32_t q_end = (hi + mis + 3u) >> 2;
                const uint32_t q_first = ((lo + mis) >> 2) + lane;
                bool ragged = false;  // some lane holds the first or last chunk of the span, and not all of it
                constexpr uint32_t kSteps = 4u * kSpanBlocks + 1u;
#pragma unroll
                for (uint32_t g0 = 0; g0 < kSteps; g0 += kStepGroup) {
                    if (((lo + mis) >> 2) + 64u * g0 >= q_end) break;  // a pass of few short runs ends early
                    uint32_t j[kStepGroup];
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) j[i] = cidx[min(q_first + 64u * (g0 + i), kChunkSlots - 1u)];
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) pin(j[i]);
                    // the entry and the key of the next one (two naturally aligned reads: a 16-byte read at an
                    // 8-byte boundary would be replayed)
                    uint2 e[kStepGroup];
                    uint32_t nxt[kStepGroup];
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) e[i] = tbl[j[i]], nxt[i] = tbl[j[i] + 1u].x;
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) pin(e[i].x), pin(e[i].y), pin(nxt[i]);
                    FieldWords<W> x[kStepGroup];
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) {
                            const int32_t p = (int32_t)((q_first + 64u * (g0 + i)) * 4u) - (int32_t)mis;
                            x[i] = load_fields<W>(win.lds, (e[i].y + (uint32_t)p * w) & bit_mask);
                        }
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) x[i].pin_all();
#pragma unroll
                    for (uint32_t i = 0; i < kStepGroup; ++i)
                        if (g0 + i < kSteps) {
                            const uint32_t q = q_first + 64u * (g0 + i);
                            const bool active = q < q_end;
                            const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                            // no run starts behind the chunk's first position
                            const bool single = active & (((((uint32_t)(p + 3)) << 1) | 1u) < nxt[i]);
                            const bool full = (p >= (int32_t)lo) & (p + 4 <= (int32_t)hi);
                            const u32x4 f = x[i].fields4();
                            const bool rle = (e[i].x & 1u) != 0u;
                            const uint32_t r = e[i].y;
                            const u32x4 v = {rle ? r : f.x, rle ? r : f.y, rle ? r : f.z, rle ? r : f.w};
                            if (single & full) sink.put4(v, p);
                            ragged = ragged | (single & !full);
                        }
                }
                if (__builtin_expect(__ballot(ragged) != 0, 0)) {
                    // the first and the last chunk of the span, where they lie inside one run: position by position
#pragma unroll
                    for (uint32_t side = 0; side < 2u; ++side) {
                        const uint32_t q = side == 0 ? (lo + mis) >> 2 : q_end - 1u;
                        const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                        const int32_t pos = p + (int32_t)(lane & 3u);
                        const uint32_t j = cidx[q];
                        const uint2 e = tbl[j];
                        const uint32_t nxt = tbl[j + 1u].x;
                        const bool single = ((((uint32_t)(p + 3)) << 1) | 1u) < nxt;
                        const bool full = p >= (int32_t)lo && p + 4 <= (int32_t)hi;
                        const uint32_t f = field_at<W>(win.lds, (e.y + (uint32_t)pos * w) & bit_mask);
                        if (lane < 4u && single && !full && pos >= (int32_t)lo && pos < (int32_t)hi)
                            sink.put1((e.x & 1u) ? e.y : f, pos);
                    }
                }
                // the chunks in which a run starts: lane r takes the one of run r + 1, value by value.  The chunk's
                // four positions lie in the run that holds its first position and the up to three behind it (runs in
                // the table hold at least one value); their entries are read together, then the four fields.
                if (key != kNoRun) {
                    const uint32_t q = (start + mis) >> 2;
                    const int32_t p = (int32_t)(q * 4u) - (int32_t)mis;
                    const uint32_t j0 = cidx[q];
                    uint2 t[5];  // entries j0 .. j0 + 4 (behind the table's end: the sentinel)
#pragma unroll
                    for (uint32_t i = 0; i < 5u; ++i) t[i] = tbl[min(j0 + i, kTableRuns + 1u)];
#pragma unroll
                    for (uint32_t i = 0; i < 5u; ++i) pin(t[i].x), pin(t[i].y);
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
                        f[k] = field_at<W>(win.lds, (cur[k].y + pk * w) & bit_mask);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) pin(f[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (p + k >= (int32_t)lo && p + k < (int32_t)hi) sink.put1((cur[k].x & 1u) ? cur[k].y : f[k], p + k);
                }
            }
            // ---- next pass (rare: more than 63 runs in the block, or bytes beyond the window) --------------
            if (hi < block_n && !alone) {  // run ridx + 1 + g becomes the first run
                ridx += 1u + g;
                if (next_known && next_count != 0u) {  // its lane has read the header already
                    kind = next_kind, rem = next_count, skip = 0, payload = next_info;
                } else {
                    const uint32_t at = __builtin_amdgcn_readfirstlane(ahead_at);
                    if (!win.holds(at & ~3u, (at & ~3u) + 24u)) win.fill(rsrc, at, (((block_n - hi) * w) >> 3) + 64u, lane);
                    auto rd = [&](uint32_t a) -> uint32_t { return __builtin_amdgcn_readfirstlane(win.dword(a)); };
                    const RunInfo first = decode_run(rd, at, end, w, n - (block_start + hi));
                    if (first.kind >= kEnd || first.count == 0) {  // cannot happen for a listed run; ends the block safely
                        kind = kRle, rem = kRemMask, payload = 0, skip = 0;
                    } else {
                        kind = first.kind, rem = first.count, skip = 0;
                        payload = first.kind == kRle ? first.value : first.payload;
                    }
                }
            }
        }
        if (hi < block_n && alone) {  // the first run continues
            rem -= take0;
            skip += take0;
        }
        first_pass = false;
        lo = hi;
    } while (lo < block_n);
}

// One wave decodes the (up to) four blocks of one tile of a page whose width is
// W.  Everything the tile needs from memory is requested at once, before the
// first value is stored: the stream bytes from the first byte any of its
// blocks needs, as far as the tile is expected to reach (into the wave's LDS
// window), and for every block the positions of the headers behind its first
// run.  Memory operations complete in order, so a load issued behind a store
// cannot be waited for without waiting for the store as well; this way no
// block waits for the stores of the one before it.  `rs4`: lane j holds the
// Resume entry of the tile's block j.
template <int W, bool kCoh, typename Sink>
__device__ __forceinline__ void expand_tile(const __amdgpu_buffer_rsrc_t rsrc, Window &win, uint2 *tbl,
                                            uint32_t *cidx, const u32x4 rs4, const uint32_t *__restrict__ hdrs,
                                            const uint32_t end, const uint32_t n, const uint32_t tile_first,
                                            Sink &sink, const uint32_t lane, const uint32_t next_from = kEndPos,
                                            const uint32_t span_lo = 0u, const uint32_t span_hi = 0xffffffffu) {
    constexpr uint32_t w = (uint32_t)W;
    constexpr uint32_t kSpans = kBlocksPerTile / kSpanBlocks;  // spans per tile
    // (the wave takes the tile's spans [span_lo, span_hi): all of them, or one where a batch of few tiles gives
    // every span a wave of its own)
    const uint32_t spans = min(min(kSpans, span_hi), (n - tile_first + kSpanValues - 1u) / kSpanValues);
    // lane j holds the Resume entry of block j; span i starts with block i * kSpanBlocks
    auto resume_of = [&](uint32_t i) {
        Resume r;
        r.run_index = __builtin_amdgcn_readlane(rs4.x, i * kSpanBlocks);
        r.rem_kind = __builtin_amdgcn_readlane(rs4.y, i * kSpanBlocks);
        r.payload = __builtin_amdgcn_readlane(rs4.z, i * kSpanBlocks);
        r.skip = __builtin_amdgcn_readlane(rs4.w, i * kSpanBlocks);
        return r;
    };
    auto length_of = [&](uint32_t i) { return min(kSpanValues, n - (tile_first + i * kSpanValues)); };

    // lane j = i * kSpanBlocks: does span i take bytes from the stream, and from where on
    uint32_t from = kEndPos;
    bool alone = true;
    if (lane < kBlocksPerTile && lane % kSpanBlocks == 0u && lane / kSpanBlocks < spans && lane / kSpanBlocks >= span_lo) {
        const uint32_t kind = rs4.y >> 31;
        alone = (rs4.y & kRemMask) >= length_of(lane / kSpanBlocks);  // the first run covers the whole span
        // bit-packed: the group that holds the span's first value; RLE: the byte behind the run
        const uint32_t at = kind == kBitpacked ? rs4.z + (rs4.w >> 3) * w : rs4.w;
        if (!(kind == kRle && alone)) from = at;  // kEndPos where the stream is exhausted
    }
    // header positions: lane r < 63 of hp[i] <- position of header run_index(i) + 1 + r
    uint32_t hp[kSpans];
#pragma unroll
    for (uint32_t i = 0; i < kSpans; ++i) {
        hp[i] = kEndPos;
        const bool wanted = i < spans && i >= span_lo && !__builtin_amdgcn_readlane((uint32_t)alone, i * kSpanBlocks);
        if (wanted && lane < 63u) {
            const uint32_t *at = &hdrs[__builtin_amdgcn_readlane(rs4.x, i * kSpanBlocks) + 1u + lane];
            hp[i] = kCoh ? load_coherent(at) : *at;
        }
    }
    // the spans lie one behind the other in the stream: the first one that needs bytes needs the earliest
    uint32_t tile_from = kEndPos, first_needing = 0;
#pragma unroll
    for (uint32_t i = kSpans; i-- > 0;) {
        const uint32_t fi = __builtin_amdgcn_readlane(from, i * kSpanBlocks);
        if (fi != kEndPos) tile_from = fi, first_needing = i;
    }
    if (tile_from != kEndPos) {
        const uint32_t values = min((spans - first_needing) * kSpanValues, n - (tile_first + first_needing * kSpanValues));
        // the estimate assumes bit-packed values throughout; where the next tile's bytes begin, if known, bounds
        // it (this tile's last value lies in the group before that byte or in the one that holds it)
        uint32_t bytes = ((values * w) >> 3) + (values >> 3) + 128u;
        if (next_from != kEndPos && next_from >= tile_from) bytes = min(bytes, next_from - tile_from + w + 16u);
        win.fill(rsrc, tile_from, bytes, lane);
    }
    for (uint32_t i = span_lo; i < spans; ++i) {
        uint32_t hpi = hp[0];
#pragma unroll
        for (uint32_t k = 1; k < kSpans; ++k) hpi = i == k ? hp[k] : hpi;
        Sink span_sink = sink.shifted(i * kSpanValues);  // (a span begins at a multiple of four positions: the same grid)
        expand_block<W, kCoh>(rsrc, win, tbl, cidx, resume_of(i), hpi, hdrs, end, n, tile_first + i * kSpanValues,
                              length_of(i), span_sink, lane);
        sink.absorb(span_sink);
    }
}

// One wave decodes tile `tile_first / kTileValues` of page `pg` (`rs4`: lane j holds the Resume entry of the
// tile's block j) with `win_lds` / `tbl` / `cidx` as its private LDS.
// `sink`: position 0 is the tile's first value.
template <bool kCoh, typename Sink>
__device__ __forceinline__ void expand_page_tile_into(const PageDev &pg, const u32x4 rs4, const uint32_t tile_first,
                                                      const uint32_t *__restrict__ run_pos, uint8_t *win_lds,
                                                      const uint32_t win_bytes, const uint32_t win_bit_mask, uint2 *tbl,
                                                      uint32_t *cidx, const uint32_t lane, Sink &sink,
                                                      const uint32_t next_from = kEndPos, const uint32_t span_lo = 0u,
                                                      const uint32_t span_hi = 0xffffffffu) {
    const uint32_t n = pg.num_values;
    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);

    Window win;
    win.lo = win.hi = 0;
    win.lds = win_lds;
    win.cap = win_bytes;
    win.bit_mask = win_bit_mask;
    if (lane < 2) tbl[kTableRuns + lane] = make_uint2(kNoRun, 0u);  // read by tbl[j + 1] only

    switch (pg.num_bits) {
#define PQ2_W(N)                                                                                                    \
    case N:                                                                                                         \
        if constexpr (N <= Sink::kMaxBits)                                                                          \
            expand_tile<N, kCoh>(rsrc, win, tbl, cidx, rs4, run_pos + pg.run_base, end, n, tile_first, sink, lane, \
                                 next_from, span_lo, span_hi);                                                       \
        break;
#ifdef PQ2_ONLY_W  // single-width build for reading the generated code
        PQ2_W(PQ2_ONLY_W)
#else
        PQ2_W(1) PQ2_W(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12) PQ2_W(13) PQ2_W(14) PQ2_W(15) PQ2_W(16)
        PQ2_W(17) PQ2_W(18) PQ2_W(19) PQ2_W(20) PQ2_W(21) PQ2_W(22) PQ2_W(23) PQ2_W(24)
        PQ2_W(25) PQ2_W(26) PQ2_W(27) PQ2_W(28) PQ2_W(29) PQ2_W(30) PQ2_W(31) PQ2_W(32)
#endif
#undef PQ2_W
        default: {  // width 0: zeros (by the wave that has the tile's first span)
            if (span_lo != 0u) break;
            const uint32_t count = min(kTileValues, n - tile_first);
            const u32x4 z = {0u, 0u, 0u, 0u};
            const uint32_t head = min((4u - sink.mis()) & 3u, count);
            if (lane < head) sink.put1(0u, (int32_t)lane);
            const uint32_t groups = (count - head) >> 2;
            for (uint32_t g = lane; g < groups; g += 64u) sink.put4(z, (int32_t)(head + 4u * g));
            const uint32_t tail = head + 4u * groups;
            if (lane < count - tail) sink.put1(0u, (int32_t)(tail + lane));
            break;
        }
    }
}
template <bool kCoh>
__device__ __forceinline__ void expand_page_tile(const PageDev &pg, const u32x4 rs4, const uint32_t tile_first,
                                                 const uint32_t *__restrict__ run_pos, uint8_t *win_lds,
                                                 const uint32_t win_bytes, const uint32_t win_bit_mask, uint2 *tbl,
                                                 uint32_t *cidx, const uint32_t lane,
                                                 const uint32_t next_from = kEndPos) {
    ExpandIndexSink sink{(out_u32 *)pg.out_addr + tile_first};
    expand_page_tile_into<kCoh>(pg, rs4, tile_first, run_pos, win_lds, win_bytes, win_bit_mask, tbl, cidx, lane, sink,
                                next_from);
}

// Selected rows: a page whose `reserved` is not 0 has one bit per tile from word `reserved - 1` of `tile_mask` on;
// tiles whose bit is clear hold no selected row and are not expanded (their values are never read).
__device__ __forceinline__ bool tile_wanted(const PageDev &pg, const uint32_t *__restrict__ tile_mask, uint32_t t) {
    if (pg.reserved == 0u || tile_mask == nullptr) return true;
    return ((tile_mask[pg.reserved - 1u + (t >> 5)] >> (t & 31u)) & 1u) != 0u;
}

#ifndef PQ2_EXPAND_WAVES
#define PQ2_EXPAND_WAVES 5
#endif
#ifndef PQ2_EXPAND_PER_SPAN_TILES
#define PQ2_EXPAND_PER_SPAN_TILES 8192  // launches of at most this many tiles give every span of a tile its own wave
#endif
// One workgroup = four waves = four consecutive tiles of the round (tiles [tile_start, tile_end) in the batch-wide
// numbering).  Dynamic LDS: four windows of `win_bytes` each, 32 spare bytes, four run tables, four chunk indexes.
__global__ __launch_bounds__(256, PQ2_EXPAND_WAVES) void expand_kernel(const PageDev *__restrict__ pages,
                                                       const uint32_t *__restrict__ tile_page,
                                                       const Resume *__restrict__ resume,
                                                       const uint32_t *__restrict__ run_pos,
                                                       uint32_t tile_start, uint32_t tile_end, uint32_t round,
                                                       uint32_t win_bytes, uint32_t win_bit_mask,
                                                       const uint32_t *__restrict__ tile_mask, uint32_t per_span) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // per_span: a wave for every span of a tile (batches of few tiles: more waves to hide each other's waits)
    constexpr uint32_t kSpans = kBlocksPerTile / kSpanBlocks;
    const uint32_t unit = blockIdx.x * 4u + wave;
    const uint32_t tile = tile_start + (per_span ? unit / kSpans : unit);
    const uint32_t span_lo = per_span ? unit % kSpans : 0u, span_hi = per_span ? span_lo + 1u : 0xffffffffu;
    if (tile >= tile_end) return;

    u32x4 rs4 = {0u, 0u, 0u, 0u};
    if (lane < kBlocksPerTile) rs4 = ((const u32x4 *)resume)[(size_t)tile * kBlocksPerTile + lane];
    const PageDev pg = pages[tile_page[tile]];
    const uint32_t tile_first = (pg.round_first(round) + (tile - pg.round_base(round))) * kTileValues;
    if (tile_first + span_lo * kSpanValues >= pg.num_values) return;
    if (!tile_wanted(pg, tile_mask, tile_first / kTileValues)) return;

    uint2 *tbl = (uint2 *)(lds + 4u * win_bytes + 32u) + wave * (kTableRuns + 2);
    uint32_t *cidx = (uint32_t *)(lds + 4u * win_bytes + 32u + 4u * (kTableRuns + 2) * 8u) + wave * kChunkSlots;
    ExpandIndexSink sink{(out_u32 *)pg.out_addr + tile_first};
    expand_page_tile_into<false>(pg, rs4, tile_first, run_pos, lds + wave * win_bytes, win_bytes, win_bit_mask, tbl, cidx,
                                 lane, sink, kEndPos, span_lo, span_hi);
}

// expand_kernel for pages of dictionary indices whose values are wanted: `pg.out_addr` is where the page's values
// go, `dicts[page]` names its dictionary (the fields of gather_kernel's descriptor; idx_addr unused), and a page
// with an index that is not below its dictionary's length gets status PQ2_OUT_OF_SPEC unless it has one already.
struct ExpandDictRef {  // = GatherPage (declared below, beside gather_kernel)
    uint64_t idx_addr, out_addr, dict_addr;
    uint32_t dict_len, num_values, first_tile, reserved;
};
template <int kWords>
__global__ __launch_bounds__(256, PQ2_EXPAND_WAVES) void expand_dict_kernel(const PageDev *__restrict__ pages,
                                                            const uint32_t *__restrict__ tile_page,
                                                            const Resume *__restrict__ resume,
                                                            const uint32_t *__restrict__ run_pos,
                                                            uint32_t tile_start, uint32_t tile_end, uint32_t round,
                                                            uint32_t win_bytes, uint32_t win_bit_mask,
                                                            const uint32_t *__restrict__ tile_mask,
                                                            const ExpandDictRef *__restrict__ dicts,
                                                            int32_t *__restrict__ status, uint32_t per_span) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kSpans = kBlocksPerTile / kSpanBlocks;
    const uint32_t unit = blockIdx.x * 4u + wave;
    const uint32_t tile = tile_start + (per_span ? unit / kSpans : unit);
    const uint32_t span_lo = per_span ? unit % kSpans : 0u, span_hi = per_span ? span_lo + 1u : 0xffffffffu;
    if (tile >= tile_end) return;

    u32x4 rs4 = {0u, 0u, 0u, 0u};
    if (lane < kBlocksPerTile) rs4 = ((const u32x4 *)resume)[(size_t)tile * kBlocksPerTile + lane];
    const uint32_t page = tile_page[tile];
    const PageDev pg = pages[page];
    const uint32_t tile_first = (pg.round_first(round) + (tile - pg.round_base(round))) * kTileValues;
    if (tile_first + span_lo * kSpanValues >= pg.num_values) return;
    if (!tile_wanted(pg, tile_mask, tile_first / kTileValues)) return;
    if (pg.num_bits > (uint32_t)ExpandDictSink<kWords>::kMaxBits) return;  // (the host sends no such batch here)

    uint2 *tbl = (uint2 *)(lds + 4u * win_bytes + 32u) + wave * (kTableRuns + 2);
    uint32_t *cidx = (uint32_t *)(lds + 4u * win_bytes + 32u + 4u * (kTableRuns + 2) * 8u) + wave * kChunkSlots;
    const ExpandDictRef ref = dicts[page];
    ExpandDictSink<kWords> sink{(out_u32 *)pg.out_addr + (size_t)tile_first * kWords,
                                (typename ExpandDictSink<kWords>::gu32 *)ref.dict_addr, ref.dict_len, false};
    expand_page_tile_into<false>(pg, rs4, tile_first, run_pos, lds + wave * win_bytes, win_bytes, win_bit_mask, tbl, cidx,
                                 lane, sink, kEndPos, span_lo, span_hi);
    if (__ballot(sink.bad) != 0ull && lane == 0u) atomicCAS(&status[page], 0, 1);  // PQ2_OUT_OF_SPEC
}

// ---------------------------------------------------------------------------
// decode_kernel: both steps in one workgroup per page
//
// Wave 0 walks the page's headers (parse_page) and publishes, in LDS, how many values the entries it has stored
// cover; waves 1..3 take the page's tiles in turn (tile t goes to wave 1 + t % 3), wait until the walk has
// passed the tile and expand it as expand_kernel does.  The walk is a chain of dependent steps that leaves the
// machine idle, and the expansion is memory traffic: running them side by side hides the first behind the
// second, and the stream bytes the expanding waves read were loaded by the walk a moment ago, so they come out
// of the cache and the stream crosses the memory bus once.  The waves of a workgroup are resident together, so
// the walk always advances and every wait ends: the walk publishes kAllParsed on every way out.
//
// Dynamic LDS: the walk's window (kWin), three stream windows of `win_bytes`, 32 spare bytes, three run
// tables, three chunk indexes, the progress word.
// ---------------------------------------------------------------------------
constexpr uint32_t kDecodeExpanders = 3;
constexpr uint32_t kMaxPolls = 1u << 24;  // of about 0.1 us each: far beyond the longest header walk of a page
#ifndef PQ2_LEAD_TILES
#define PQ2_LEAD_TILES 5
#endif
constexpr uint32_t kLeadTiles = PQ2_LEAD_TILES;  // tiles the walk may be ahead of the expansion (0: no limit)
constexpr uint32_t kCuTickets = 4096;     // entries of the per-compute-unit ticket table (see decode_kernel)
// Which wave of the workgroup walks the headers.  The walk issues an instruction whenever its SIMD lets it, and
// the workgroups that share a compute unit would all put wave 0 on the same SIMD, where the walks would take
// turns; so every workgroup draws a ticket from its compute unit's counter and gives the walk to the wave that
// sits on SIMD `ticket % 4`: four resident workgroups, four SIMDs.  (A wrong guess about the placement costs
// speed, nothing else: any wave can take either role.)
__device__ __forceinline__ uint32_t pick_walker(uint32_t *cu_tickets, uint32_t *scratch /* 8 words of LDS */,
                                                const uint32_t wave, const uint32_t lane) {
    const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const uint32_t xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
    if (lane == 0) scratch[wave] = (hw_id >> 4) & 3u;                   // SIMD_ID
    if (threadIdx.x == 0)  // CU_ID, SH_ID, SE_ID of this compute unit | accelerator die
        scratch[4] = atomicAdd(&cu_tickets[(((hw_id >> 8) & 0xffu) | (xcc_id << 8)) & (kCuTickets - 1u)], 1u);
    __syncthreads();
    const uint32_t target = scratch[4] & 3u;
    uint32_t walker = 0;
    for (uint32_t k = 4; k-- > 0;)
        if (scratch[k] == target) walker = k;
    return __builtin_amdgcn_readfirstlane(walker);
}

template <bool kWide>
__global__ __launch_bounds__(256, 4) void decode_kernel(const PageDev *__restrict__ pages, uint32_t n_pages,
                                                         Resume *__restrict__ resume, uint32_t *__restrict__ run_pos,
                                                         int32_t *__restrict__ status, ParseState *__restrict__ state,
                                                         uint32_t *__restrict__ cu_tickets, uint32_t win_bytes,
                                                         uint32_t win_bit_mask,
                                                         const uint32_t *__restrict__ tile_mask) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    uint8_t *const tables = lds + kWin + kDecodeExpanders * win_bytes + 32u;
    uint32_t *const parsed = (uint32_t *)(tables + kDecodeExpanders * ((kTableRuns + 2) * 8u + kChunkSlots * 4u));
    if (threadIdx.x < 4u) parsed[threadIdx.x] = 0u;  // the progress word; tiles done by each expanding wave
    const uint32_t walker = pick_walker(cu_tickets, parsed + 4, wave, lane);  // with a workgroup barrier inside

    if (wave == walker) {
        // (pages whose stream is small beside their values -- long RLE runs -- have nothing to keep in the cache)
        const bool throttled = pg.in_len / 512u >= (pg.num_values + kTileValues - 1u) / kTileValues;
        parse_page<kWide>(pg, page, lds, resume, nullptr, run_pos, status, state, 0u, lane, [&](uint32_t covered) {
            // release: the entries stored before are visible to whoever sees the new number
            if (lane == 0) __hip_atomic_store(parsed, covered, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // The walk does not run further ahead of the expansion than kLeadTiles tiles: the expanding waves read
            // the stream a second time, and what lies between the two reads of a byte (of all pages resident on
            // the GPU, with the values written meanwhile) has to fit the last-level cache for the second read not
            // to reach memory.  The wave that holds the lowest unfinished tile never waits for the walk here
            // (the walk is past that tile by at least one tile), so the wait ends; it is bounded all the same.
            if (kLeadTiles != 0u && throttled && covered != kAllParsed) {
                for (uint32_t polls = 0; polls < kMaxPolls; ++polls) {
                    uint32_t lowest = 0xffffffffu;  // the lowest tile some expanding wave has not finished
#pragma unroll
                    for (uint32_t k = 0; k < kDecodeExpanders; ++k)
                        lowest = min(lowest, k + kDecodeExpanders * __hip_atomic_load(parsed + 1u + k, __ATOMIC_RELAXED,
                                                                                       __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (covered / kTileValues <= lowest + kLeadTiles) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        });
        return;
    }
    const uint32_t e = wave - (wave > walker ? 1u : 0u);
    uint2 *tbl = (uint2 *)tables + e * (kTableRuns + 2);
    uint32_t *cidx = (uint32_t *)(tables + kDecodeExpanders * (kTableRuns + 2) * 8u) + e * kChunkSlots;
    const uint32_t n = pg.num_values;
    const uint32_t n_tiles = (n + kTileValues - 1u) / kTileValues;
    for (uint32_t t = e; t < n_tiles; t += kDecodeExpanders) {
        if (!tile_wanted(pg, tile_mask, t)) {  // done without a look: the walk counts on this number only
            if (lane == 0)
                __hip_atomic_store(parsed + 1u + e, t / kDecodeExpanders + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        const uint32_t need = min((t + 1u) * kTileValues, n);
        uint32_t polls = 0, covered;
        while ((covered = __hip_atomic_load(parsed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
            __builtin_amdgcn_s_sleep(4);
            if (++polls == kMaxPolls) {  // never seen; a wait that cannot end would hang the GPU
                if (lane == 0) status[page] = 6;  // PQ2_DEVICE_ERROR
                return;
            }
        }
        // the next tile's first Resume entry rides along (lane kBlocksPerTile) once the walk has covered that
        // tile's first value: it says where this tile's bytes end
        const bool next_known = t + 1u < n_tiles && covered > need;
        // the tile's four Resume entries, one per lane, past the vector cache (the walk wrote them just now)
        const uint32_t *rs = (const uint32_t *)(resume + (size_t)(pg.round_base(0) + t) * kBlocksPerTile);
        u32x4 rs4 = {0u, 0u, 0u, 0u};
        if (lane < kBlocksPerTile + (next_known ? 1u : 0u)) {
            rs4.x = load_coherent(rs + 4u * lane);
            rs4.y = load_coherent(rs + 4u * lane + 1u);
            rs4.z = load_coherent(rs + 4u * lane + 2u);
            rs4.w = load_coherent(rs + 4u * lane + 3u);
        }
        uint32_t next_from = kEndPos;
        if (next_known) {
            const uint32_t rem_kind = __builtin_amdgcn_readlane(rs4.y, kBlocksPerTile);
            const uint32_t payload = __builtin_amdgcn_readlane(rs4.z, kBlocksPerTile);
            const uint32_t skip = __builtin_amdgcn_readlane(rs4.w, kBlocksPerTile);
            // bit-packed: the group that holds the next tile's first value; RLE: the byte behind that run
            next_from = (rem_kind >> 31) == kBitpacked ? payload + (skip >> 3) * pg.num_bits : skip;
        }
        if (lane >= kBlocksPerTile) rs4 = u32x4{0u, 0u, 0u, 0u};
        expand_page_tile<true>(pg, rs4, t * kTileValues, run_pos, lds + kWin + e * win_bytes, win_bytes, win_bit_mask,
                               tbl, cidx, lane, next_from);
        // (no release: the walk only counts tiles with this, and a wider fence would wait for the stores to drain)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0)
            __hip_atomic_store(parsed + 1u + e, t / kDecodeExpanders + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ---------------------------------------------------------------------------
// bitunpack_kernel: `length` values of width w from a bit-packed buffer
// (bitpacked::Decoder<u32>, src/encoding/bitpacked/decode.rs:28-86).
// ---------------------------------------------------------------------------
// A wave takes the values in blocks of 2048: the 256 W bytes of a block (they begin at a whole byte) come into the
// wave's LDS with coalesced dwordx4 loads that are all in flight together -- every packed byte is requested once,
// whatever the width -- and every lane then takes 4 values per step out of LDS (fields4_at) and stores one dwordx4:
// 1 KiB of contiguous output per wave instruction.  A last block that is not full, and an output that is not
// 16-byte aligned, go value group by value group straight from memory (emit_bp).
constexpr uint32_t kUnpackBlock = 2048;
__host__ __device__ constexpr uint32_t unpack_pieces(uint32_t w) { return 16u * w + 2u; }  // of 16 bytes: the block, its misalignment, 8 bytes of slack
template <int W>
__device__ __forceinline__ void unpack_block_lds(const __amdgpu_buffer_rsrc_t rsrc, const uint32_t a, uint8_t *lds,
                                                 out_u32 *out, const uint32_t lane) {
    constexpr uint32_t kPieces = unpack_pieces((uint32_t)W);
    constexpr uint32_t kRounds = (kPieces + 63u) / 64u;
    const uint32_t a0 = a & ~15u;
    u32x4 v[kRounds];
#pragma unroll
    for (uint32_t i = 0; i < kRounds; ++i)
        if (lane + 64u * i < kPieces) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a0 + (lane + 64u * i) * 16u, 0, 0);
#pragma unroll
    for (uint32_t i = 0; i < kRounds; ++i)
        if (lane + 64u * i < kPieces) *(u32x4 *)(lds + (lane + 64u * i) * 16u) = v[i];
    wave_sync();
    const uint32_t bit0 = (a - a0) * 8u;
#pragma unroll 2
    for (uint32_t s = 0; s < kUnpackBlock / 256u; ++s) {
        const uint32_t p = (s * 64u + lane) * 4u;
        store4(fields4_at<W>(lds, bit0 + p * (uint32_t)W), out + p);
    }
    wave_sync();  // the next block's bytes come into the same LDS
}

template <int W>
__global__ __launch_bounds__(256, 6) void bitunpack_kernel(uint64_t in_addr, uint32_t in_len, uint32_t length,
                                                           uint32_t *__restrict__ out, uint32_t wave_lds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t unpack_lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t delta = (uint32_t)(in_addr & 15u);
    const uint32_t end = delta + in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(in_addr - delta, (end + 3u) & ~3u);
    const uint32_t n_blocks = (length + kUnpackBlock - 1) / kUnpackBlock;
    const bool aligned = ((uintptr_t)out & 15u) == 0u;
    uint8_t *const lds = unpack_lds + wave * wave_lds;
    for (uint32_t block = blockIdx.x * 4u + wave; block < n_blocks; block += gridDim.x * 4u) {
        const uint32_t start = block * kUnpackBlock;
        const uint32_t cnt = min(kUnpackBlock, length - start);
        out_u32 *const o = (out_u32 *)(out + start);
        if (cnt < kUnpackBlock || !aligned)
            emit_bp<W>(rsrc, delta, start, o, cnt, lane);
        else
            unpack_block_lds<W>(rsrc, delta + block * (kUnpackBlock / 8u) * (uint32_t)W, lds, o, lane);
    }
}

}  // namespace pq2

namespace pq2 {

// ---------------------------------------------------------------------------
// gather_kernel: dictionary indices -> dictionary values (the consumer loop of
// a dictionary-encoded page: tests/it/read/primitive.rs:99-108, with
// PrimitivePageDict::value, tests/it/read/dictionary/primitive.rs:20-27: an
// index that is not below the dictionary's length is an error of the page).
// One workgroup per 4096 values of one page; kWords dwords per value.
// ---------------------------------------------------------------------------
struct GatherPage {  // must match the host's
    uint64_t idx_addr;   // the page's decoded indices (u32)
    uint64_t out_addr;   // the page's values
    uint64_t dict_addr;  // the page's dictionary
    uint32_t dict_len;
    uint32_t num_values;
    uint32_t first_tile;  // batch-wide number of the page's first 4096-value tile
    uint32_t reserved;
};

static_assert(sizeof(GatherPage) == sizeof(ExpandDictRef), "expand_dict_kernel reads the same descriptors");

constexpr uint32_t kGatherTileValues = 65536;  // values of one page per workgroup of gather_kernel

// kLds: the page's dictionary (at most `lds_bytes`) is copied into LDS first and the lookups go there: a wave's
// 64 lookups at unrelated addresses take the vector cache one address at a time, LDS serves them by bank.
template <int kWords, bool kLds>
__global__ __launch_bounds__(256) void gather_kernel(const GatherPage *__restrict__ pages, uint32_t n_pages,
                                                     int32_t *__restrict__ status, uint32_t lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // the last page whose first tile is not behind this one (pages without values share their successor's)
    const uint32_t tile = blockIdx.x;
    uint32_t lo = 0, hi = n_pages;  // pages[lo].first_tile <= tile < pages[hi].first_tile
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pages[mid].first_tile <= tile) lo = mid; else hi = mid;
    }
    const GatherPage pg = pages[lo];
    const uint32_t first = (tile - pg.first_tile) * kGatherTileValues;
    if (first >= pg.num_values) return;
    const uint32_t count = min(kGatherTileValues, pg.num_values - first);
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    gu32 *const idx = (gu32 *)pg.idx_addr + first;
    gu32 *const dict = (gu32 *)pg.dict_addr;
    out_u32 *const out = (out_u32 *)pg.out_addr + (size_t)first * kWords;
    const bool in_lds = kLds && (size_t)pg.dict_len * kWords * 4u <= lds_bytes;  // the same for the whole workgroup
    if (in_lds) {
        const uint32_t words = pg.dict_len * kWords;
        for (uint32_t i = threadIdx.x; i < words; i += 256u) ((uint32_t *)lds)[i] = dict[i];
        __syncthreads();
    }
    bool bad = false;
    auto value_word = [&](uint32_t j, bool ok, int c) -> uint32_t {
        if (!ok) return 0u;
        return in_lds ? ((const uint32_t *)lds)[j * kWords + c] : dict[(size_t)j * kWords + c];
    };
    // Four values per thread and step where the page's indices and values are 16-byte aligned (every load and
    // store a dwordx4); otherwise, and for the last values of the page, one value per thread.
    uint32_t done = 0;
    if (((pg.idx_addr | pg.out_addr) & 15u) == 0u) {
        const uint32_t quads = count >> 2;
#pragma unroll 2
        for (uint32_t q = threadIdx.x; q < quads; q += 256u) {
            const u32x4 j = ((gu32x4 *)idx)[q];
            const bool ok[4] = {j.x < pg.dict_len, j.y < pg.dict_len, j.z < pg.dict_len, j.w < pg.dict_len};
            bad |= !(ok[0] & ok[1] & ok[2] & ok[3]);
            const uint32_t jj[4] = {j.x, j.y, j.z, j.w};
            uint32_t v[4 * kWords];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < kWords; ++c) v[k * kWords + c] = value_word(jj[k], ok[k], c);
#pragma unroll
            for (int c = 0; c < kWords; ++c) {
                const u32x4 o = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
                *((out_u32x4 *)out + (size_t)q * kWords + c) = o;
            }
        }
        done = quads << 2;
    }
    for (uint32_t i = done + threadIdx.x; i < count; i += 256u) {
        const uint32_t j = idx[i];
        const bool ok = j < pg.dict_len;
        bad |= !ok;
#pragma unroll
        for (int c = 0; c < kWords; ++c) out[(size_t)i * kWords + c] = value_word(j, ok, c);
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63u) == 0u) atomicCAS(&status[lo], 0, 1);  // PQ2_OUT_OF_SPEC
}

// ---------------------------------------------------------------------------
// optional_kernel: definition levels + dictionary indices -> nullable values
// (deserialize_levels, tests/it/read/utils.rs:52-66, over an OptionalDictionary
// page, tests/it/read/primitive.rs:104-110): row r is valid where its level
// equals the maximum, and the k-th valid row takes the dictionary value of the
// page's k-th index.  One workgroup per page walks the page's rows 4096 at a
// time (four quads of four consecutive rows per thread, all of their loads in
// flight together), carrying the count of valid rows: a prefix sum over the
// lanes of a wave, the wave totals through LDS, one barrier per step.
// ---------------------------------------------------------------------------
// Which page segment `seg` belongs to: the last one whose first segment is not behind it (pages without rows share
// their successor's).
__device__ __forceinline__ uint32_t optional_page_of(const OptionalPage *__restrict__ pages, uint32_t n_pages, uint32_t seg) {
    uint32_t lo = 0, hi = n_pages;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pages[mid].first_seg <= seg) lo = mid; else hi = mid;
    }
    return lo;
}

// A batch of few pages is cut into segments of `seg_rows` rows (a multiple of 4096), one workgroup each, so that a
// few large pages still fill the GPU.  A segment has to know how many valid rows lie before it in its page: this
// kernel counts the valid rows of every segment.  The workgroup of a page's first segment also takes over the index
// stream's status where the page's own is 0 (before any lookup can report an index out of range).
__global__ __launch_bounds__(256) void optional_count_kernel(const OptionalPage *__restrict__ pages, uint32_t n_pages,
                                                             uint32_t seg_rows, uint32_t *__restrict__ seg_valid,
                                                             int32_t *__restrict__ status) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t seg = blockIdx.x;
    const uint32_t page = optional_page_of(pages, n_pages, seg);
    const OptionalPage pg = pages[page];
    const uint32_t first = (seg - pg.first_seg) * seg_rows;
    if (first == 0u && threadIdx.x == 0 && status[page] == 0) status[page] = status[pg.idx_status_at];
    if (first >= pg.rows) {  // (a page without rows has one segment all the same)
        if (threadIdx.x == 0) seg_valid[seg] = 0u;
        return;
    }
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    gu32 *const level = (gu32 *)pg.level_addr;  // 16-byte aligned
    const uint32_t last = min(pg.rows, first + seg_rows);
    uint32_t count = 0;
    for (uint32_t r = first + 4u * threadIdx.x; r < last; r += 1024u) {
        if (r + 3u < last) {
            const u32x4 lv = ((gu32x4 *)level)[r >> 2];
            count += (lv.x == pg.max_level) + (lv.y == pg.max_level) + (lv.z == pg.max_level) + (lv.w == pg.max_level);
        } else {
            for (uint32_t k = r; k < last; ++k) count += level[k] == pg.max_level;
        }
    }
    count = wave_inclusive_sum(count);
    if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = count;
    __syncthreads();
    if (threadIdx.x == 0) seg_valid[seg] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// seg_rows = 0: one workgroup per page (blockIdx.x = page); else one per segment of seg_rows rows, `seg_valid` from
// optional_count_kernel.
template <int kWords>
__global__ __launch_bounds__(256) void optional_kernel(const OptionalPage *__restrict__ pages, uint32_t n_pages,
                                                       int32_t *__restrict__ status, uint32_t seg_rows,
                                                       const uint32_t *__restrict__ seg_valid) {
    constexpr uint32_t kQuads = 4;                    // quads of rows per thread and step
    constexpr uint32_t kStepRows = 1024u * kQuads;    // rows of a page per step: quad q of thread t = rows q * 1024 + 4 t ...
    __shared__ uint32_t wave_total[2][kQuads][4];
    __shared__ uint32_t valid_before;
    const uint32_t page = seg_rows == 0u ? blockIdx.x : optional_page_of(pages, n_pages, blockIdx.x);
    if (page >= n_pages) return;
    const OptionalPage pg = pages[page];
    // the rows of this workgroup: [row_first, row_last)
    const uint32_t row_first = seg_rows == 0u ? 0u : (blockIdx.x - pg.first_seg) * seg_rows;
    if (seg_rows != 0u && row_first >= pg.rows) return;
    const uint32_t row_last = seg_rows == 0u ? pg.rows : min(pg.rows, row_first + seg_rows);
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    gu32 *const level = (gu32 *)pg.level_addr;
    gu32 *const idx = (gu32 *)pg.idx_addr;
    gu32 *const dict = (gu32 *)pg.dict_addr;
    out_u32 *const out = (out_u32 *)pg.out_addr;
    __attribute__((address_space(1))) uint8_t *const validity =
        (__attribute__((address_space(1))) uint8_t *)pg.validity_addr;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t used = 0;  // valid rows before this step = indices consumed
    if (seg_rows != 0u && row_first != 0u) {  // those of the page's segments before this one
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (uint32_t sg = pg.first_seg; sg < blockIdx.x; ++sg) sum += seg_valid[sg];
            valid_before = sum;
        }
        __syncthreads();
        used = valid_before;
    }
    bool bad = false;
    uint32_t flip = 0;
    // the four levels of rows r0 .. r0 + 3 (rows behind the page's end: a level that is not the maximum)
    auto load_levels = [&](uint32_t r0) {
        u32x4 lv = {~pg.max_level, ~pg.max_level, ~pg.max_level, ~pg.max_level};
        if (r0 + 3u < pg.rows) {
            lv = ((gu32x4 *)level)[r0 >> 2];
        } else {
            if (r0 < pg.rows) lv.x = level[r0];
            if (r0 + 1u < pg.rows) lv.y = level[r0 + 1u];
            if (r0 + 2u < pg.rows) lv.z = level[r0 + 2u];
        }
        return lv;
    };
    // whole quads of rows are written with one store each where the page's values and validity bytes are aligned
    const bool aligned = ((pg.out_addr & 15u) | (pg.validity_addr & 3u)) == 0u;
    u32x4 lv_next[kQuads];
#pragma unroll
    for (uint32_t q = 0; q < kQuads; ++q) lv_next[q] = load_levels(row_first + q * 1024u + 4u * threadIdx.x);
    for (uint32_t base = row_first; base < row_last; base += kStepRows, flip ^= 1u) {
        uint32_t r0[kQuads], c[kQuads], incl[kQuads];
        bool v[kQuads][4];
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q) {
            r0[q] = base + q * 1024u + 4u * threadIdx.x;
            const u32x4 lv = lv_next[q];
            lv_next[q] = load_levels(r0[q] + kStepRows);  // the next step's levels are on their way while this step works
            const uint32_t l4[4] = {lv.x, lv.y, lv.z, lv.w};
            c[q] = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                v[q][k] = r0[q] + k < pg.rows && l4[k] == pg.max_level;
                c[q] += v[q][k] ? 1u : 0u;
            }
            incl[q] = wave_inclusive_sum(c[q]);
            if (lane == 63u) wave_total[flip][q][wave] = incl[q];
        }
        __syncthreads();  // (the other buffer is written in the next step, this one again only after that step's barrier)
        bool present[kQuads][4];
        uint32_t j[kQuads][4];
        uint32_t before = used;  // valid rows before the quads q of all threads
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q) {
            uint32_t below = 0, all = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t t = wave_total[flip][q][k];
                below += k < wave ? t : 0u;
                all += t;
            }
            uint32_t rank = before + below + incl[q] - c[q];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                present[q][k] = v[q][k] && rank < pg.idx_values;  // behind the last index: values.next() is None
                j[q][k] = 0;
                if (present[q][k]) j[q][k] = idx[rank];
                rank += v[q][k] ? 1u : 0u;
            }
            before += all;
        }
        used = before;
        uint32_t val[kQuads][4][kWords];
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q)
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const bool ok = j[q][k] < pg.dict_len;
                bad |= present[q][k] && !ok;
#pragma unroll
                for (int w = 0; w < kWords; ++w) val[q][k][w] = 0u;
                if (present[q][k] && ok) {
#pragma unroll
                    for (int w = 0; w < kWords; ++w) val[q][k][w] = dict[(size_t)j[q][k] * kWords + w];
                }
            }
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q) {
            if (aligned && r0[q] + 3u < pg.rows) {
                uint32_t flat[4 * kWords];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
                    for (int w = 0; w < kWords; ++w) flat[k * kWords + w] = val[q][k][w];
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    const u32x4 o = {flat[4 * w], flat[4 * w + 1], flat[4 * w + 2], flat[4 * w + 3]};
                    *((out_u32x4 *)out + (size_t)(r0[q] >> 2) * kWords + w) = o;
                }
                *(__attribute__((address_space(1))) uint32_t *)(validity + r0[q]) =
                    (present[q][0] ? 1u : 0u) | (present[q][1] ? 0x100u : 0u) | (present[q][2] ? 0x10000u : 0u) |
                    (present[q][3] ? 0x1000000u : 0u);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    if (r0[q] + k >= pg.rows) break;
#pragma unroll
                    for (int w = 0; w < kWords; ++w) out[(size_t)(r0[q] + k) * kWords + w] = val[q][k][w];
                    validity[r0[q] + k] = present[q][k] ? 1 : 0;
                }
            }
        }
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (seg_rows != 0u) {  // (optional_count_kernel has merged the index stream's status)
        if (threadIdx.x == 0 && any_bad) atomicCAS(&status[page], 0, 1);  // PQ2_OUT_OF_SPEC
        return;
    }
    if (threadIdx.x == 0 && status[page] == 0) {
        const int32_t of_indices = status[pg.idx_status_at];
        status[page] = of_indices != 0 ? of_indices : (any_bad ? 1 : 0);  // PQ2_OUT_OF_SPEC
    }
}

// ---------------------------------------------------------------------------
// select_kernel: copies the selected row ranges of decoded pages to their places
// (SliceFilteredIter, src/deserialize/utils.rs:92-148).  One workgroup per piece
// of at most 16384 values of one interval.
// ---------------------------------------------------------------------------
struct SelectPiece {  // must match the host's
    uint64_t src_addr, dst_addr;  // u32 values
    uint32_t count, reserved;
};

__global__ __launch_bounds__(256) void select_kernel(const SelectPiece *__restrict__ pieces) {
    const SelectPiece pc = pieces[blockIdx.x];
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    gu32 *const src = (gu32 *)pc.src_addr;
    out_u32 *const dst = (out_u32 *)pc.dst_addr;
    uint32_t done = 0;
    if (((pc.src_addr | pc.dst_addr) & 15u) == 0u) {
        const uint32_t quads = pc.count >> 2;
        for (uint32_t q = threadIdx.x; q < quads; q += 256u) ((out_u32x4 *)dst)[q] = ((gu32x4 *)src)[q];
        done = quads << 2;
    }
    for (uint32_t i = done + threadIdx.x; i < pc.count; i += 256u) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// filtered_kernel: the definition levels of selected rows as validity bytes, and the counts of valid rows in and
// between the selected intervals (FilteredHybridBitmapIter, src/deserialize/filtered_rle.rs:53-274: its Bitmap /
// Repeated items are the bytes, its Skipped(set_count) items the counts).  One workgroup per piece of at most
// 16384 rows of a decoded page: a piece of an interval writes its rows' bytes and adds its valid rows to the
// interval's second counter, a piece of the rows in front of an interval only counts, into the first.
// ---------------------------------------------------------------------------
struct FilterPiece {  // must match the host's
    uint64_t src_addr;  // the piece's first level (u32)
    uint64_t dst_addr;  // where its validity bytes go; 0: rows that are skipped
    uint32_t count;
    uint32_t counter;   // which word of the counts its valid rows are added to
};

__global__ __launch_bounds__(256) void filtered_kernel(const FilterPiece *__restrict__ pieces, uint32_t *__restrict__ counts) {
    __shared__ uint32_t wave_sum[4];
    const FilterPiece pc = pieces[blockIdx.x];
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    typedef __attribute__((address_space(1))) uint8_t gu8;
    gu32 *const src = (gu32 *)pc.src_addr;
    gu8 *const dst = (gu8 *)pc.dst_addr;
    uint32_t valid = 0, done = 0;
    if ((pc.src_addr & 15u) == 0u && (pc.dst_addr & 3u) == 0u) {  // four rows a thread: one load, one store
        const uint32_t quads = pc.count >> 2;
        for (uint32_t q = threadIdx.x; q < quads; q += 256u) {
            const u32x4 lv = ((gu32x4 *)src)[q];
            const uint32_t b = (lv.x == 1u ? 1u : 0u) | (lv.y == 1u ? 0x100u : 0u) | (lv.z == 1u ? 0x10000u : 0u) |
                               (lv.w == 1u ? 0x1000000u : 0u);
            valid += (uint32_t)__popc(b);
            if (dst) *(__attribute__((address_space(1))) uint32_t *)(dst + 4u * q) = b;
        }
        done = quads << 2;
    }
    for (uint32_t i = done + threadIdx.x; i < pc.count; i += 256u) {
        const uint32_t v = src[i] == 1u ? 1u : 0u;
        valid += v;
        if (dst) dst[i] = (uint8_t)v;
    }
    valid = wave_inclusive_sum(valid);
    if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = valid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t all = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        if (all) atomicAdd(&counts[pc.counter], all);
    }
}

// ---------------------------------------------------------------------------
// front_kernel: split_buffer_v1 (src/page/mod.rs:350-398) and the width byte of
// dictionary indices (src/deserialize/utils.rs:13-27) for page buffers that lie
// in device memory.  The host has built three stream descriptors per page
// (repetition levels, definition levels, indices) that span the whole page
// buffer; one thread per page reads the length prefixes and writes the streams'
// real address, length and, for the indices, bit width into them before the
// decode kernels read them.  A page that fails keeps its three streams empty
// (they decode to zeros) and gets its status here.
// ---------------------------------------------------------------------------
struct RawPageDev {  // must match the host's
    uint64_t buf_addr;
    uint32_t buf_len;
    uint32_t what;  // bit 0: repetition levels, bit 1: definition levels, bit 2: dictionary indices; bits 8..15: the
                    // largest index width the batch was set up for
};

__global__ __launch_bounds__(256) void front_kernel(PageDev *__restrict__ streams, const RawPageDev *__restrict__ raw,
                                                    uint32_t n_pages, int32_t *__restrict__ front_status) {
    const uint32_t page = blockIdx.x * 256u + threadIdx.x;
    if (page >= n_pages) return;
    const RawPageDev rp = raw[page];
    const uint8_t *const buf = (const uint8_t *)rp.buf_addr;
    PageDev *const st = streams + 3u * (size_t)page;
    uint32_t cursor = 0;
    int32_t code = 0;
    uint64_t addr[3] = {0, 0, 0};
    uint32_t len[3] = {0, 0, 0};
    uint32_t width = st[2].num_bits;
    for (uint32_t level = 0; level < 2u && code == 0; ++level) {
        if (((rp.what >> level) & 1u) == 0u) continue;
        if (rp.buf_len - cursor < 4u) {  // get_length: None (src/encoding/mod.rs:14-21)
            code = 1;
            break;
        }
        const uint32_t l = (uint32_t)buf[cursor] | ((uint32_t)buf[cursor + 1u] << 8) | ((uint32_t)buf[cursor + 2u] << 16) |
                           ((uint32_t)buf[cursor + 3u] << 24);
        if (l > rp.buf_len - cursor - 4u) {  // buffer.get(4..4 + length): None
            code = 1;
            break;
        }
        addr[level] = rp.buf_addr + cursor + 4u;
        len[level] = l;
        cursor += 4u + l;
    }
    if (code == 0 && (rp.what & 4u) != 0u) {
        if (rp.buf_len == cursor) {
            code = 5;  // indices_buffer[0] on an empty buffer
        } else {
            width = buf[cursor];
            if (width > 32u) code = 1;
            else if (width > ((rp.what >> 8) & 0xffu)) code = 2;
            addr[2] = rp.buf_addr + cursor + 1u;
            len[2] = rp.buf_len - cursor - 1u;
        }
    }
    front_status[page] = code;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        st[k].in_addr = code == 0 ? addr[k] : 0ull;
        st[k].in_len = code == 0 ? len[k] : 0u;
    }
    if (code == 0) st[2].num_bits = width;
}

// ---------------------------------------------------------------------------
// bitunpack64_kernel: `length` values of width w (1..64) from a bit-packed
// buffer into u64 (bitpacked::Decoder<u64>, decode.rs:28-86; unpack64,
// unpack.rs:21-95: LSB-first fields of a little-endian byte stream).  As
// bitunpack_kernel: a wave takes blocks of 1024 values, whose 128 w bytes come
// into its LDS with coalesced dwordx4 loads that are all in flight together;
// every lane then takes two values per step out of LDS (three aligned dwords,
// two funnel shifts, a mask each) and stores them as one dwordx4.  A last block
// that is not full, and an output that is not 16-byte aligned, go one value per
// lane straight from memory.
// ---------------------------------------------------------------------------
constexpr uint32_t kUnpack64Block = 1024;
__host__ __device__ constexpr uint32_t unpack64_pieces(uint32_t w) { return 8u * w + 2u; }  // of 16 bytes: the block, its misalignment, slack
constexpr uint32_t kUnpack64Rounds = (unpack64_pieces(64u) + 63u) / 64u;

__device__ __forceinline__ uint64_t field64(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t sh, uint64_t mask) {
    const uint32_t lo = __builtin_amdgcn_alignbit(x1, x0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbit(x2, x1, sh);
    return (((uint64_t)hi << 32) | lo) & mask;
}

__global__ __launch_bounds__(256, 6) void bitunpack64_kernel(uint64_t in_addr, uint32_t in_len, uint32_t w,
                                                             uint32_t length, uint64_t *__restrict__ out,
                                                             uint32_t wave_lds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t unpack64_lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t delta = (uint32_t)(in_addr & 15u);
    const uint32_t end = delta + in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(in_addr - delta, (end + 3u) & ~3u);
    const uint64_t mask = w == 64u ? ~0ull : ((1ull << w) - 1ull);
    const uint32_t n_blocks = (length + kUnpack64Block - 1u) / kUnpack64Block;
    const bool aligned = ((uintptr_t)out & 15u) == 0u;
    uint8_t *const lds = unpack64_lds + wave * wave_lds;
    const uint32_t pieces = unpack64_pieces(w);
    for (uint32_t block = blockIdx.x * 4u + wave; block < n_blocks; block += gridDim.x * 4u) {
        const uint32_t start = block * kUnpack64Block;
        const uint32_t cnt = min(kUnpack64Block, length - start);
        if (cnt < kUnpack64Block || !aligned) {
            for (uint32_t i = start + lane; i < start + cnt; i += 64u) {
                const uint64_t bit = (uint64_t)delta * 8u + (uint64_t)i * w;
                const uint32_t d = (uint32_t)(bit >> 5) * 4u;  // byte offset of the first dword (the buffer is < 4 GiB)
                out[i] = field64(__builtin_amdgcn_raw_buffer_load_b32(rsrc, d, 0, 0),
                                 __builtin_amdgcn_raw_buffer_load_b32(rsrc, d + 4u, 0, 0),
                                 __builtin_amdgcn_raw_buffer_load_b32(rsrc, d + 8u, 0, 0), (uint32_t)bit & 31u, mask);
            }
            continue;
        }
        const uint32_t a = delta + block * (kUnpack64Block / 8u) * w;  // the block's first byte
        const uint32_t a0 = a & ~15u;
        u32x4 v[kUnpack64Rounds];
#pragma unroll
        for (uint32_t i = 0; i < kUnpack64Rounds; ++i)
            if (lane + 64u * i < pieces) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a0 + (lane + 64u * i) * 16u, 0, 0);
#pragma unroll
        for (uint32_t i = 0; i < kUnpack64Rounds; ++i)
            if (lane + 64u * i < pieces) *(u32x4 *)(lds + (lane + 64u * i) * 16u) = v[i];
        wave_sync();
        const uint32_t bit0 = (a - a0) * 8u;
        typedef __attribute__((address_space(1))) u32x4 g_u32x4;
#pragma unroll 2
        for (uint32_t s = 0; s < kUnpack64Block / 128u; ++s) {
            const uint32_t p = (s * 64u + lane) * 2u;
            uint64_t f[2];
#pragma unroll
            for (uint32_t k = 0; k < 2u; ++k) {
                const uint32_t bit = bit0 + (p + k) * w;
                const uint32_t *src = (const uint32_t *)(lds + ((bit >> 3) & ~3u));
                f[k] = field64(src[0], src[1], src[2], bit & 31u, mask);
            }
            const u32x4 o = {(uint32_t)f[0], (uint32_t)(f[0] >> 32), (uint32_t)f[1], (uint32_t)(f[1] >> 32)};
            *(g_u32x4 *)(out + start + p) = o;
        }
        wave_sync();  // the next block's bytes come into the same LDS
    }
}

// ---------------------------------------------------------------------------
// delta_kernel: DELTA_BINARY_PACKED streams into i64 (delta_bitpacked::Decoder,
// src/encoding/delta_bitpacked/decoder.rs:121-244, over Block, :9-118,
// zigzag_leb128.rs:5-8 and uleb128.rs:3-23).  One wave per stream.  The blocks
// of a stream follow each other (where one begins is known only once the widths
// of the one before are read), the inside of a block is lane-parallel: up to 64
// mini-blocks have their widths read, their byte ranges summed and checked in
// one step, then every lane takes one value of the block, adds the block's
// minimum delta and a 64-bit wave prefix sum turns the deltas into values.
// ---------------------------------------------------------------------------
struct DeltaPage {
    uint64_t in_addr;
    uint64_t out_addr;  // int64 values
    uint32_t in_len;
    uint32_t cap;  // values the output holds
};
struct DeltaResult {
    int32_t status;
    uint32_t consumed;    // Decoder::consumed_bytes() once every value is taken (decoder.rs:182-185)
    uint64_t num_values;  // total_count of the header
};

#define PQ2_DPP64(V, CTRL, ROWS)                                                                                \
    (((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)((V) >> 32), CTRL, ROWS, 0xf, false) << 32) | \
     (uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)(V), CTRL, ROWS, 0xf, false))
// wave_inclusive_sum for 64-bit terms (the sums wrap, as the reference's wrapping adds do)
__device__ __forceinline__ uint64_t wave_inclusive_sum64(uint64_t v) {
    v += PQ2_DPP64(v, 0x111, 0xf);
    v += PQ2_DPP64(v, 0x112, 0xf);
    v += PQ2_DPP64(v, 0x114, 0xf);
    v += PQ2_DPP64(v, 0x118, 0xf);
    v += PQ2_DPP64(v, 0x142, 0xa);
    v += PQ2_DPP64(v, 0x143, 0xc);
    return v;
}
#undef PQ2_DPP64

__device__ __forceinline__ uint64_t lane_value64(uint64_t v, uint32_t lane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((uint32_t)v, lane);
}

struct Varint {
    uint64_t value;
    uint32_t used;
    uint32_t panics;
};
// uleb128::decode (uleb128.rs:3-23) on a window that holds one stream byte per lane (bit 8 set: behind the
// stream's end), from lane `o` on; o + 9 < 64.
__device__ __forceinline__ Varint varint_at(uint32_t window, uint32_t o) {
    Varint r = {0ull, 0u, 0u};
    for (uint32_t i = 0; i < 10u; ++i) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane(window, o + i);
        if (b & 0x100u) break;
        r.used += 1u;
        if (i == 9u && (b & 0xffu) > 1u) {  // `shift == 63 && *byte > 1`
            r.panics = 1u;
            break;
        }
        r.value |= (uint64_t)(b & 0x7fu) << (7u * i);
        if ((b & 0x80u) == 0u) break;
    }
    return r;
}
__device__ __forceinline__ uint64_t zigzag(uint64_t u) { return (u >> 1) ^ (0ull - (u & 1ull)); }

// A stream behind its header, as every lane of a wave holds it.
struct DeltaStream {
    __amdgpu_buffer_rsrc_t rsrc;  // the stream from its 16-byte aligned base on ("a-offsets")
    uint32_t delta, end;          // a-offsets of the first byte and behind the last one
    uint64_t *out;
    uint64_t num_mini_blocks, per_mini, block_most;  // block_most: values of a whole block (<= block_size)
    uint32_t per_mini32;                             // per_mini, capped: value indices stay below 2^31
    uint32_t block_size_small;                       // block_size where it is at most 4096, else 0
    // where the decode stands
    uint32_t pos;        // a-offset of the next block
    uint32_t remaining;  // deltas still to take
    uint32_t done;       // deltas taken
    uint32_t empties;
    uint64_t acc;        // the last value
};

__device__ __forceinline__ uint32_t delta_byte_at(const DeltaStream &s, uint32_t a) {
    return (__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, a & ~3u, 0, 0) >> ((a & 3u) * 8u)) & 0xffu;
}
// one stream byte per lane from `a` on; bit 8 set behind the stream's end
__device__ __forceinline__ uint32_t delta_window_at(const DeltaStream &s, uint32_t a, uint32_t lane) {
    return a + lane < s.end ? delta_byte_at(s, a + lane) : 0x100u;
}

// Decoder::try_new, decoder.rs:136-180.  Returns the stream's status, or -1 with `s` set up where blocks follow.
__device__ int32_t delta_open(const DeltaPage pg, uint32_t lane, bool store_first, DeltaStream &s,
                              uint64_t &num_values) {
    s.delta = (uint32_t)(pg.in_addr & 15u);
    s.end = s.delta + pg.in_len;
    s.rsrc = make_rsrc(pg.in_addr - s.delta, (s.end + 3u) & ~3u);
    s.out = (uint64_t *)pg.out_addr;
    const uint32_t window = delta_window_at(s, s.delta, lane);
    const Varint block_size = varint_at(window, 0u);
    if (block_size.panics || (block_size.value & 127ull) != 0ull) return 5;  // assert_eq!(block_size % 128, 0)
    uint32_t o = block_size.used;
    const Varint num_mini_blocks = varint_at(window, o);
    if (num_mini_blocks.panics) return 5;
    o += num_mini_blocks.used;
    const Varint total_count = varint_at(window, o);
    if (total_count.panics) return 5;
    o += total_count.used;
    const Varint first = varint_at(window, o);
    if (first.panics) return 5;
    o += first.used;
    if (num_mini_blocks.value == 0ull) return 5;  // block_size / num_mini_blocks
    s.num_mini_blocks = num_mini_blocks.value;
    s.per_mini = block_size.value / num_mini_blocks.value;
    if ((s.per_mini & 7ull) != 0ull) return 5;  // assert_eq!(values_per_mini_block % 8, 0)
    num_values = total_count.value;
    if (total_count.value > pg.cap) return 2;
    s.pos = s.delta + o;
    s.acc = zigzag(first.value);
    if (total_count.value != 0ull && lane == 0u && store_first) s.out[0] = s.acc;
    s.remaining = total_count.value != 0ull ? (uint32_t)total_count.value - 1u : 0u;
    s.done = 0u, s.empties = 0u;
    s.block_most = s.num_mini_blocks * s.per_mini;
    s.per_mini32 = s.per_mini > 0x80000000ull ? 0x80000000u : (uint32_t)s.per_mini;
    s.block_size_small = block_size.value <= 4096ull ? (uint32_t)block_size.value : 0u;
    return s.remaining == 0u ? 0 : -1;
}

// The blocks of a stream from where `s` stands, one after the other, by one wave; returns the status.  This is
// the path for every shape of stream and the one that finds what is wrong with a stream.  `tbl_start` /
// `tbl_bits`: 64 LDS words each.
__device__ int32_t delta_blocks(DeltaStream &s, uint32_t lane, uint32_t *tbl_start, uint32_t *tbl_bits) {
    const uint32_t end = s.end;
    const uint32_t per_mini32 = s.per_mini32;
    // a mini-block's index from a value's index: a shift where the mini-blocks hold a power of two of values
    const bool minis_pow2 = (per_mini32 & (per_mini32 - 1u)) == 0u;
    const uint32_t mini_shift = per_mini32 != 0u ? (uint32_t)__builtin_ctz(per_mini32) : 0u;
    uint32_t window = s.remaining != 0u ? delta_window_at(s, s.pos, lane) : 0u;
    while (s.remaining != 0u) {
        // Block::try_new, decoder.rs:26-64; `window` holds the 64 bytes from `pos` on
        const Varint md = varint_at(window, 0u);
        if (md.panics) return 5;
        const uint64_t min_delta = zigzag(md.value);
        const uint32_t widths_at = s.pos + md.used;
        if (s.num_mini_blocks > (uint64_t)(end - widths_at)) return 1;  // "Block must contain at least..."
        const uint32_t block_len = (uint64_t)s.remaining < s.block_most ? s.remaining : (uint32_t)s.block_most;
        // the first mini-block is opened by try_new, each further one after the last value of the one before
        const uint32_t needed = block_len == 0u ? 1u : (block_len - 1u) / per_mini32 + 1u;
        uint32_t at = widths_at + (uint32_t)s.num_mini_blocks;  // where the next mini-block's bytes begin
        uint32_t next_window = 0u;
        for (uint32_t g = 0; g < needed; g += 64u) {
            // Block::advance_miniblock, decoder.rs:66-91, for the mini-blocks g .. g + 63
            const uint32_t j = g + lane;
            uint32_t bits;
            if (g == 0u && md.used + needed <= 64u)  // the widths are in the window already
                bits = (uint32_t)__shfl((int)window, (int)((md.used + lane) & 63u)) & 0xffu;
            else
                bits = j < needed ? delta_byte_at(s, widths_at + j) : 0u;
            if (j >= needed) bits = 0u;
            uint64_t length = bits != 0u ? (s.per_mini * bits + 7ull) / 8ull : 0ull;  // ceil8(values_per_mini_block * num_bits)
            if (length > 0x100000000ull) length = 0x100000000ull;                     // more than any stream holds
            const uint64_t upto = wave_inclusive_sum64(length);
            const bool short_of = bits != 0u && upto > (uint64_t)(end - at);  // "block must contain at least miniblock_length bytes"
            // bitpacked::Decoder::<u64>::try_new(..).unwrap() unpacks the first pack, unless there are no bytes at
            // all (decode.rs:46-50), and unpack64 has no width above 64
            const uint64_t bad = __ballot(short_of || (bits > 64u && length != 0ull));
            if (bad != 0ull) return ((__ballot(short_of) >> __builtin_ctzll(bad)) & 1ull) != 0ull ? 1 : 5;
            const bool narrow = __ballot(bits > 32u) == 0ull;  // then a field and its shift fit two dwords
            tbl_start[lane] = at + (uint32_t)(upto - length);
            tbl_bits[lane] = bits;
            wave_sync();
            at += (uint32_t)lane_value64(upto, 63u);
            // the next block's header is asked for before this block's values: its latency is the chain from
            // block to block, and loads return in the order they were issued
            if (g + 64u >= needed && s.remaining != block_len) next_window = delta_window_at(s, at, lane);
            // Iterator for Block (:93-118) and for Decoder (:216-243): the values of these mini-blocks
            const uint64_t group_end = ((uint64_t)g + 64ull) * per_mini32;
            const uint32_t v_end = group_end < block_len ? (uint32_t)group_end : block_len;
            auto term_of = [&](uint32_t v) -> uint64_t {  // min_delta + the packed value, 0 behind the last one
                if (v >= v_end) return 0ull;
                const uint32_t mini = minis_pow2 ? v >> mini_shift : v / per_mini32;
                const uint32_t w = tbl_bits[mini - g];
                const uint64_t bit = (uint64_t)tbl_start[mini - g] * 8ull + (uint64_t)(v - mini * per_mini32) * w;
                const uint32_t d = (uint32_t)(bit >> 5) * 4u;
                const uint32_t sh = (uint32_t)bit & 31u;
                const uint32_t x0 = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, d, 0, 0);
                const uint32_t x1 = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, d + 4u, 0, 0);
                const uint32_t x2 = narrow ? 0u : __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, d + 8u, 0, 0);
                const uint64_t field = ((uint64_t)__builtin_amdgcn_alignbit(x2, x1, sh) << 32) |
                                       __builtin_amdgcn_alignbit(x1, x0, sh);
                const uint64_t mask = w == 0u ? 0ull : (~0ull >> (64u - w));
                return min_delta + (field & mask);
            };
            // two steps of 64 values at a time: the loads of both are in flight before the first store
            for (uint32_t base = g * per_mini32; base < v_end; base += 128u) {
                const uint32_t v = base + lane;
                const uint64_t term0 = term_of(v);
                const uint64_t term1 = term_of(v + 64u);
                const uint64_t value0 = s.acc + wave_inclusive_sum64(term0);
                const uint64_t value1 = lane_value64(value0, 63u) + wave_inclusive_sum64(term1);
                if (v < v_end) s.out[1ull + s.done + v] = value0;
                if (v + 64u < v_end) s.out[1ull + s.done + v + 64u] = value1;
                s.acc = lane_value64(value1, 63u);  // the lanes behind v_end add nothing
            }
            wave_sync();
        }
        s.pos = at;
        window = next_window;
        if (block_len == 0u && ++s.empties == 2u) return 1;  // "Missing block" (decoder.rs:199-206)
        s.done += block_len;
        s.remaining -= block_len;
    }
    return 0;
}

// One workgroup of four waves per stream.  Streams of the usual shape (blocks of at most 512 values in at most 16
// mini-blocks) go chunk by chunk through two LDS buffers.  Wave 0, the walker, loads 8 KiB of the stream into a
// buffer -- the only time these bytes are read from memory --, walks the block headers inside it (a chain of LDS
// reads instead of one of memory round trips) and lists every block that lies in the chunk whole and is
// well-formed; then it goes on to the next chunk in the other buffer.  Waves 1..3, the expanders, follow one chunk
// behind: they share out the listed blocks, sum every block's deltas, wait for each other, each takes a prefix sum
// over the block sums (which gives every block its first value, and the chunk's last value), and they extract
// the deltas once more and store the values.  Where the walk lists nothing -- a block that is cut, malformed or
// absurd -- and for every other shape of stream, wave 0 goes on alone with delta_blocks from that block on, which
// also decides the stream's status: the chunked path accepts only what delta_blocks accepts.
//
// Blocks of 128 values in 1, 2 or 4 mini-blocks (`brief_shape`) have a shorter walk (the varint and the width bytes
// lie in 12 bytes from a block's start: every lane does the same arithmetic on them, one readfirstlane per block)
// and a shorter list (one 16-byte entry a block), and their expansion gives a lane two neighbouring values, stored
// with one instruction through a buffer descriptor over the output; blocks whose widths are all at most
// kDeltaNarrowBits take their running sums in 32 bits.  The same protocol, the same fallback.
//
// The waves of a workgroup are resident together, so every wait below ends: the walker waits for expanders that
// have all they need, the expanders for a walker that always advances or says that it has stopped; the waits are
// bounded all the same (status PQ2_DEVICE_ERROR).
#ifndef PQ2_DELTA_CHUNK
#define PQ2_DELTA_CHUNK 8192
#endif
#ifndef PQ2_DELTA_LISTED
#define PQ2_DELTA_LISTED 32
#endif
constexpr uint32_t kDeltaChunk = PQ2_DELTA_CHUNK;    // bytes of the stream per LDS buffer (a multiple of 1024)
constexpr uint32_t kDeltaListed = PQ2_DELTA_LISTED;  // blocks listed per chunk (at most 64)
constexpr uint32_t kDeltaMinis = 16;                 // mini-blocks per block on the chunked path
// values per block on the chunked path: a block (10 + 16 + 8 bytes per value at most) fits a chunk from any start
constexpr uint32_t kDeltaBlockMost = kDeltaChunk / 16u;
static_assert(10u + kDeltaMinis + kDeltaBlockMost * 8u + 15u <= kDeltaChunk && kDeltaListed <= 64u, "delta chunk");
constexpr uint32_t kDeltaExpanders = 3;

struct alignas(16) DeltaLists {  // what the walker says about one chunk
    uint64_t blk_min[kDeltaListed];                      // min_delta
    uint64_t blk_sum[kDeltaListed];                      // (by the expanders) the sum of the block's deltas
    uint32_t blk_len[kDeltaListed];                      // values
    uint32_t blk_first[kDeltaListed];                    // index of the first one among the stream's deltas
    union {
        uint16_t mini_start[kDeltaListed * kDeltaMinis];  // byte offset in the chunk
        // blocks of 128 values in at most 4 mini-blocks, one entry a block: the eight stream bytes from the
        // block's start on (x, y: min_delta's varint and what follows it), the four widths (z), the byte offset
        // of the first mini-block in the chunk and, from bit 16 on, the bytes of the varint (w)
        u32x4 brief[kDeltaListed];
    };
    uint8_t mini_bits[kDeltaListed * kDeltaMinis];
    uint32_t listed;  // blocks; 0: the walker has stopped here and takes the rest by itself
    uint32_t last;    // the stream ends with this chunk
    uint32_t first;   // blocks of 128: index of the first listed block's first value among the stream's deltas
    uint32_t left;    // ... and the deltas of the stream from there on
};
constexpr uint32_t kDeltaNarrowBits = 24;  // 128 deltas of at most that many bits: their sum fits 32 bits
// uleb128::decode over four bytes of a varint (the low seven bits of each, the first byte lowest)
__device__ __forceinline__ uint32_t varint_half(uint32_t b) {
    return (b & 0x7fu) | ((b >> 1) & 0x3f80u) | ((b >> 2) & 0x1fc000u) | ((b >> 3) & 0xfe00000u);
}

// (six workgroups on a compute unit: 80 registers a lane, a handful of rarely used values kept in scratch)
#ifndef PQ2_DELTA_WAVES
#define PQ2_DELTA_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PQ2_DELTA_WAVES, PQ2_DELTA_WAVES))) void delta_kernel(const DeltaPage *__restrict__ pages, uint32_t n_pages,
                                                    DeltaResult *__restrict__ results) {
    __shared__ __attribute__((aligned(16))) uint8_t chunks[2][kDeltaChunk + 32];
    __shared__ DeltaLists lists[2];
    __shared__ uint32_t tbl_start[64], tbl_bits[64];
    __shared__ uint32_t walked;                       // chunks whose lists are complete
    __shared__ uint32_t summed;                       // sum phases finished, over all expanders and chunks
    __shared__ uint32_t expanded[kDeltaExpanders];    // chunks every expander is done with
    __shared__ uint64_t acc_after;                    // the last value of the chunks expander 0 is done with

    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    DeltaStream s;
    uint64_t num_values = 0ull;
    // every wave reads the header for itself: all four hold the same stream state without an exchange
    int32_t status = delta_open(pages[page], lane, wave == 0u, s, num_values);
    auto report = [&](int32_t st) {  // (by the walker, or by every wave alike)
        if (wave == 0u && lane == 0u) {
            DeltaResult r;
            r.status = st;
            r.consumed = st == 0 ? s.pos - s.delta : 0u;
            r.num_values = num_values;
            results[page] = r;
        }
    };
    if (status >= 0) {
        report(status);
        return;
    }
    const uint32_t per_mini32 = s.per_mini32;
    const bool chunked = s.block_size_small != 0u && s.block_size_small <= kDeltaBlockMost &&
                         s.num_mini_blocks <= kDeltaMinis && per_mini32 != 0u;
    // blocks of 128 values in 1, 2 or 4 mini-blocks (what the writers at large produce): a shorter walk, shorter lists
    // (and values whose byte offsets in the output fit 32 bits: they are stored through a buffer descriptor)
    const bool brief_shape = s.block_most == 128ull && s.num_mini_blocks <= 4ull && num_values <= 0x10000000ull;
    // bytes of a mini-block = its width << per8_shift (a mini-block holds a multiple of 8 values, here 32, 64 or 128)
    const uint32_t per8_shift = brief_shape ? (uint32_t)__builtin_ctz(per_mini32) - 3u : 0u;
    if (chunked) {
        if (threadIdx.x < 4u) *(u32x4 *)(chunks[threadIdx.x >> 1] + kDeltaChunk + (threadIdx.x & 1u) * 16u) = u32x4{0u, 0u, 0u, 0u};
        if (threadIdx.x == 0u) walked = 0u, summed = 0u, expanded[0] = 0u, expanded[1] = 0u, expanded[2] = 0u;
        __syncthreads();
        // waits until *word >= want; false if that never comes
        auto wait_for = [&](const uint32_t *word, uint32_t want) -> bool {
            for (uint32_t polls = 0; polls < kMaxPolls; ++polls) {
                if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return true;
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0u) results[page].status = 6;  // PQ2_DEVICE_ERROR; never seen
            return false;
        };
        if (wave == 0u) {
            // ---- the walker ----
            // The walk is one chain of dependent steps, a few dozen instructions per block, and sets the pace of
            // the workgroup: its instructions go before those of the waves it shares the SIMD with.
            __builtin_amdgcn_s_setprio(3);
            const uint32_t block_most = (uint32_t)s.block_most;
            const uint32_t minis = (uint32_t)s.num_mini_blocks;
            for (uint32_t c = 0;; ++c) {
                uint8_t *const chunk = chunks[c & 1u];
                DeltaLists &ls = lists[c & 1u];
                if (c >= 2u)  // the buffer's chunk before has been expanded
                    for (uint32_t e = 0; e < kDeltaExpanders; ++e)
                        if (!wait_for(&expanded[e], c - 1u)) return;
                const uint32_t chunk_a = s.pos & ~15u;
                const uint32_t chunk_hi = chunk_a + kDeltaChunk;
                {
                    constexpr uint32_t kPieces = kDeltaChunk / 1024u;
                    u32x4 piece[kPieces];
#pragma unroll
                    for (uint32_t i = 0; i < kPieces; ++i)
                        piece[i] = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, chunk_a + i * 1024u + lane * 16u, 0, 0);
#pragma unroll
                    for (uint32_t i = 0; i < kPieces; ++i) *(u32x4 *)(chunk + i * 1024u + lane * 16u) = piece[i];
                }
                wave_sync();
                // Block::try_new and advance_miniblock as in delta_blocks, out of the chunk; anything but a
                // well-formed block that lies in the chunk whole ends the list
                uint32_t listed = 0u;
                if (brief_shape) {
                    // The same checks, every lane alike and without branches but one: the varint (up to 8 bytes; a
                    // longer one ends the list) and the four width bytes lie in 12 bytes from the block's start,
                    // read as four dwords.  A block whose first 12 bytes are not all in the stream ends the list
                    // as well.  Only the stream's last block may open fewer than all of its mini-blocks.
                    if (lane == 0u) ls.first = s.done, ls.left = s.remaining;
                    // (the stream's state is the same in every lane: into scalar registers, and the loop's
                    // conditions with it)
                    const uint32_t end = __builtin_amdgcn_readfirstlane(s.end);
                    const uint32_t remaining = __builtin_amdgcn_readfirstlane(s.remaining);
                    const uint32_t mini_shift = per8_shift + 3u;
                    const uint32_t blocks_left = (remaining + 127u) >> 7;
                    const uint32_t want = blocks_left < kDeltaListed ? blocks_left : kDeltaListed;
                    const uint32_t tail = remaining - (blocks_left - 1u) * 128u;  // values of the last block
                    const uint32_t tail_minis = ((tail - 1u) >> mini_shift) + 1u;
                    const uint32_t tail_mask = tail_minis < 4u ? (1u << (8u * tail_minis)) - 1u : ~0u;
                    const uint32_t whole_mask = minis < 4u ? (1u << (8u * minis)) - 1u : ~0u;
                    const uint32_t chunk_lo = __builtin_amdgcn_readfirstlane(chunk_a);
                    const uint32_t limit = end < chunk_lo + kDeltaChunk ? end : chunk_lo + kDeltaChunk;  // a listed block ends at or before it
                    uint32_t pos = __builtin_amdgcn_readfirstlane(s.pos);
                    for (; listed < want; ++listed) {
                        if (pos + 12u > end) break;
                        const uint32_t a = pos - chunk_lo;
                        const uint32_t *p = (const uint32_t *)(chunk + (a & ~3u));
                        const uint32_t x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
                        const uint32_t sh = a * 8u;  // (the funnel shift takes the low five bits)
                        const uint32_t b0 = __builtin_amdgcn_alignbit(x1, x0, sh);
                        const uint32_t b1 = __builtin_amdgcn_alignbit(x2, x1, sh);
                        const uint32_t b2 = __builtin_amdgcn_alignbit(x3, x2, sh);
                        // the varint's last byte is the first one without bit 7; `behind`: 8 * the varint's bytes
                        const uint32_t e0 = ~b0 & 0x80808080u, e1 = ~b1 & 0x80808080u;
                        const uint32_t behind = (uint32_t)__builtin_ctzll(((uint64_t)e1 << 32) | e0) + 1u;
                        // the four bytes behind the varint
                        const uint32_t lo = behind < 32u ? b0 : behind < 64u ? b1 : b2;
                        const uint32_t hi = behind < 32u ? b1 : b2;
                        uint32_t widths = __builtin_amdgcn_alignbit(hi, lo, behind);
                        widths &= listed + 1u == blocks_left ? tail_mask : whole_mask;
                        const uint32_t over = (((widths & 0x7f7f7f7fu) + 0x3f3f3f3fu) | widths) & 0x80808080u;  // a width above 64
                        const uint32_t total = __builtin_amdgcn_sad_u8(widths, 0u, 0u) << per8_shift;
                        const uint32_t at = pos + minis + (behind >> 3);
                        const uint32_t next = (e0 | e1) == 0u || over != 0u ? ~0u : at + total;
                        const uint32_t next_pos = __builtin_amdgcn_readfirstlane(next);
                        if (next_pos > limit) break;
                        // (by every lane alike, to one address: a branch on the lane would make a loop of lane masks
                        // out of this one, whose conditions are all scalar)
                        ls.brief[listed] = u32x4{b0, b1, widths, (at - chunk_lo) | (behind << 13)};
                        pos = next_pos;
                    }
                    const uint32_t taken = listed == blocks_left ? remaining : listed * 128u;
                    s.pos = pos;
                    s.done += taken;
                    s.remaining -= taken;
                } else
                while (s.remaining != 0u && listed < kDeltaListed) {
                    if (!(s.pos + 64u <= chunk_hi || s.end <= chunk_hi)) break;
                    const uint32_t a = s.pos + lane;
                    const uint32_t window = a < s.end ? chunk[a - chunk_a] : 0x100u;
                    const Varint md = varint_at(window, 0u);
                    if (md.panics) break;
                    const uint32_t widths_at = s.pos + md.used;
                    if (minis > s.end - widths_at) break;
                    const uint32_t block_len = s.remaining < block_most ? s.remaining : block_most;
                    // (a whole block opens all of its mini-blocks: the division is for the stream's last block)
                    const uint32_t needed = block_len == block_most ? minis : (block_len - 1u) / per_mini32 + 1u;
                    const uint32_t at = widths_at + minis;
                    // (the shuffle by all lanes: a lane that is switched off hands nothing over)
                    const uint32_t behind = (uint32_t)__shfl((int)window, (int)((md.used + lane) & 63u)) & 0xffu;
                    const uint32_t bits = lane < needed ? behind : 0u;
                    const uint32_t length = bits != 0u ? (per_mini32 * bits + 7u) / 8u : 0u;
                    const uint32_t upto = wave_inclusive_sum(length);
                    if (__ballot((bits != 0u && upto > s.end - at) || bits > 64u) != 0ull) break;
                    const uint32_t total = __builtin_amdgcn_readlane(upto, 63);
                    if (at + total > chunk_hi) break;
                    if (lane < kDeltaMinis) {
                        ls.mini_start[listed * kDeltaMinis + lane] = (uint16_t)(at + upto - length - chunk_a);
                        ls.mini_bits[listed * kDeltaMinis + lane] = (uint8_t)bits;
                    }
                    if (lane == 0u) {
                        ls.blk_min[listed] = zigzag(md.value);
                        ls.blk_len[listed] = block_len;
                        ls.blk_first[listed] = s.done;
                    }
                    s.pos = at + total;
                    s.done += block_len;
                    s.remaining -= block_len;
                    ++listed;
                }
                const bool last = s.remaining == 0u;
                if (lane == 0u) {
                    ls.listed = listed, ls.last = last ? 1u : 0u;
                    // release: the lists and the chunk are visible to whoever sees the new number
                    __hip_atomic_store(&walked, c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (listed == 0u) {
                    // the rest is for delta_blocks, from the last value of the chunks before
                    if (c != 0u) {
                        if (!wait_for(&expanded[0], c)) return;
                        s.acc = acc_after;
                    }
                    break;
                }
                if (last) {
                    report(0);
                    return;
                }
            }
        } else {
            // ---- the expanders ----
            const uint32_t e = wave - 1u;
            const bool minis_pow2 = (per_mini32 & (per_mini32 - 1u)) == 0u;
            const uint32_t mini_shift = (uint32_t)__builtin_ctz(per_mini32);
            uint64_t acc = s.acc;  // the last value before the chunk
            const bool small_blocks = s.block_most <= 128ull;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc((uint64_t)s.out, brief_shape ? (uint32_t)num_values * 8u : 0u);
            for (uint32_t c = 0;; ++c) {
                const uint8_t *const chunk = chunks[c & 1u];
                DeltaLists &ls = lists[c & 1u];
                if (!wait_for(&walked, c + 1u)) return;
                const uint32_t listed = __builtin_amdgcn_readfirstlane(ls.listed);  // (the same for every lane: into a scalar register)
                if (listed == 0u) return;
                // min_delta + the packed value v of listed block k
                auto term_of = [&](uint32_t k, uint32_t v, uint64_t min_delta) -> uint64_t {
                    const uint32_t mini = minis_pow2 ? v >> mini_shift : v / per_mini32;
                    const uint32_t w = ls.mini_bits[k * kDeltaMinis + mini];
                    const uint32_t bit = (uint32_t)ls.mini_start[k * kDeltaMinis + mini] * 8u + (v - mini * per_mini32) * w;
                    const uint32_t *src = (const uint32_t *)(chunk + ((bit >> 5) << 2));
                    const uint32_t sh = bit & 31u;
                    const uint32_t x0 = src[0], x1 = src[1], x2 = w > 32u ? src[2] : 0u;
                    const uint64_t field = ((uint64_t)__builtin_amdgcn_alignbit(x2, x1, sh) << 32) |
                                           __builtin_amdgcn_alignbit(x1, x0, sh);
                    const uint64_t mask = w == 0u ? 0ull : (~0ull >> (64u - w));
                    return min_delta + (field & mask);
                };
                // Blocks of at most 128 values (the usual size) are extracted once: the running sums inside the
                // block stay in registers until the block's first value is known.
                constexpr uint32_t kKept = (kDeltaListed + kDeltaExpanders - 1u) / kDeltaExpanders;  // blocks per wave
                uint64_t kept[kKept][2];
                const uint32_t first0 = brief_shape ? __builtin_amdgcn_readfirstlane(ls.first) : 0u;
                const uint32_t left0 = brief_shape ? __builtin_amdgcn_readfirstlane(ls.left) : 0u;
                if (brief_shape) {
                    // A lane takes the values 2 lane and 2 lane + 1 of a block (they lie in one mini-block, one
                    // behind the other; the lane stores them with one instruction): 8 * their mini-block's
                    // number, and the first one's number inside it.  kept[.][0] and [1]: the sums up to each.
                    const uint32_t mini_shift = per8_shift + 3u;
                    const uint32_t sh8 = ((2u * lane) >> mini_shift) * 8u;
                    const uint32_t idx = (2u * lane) & (per_mini32 - 1u);
#pragma unroll
                    for (uint32_t i = 0; i < kKept; ++i) {
                        const uint32_t k = e + i * kDeltaExpanders;
                        kept[i][0] = kept[i][1] = 0ull;
                        if (k < listed) {
                            const u32x4 entry = ls.brief[k];
                            const uint32_t widths = __builtin_amdgcn_readfirstlane(entry.z);
                            const uint32_t at = __builtin_amdgcn_readfirstlane(entry.w) & 0xffffu;
                            const uint32_t used = __builtin_amdgcn_readfirstlane(entry.w) >> 16;
                            uint64_t min_delta;  // zigzag_leb128::decode of the varint's `used` bytes
                            if (used <= 4u) {
                                const uint32_t u = varint_half(__builtin_amdgcn_readfirstlane(entry.x) & (~0u >> (32u - 8u * used)));
                                min_delta = (uint64_t)(int64_t)(int32_t)((u >> 1) ^ (0u - (u & 1u)));
                            } else {
                                const uint64_t bytes = (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(entry.y) << 32) |
                                                        (uint32_t)__builtin_amdgcn_readfirstlane(entry.x)) & (~0ull >> (64u - 8u * used));
                                min_delta = zigzag((uint64_t)varint_half((uint32_t)bytes) |
                                                   ((uint64_t)varint_half((uint32_t)(bytes >> 32)) << 28));
                            }
                            const uint32_t len = left0 - k * 128u < 128u ? left0 - k * 128u : 128u;
                            // the packed value 2 lane: its width and its first bit in the chunk
                            const uint32_t w = (widths >> sh8) & 0xffu;
                            const uint32_t minis_before = __builtin_amdgcn_sad_u8(widths & ~(~0u << sh8), 0u, 0u);
                            const uint32_t bit = ((at + (minis_before << per8_shift)) << 3) + idx * w;
                            const bool narrow = (((widths & 0x7f7f7f7fu) + (0x80u - 1u - kDeltaNarrowBits) * 0x01010101u) & 0x80808080u) == 0u;
                            if (narrow) {
                                // 128 deltas of at most 24 bits: the sums of the packed values fit 32 bits, and
                                // min_delta comes in by a multiplication.  Both values lie in three dwords.
                                const uint32_t *src = (const uint32_t *)(chunk + ((bit >> 5) << 2));
                                const uint32_t x0 = src[0], x1 = src[1], x2 = src[2];
                                const uint32_t mask = (1u << w) - 1u;
                                const uint32_t behind = (bit & 31u) + w;
                                uint32_t field0 = __builtin_amdgcn_alignbit(x1, x0, bit) & mask;
                                uint32_t field1 = __builtin_amdgcn_alignbit(behind < 32u ? x1 : x2, behind < 32u ? x0 : x1, behind) & mask;
                                uint32_t n0 = 2u * lane + 1u, n1 = 2u * lane + 2u;  // terms up to each value
                                if (len < 128u) {
                                    field0 = n0 <= len ? field0 : 0u, field1 = n1 <= len ? field1 : 0u;
                                    n0 = n0 < len ? n0 : len, n1 = n1 < len ? n1 : len;
                                }
                                const uint32_t upto = wave_inclusive_sum(field0 + field1);
                                kept[i][0] = (uint64_t)n0 * min_delta + (upto - field1);
                                kept[i][1] = (uint64_t)n1 * min_delta + upto;
                            } else {
                                uint64_t term[2];
#pragma unroll
                                for (uint32_t h = 0; h < 2u; ++h) {
                                    const uint32_t from = bit + h * w;
                                    const uint32_t *src = (const uint32_t *)(chunk + ((from >> 5) << 2));
                                    const uint32_t x0 = src[0], x1 = src[1], x2 = w > 32u ? src[2] : 0u;
                                    const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(x2, x1, from) << 32) |
                                                       __builtin_amdgcn_alignbit(x1, x0, from);
                                    const uint64_t mask = w == 0u ? 0ull : (~0ull >> (64u - w));
                                    term[h] = 2u * lane + h < len ? min_delta + (x & mask) : 0ull;
                                }
                                kept[i][1] = wave_inclusive_sum64(term[0] + term[1]);
                                kept[i][0] = kept[i][1] - term[1];
                            }
                            if (lane == 63u) ls.blk_sum[k] = kept[i][1];
                        }
                    }
                } else if (small_blocks) {
#pragma unroll
                    for (uint32_t i = 0; i < kKept; ++i) {
                        const uint32_t k = e + i * kDeltaExpanders;
                        kept[i][0] = kept[i][1] = 0ull;
                        if (k < listed) {
                            const uint32_t len = __builtin_amdgcn_readfirstlane(ls.blk_len[k]);
                            const uint64_t min_delta = ls.blk_min[k];
                            const uint64_t term0 = lane < len ? term_of(k, lane, min_delta) : 0ull;
                            const uint64_t term1 = lane + 64u < len ? term_of(k, lane + 64u, min_delta) : 0ull;
                            kept[i][0] = wave_inclusive_sum64(term0);
                            kept[i][1] = lane_value64(kept[i][0], 63u) + wave_inclusive_sum64(term1);
                            if (lane == 63u) ls.blk_sum[k] = kept[i][1];
                        }
                    }
                } else {
                    for (uint32_t k = e; k < listed; k += kDeltaExpanders) {  // the sums of this wave's blocks
                        const uint32_t len = __builtin_amdgcn_readfirstlane(ls.blk_len[k]);
                        const uint64_t min_delta = ls.blk_min[k];
                        uint64_t sum = 0ull;
                        for (uint32_t v = lane; v < len; v += 64u) sum += term_of(k, v, min_delta);
                        sum = wave_inclusive_sum64(sum);
                        if (lane == 63u) ls.blk_sum[k] = sum;
                    }
                }
                // all three have stored their sums (release / acquire through the counter)
                if (lane == 0u) __hip_atomic_fetch_add(&summed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!wait_for(&summed, kDeltaExpanders * (c + 1u))) return;
                // the value before every listed block's first, lane k for block k
                const uint64_t own = lane < listed ? ls.blk_sum[lane] : 0ull;
                const uint64_t upto = wave_inclusive_sum64(own);
                const uint64_t before_block = acc + upto - own;
                acc += lane_value64(upto, 63u);
                if (brief_shape) {
#pragma unroll
                    for (uint32_t i = 0; i < kKept; ++i) {
                        const uint32_t k = e + i * kDeltaExpanders;
                        if (k < listed) {
                            const uint64_t before = lane_value64(before_block, k);
                            const uint64_t v0 = before + kept[i][0], v1 = before + kept[i][1];
                            const uint32_t at = (1u + first0 + k * 128u + 2u * lane) * 8u;
                            if (left0 - k * 128u >= 128u) {
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32)},
                                                                       out_rsrc, at, 0, 0);
                            } else {
                                // the stream's last block: value by value through the descriptor, which drops
                                // a store behind the stream's last value
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)v0, (uint32_t)(v0 >> 32)}, out_rsrc, at, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)v1, (uint32_t)(v1 >> 32)}, out_rsrc, at + 8u, 0, 0);
                            }
                        }
                    }
                } else if (small_blocks) {
#pragma unroll
                    for (uint32_t i = 0; i < kKept; ++i) {
                        const uint32_t k = e + i * kDeltaExpanders;
                        if (k < listed) {
                            const uint32_t len = __builtin_amdgcn_readfirstlane(ls.blk_len[k]);
                            uint64_t *dst = s.out + 1ull + __builtin_amdgcn_readfirstlane(ls.blk_first[k]);
                            const uint64_t before = lane_value64(before_block, k);
                            if (lane < len) dst[lane] = before + kept[i][0];
                            if (lane + 64u < len) dst[lane + 64u] = before + kept[i][1];
                        }
                    }
                } else
                for (uint32_t k = e; k < listed; k += kDeltaExpanders) {
                    const uint32_t len = __builtin_amdgcn_readfirstlane(ls.blk_len[k]);
                    const uint64_t min_delta = ls.blk_min[k];
                    uint64_t *dst = s.out + 1ull + __builtin_amdgcn_readfirstlane(ls.blk_first[k]);
                    uint64_t before = lane_value64(before_block, k);
                    for (uint32_t base = 0; base < len; base += 64u) {
                        const uint32_t v = base + lane;
                        const uint64_t value = before + wave_inclusive_sum64(v < len ? term_of(k, v, min_delta) : 0ull);
                        if (v < len) dst[v] = value;
                        before = lane_value64(value, 63u);
                    }
                }
                const bool last = __builtin_amdgcn_readfirstlane(ls.last) != 0u;
                if (lane == 0u) {
                    if (e == 0u) acc_after = acc;
                    // release: this wave reads the buffer no more (and acc_after is in place)
                    __hip_atomic_store(&expanded[e], c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (last) return;
            }
        }
    }
    if (wave != 0u) return;
    status = delta_blocks(s, lane, tbl_start, tbl_bits);
    report(status);
}

}  // namespace pq2

// ---- launchers (called from pq2_api.cpp) ----------------------------------------
extern "C" hipError_t pq2_launch_parse(const void *pages, uint32_t n_pages, void *resume, uint32_t *tile_page,
                                       uint32_t *run_pos, int32_t *status, void *state, uint32_t round,
                                       uint32_t max_bits, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    if (max_bits > 16u)
        hipLaunchKernelGGL(pq2::parse_kernel<true>, dim3(n_pages), dim3(64), 0, stream, (const pq2::PageDev *)pages,
                           n_pages, (pq2::Resume *)resume, tile_page, run_pos, status, (pq2::ParseState *)state, round);
    else
        hipLaunchKernelGGL(pq2::parse_kernel<false>, dim3(n_pages), dim3(64), 0, stream, (const pq2::PageDev *)pages,
                           n_pages, (pq2::Resume *)resume, tile_page, run_pos, status, (pq2::ParseState *)state, round);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_probe(const void *pages, uint32_t n_pages, void *resume, uint32_t *tile_page,
                                       uint32_t *run_pos, int32_t *status, void *state, uint32_t passes, uint32_t max_bits,
                                       hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    if (max_bits > 16u)
        hipLaunchKernelGGL(pq2::probe_kernel<true>, dim3(n_pages), dim3(64), 0, stream, (const pq2::PageDev *)pages,
                           n_pages, (pq2::Resume *)resume, tile_page, run_pos, status, (pq2::ParseState *)state, passes);
    else
        hipLaunchKernelGGL(pq2::probe_kernel<false>, dim3(n_pages), dim3(64), 0, stream, (const pq2::PageDev *)pages,
                           n_pages, (pq2::Resume *)resume, tile_page, run_pos, status, (pq2::ParseState *)state, passes);
    return hipGetLastError();
}

// The tiles [tile_start, tile_start + n_tiles) are those of round `round`; `max_bits` is the largest width of
// a page in the batch: the stream window of a wave is sized for a whole tile of bit-packed values of that width.
// `dicts` (may be null): the values are dictionary indices of at most pq2_expand_dict_max_bits() bits and what is
// written are the dictionaries' values of `value_words` dwords (1, 2 or 3) at the pages' out_addr
// (expand_dict_kernel; `dicts`: one gather descriptor per page, `status`: the pages' statuses).
extern "C" uint32_t pq2_expand_dict_max_bits() { return (uint32_t)pq2::ExpandDictSink<1>::kMaxBits; }
extern "C" hipError_t pq2_launch_expand(const void *pages, const uint32_t *tile_page, uint32_t tile_start,
                                        uint32_t n_tiles, const void *resume, const uint32_t *run_pos, uint32_t round,
                                        uint32_t max_bits, const uint32_t *tile_mask, const void *dicts,
                                        uint32_t value_words, int32_t *status, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    uint32_t win_bytes = (pq2::kTileValues / 8u) * (max_bits ? max_bits : 1u) + 256u;
    if (win_bytes < 2048u) win_bytes = 2048u;
    // a workgroup gets at most 64 KiB: the widest pages then refill their window inside a tile
    const uint32_t most = (65536u - 32u - 4u * (pq2::kTableRuns + 2u) * 8u - 4u * pq2::kChunkSlots * 4u) / 4u;
    if (win_bytes > most) win_bytes = most;
    win_bytes &= ~15u;
    uint32_t pow2 = 2048u;
    while (pow2 < win_bytes) pow2 <<= 1;
    const uint32_t lds_bytes = 4u * win_bytes + 32u + 4u * (pq2::kTableRuns + 2u) * 8u + 4u * pq2::kChunkSlots * 4u;
    // a batch of few tiles: a wave for every span of a tile
    static const uint32_t per_span_most = [] {
        const char *v = getenv("PQ2_EXPAND_PER_SPAN_TILES");
        return v ? (uint32_t)atoi(v) : PQ2_EXPAND_PER_SPAN_TILES;
    }();
    const uint32_t per_span = n_tiles <= per_span_most ? 1u : 0u;
    const uint32_t units = per_span ? n_tiles * (pq2::kBlocksPerTile / pq2::kSpanBlocks) : n_tiles;
#define PQ2_EXPAND_DICT(WORDS)                                                                                      \
    hipLaunchKernelGGL(pq2::expand_dict_kernel<WORDS>, dim3((units + 3u) / 4u), dim3(256), lds_bytes, stream,         \
                       (const pq2::PageDev *)pages, tile_page, (const pq2::Resume *)resume, run_pos, tile_start,        \
                       tile_start + n_tiles, round, win_bytes, pow2 * 8u - 1u, tile_mask,                               \
                       (const pq2::ExpandDictRef *)dicts, status, per_span)
    if (dicts && value_words == 1u)
        PQ2_EXPAND_DICT(1);
    else if (dicts && value_words == 2u)
        PQ2_EXPAND_DICT(2);
    else if (dicts && value_words == 3u)
        PQ2_EXPAND_DICT(3);
    else if (dicts)
        return hipErrorInvalidValue;
    else
        hipLaunchKernelGGL(pq2::expand_kernel, dim3((units + 3u) / 4u), dim3(256), lds_bytes, stream,
                           (const pq2::PageDev *)pages, tile_page, (const pq2::Resume *)resume, run_pos, tile_start,
                           tile_start + n_tiles, round, win_bytes, pow2 * 8u - 1u, tile_mask, per_span);
#undef PQ2_EXPAND_DICT
    return hipGetLastError();
}

// One workgroup per page; `max_bits` as for pq2_launch_expand.
extern "C" uint32_t pq2_cu_ticket_bytes() { return pq2::kCuTickets * 4u; }

extern "C" hipError_t pq2_launch_decode(const void *pages, uint32_t n_pages, void *resume, uint32_t *run_pos,
                                        int32_t *status, void *state, uint32_t *cu_tickets, uint32_t max_bits,
                                        const uint32_t *tile_mask, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    constexpr uint32_t kFixed = pq2::kWin + 32u +
                                pq2::kDecodeExpanders * ((pq2::kTableRuns + 2u) * 8u + pq2::kChunkSlots * 4u) + 64u;
    uint32_t win_bytes = (pq2::kTileValues / 8u) * (max_bits ? max_bits : 1u) + 256u;
    if (win_bytes < 2048u) win_bytes = 2048u;
    if (win_bytes > 16384u) win_bytes = 16384u;  // the widest pages refill their window inside a tile
    win_bytes &= ~15u;
    uint32_t pow2 = 2048u;
    while (pow2 < win_bytes) pow2 <<= 1;
    // positions masked with `pow2 * 8 - 1` behind the last window's base stay inside the allocation
    uint32_t lds_bytes = kFixed + pq2::kDecodeExpanders * win_bytes;
    const uint32_t reach = pq2::kWin + (pq2::kDecodeExpanders - 1u) * win_bytes + pow2 + 64u;
    if (lds_bytes < reach) lds_bytes = reach;
    if (max_bits > 16u)
        hipLaunchKernelGGL(pq2::decode_kernel<true>, dim3(n_pages), dim3(256), lds_bytes, stream,
                           (const pq2::PageDev *)pages, n_pages, (pq2::Resume *)resume, run_pos, status,
                           (pq2::ParseState *)state, cu_tickets, win_bytes, pow2 * 8u - 1u, tile_mask);
    else
        hipLaunchKernelGGL(pq2::decode_kernel<false>, dim3(n_pages), dim3(256), lds_bytes, stream,
                           (const pq2::PageDev *)pages, n_pages, (pq2::Resume *)resume, run_pos, status,
                           (pq2::ParseState *)state, cu_tickets, win_bytes, pow2 * 8u - 1u, tile_mask);
    return hipGetLastError();
}

// seg_rows = 0: one workgroup per page.  Else the pages are cut into `n_segs` segments of seg_rows rows
// (OptionalPage::first_seg) and `seg_valid` has room for n_segs words.
extern "C" hipError_t pq2_launch_optional(const void *pages, uint32_t n_pages, uint32_t value_words, int32_t *status,
                                          uint32_t seg_rows, uint32_t n_segs, uint32_t *seg_valid, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    const pq2::OptionalPage *p = (const pq2::OptionalPage *)pages;
    const uint32_t grid = seg_rows == 0u ? n_pages : n_segs;
    if (grid == 0u) return hipSuccess;
    if (seg_rows != 0u)
        hipLaunchKernelGGL(pq2::optional_count_kernel, dim3(grid), dim3(256), 0, stream, p, n_pages, seg_rows, seg_valid, status);
    if (value_words == 1)
        hipLaunchKernelGGL(pq2::optional_kernel<1>, dim3(grid), dim3(256), 0, stream, p, n_pages, status, seg_rows, seg_valid);
    else if (value_words == 2)
        hipLaunchKernelGGL(pq2::optional_kernel<2>, dim3(grid), dim3(256), 0, stream, p, n_pages, status, seg_rows, seg_valid);
    else
        hipLaunchKernelGGL(pq2::optional_kernel<3>, dim3(grid), dim3(256), 0, stream, p, n_pages, status, seg_rows, seg_valid);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_select(const void *pieces, uint32_t n_pieces, hipStream_t stream) {
    if (n_pieces == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::select_kernel, dim3(n_pieces), dim3(256), 0, stream, (const pq2::SelectPiece *)pieces);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_filtered(const void *pieces, uint32_t n_pieces, uint32_t *counts, hipStream_t stream) {
    if (n_pieces == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::filtered_kernel, dim3(n_pieces), dim3(256), 0, stream, (const pq2::FilterPiece *)pieces, counts);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_front(void *streams, const void *raw, uint32_t n_pages, int32_t *front_status,
                                       hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::front_kernel, dim3((n_pages + 255u) / 256u), dim3(256), 0, stream, (pq2::PageDev *)streams,
                       (const pq2::RawPageDev *)raw, n_pages, front_status);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_bitunpack64(const void *d_packed, uint32_t packed_len, uint32_t num_bits,
                                             uint32_t length, uint64_t *d_out, hipStream_t stream) {
    if (length == 0) return hipSuccess;
    const uint32_t groups = (length + 4u * pq2::kUnpack64Block - 1u) / (4u * pq2::kUnpack64Block);  // of four blocks: a workgroup's turn
    const uint32_t wave_lds = (pq2::unpack64_pieces(num_bits ? num_bits : 1u) * 16u + 63u) & ~63u;
    hipLaunchKernelGGL(pq2::bitunpack64_kernel, dim3(groups < 65536u ? groups : 65536u), dim3(256), 4u * wave_lds, stream,
                       (uint64_t)(uintptr_t)d_packed, packed_len, num_bits, length, d_out, wave_lds);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_delta(const void *pages, uint32_t n_pages, void *results, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::delta_kernel, dim3(n_pages), dim3(256), 0, stream, (const pq2::DeltaPage *)pages, n_pages,
                       (pq2::DeltaResult *)results);
    return hipGetLastError();
}

extern "C" uint32_t pq2_gather_tile_values() { return pq2::kGatherTileValues; }

// `max_dict_bytes`: the largest dictionary of the batch; dictionaries of up to 48 KiB are looked up in LDS.
extern "C" hipError_t pq2_launch_gather(const void *pages, uint32_t n_pages, uint32_t n_tiles, uint32_t value_words,
                                        uint64_t max_dict_bytes, int32_t *status, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    const pq2::GatherPage *p = (const pq2::GatherPage *)pages;
    const bool use_lds = max_dict_bytes > 0 && max_dict_bytes <= 49152u;
    const uint32_t lds = use_lds ? (uint32_t)((max_dict_bytes + 15u) & ~15ull) : 0u;
#define PQ2_GATHER(WORDS)                                                                                         \
    if (use_lds)                                                                                                  \
        hipLaunchKernelGGL((pq2::gather_kernel<WORDS, true>), dim3(n_tiles), dim3(256), lds, stream, p, n_pages,  \
                           status, lds);                                                                          \
    else                                                                                                          \
        hipLaunchKernelGGL((pq2::gather_kernel<WORDS, false>), dim3(n_tiles), dim3(256), 0, stream, p, n_pages,   \
                           status, 0u);
    if (value_words == 1) {
        PQ2_GATHER(1)
    } else if (value_words == 2) {
        PQ2_GATHER(2)
    } else {
        PQ2_GATHER(3)
    }
#undef PQ2_GATHER
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_bitunpack(const void *d_packed, uint32_t packed_len, uint32_t num_bits,
                                           uint32_t length, uint32_t *d_out, hipStream_t stream) {
    if (length == 0) return hipSuccess;
    const uint32_t n_groups = (length + 4u * pq2::kUnpackBlock - 1) / (4u * pq2::kUnpackBlock);  // of four blocks: a workgroup's turn
    const uint32_t grid = n_groups < 16384u ? n_groups : 16384u;
    const uint32_t wave_lds = (pq2::unpack_pieces(num_bits ? num_bits : 1u) * 16u + 63u) & ~63u;
    switch (num_bits) {
#define PQ2_W(N)                                                                                                      \
    case N:                                                                                                           \
        hipLaunchKernelGGL(pq2::bitunpack_kernel<N>, dim3(grid), dim3(256), 4u * wave_lds, stream,                    \
                           (uint64_t)(uintptr_t)d_packed, packed_len, length, d_out, wave_lds);                       \
        break;
        PQ2_W(1) PQ2_W(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12) PQ2_W(13) PQ2_W(14) PQ2_W(15) PQ2_W(16)
        PQ2_W(17) PQ2_W(18) PQ2_W(19) PQ2_W(20) PQ2_W(21) PQ2_W(22) PQ2_W(23) PQ2_W(24)
        PQ2_W(25) PQ2_W(26) PQ2_W(27) PQ2_W(28) PQ2_W(29) PQ2_W(30) PQ2_W(31) PQ2_W(32)
#undef PQ2_W
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
