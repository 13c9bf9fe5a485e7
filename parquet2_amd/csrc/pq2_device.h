// Device-side building blocks shared by the pq2hip kernels (gfx950 only).
//
// Coordinates.  A page's stream starts at an arbitrary byte address.  All
// device code addresses it through a buffer resource whose base is that
// address rounded down to 16 bytes; "a-coordinates" are byte offsets from that
// base, so the stream occupies a-offsets [delta, delta + in_len) with
// delta = address & 15.  Every load is an aligned dword (or 4 aligned dwords)
// and is range-checked by the hardware against the descriptor: reads past the
// end return 0 instead of faulting, whatever the stream contains.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pq2 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Output values live in global memory.  The address arrives as an integer inside a descriptor, so the pointer
// type has to say it: a generic pointer would make every store a flat_store, which is also counted by the
// LDS / scalar-memory counter and holds back the waits of LDS reads.
typedef __attribute__((address_space(1))) uint32_t out_u32;
typedef __attribute__((address_space(1))) u32x4 out_u32x4;

constexpr uint32_t kBlockValues = 1024;    // values per wave-block (PQ2_BLOCK_VALUES)
constexpr uint32_t kBlocksPerTile = 4;     // one 256-thread workgroup = 4 wave-blocks
constexpr uint32_t kTileValues = kBlockValues * kBlocksPerTile;
constexpr uint32_t kEndPos = 0xffffffffu;  // resume position meaning "stream exhausted: zeros"

// A batch is decoded in up to kRounds rounds so that finding the headers of
// one part of every page (parse_kernel) overlaps with producing the values of
// the part before it (expand_kernel, on another stream).  Round r of a page
// covers its tiles [round_tile[r], round_tile[r + 1]) (round_tile[0] = 0, the
// last round ends with the page); tiles are numbered round-major over the
// batch (all tiles of round 0 of all pages, then round 1, ...), so that the
// tiles of one round form one contiguous grid.
constexpr uint32_t kRounds = 4;

// Device-side page descriptor (72 bytes), built by the host from pq2_page_desc.
struct PageDev {
    uint64_t in_addr;             // device address of the first stream byte
    uint64_t out_addr;            // device address of the page's first output value
    uint64_t run_base;            // index of the page's first entry in the header-position list
    uint32_t in_len;              // stream bytes
    uint32_t num_values;          // values to produce (0 if the page is skipped)
    uint32_t num_bits;            // 0..32
    uint32_t round_tile[kRounds];  // the page's first tile of round r (tile = 4096 values); [0] = 0, ascending
    uint32_t tile_base[kRounds];   // batch-wide number of that tile
    uint32_t reserved;

    // The members are only ever indexed with constants: a register-resident copy of the descriptor stays in
    // (scalar) registers.
    static_assert(kRounds == 4, "the accessors spell out four rounds");
    __device__ __forceinline__ uint32_t round_first(uint32_t r) const {
        return r == 0 ? round_tile[0] : r == 1 ? round_tile[1] : r == 2 ? round_tile[2] : round_tile[3];
    }
    __device__ __forceinline__ uint32_t round_base(uint32_t r) const {
        return r == 0 ? tile_base[0] : r == 1 ? tile_base[1] : r == 2 ? tile_base[2] : tile_base[3];
    }
    // batch-wide number of the page's tile t
    __device__ __forceinline__ uint32_t tile_number(uint32_t t) const {
        // the step of the numbering at each round boundary (wave-uniform), added where t lies beyond it
        const uint32_t s1 = tile_base[1] - tile_base[0] - (round_tile[1] - round_tile[0]);
        const uint32_t s2 = tile_base[2] - tile_base[1] - (round_tile[2] - round_tile[1]);
        const uint32_t s3 = tile_base[3] - tile_base[2] - (round_tile[3] - round_tile[2]);
        uint32_t number = tile_base[0] + t;
        number += t >= round_tile[1] ? s1 : 0u;
        number += t >= round_tile[2] ? s2 : 0u;
        number += t >= round_tile[3] ? s3 : 0u;
        return number;
    }
};

// What parse_kernel keeps of a page between two rounds.
struct ParseState {
    uint32_t pos;       // a-offset of the next header
    uint32_t produced;  // values the listed runs yield so far
    uint32_t n_runs;    // entries of the header-position list so far
    uint32_t done;      // the page's Resume entries and header list are complete
};

// The header-position list: for every page, the a-offsets of the headers of
// its runs that yield at least one value, in stream order, closed by kEndPos.
// parse_kernel writes it (4 bytes per run); expand_kernel reads the entries of
// the runs that start inside its block, one per lane, and decodes those
// headers in parallel.
//
// Where a 1024-value block starts inside the stream: the run that covers the
// block's first value, as far as the block needs it.  Written by parse_kernel,
// read by expand_kernel (16 bytes per 4 KiB of output).
struct Resume {
    uint32_t run_index;  // index of this run in the page's header-position list
    uint32_t rem_kind;   // values of the run from the block's first value on | kind << 31
    uint32_t payload;    // bit-packed: a-offset of the run's payload; RLE: the value
    uint32_t skip;       // bit-packed: values of the run that precede the block; RLE: a-offset of the byte
                         // behind the run (where the stream bytes the block may need begin), kEndPos if none
};
constexpr uint32_t kRemMask = 0x7fffffffu;

enum RunKind : uint32_t { kBitpacked = 0, kRle = 1, kEnd = 2, kPanic = 3 };

// What ring_kernel<., kWords, true> needs of a flat nullable dictionary page beside its PageDev and RingDict: the
// page's validity bitmap (row r at bit r % 32 of word r / 32, bits behind the page's rows 0), where its rows go, and
// the row directory rowdir_kernel makes of the bitmap: entry m >= 1 is the row of the page's valid row number
// kRowDirStep * m (counted from 0), or `rows` if the page has no more valid rows; entry 0 is 0.
struct RingNullable {  // must match the host's
    uint64_t bitmap_addr;    // 4-byte aligned
    uint64_t validity_addr;  // one byte per row
    uint64_t out_addr;       // kWords dwords per row
    uint64_t rowdir_addr;    // idx_values / kRowDirStep + 3 words
    uint32_t rows, idx_values;
};
constexpr uint32_t kRowDirStep = 2048;  // = the values of a span of ring_kernel

// A nullable dictionary page after its two streams are decoded (optional_kernel, optional_bits_kernel).
struct OptionalPage {  // must match the host's
    uint64_t level_addr;  // decoded levels: u32, one per row, 16-byte aligned; or the validity bitmap, row r at bit r % 32 of word r / 32
    uint64_t idx_addr;    // decoded indices (u32)
    uint64_t out_addr, validity_addr, dict_addr;
    uint32_t rows, idx_values, dict_len, max_level;
    uint32_t idx_status_at;  // where the index stream's status lies in the status array
    uint32_t first_seg;      // batch-wide number of the page's first segment of `seg_rows` rows
};

struct RunInfo {
    uint32_t kind;     // RunKind
    uint32_t count;    // values the run yields, already clamped to `remaining`
    uint32_t payload;  // a-offset of the bit-packed payload
    uint32_t value;    // RLE value (unmasked, as the reference)
    uint32_t next;     // a-offset of the next header
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(uint64_t base16, uint32_t num_records) {
    // 0x00020000: DATA_FORMAT = 32 (raw dword access) for gfx9-family descriptors
    return __builtin_amdgcn_make_buffer_rsrc((void *)base16, 0, (int)num_records, 0x00020000);
}

__device__ __forceinline__ uint32_t ceil8(uint32_t v) { return (v + 7u) >> 3; }

// Rare headers: varints of 3..10 bytes, and bit-packed runs cut by the end of
// the stream.  Same contract as decode_run below.
template <typename Rd>
__device__ inline RunInfo decode_run_slow(Rd rd, uint32_t pos, uint32_t end, uint32_t w, uint32_t remaining) {
    RunInfo r;
    r.kind = kEnd;
    r.count = 0;
    r.payload = 0;
    r.value = 0;
    r.next = end;
    const uint32_t a0 = pos & ~3u;
    const uint32_t sh = (pos & 3u) * 8u;
    const uint32_t d0 = rd(a0), d1 = rd(a0 + 4), d2 = rd(a0 + 8), d3 = rd(a0 + 12);
    const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sh);  // 12 stream bytes starting at pos
    const uint32_t x1 = __builtin_amdgcn_alignbit(d2, d1, sh);
    const uint32_t x2 = __builtin_amdgcn_alignbit(d3, d2, sh);

    // uleb128::decode, src/encoding/uleb128.rs:3-23
    uint64_t ind = 0;
    uint32_t c = 0;
    bool done = false;
    for (uint32_t i = 0; i < 10 && !done; ++i) {
        if (pos + i >= end) return r;  // the varint runs into the end of the stream: consumed == len
        const uint32_t word = i < 4 ? x0 : (i < 8 ? x1 : x2);
        const uint32_t b = (word >> ((i & 3u) * 8u)) & 0xffu;
        c = i + 1;
        if (i == 9 && b > 1) {  // uleb128.rs:10-12
            r.kind = kPanic;
            return r;
        }
        ind |= (uint64_t)(b & 0x7fu) << (7u * i);
        done = (b & 0x80u) == 0;
    }
    const uint32_t p = pos + c;
    if (p >= end) return r;  // src/encoding/hybrid_rle/decoder.rs:44-46
    const uint32_t left = end - p;
    const uint64_t n = ind >> 1;
    if (ind & 1u) {
        if ((n >> 59) != 0 && n > 0xffffffffffffffffull / w) {  // (ind >> 1) * num_bits overflows usize
            r.kind = kPanic;
            return r;
        }
        const uint64_t bytes = n * w;
        uint32_t nbytes;
        uint64_t k;
        if (bytes > left) {  // decoder.rs:51 clamp: only whole values inside the remaining bytes
            nbytes = left;
            k = (uint64_t)(left / w) * 8u + ((left % w) * 8u) / w;  // left * 8 / w
        } else {
            nbytes = (uint32_t)bytes;
            k = n * 8u;
        }
        r.kind = kBitpacked;
        r.count = k < remaining ? (uint32_t)k : remaining;
        r.payload = p;
        r.next = p + nbytes;
    } else {
        const uint32_t rb = ceil8(w);
        if (left < rb) {  // split_at(rle_bytes) panics, decoder.rs:59-60
            r.kind = kPanic;
            return r;
        }
        const uint32_t b0 = p & ~3u;
        uint32_t v = __builtin_amdgcn_alignbit(rd(b0 + 4), rd(b0), (p & 3u) * 8u);
        if (rb < 4) v &= (1u << (rb * 8u)) - 1u;
        r.kind = kRle;
        r.count = n < remaining ? (uint32_t)n : remaining;
        r.value = v;
        r.next = p + rb;
    }
    return r;
}

// Decodes the run whose header starts at a-offset `pos`.  `rd(a)` returns the
// aligned dword at a-offset `a` (a % 4 == 0); up to 24 bytes from pos & ~3 are
// read.  Everything here is wave-uniform.
//
// Follows hybrid_rle::Decoder::next (src/encoding/hybrid_rle/decoder.rs:30-64),
// uleb128::decode (src/encoding/uleb128.rs:3-23) and read_next
// (src/encoding/hybrid_rle/mod.rs:41-64):
//   - an exhausted stream, or a header that consumes the last byte, ends the
//     stream (kEnd): the caller pads with zeros (mod.rs:95);
//   - bit-packed: bytes = min((ind >> 1) * w, left); values = min(bytes*8/w, remaining);
//   - RLE: the value is the first min(4, ceil8(w)) bytes, little endian, not
//     masked; fewer than ceil8(w) bytes left is a panic in the reference.
// Headers of one or two bytes whose run lies inside the stream take the short
// path below; everything else goes through decode_run_slow.
template <typename Rd>
__device__ __forceinline__ RunInfo decode_run(Rd rd, uint32_t pos, uint32_t end, uint32_t w, uint32_t remaining) {
    RunInfo r;
    if (pos >= end) {
        r.kind = kEnd;
        r.count = 0;
        r.payload = 0;
        r.value = 0;
        r.next = end;
        return r;
    }
    const uint32_t a0 = pos & ~3u;
    const uint32_t sh = (pos & 3u) * 8u;
    const uint32_t d0 = rd(a0), d1 = rd(a0 + 4), d2 = rd(a0 + 8);
    const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sh);  // stream bytes pos .. pos+3
    const uint32_t x1 = __builtin_amdgcn_alignbit(d2, d1, sh);  // stream bytes pos+4 .. pos+7
    uint32_t ind, c, rest;  // header value, header bytes, the 4 stream bytes after the header
    if ((x0 & 0x80u) == 0) {
        ind = x0 & 0x7fu;
        c = 1;
        rest = __builtin_amdgcn_alignbit(x1, x0, 8);
    } else if ((x0 & 0x8000u) == 0) {
        ind = (x0 & 0x7fu) | ((x0 >> 1) & 0x3f80u);
        c = 2;
        rest = __builtin_amdgcn_alignbit(x1, x0, 16);
    } else {
        return decode_run_slow(rd, pos, end, w, remaining);
    }
    const uint32_t p = pos + c;
    const uint32_t n = ind >> 1;
    if (ind & 1u) {
        const uint32_t bytes = n * w;  // < 2^13 * 32
        if (p >= end || bytes > end - p) return decode_run_slow(rd, pos, end, w, remaining);
        const uint32_t k = n * 8u;
        r.kind = kBitpacked;
        r.count = k < remaining ? k : remaining;
        r.payload = p;
        r.value = 0;
        r.next = p + bytes;
    } else {
        const uint32_t rb = ceil8(w);
        if (p >= end || rb > end - p) return decode_run_slow(rd, pos, end, w, remaining);
        r.kind = kRle;
        r.count = n < remaining ? n : remaining;
        r.payload = 0;
        r.value = rb < 4 ? (rest & ((1u << (rb * 8u)) - 1u)) : rest;
        r.next = p + rb;
    }
    return r;
}

// One lane's view of a run: what expand_kernel needs to place and produce its values.
struct LaneRun {
    uint32_t kind;   // kBitpacked | kRle
    uint32_t count;  // values the run yields before any clamp to the page length (saturated at 2^31 - 1)
    uint32_t info;   // bit-packed: a-offset of the payload; RLE: the value
};

// Per-lane header decode out of an LDS image: `lds + off` is the header's first
// byte and at least 24 bytes from off & ~3 are readable.  Same arithmetic as
// decode_run, for a header parse_kernel has already accepted (it lies inside
// the stream and its run yields values), so no end-of-stream case remains but
// the clamp of a bit-packed payload cut by the end of the stream.
template <int W>
__device__ __forceinline__ LaneRun decode_lane(const uint8_t *lds, uint32_t off, uint32_t hp, uint32_t end) {
    constexpr uint32_t w = (uint32_t)W;
    constexpr uint32_t rb = (w + 7u) >> 3;
    const uint32_t *src = (const uint32_t *)(lds + (off & ~3u));
    const uint32_t sh = (off & 3u) * 8u;
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
    const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sh);
    const uint32_t x1 = __builtin_amdgcn_alignbit(d2, d1, sh);
    LaneRun r;
    if (__builtin_expect((x0 & 0x8080u) != 0x8080u, 1)) {
        // a header of one or two bytes (nearly all are): 32-bit arithmetic does, (ind >> 1) * w < 2^13 * 32
        const bool two = (x0 & 0x80u) != 0u;
        const uint32_t ind = two ? (x0 & 0x7fu) | ((x0 >> 1) & 0x3f80u) : x0 & 0x7fu;
        const uint32_t rest = two ? __builtin_amdgcn_alignbit(x1, x0, 16) : __builtin_amdgcn_alignbit(x1, x0, 8);
        const uint32_t p = hp + (two ? 2u : 1u);
        const uint32_t n = ind >> 1;
        const uint32_t left = end - p;
        uint32_t k = n * 8u;
        if (__builtin_expect((ind & 1u) != 0u && n * w > left, 0))  // decoder.rs:51: only whole values inside the remaining bytes
            k = (left / w) * 8u + ((left % w) * 8u) / w;
        r.kind = (ind & 1u) ? kBitpacked : kRle;
        r.count = (ind & 1u) ? k : n;
        r.info = (ind & 1u) ? p : (rb < 4 ? (rest & ((1u << ((rb & 3u) * 8u)) - 1u)) : rest);
        return r;
    }
    // 3..10 header bytes
    uint64_t ind = 0;
    uint32_t c = 0;
    {
        const uint32_t d3 = src[3], d4 = src[4];
        const uint32_t x2 = __builtin_amdgcn_alignbit(d3, d2, sh);
        const uint32_t x3 = __builtin_amdgcn_alignbit(d4, d3, sh);
        bool done = false;
        for (uint32_t i = 0; i < 10 && !done; ++i) {
            const uint32_t word = i < 4 ? x0 : (i < 8 ? x1 : x2);
            const uint32_t b = (word >> ((i & 3u) * 8u)) & 0xffu;
            c = i + 1;
            ind |= (uint64_t)(b & 0x7fu) << (7u * i);
            done = (b & 0x80u) == 0;
        }
        // the 4 bytes that follow a header of c = 3..10 bytes
        const uint32_t lo_w = c < 4 ? x0 : (c < 8 ? x1 : x2);
        const uint32_t hi_w = c < 4 ? x1 : (c < 8 ? x2 : x3);
        const uint32_t rest = __builtin_amdgcn_alignbit(hi_w, lo_w, (c & 3u) * 8u);
        const uint32_t p = hp + c;
        const uint64_t n = ind >> 1;
        if (ind & 1u) {
            const uint32_t left = end - p;
            uint64_t k;
            if (n > 0x0fffffffull || n * w > left)  // decoder.rs:51: only whole values inside the remaining bytes
                k = (uint64_t)(left / w) * 8u + ((left % w) * 8u) / w;
            else
                k = n * 8u;
            r.kind = kBitpacked;
            r.count = k < 0x7fffffffull ? (uint32_t)k : 0x7fffffffu;
            r.info = p;
        } else {
            r.kind = kRle;
            r.count = n < 0x7fffffffull ? (uint32_t)n : 0x7fffffffu;
            r.info = rb < 4 ? (rest & ((1u << ((rb & 3u) * 8u)) - 1u)) : rest;
        }
    }
    return r;
}

// Inclusive prefix operations over the 64 lanes of a wave with DPP row shifts and row broadcasts: six
// instructions (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then the last lane of row 0 / 2 into row 1 / 3
// and the last lane of row 1 into rows 2 and 3).  0 is the identity of both operations.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// the sum with every partial result capped at `cap` (an associative operation on values <= cap < 2^31)
__device__ __forceinline__ uint32_t wave_inclusive_sum_saturating(uint32_t v, uint32_t cap) {
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false), cap);
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false), cap);
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false), cap);
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false), cap);
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false), cap);
    v = min(v + (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false), cap);
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false));
    return v;
}

// Orders the LDS operations of one wave as the program has them (they execute in order; this keeps the
// compiler from moving them) where a region of LDS is private to the wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A load that sees what another wave of the same GPU has stored (and made visible with a release) although this
// compute unit's vector cache may hold an older copy of the line.
__device__ __forceinline__ uint32_t load_coherent(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- stores -------------------------------------------------------------------

__device__ __forceinline__ void store4(u32x4 v, out_u32 *p) {
#ifdef PQ2_EXPERIMENT_NO_STORES  // timing experiment only: results are not written
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u && v.z == 0x0fedcba9u) *(out_u32x4 *)p = v;
#elif defined(PQ2_NT_STORES)
    __builtin_nontemporal_store(v, (out_u32x4 *)p);
#else
    *(out_u32x4 *)p = v;
#endif
}

// ---- LDS-staged stream window --------------------------------------------------
// Each wave of expand_kernel owns `cap` bytes of LDS holding stream bytes
// [lo, hi) (a-coordinates, lo 16-byte aligned).  A fill is a series of coalesced
// dwordx4 loads of the whole wave (1 KiB each) that are all in flight together --
// one memory round trip -- after which run headers and bit-packed fields are
// read from LDS.  The region is private to the wave: LDS operations of one wave
// execute in order, so no barrier is needed between the fill and the reads.
struct Window {
    uint32_t lo, hi;    // resident a-offsets [lo, hi)
    uint8_t *lds;       // this wave's region
    uint32_t cap;       // its size in bytes (a multiple of 16, at least 2048)
    uint32_t bit_mask;  // 8 * (the power of two >= cap) - 1: keeps any bit position inside the workgroup's LDS

    // makes [from & ~15, ...) resident: `need` bytes from `from`, as far as the region goes
    __device__ __forceinline__ void fill(__amdgpu_buffer_rsrc_t rsrc, uint32_t from, uint32_t need, uint32_t lane) {
        lo = from & ~15u;
        const uint32_t bytes = min((need + (from & 15u) + 1023u) & ~1023u, cap);
        for (uint32_t base = 0; base < bytes; base += 4096u) {  // four pieces at a time
            u32x4 v[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t at = base + i * 1024u + lane * 16u;
                if (at < bytes) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lo + at, 0, 0);
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t at = base + i * 1024u + lane * 16u;
                if (at < bytes) *(u32x4 *)(lds + at) = v[i];
            }
        }
        hi = lo + bytes;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ bool holds(uint32_t a_lo, uint32_t a_hi) const { return a_lo >= lo && a_hi <= hi; }
    __device__ __forceinline__ uint32_t dword(uint32_t a) const { return *(const uint32_t *)(lds + (a - lo)); }
};

// One W-bit field at LDS bit position `bit` (LSB-first, src/encoding/bitpacked/unpack.rs:40-67).
template <int W>
__device__ __forceinline__ uint32_t field_at(const uint8_t *lds, uint32_t bit) {
    const uint32_t *src = (const uint32_t *)(lds + ((bit >> 3) & ~3u));
    const uint32_t x = __builtin_amdgcn_alignbit(src[1], src[0], bit & 31u);
    constexpr uint32_t mask = W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u);
    return x & mask;
}

// Keeps a value in a register at this point of the program: a load that produces it has been issued, and is
// waited for, here and not inside a later branch.
__device__ __forceinline__ void pin(uint32_t &x) { asm volatile("" : "+v"(x)); }

// The dwords that hold four consecutive W-bit fields starting at LDS bit position `bit`, read now, extracted later
// (fields4_at in two halves, so that the reads of several chunks can be in flight together).
template <int W>
struct FieldWords {
    static constexpr int NY = (4 * W + 31) / 32;
    uint32_t x[NY + 1];
    uint32_t s;
    __device__ __forceinline__ void pin_all() {  // from the dword read last to the one read first: one wait
#pragma unroll
        for (int i = NY; i >= 0; --i) pin(x[i]);
    }
    __device__ __forceinline__ u32x4 fields4() const {
        uint32_t y[NY];
#pragma unroll
        for (int i = 0; i < NY; ++i) y[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], s);
        constexpr uint32_t mask = W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lo = (i * W) >> 5;
            const int shf = (i * W) & 31;
            if (shf + W <= 32)
                o[i] = W == 32 ? y[lo] : ((y[lo] >> shf) & mask);
            else
                o[i] = __builtin_amdgcn_alignbit(y[lo + 1 < NY ? lo + 1 : lo], y[lo], shf) & mask;
        }
        const u32x4 res = {o[0], o[1], o[2], o[3]};
        return res;
    }
};
template <int W>
__device__ __forceinline__ FieldWords<W> load_fields(const uint8_t *lds, uint32_t bit) {
    FieldWords<W> f;
    const uint32_t *src = (const uint32_t *)(lds + ((bit >> 3) & ~3u));
#pragma unroll
    for (int i = 0; i <= FieldWords<W>::NY; ++i) f.x[i] = src[i];
    f.s = bit & 31u;
    return f;
}

// Four consecutive W-bit fields starting at LDS bit position `bit`: ceil(4W / 32) + 1 dword reads, one funnel
// shift per dword to bring the first field to bit 0, then one extraction per field at a constant position.
template <int W>
__device__ __forceinline__ u32x4 fields4_at(const uint8_t *lds, uint32_t bit) {
    constexpr int NY = (4 * W + 31) / 32;
    const uint32_t *src = (const uint32_t *)(lds + ((bit >> 3) & ~3u));
    const uint32_t s = bit & 31u;
    uint32_t x[NY + 1];
#pragma unroll
    for (int i = 0; i <= NY; ++i) x[i] = src[i];
    uint32_t y[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) y[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], s);
    constexpr uint32_t mask = W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u);
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lo = (i * W) >> 5;
        const int shf = (i * W) & 31;
        if (shf + W <= 32)
            o[i] = W == 32 ? y[lo] : ((y[lo] >> shf) & mask);
        else
            o[i] = __builtin_amdgcn_alignbit(y[lo + 1 < NY ? lo + 1 : lo], y[lo], shf) & mask;
    }
    const u32x4 res = {o[0], o[1], o[2], o[3]};
    return res;
}

// ---- direct-from-memory extraction (bitunpack_kernel: no headers, no staging) -------------

// Four consecutive values starting at run-internal index v, straight from the buffer.
template <int W>
__device__ __forceinline__ u32x4 extract4(__amdgpu_buffer_rsrc_t rsrc, uint32_t payload, uint32_t v) {
    constexpr int NY = (4 * W + 31) / 32;
    const uint32_t byte_off = payload + (v >> 3) * (uint32_t)W;
    const uint32_t bit_in = (byte_off & 3u) * 8u + (v & 7u) * (uint32_t)W;
    const uint32_t d = (byte_off >> 2) + (bit_in >> 5);
    const uint32_t s = bit_in & 31u;
    uint32_t x[NY + 1];
#pragma unroll
    for (int i = 0; i <= NY; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (d + i) * 4u, 0, 0);
    uint32_t y[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) y[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], s);
    constexpr uint32_t mask = W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u);
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lo = (i * W) >> 5;
        const int shf = (i * W) & 31;
        if (shf + W <= 32) {
            o[i] = W == 32 ? y[lo] : ((y[lo] >> shf) & mask);
        } else {
            o[i] = __builtin_amdgcn_alignbit(y[lo + 1 < NY ? lo + 1 : lo], y[lo], shf) & mask;
        }
    }
    const u32x4 res = {o[0], o[1], o[2], o[3]};
    return res;
}

__device__ __forceinline__ uint32_t extract1(__amdgpu_buffer_rsrc_t rsrc, uint32_t payload, uint32_t v, uint32_t w) {
    const uint32_t byte_off = payload + (v >> 3) * w;
    const uint32_t bit_in = (byte_off & 3u) * 8u + (v & 7u) * w;
    const uint32_t d = (byte_off >> 2) + (bit_in >> 5);
    const uint32_t s = bit_in & 31u;
    const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(rsrc, d * 4u, 0, 0);
    const uint32_t hi = __builtin_amdgcn_raw_buffer_load_b32(rsrc, d * 4u + 4u, 0, 0);
    const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, s);
    return w == 32 ? x : (x & ((1u << w) - 1u));
}

template <int W>
__device__ __forceinline__ void emit_bp(__amdgpu_buffer_rsrc_t rsrc, uint32_t payload, uint32_t v_begin,
                                        out_u32 *out, uint32_t count, uint32_t lane) {
    const uint32_t head = (4u - (uint32_t)(((uintptr_t)out >> 2) & 3u)) & 3u;
    const uint32_t h = head < count ? head : count;
    if (lane < h) out[lane] = extract1(rsrc, payload, v_begin + lane, W);
    const uint32_t groups = (count - h) >> 2;
    for (uint32_t g = lane; g < groups; g += 64u)
        store4(extract4<W>(rsrc, payload, v_begin + h + 4u * g), out + h + 4u * g);
    const uint32_t tail = h + 4u * groups;
    if (lane < count - tail) out[tail + lane] = extract1(rsrc, payload, v_begin + tail + lane, W);
}

__device__ __forceinline__ void emit_bp_dispatch(uint32_t w, __amdgpu_buffer_rsrc_t rsrc, uint32_t payload,
                                                 uint32_t v_begin, out_u32 *out, uint32_t count, uint32_t lane) {
    switch (w) {
#define PQ2_W(N)                                             \
    case N:                                                  \
        emit_bp<N>(rsrc, payload, v_begin, out, count, lane); \
        break;
        PQ2_W(1) PQ2_W(2) PQ2_W(3) PQ2_W(4) PQ2_W(5) PQ2_W(6) PQ2_W(7) PQ2_W(8)
        PQ2_W(9) PQ2_W(10) PQ2_W(11) PQ2_W(12) PQ2_W(13) PQ2_W(14) PQ2_W(15) PQ2_W(16)
        PQ2_W(17) PQ2_W(18) PQ2_W(19) PQ2_W(20) PQ2_W(21) PQ2_W(22) PQ2_W(23) PQ2_W(24)
        PQ2_W(25) PQ2_W(26) PQ2_W(27) PQ2_W(28) PQ2_W(29) PQ2_W(30) PQ2_W(31) PQ2_W(32)
#undef PQ2_W
        default:
            break;
    }
}

// ---- the chase loop of the header walk (parse_page in pq2_kernels.hip, ring_walk in pq2_ring.hip) ------------
// One pass per run: the header at `pos` (already requested from LDS) gives the next header's position, whose
// 8 bytes are requested at once; (position, bytes) of the current run are parked in lane `m` of r_pos / r_lo /
// r_hi.  Runs with headers of one or two bytes that end at or before `chase_limit` keep the loop going; it stops
// behind the record that does not (and behind the 64th).  The caller provides the variables it names: pos, m,
// r_pos, r_lo, r_hi, lo, hi, groups, t, bytes, nm1, chk, addr, exec_all, m0_kept, w, rb, chase_limit and voff
// (LDS address of stream byte 0 as far as this stretch of the walk is concerned).  HI_READ / HI_PARK: the instructions that move the second four
// bytes behind a header into the record, or nothing.  The statement uses M0 (the lane select of v_writelane) and
// narrows EXEC to lane 0; it puts both back before it ends, so neither is named as clobbered.
// (`m` is kept minus 64 inside: the 64th increment sets the carry, and its low six bits, which select the lane of
// v_writelane, are those of the record's number)
#define PQ2_CHASE_RUNS "64"
#define PQ2_CHASE(HI_READ, HI_PARK) \
asm volatile( \
                "s_mov_b32 %[m0_kept], m0\n\t" \
                "s_mov_b64 %[exec_all], exec\n\t" \
                "s_mov_b64 exec, 1\n\t" \
                "s_sub_u32 %[m], %[m], " PQ2_CHASE_RUNS "\n\t" \
                "v_add_u32 %[addr], %[pos], %[voff]\n\t" \
                "ds_read_b64 v[2:3], %[addr]\n" \
                "1:\n\t" \
                "s_waitcnt lgkmcnt(0)\n\t" \
                "v_readfirstlane_b32 %[lo], v2\n\t" \
                "s_bfe_u32 %[groups], %[lo], 0x60001\n\t" \
                "s_lshr_b32 %[t], %[lo], 2\n\t" \
                "s_and_b32 %[t], %[t], 0x1fc0\n\t" \
                "s_bitcmp1_b32 %[lo], 7\n\t" \
                "s_cselect_b32 %[t], %[t], 0\n\t" \
                "s_or_b32 %[groups], %[groups], %[t]\n\t" \
                "s_mul_i32 %[bytes], %[groups], %[w]\n\t" \
                "s_bitcmp1_b32 %[lo], 0\n\t" \
                "s_cselect_b32 %[bytes], %[bytes], %[rb]\n\t" \
                "s_bitcmp1_b32 %[lo], 7\n\t" \
                "s_addc_u32 %[nm1], %[pos], %[bytes]\n\t" \
                HI_READ \
                "v_add_u32 %[addr], %[nm1], %[voff1]\n\t" \
                "ds_read_b64 v[2:3], %[addr]\n\t" \
                "s_mov_b32 m0, %[m]\n\t" \
                "v_writelane_b32 %[r_pos], %[pos], m0\n\t" \
                "v_writelane_b32 %[r_lo], %[lo], m0\n\t" \
                HI_PARK \
                "s_add_u32 %[pos], %[nm1], 1\n\t" \
                "s_andn2_b32 %[t], 0x8080, %[lo]\n\t" \
                "s_cselect_b32 %[chk], %[nm1], -1\n\t" \
                "s_add_u32 %[m], %[m], 1\n\t" \
                "s_cselect_b32 %[chk], -1, %[chk]\n\t" \
                "s_cmp_le_u32 %[chk], %[limit]\n\t" \
                "s_cbranch_scc1 1b\n\t" \
                "s_waitcnt lgkmcnt(0)\n\t" \
                "s_mov_b64 exec, %[exec_all]\n\t" \
                "s_mov_b32 m0, %[m0_kept]\n\t" \
                "s_add_u32 %[m], %[m], " PQ2_CHASE_RUNS \
                : [pos] "+s"(pos), [m] "+s"(m), [r_pos] "+v"(r_pos), [r_lo] "+v"(r_lo), [r_hi] "+v"(r_hi), \
                  [lo] "=&s"(lo), [hi] "=&s"(hi), [groups] "=&s"(groups), [t] "=&s"(t), [bytes] "=&s"(bytes), \
                  [nm1] "=&s"(nm1), [chk] "=&s"(chk), [addr] "=&v"(addr), [exec_all] "=&s"(exec_all), \
                  [m0_kept] "=&s"(m0_kept) \
                : [w] "s"(w), [rb] "s"(rb), [limit] "s"(chase_limit), [voff] "v"(voff), \
                  [voff1] "v"(voff + 1u) \
                : "v2", "v3", "scc", "memory");

constexpr uint32_t kExplicit = 0x8080u;  // `lo` of a parked record that carries (kind, count, info) itself

// What the chase parked for one run (record of lane i), decoded by lane i: the run's kind, the values it holds
// before any clamp (mod.rs:41-64), the RLE value (not masked to the bit width) or the a-offset of the bit-packed
// payload, and the a-offset of the byte behind an RLE run.  `rb`: bytes of an RLE value, `rb_mask`: the first
// min(4, rb) of them.  Records that carry the explicit marker were decoded by decode_run already.
struct ParkedRun {
    uint32_t kind, count, info, after;
};
__device__ __forceinline__ ParkedRun decode_parked(uint32_t r_pos, uint32_t r_lo, uint32_t r_hi, uint32_t r_info,
                                                   uint32_t rb, uint32_t rb_mask) {
    ParkedRun p;
    const uint32_t two = (r_lo >> 7) & 1u;
    const uint32_t groups = ((r_lo >> 1) & 0x3fu) | (two ? (r_lo >> 2) & 0x1fc0u : 0u);
    const uint32_t rest = two ? __builtin_amdgcn_alignbit(r_hi, r_lo, 16) : __builtin_amdgcn_alignbit(r_hi, r_lo, 8);
    p.kind = (r_lo & 1u) ^ 1u;
    p.count = p.kind == kRle ? groups : groups * 8u;
    p.info = p.kind == kRle ? (rest & rb_mask) : r_pos + 1u + two;
    p.after = r_pos + 1u + two + rb;
    if ((~r_lo & kExplicit) == 0u) {
        p.kind = (r_lo >> 16) & 3u;
        p.count = r_hi;
        p.info = r_info;
        p.after = r_pos + ((r_lo >> 20) & 0xffu);
    }
    return p;
}

}  // namespace pq2
