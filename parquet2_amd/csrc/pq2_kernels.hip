// pq2hip kernels (gfx950 / MI355X).
//
// A batch of hybrid-RLE pages is decoded by two kernels:
//
//   parse_kernel   one wave per page.  Walks the run headers (a serial chain:
//                  the position of a header is only known once the previous
//                  one is decoded) with the stream staged through an LDS ring,
//                  and records, for every 1024-value block of the output,
//                  where in the stream that block starts (Resume).  8 bytes are
//                  written per 4 KiB of output; no run table is materialised.
//   expand_kernel  one wave per 1024-value block, four blocks per workgroup.
//                  Re-decodes the (few) headers that cover its block, starting
//                  from the Resume entry, and writes the values: RLE runs as
//                  splat dwordx4 stores, bit-packed runs through the
//                  per-width extract4<W> bodies.  All HBM traffic of the decode
//                  (payload reads, value stores) happens here.
//
// bitunpack_kernel is the header-less special case (bitpacked::Decoder<u32>).
#include "pq2_device.h"

namespace pq2 {

constexpr uint32_t kHalf = 4096;  // bytes per ring half
constexpr uint32_t kRing = 2 * kHalf;

// ---------------------------------------------------------------------------
// parse_kernel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void parse_kernel(const PageDev *__restrict__ pages, uint32_t n_pages,
                                                   Resume *__restrict__ resume, uint32_t *__restrict__ tile_page,
                                                   int32_t *__restrict__ status) {
    // ring[a & (kRing-1)] holds stream byte a for a in [win_lo, win_lo + kRing)
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];

    const uint32_t lane = threadIdx.x;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const PageDev pg = pages[page];
    const uint32_t n = pg.num_values;
    if (n == 0) {
        if (lane == 0) status[page] = 0;
        return;
    }
    const uint32_t w = pg.num_bits;
    const uint32_t n_blocks = (n + kBlockValues - 1) / kBlockValues;
    Resume *const entries = resume + (size_t)pg.tile_base * kBlocksPerTile;
    // expand_kernel finds its page through tile_page[tile]
    for (uint32_t t = lane; t < (n_blocks + kBlocksPerTile - 1) / kBlocksPerTile; t += 64) tile_page[pg.tile_base + t] = page;

    // fills entries [from, n_blocks) with "stream exhausted"
    auto fill_end = [&](uint32_t from) {
        for (uint32_t b = from + lane; b < n_blocks; b += 64) {
            Resume e;
            e.pos = kEndPos;
            e.skip = 0;
            entries[b] = e;
        }
    };

    if (w == 0 || pg.in_len == 0 || w > 32) {  // decoder.rs:31-37: nothing to iterate; mod.rs:95 pads with zeros
        fill_end(0);
        if (lane == 0) status[page] = 0;
        return;
    }

    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint64_t base16 = pg.in_addr - delta;
    const uint32_t end = delta + pg.in_len;  // a-offset one past the last stream byte
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base16, (end + 3u) & ~3u);

    u32x4 pre[4];              // prefetched half [win_lo + kRing, win_lo + kRing + kHalf)
    uint32_t win_lo = 0;       // a-offset (multiple of kHalf) of the first resident byte
    bool have_window = false;  // false until the first load

    auto load_half = [&](uint32_t half_a, u32x4 *r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, half_a + (uint32_t)i * 1024u + lane * 16u, 0, 0);
    };
    auto store_half = [&](uint32_t half_a, const u32x4 *r) {
        const uint32_t slot = half_a & (kRing - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4 *)&ring[slot + (uint32_t)i * 1024u + lane * 16u] = r[i];
    };
    // makes [pos & ~3, (pos & ~3) + 24) resident: decode_run reads at most 22 bytes from pos
    auto ensure = [&](uint32_t pos) {
        const uint32_t need_hi = (pos & ~3u) + 24u;
        if (have_window && pos >= win_lo && need_hi <= win_lo + kRing) return;
        __syncthreads();  // every earlier ring read of the wave is done before the ring is overwritten
        if (have_window && pos >= win_lo && need_hi <= win_lo + kRing + kHalf) {
            // slide by one half: the prefetched half replaces the oldest one
            store_half(win_lo + kRing, pre);
            win_lo += kHalf;
        } else {
            // first load, or a run that jumped beyond the prefetched half
            win_lo = pos & ~(kHalf - 1);
            u32x4 a[4], b[4];
            load_half(win_lo, a);
            load_half(win_lo + kHalf, b);
            store_half(win_lo, a);
            store_half(win_lo + kHalf, b);
            have_window = true;
        }
        load_half(win_lo + kRing, pre);  // lands while the chase works through the window
        __syncthreads();
    };
    auto rd = [&](uint32_t a) -> uint32_t {
        return __builtin_amdgcn_readfirstlane(*(const uint32_t *)&ring[a & (kRing - 1)]);
    };

    uint32_t pos = delta;
    uint32_t produced = 0;
    uint32_t next_block = 0;  // blocks < next_block have their entry; next_block * 1024 >= produced
    int32_t st = 0;
    while (produced < n) {
        ensure(pos);
        const RunInfo run = decode_run(rd, pos, end, w, n - produced);
        if (run.kind >= kEnd) {
            if (run.kind == kPanic) st = 5;  // PQ2_REFERENCE_WOULD_PANIC
            break;
        }
        if (run.count > 0) {
            const uint32_t last_block = (produced + run.count - 1) / kBlockValues;
            for (uint32_t b = next_block + lane; b <= last_block; b += 64) {
                Resume e;
                e.pos = pos;
                e.skip = b * kBlockValues - produced;
                entries[b] = e;
            }
            next_block = last_block + 1;
        }
        produced += run.count;
        pos = run.next;
    }
    fill_end(next_block);
    if (lane == 0) status[page] = st;
}

// ---------------------------------------------------------------------------
// expand_kernel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 8) void expand_kernel(const PageDev *__restrict__ pages,
                                                     const uint32_t *__restrict__ tile_page,
                                                     const Resume *__restrict__ resume) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kBlocksPerTile * kWinBytes];

    const uint32_t tile = blockIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;

    const PageDev pg = pages[tile_page[tile]];
    const uint32_t n = pg.num_values;
    const uint32_t block = (tile - pg.tile_base) * kBlocksPerTile + wave;
    const uint32_t block_start = block * kBlockValues;
    if (block_start >= n) return;
    const uint32_t block_n = min(kBlockValues, n - block_start);
    const uint32_t w = pg.num_bits;

    const Resume rs = resume[(size_t)tile * kBlocksPerTile + wave];
    uint32_t pos = __builtin_amdgcn_readfirstlane(rs.pos);
    uint32_t skip = __builtin_amdgcn_readfirstlane(rs.skip);

    const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
    const uint32_t end = delta + pg.in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);

    Window win;
    win.lo = win.hi = 0;
    win.lds = stage + wave * kWinBytes;
    auto rd = [&](uint32_t a) -> uint32_t { return __builtin_amdgcn_readfirstlane(win.dword(a)); };

    // values of one bit-packed span that are guaranteed to fit the window together
    const uint32_t max_vals = w ? ((kWinBytes - 64u) / w) * 8u : 0u;

    uint32_t *const out = (uint32_t *)pg.out_addr + block_start;
    uint32_t o = 0;  // values of this block already written
    while (o < block_n) {
        if (pos == kEndPos) {
            emit_rle(0u, out + o, block_n - o, lane);
            break;
        }
        if (!win.holds(pos & ~3u, (pos & ~3u) + 24u))  // header + RLE value; normally one fill serves the block
            win.fill(rsrc, pos, (((block_n - o) * w) >> 3) + 64u, lane);
        const uint32_t run_start = block_start + o - skip;
        const RunInfo run = decode_run(rd, pos, end, w, n - run_start);
        if (run.kind >= kEnd) {  // exhausted (or a panic path, reported by parse_kernel): zeros
            emit_rle(0u, out + o, block_n - o, lane);
            break;
        }
        const uint32_t avail = run.count > skip ? run.count - skip : 0u;
        uint32_t take = min(avail, block_n - o);
        if (run.kind == kRle) {
            if (take > 0) emit_rle(run.value, out + o, take, lane);
            o += take;
        } else {
            while (take > 0) {  // normally one pass; more only when the span's bytes exceed the window
                const uint32_t chunk = min(take, max_vals);
                const uint32_t b_lo = run.payload + (skip >> 3) * w;
                const uint32_t b_hi = run.payload + ((skip + chunk + 7u) >> 3) * w + 8u;
                if (!win.holds(b_lo & ~3u, b_hi)) win.fill(rsrc, b_lo, b_hi - b_lo + 16u, lane);
                emit_bp_lds_dispatch(w, win.lds, span_bit0(win, run.payload, skip, w), out + o, chunk, lane);
                o += chunk;
                skip += chunk;
                take -= chunk;
            }
        }
        skip = 0;
        pos = run.next;
    }
}

// ---------------------------------------------------------------------------
// bitunpack_kernel: `length` values of width w from a bit-packed buffer
// (bitpacked::Decoder<u32>, src/encoding/bitpacked/decode.rs:28-86).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bitunpack_kernel(uint64_t in_addr, uint32_t in_len, uint32_t w,
                                                        uint32_t length, uint32_t *__restrict__ out) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t delta = (uint32_t)(in_addr & 15u);
    const uint32_t end = delta + in_len;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(in_addr - delta, (end + 3u) & ~3u);
    const uint32_t n_blocks = (length + kBlockValues - 1) / kBlockValues;
    for (uint32_t block = blockIdx.x * kBlocksPerTile + wave; block < n_blocks;
         block += gridDim.x * kBlocksPerTile) {
        const uint32_t start = block * kBlockValues;
        const uint32_t cnt = min(kBlockValues, length - start);
        emit_bp_dispatch(w, rsrc, delta, start, out + start, cnt, lane);
    }
}

}  // namespace pq2

// ---- launchers (called from pq2_api.cpp) ----------------------------------------
extern "C" hipError_t pq2_launch_parse(const void *pages, uint32_t n_pages, void *resume, uint32_t *tile_page,
                                       int32_t *status, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::parse_kernel, dim3(n_pages), dim3(64), 0, stream, (const pq2::PageDev *)pages, n_pages,
                       (pq2::Resume *)resume, tile_page, status);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_expand(const void *pages, const uint32_t *tile_page, uint32_t n_tiles,
                                        const void *resume, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::expand_kernel, dim3(n_tiles), dim3(256), 0, stream, (const pq2::PageDev *)pages,
                       tile_page, (const pq2::Resume *)resume);
    return hipGetLastError();
}

extern "C" hipError_t pq2_launch_bitunpack(const void *d_packed, uint32_t packed_len, uint32_t num_bits,
                                           uint32_t length, uint32_t *d_out, hipStream_t stream) {
    if (length == 0) return hipSuccess;
    const uint32_t n_tiles = (length + pq2::kTileValues - 1) / pq2::kTileValues;
    const uint32_t grid = n_tiles < 16384u ? n_tiles : 16384u;
    hipLaunchKernelGGL(pq2::bitunpack_kernel, dim3(grid), dim3(256), 0, stream, (uint64_t)(uintptr_t)d_packed,
                       packed_len, num_bits, length, d_out);
    return hipGetLastError();
}
