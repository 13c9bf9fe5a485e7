// Host side of libpq2hip: the C ABI of include/pq2hip.h.
//
// A context owns, for one GPU: small device arrays for the per-batch page
// descriptors / tile prefix / per-page status, the Resume table (16 bytes per
// 1024 output values), a pinned staging buffer for the descriptor upload and,
// for the host-buffer entry points, device arenas and copy streams.  All of
// them grow on demand and are reused, so a steady stream of batches performs
// no allocation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pq2hip.h"

extern "C" {
hipError_t pq2_launch_parse(const void *pages, uint32_t n_pages, void *resume, uint32_t *tile_page, uint32_t *run_pos,
                            int32_t *status, void *state, uint32_t round, uint32_t max_bits, hipStream_t stream);
hipError_t pq2_launch_expand(const void *pages, const uint32_t *tile_page, uint32_t tile_start, uint32_t n_tiles,
                             const void *resume, const uint32_t *run_pos, uint32_t round, uint32_t max_bits,
                             const uint32_t *tile_mask, const void *dicts, uint32_t value_words, int32_t *status,
                             hipStream_t stream);
uint32_t pq2_expand_dict_max_bits();
hipError_t pq2_launch_bitunpack(const void *d_packed, uint32_t packed_len, uint32_t num_bits, uint32_t length,
                                uint32_t *d_out, hipStream_t stream);
hipError_t pq2_launch_decode(const void *pages, uint32_t n_pages, void *resume, uint32_t *run_pos, int32_t *status,
                             void *state, uint32_t *cu_tickets, uint32_t max_bits, const uint32_t *tile_mask,
                             hipStream_t stream);
uint32_t pq2_cu_ticket_bytes();
hipError_t pq2_launch_gather(const void *pages, uint32_t n_pages, uint32_t n_tiles, uint32_t value_words,
                             uint64_t max_dict_bytes, int32_t *status, hipStream_t stream);
uint32_t pq2_gather_tile_values();
hipError_t pq2_launch_front(void *streams, const void *raw, uint32_t n_pages, int32_t *front_status, hipStream_t stream);
hipError_t pq2_launch_select(const void *pieces, uint32_t n_pieces, hipStream_t stream);
hipError_t pq2_launch_filtered(const void *pieces, uint32_t n_pieces, uint32_t *counts, hipStream_t stream);
hipError_t pq2_launch_optional(const void *pages, uint32_t n_pages, uint32_t value_words, int32_t *status,
                               uint32_t seg_rows, uint32_t n_segs, uint32_t *seg_valid,
                               hipStream_t stream);
hipError_t pq2_launch_optional_bits(const void *pages, uint32_t n_pages, uint32_t value_words, int32_t *status,
                                    const int32_t *level_status, uint64_t dict_bytes, hipStream_t stream);
hipError_t pq2_launch_bitunpack64(const void *d_packed, uint32_t packed_len, uint32_t num_bits, uint32_t length,
                                  uint64_t *d_out, hipStream_t stream);
hipError_t pq2_launch_delta(const void *pages, uint32_t n_pages, void *results, hipStream_t stream);
int32_t pq2_noalloc_setup(int32_t device, uint8_t **out);  // pq2_noalloc.hip
hipError_t pq2_launch_ring(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets, uint32_t epoch,
                           uint32_t max_bits, const uint32_t *tile_mask, const uint32_t *go, hipEvent_t begin, hipEvent_t end,
                           hipStream_t stream);  // pq2_ring.hip
hipError_t pq2_launch_density(const void *pages, uint32_t n_pages, void *seen, float scout_us, float ring_fixed_us,
                              float per_header_us, float values_us, float turns, uint32_t *go, uint32_t *host_words,
                              uint32_t seq, hipStream_t stream);  // pq2_density.hip
uint32_t pq2_ring_ticket_bytes();
uint32_t pq2_ring_dict_max_bits();  // pq2_ring_dict.hip
uint32_t pq2_ring_dict_max_bytes();
hipError_t pq2_launch_ring_bits(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets, uint32_t epoch,
                                uint32_t *bit_meta, hipStream_t stream);  // pq2_ring.hip
hipError_t pq2_launch_ring_dict(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets,
                                uint32_t epoch, const uint32_t *tile_mask, const void *dicts, uint32_t value_words, uint32_t max_dict_bytes,
                                const void *nulls, const uint32_t *go, hipStream_t stream);
uint32_t pq2_ring_nullable_max_dict_bytes();
uint32_t pq2_ring_nullable_max_bits();
hipError_t pq2_launch_rowdir(const void *nulls, uint32_t n_pages, hipStream_t stream);  // pq2_optional.hip
uint32_t pq2_rowdir_step();
hipError_t pq2_launch_nullable_status(int32_t *status, const int32_t *level_status, uint32_t n_pages, hipStream_t stream);
uint32_t pq2_scout_tile_bytes();  // pq2_scout.hip
size_t pq2_scout_table_bytes(uint64_t n_stiles);
size_t pq2_scout_words_bytes(uint64_t n_stiles, uint64_t n_pages);
size_t pq2_scout_rounds_at(uint64_t n_stiles, uint64_t n_pages);
hipError_t pq2_launch_scout(const void *pages, const void *spages, uint32_t n_pages, uint32_t n_stiles, void *table,
                            void *words, void *resume, uint32_t *tile_page, uint32_t *run_pos, int32_t *status, void *state,
                            uint32_t max_bits, hipStream_t stream);
}

namespace {

constexpr uint32_t kTileValues = 4096;  // 4 blocks of PQ2_BLOCK_VALUES
constexpr uint32_t kBlocksPerTile = 4;
constexpr uint32_t kRounds = 4;          // must match pq2::kRounds
constexpr uint64_t kMinRoundedTiles = 16384;  // batches below 64 Mi values are decoded in one round
constexpr size_t kParseStateBytes = 16;  // sizeof(pq2::ParseState)
constexpr uint64_t kFusedMinPages = 64;  // batches of at least this many pages take one workgroup per page ...
constexpr uint64_t kScoutMaxPages = 96;  // ... unless they have fewer than this many: then the scout kernels walk the headers

struct ScoutPage {  // must match pq2::ScoutPage
    uint32_t tile_base, n_tiles;
};

struct PageDev {  // must match pq2::PageDev
    uint64_t in_addr;
    uint64_t out_addr;
    uint64_t run_base;
    uint32_t in_len;
    uint32_t num_values;
    uint32_t num_bits;
    uint32_t round_tile[kRounds];
    uint32_t tile_base[kRounds];
    uint32_t reserved;
};
static_assert(sizeof(PageDev) == 72, "PageDev layout");

struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct GatherPage {  // must match pq2::GatherPage
    uint64_t idx_addr, out_addr, dict_addr;
    uint32_t dict_len, num_values, first_tile, reserved;
};
static_assert(sizeof(GatherPage) == 40, "GatherPage layout");

struct OptionalPage {  // must match pq2::OptionalPage
    uint64_t level_addr, idx_addr, out_addr, validity_addr, dict_addr;
    uint32_t rows, idx_values, dict_len, max_level, idx_status_at, first_seg;
};
constexpr uint64_t kOptionalSegRows = 65536;     // batches of few pages: one workgroup of optional_kernel per so many rows
constexpr uint64_t kOptionalWholePages = 256;    // from this many pages on: one workgroup per page
static_assert(sizeof(OptionalPage) == 64, "OptionalPage layout");

struct SelectPiece {  // must match pq2::SelectPiece
    uint64_t src_addr, dst_addr;
    uint32_t count, reserved;
};
static_assert(sizeof(SelectPiece) == 24, "SelectPiece layout");
struct FilterPiece {  // must match pq2::FilterPiece
    uint64_t src_addr, dst_addr;
    uint32_t count, counter;
};
static_assert(sizeof(FilterPiece) == 24, "FilterPiece layout");
constexpr uint64_t kSelectPieceValues = 16384;

struct RawPageDev {  // must match pq2::RawPageDev
    uint64_t buf_addr;
    uint32_t buf_len, what;
};
static_assert(sizeof(RawPageDev) == 16, "RawPageDev layout");

struct DeltaPage {  // must match pq2::DeltaPage
    uint64_t in_addr, out_addr;
    uint32_t in_len, cap;
};
static_assert(sizeof(DeltaPage) == 24, "DeltaPage layout");
struct DeltaResult {  // must match pq2::DeltaResult
    int32_t status;
    uint32_t consumed;
    uint64_t num_values;
};
static_assert(sizeof(DeltaResult) == 16, "DeltaResult layout");

// What pq2_decode_selected_batch_device adds to a batch: one bit per tile of every page that says whether the
// tile holds a selected row (`at[i]`: the page's first word in `words`, plus one; 0: every tile).
struct TileMask {
    std::vector<uint32_t> words;
    std::vector<uint32_t> at;
};

// What pq2_decode_raw_pages_v1_device adds to a batch: front_kernel patches the uploaded stream descriptors.
// What pq2_decode_dict_batch_device adds to a batch when the dictionary lookup happens inside ring_kernel: the
// per-page records (device) and the dictionaries' size.
struct FusedDict {
    const void *d_dicts;
    uint32_t value_words, max_dict_bytes;
    const void *d_nulls = nullptr;  // RingNullable records: the values go to the rows of flat nullable pages
};

// What pq2_decode_dict_batch_device adds to a batch that does not take ring_kernel: where expand_kernel carries
// the values (the scout kernels or parse_kernel before it, one round) it looks the indices up itself and writes
// the dictionaries' values to `d_values` (at out_off * value_words * 4 bytes) instead of indices to d_out; the
// context's `expand_looked_up` says whether that happened.
struct ExpandLookup {
    const void *d_dicts;  // gather descriptors (device)
    uint32_t value_words;
    void *d_values;
};

struct RingNullable {  // must match pq2::RingNullable
    uint64_t bitmap_addr, validity_addr, out_addr, rowdir_addr;
    uint32_t rows, idx_values;
};
static_assert(sizeof(RingNullable) == 40, "RingNullable layout");

// What pq2_decode_validity_bitmap_batch_device adds to a batch: the pages come out as validity bitmaps (out_off: the
// byte at which a page's bitmap begins) and ring_kernel reports two words per page, rows and set bits.
struct BitmapOut {
    uint64_t bitmap_bytes;
    uint32_t *d_meta;
};

struct FrontJob {
    const void *d_raw;
    uint32_t n_pages;
    int32_t *d_front_status;
};

}  // namespace

struct pq2_ctx {
    int device = 0;
    std::string last_error;
    // per-batch metadata: [PageDev x n] in one pinned block, mirrored on the device
    void *h_meta = nullptr;
    size_t h_meta_cap = 0;
    hipEvent_t meta_copied = nullptr;  // the previous upload has left h_meta
    bool meta_in_flight = false;
    DevBuf d_meta, d_status, d_resume, d_tile_page, d_run_pos, d_state;
    DevBuf d_tickets;  // per-compute-unit counters: decode_kernel's (zeroed once; only their low bits matter), then ring_kernel's
    uint32_t ring_epoch = 0;  // ring_kernel launches so far: which half of its counters a launch counts in
    uint64_t pages_total = 0;  // pages handed to pq2_decode_batch_device / _host on this context so far
    DevBuf d_scout_table, d_scout_words;  // the scout kernels' tables and per-tile words
    uint64_t last_scout_tiles = 0;        // stream tiles of the last batch, if the scout kernels walked it
    bool last_scout = false;
    // pq2_decode_dict_batch_device: the decoded indices, and the per-page gather descriptors (page-locked + device)
    DevBuf d_indices, d_gather, d_seg_valid;
    DevBuf d_bit_meta;  // pq2_decode_validity_bitmap_batch_device: rows and set bits of every page
    bool fused_declined = false;  // enqueue_batch: the density probe sent a batch with the lookup inside ring_kernel elsewhere
    bool expand_looked_up = false;  // enqueue_batch: expand_kernel has written dictionary values (ExpandLookup)
    DevBuf d_rowdir;    // pq2_decode_optional_dict_batch_device: the row directories of flat nullable pages (rowdir_kernel)
    // density_kernel (pq2_density.hip): its two counters and the verdict word on the device, the verdict as the host
    // reads it (page-locked, written by the device), the event behind the kernel, a count of its launches
    DevBuf d_density;
    uint32_t *h_density = nullptr;
    hipEvent_t density_done = nullptr;
    uint32_t density_seq = 0;
    bool probe_density = true;  // PQ2_DENSITY_PROBE=0: the descriptors alone decide (A/B runs)
    bool optional_bits = true, optional_fused = true;  // PQ2_OPTIONAL_BITS=0, PQ2_OPTIONAL_FUSED=0 (A/B runs, tests)
    uint32_t last_density_headers = 0;  // largest estimated header count of a page, if the last batch was probed
    void *h_gather = nullptr;
    size_t h_gather_cap = 0;
    hipEvent_t gather_copied = nullptr;
    bool gather_in_flight = false;
    // Where the rounds of a page end, as fractions of its tiles (the last round ends with the page).  The parts
    // grow because round r + 1 is parsed while round r is expanded, and parsing is the faster of the two; the
    // first part is small because nothing hides its parse.  Tuning knobs: PQ2_ROUND_SPLIT="a,b,c",
    // PQ2_MIN_ROUNDED_TILES.
    double round_split[kRounds - 1] = {0.25, 1.0, 1.0};
    uint64_t min_rounded_tiles = kMinRoundedTiles;
    // How a batch is scheduled.  kPerPage: decode_kernel, one workgroup per page, the header walk and the
    // expansion of a page side by side (batches with enough pages to fill the GPU that way).  kPerTile:
    // parse_kernel + expand_kernel, one wave per 4096 values, in rounds (few, large pages).  PQ2_MODE=page|tile
    // forces one of them.
    // kRing: ring_kernel, one workgroup per page like kPerPage, but the stream is staged once through an LDS ring
    // that the walking wave fills and all four waves read, and nothing passes through device memory on its way
    // from the walk to the expansion (PQ2_MODE=ring; what kAuto picks for batches of many pages).
    // kScout: the scout kernels walk the headers of all pages in parallel over tiles of the streams, then
    // expand_kernel in one round (PQ2_MODE=scout; what kAuto picks for batches of few pages).
    enum Mode { kAuto, kPerPage, kPerTile, kRing, kScout } mode = kAuto;
    bool one_expand_stream = true;  // the expand launches of a batch follow each other (PQ2_EXPAND_STREAMS=2: they take turns on two streams)
    bool split_given = false;       // PQ2_ROUND_SPLIT was set
    // rounds: parse_kernel of round r + 1 runs on parse_stream while expand_kernel of round r runs on expand_stream
    // (two of them, taken in turn: the expand launches of one batch do not depend on each other, so the next one
    // starts while the last waves of the previous one drain)
    hipStream_t parse_stream = nullptr, expand_stream[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, parsed[4] = {nullptr, nullptr, nullptr, nullptr}, parse_joined = nullptr,
               expand_joined[2] = {nullptr, nullptr};
    std::vector<int32_t> host_status;  // statuses decided on the host (bad num_bits)
    uint64_t last_n_pages = 0;
    const char *last_schedule = "";  // the kernel that carried the values of the last batch
    hipStream_t last_stream = nullptr;
    bool busy = false;  // a batch may still be running on last_stream
    // optional kernel timing (pq2_ctx_enable_profiling): events around parse and expand of every batch
    bool profiling = false;
    struct Stamp {  // one kernel launch
        hipEvent_t begin, end;
        bool expand;
    };
    std::vector<Stamp> stamps;
    uint64_t stamped_batches = 0;
    // host-buffer path: device arenas; large batches go in chunks through three lanes (child contexts with a
    // stream each), so that the upload of one chunk, the kernels of another and the download of a third overlap
    DevBuf d_in, d_out;
    pq2_ctx *lanes[3] = {nullptr, nullptr, nullptr};
    int32_t *h_status = nullptr;  // page-locked: a copy into the caller's (pageable) array would block every chunk
    size_t h_status_cap = 0;
    hipStream_t lane_stream = nullptr;  // of a child context
    // pq2_decode_optional_dict_batch_device: the child context that turns the level streams into validity bitmaps on a
    // stream of its own while this one decodes the indices, and the events between the two streams
    pq2_ctx *bits_lane = nullptr;
    hipEvent_t bits_fork = nullptr, bits_joined = nullptr;
};

namespace {

int32_t fail(pq2_ctx *ctx, int32_t code, const std::string &msg) {
    if (ctx) ctx->last_error = msg;
    return code;
}

int32_t hip_fail(pq2_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, PQ2_DEVICE_ERROR, std::string(what) + ": " + hipGetErrorString(e));
}

#define PQ2_HIP(ctx, call)                                    \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int32_t reserve(pq2_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return PQ2_OK;
    size_t cap = std::max(bytes, b.cap + b.cap / 2);
    cap = (cap + 4095) & ~size_t(4095);
    if (b.ptr) PQ2_HIP(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    PQ2_HIP(ctx, hipMalloc(&b.ptr, cap));
    b.cap = cap;
    return PQ2_OK;
}

int32_t reserve_pinned(pq2_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_meta_cap) return PQ2_OK;
    if (ctx->meta_in_flight) {
        PQ2_HIP(ctx, hipEventSynchronize(ctx->meta_copied));
        ctx->meta_in_flight = false;
    }
    const size_t cap = (std::max(bytes, ctx->h_meta_cap * 2) + 4095) & ~size_t(4095);  // geometric growth
    if (ctx->h_meta) PQ2_HIP(ctx, hipHostFree(ctx->h_meta));
    ctx->h_meta = nullptr;
    ctx->h_meta_cap = 0;
    PQ2_HIP(ctx, hipHostMalloc(&ctx->h_meta, cap, hipHostMallocDefault));
    ctx->h_meta_cap = cap;
    return PQ2_OK;
}

// The small per-batch descriptor arrays of the entry points that add a kernel behind the decode (gather, nullable
// pages, selected rows, raw pages) go through one page-locked block and one device block of the context.
// stage_block: the page-locked block, at least `bytes` large and no longer read by an earlier upload.
int32_t stage_block(pq2_ctx *ctx, size_t bytes, size_t device_bytes) {
    if (ctx->gather_in_flight) {  // the previous batch's upload still reads the page-locked block
        PQ2_HIP(ctx, hipEventSynchronize(ctx->gather_copied));
        ctx->gather_in_flight = false;
    }
    if (bytes > ctx->h_gather_cap) {
        const size_t cap = (std::max(bytes, ctx->h_gather_cap * 2) + 4095) & ~size_t(4095);  // geometric growth
        if (ctx->h_gather) PQ2_HIP(ctx, hipHostFree(ctx->h_gather));
        ctx->h_gather = nullptr;
        ctx->h_gather_cap = 0;
        PQ2_HIP(ctx, hipHostMalloc(&ctx->h_gather, cap, hipHostMallocDefault));
        ctx->h_gather_cap = cap;
    }
    if (!ctx->gather_copied) PQ2_HIP(ctx, hipEventCreateWithFlags(&ctx->gather_copied, hipEventDisableTiming));
    return reserve(ctx, ctx->d_gather, std::max(bytes, device_bytes));
}
// upload_block: the first `bytes` of the page-locked block to the device block, on `stream`.
int32_t upload_block(pq2_ctx *ctx, size_t bytes, hipStream_t stream) {
    PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_gather.ptr, ctx->h_gather, bytes, hipMemcpyHostToDevice, stream));
    PQ2_HIP(ctx, hipEventRecord(ctx->gather_copied, stream));
    ctx->gather_in_flight = true;
    return PQ2_OK;
}

// Does a batch of `n_pages` pages whose streams' addresses the host knows go through ring_kernel?
bool takes_ring(const pq2_ctx *ctx, uint64_t n_pages) {
    return ctx->mode == pq2_ctx::kRing || (ctx->mode == pq2_ctx::kAuto && n_pages >= kScoutMaxPages);
}

// kAuto, a batch of index or level pages: the scout kernels (true) or ring_kernel?  Below kScoutMaxPages pages the
// scout kernels (streams of a few hundred bytes aside, see below).  From there on ring_kernel, whose time is that of its slowest page (one workgroup a page; with
// more pages than workgroup slots the pages' walks hide behind each other's expansion) -- unless the batch is small
// enough for the scout kernels, whose time follows the bytes of the whole batch, to be done before one such page is:
// about a hundred to three hundred large pages.  Both sides of the comparison are what was measured on an MI355X
// for pages of long runs (microseconds; the scout kernels: 40 + 2.8 per MB of streams + the values at 3.5 TB/s;
// ring_kernel: 30 + the largest page at 0.57 per KB of stream or at 9.5 GB/s of values, whichever is more).  Pages of
// very short runs take ring_kernel far longer than that, which only makes the choice of the scout kernels righter;
// where such pages come in larger batches the caller has to say so (pq2_ctx_set_schedule).
constexpr uint64_t kRingSlots = 1024;  // workgroups of ring_kernel that are resident at a time
struct BatchBytes {
    double in_total = 0, out_total = 0, in_max = 0, out_max = 0;
};
BatchBytes batch_bytes(const pq2_page_desc *descs, uint64_t n_pages) {
    BatchBytes b;
    for (uint64_t i = 0; i < n_pages; ++i) {
        const double in = double(descs[i].in_len), out = 4.0 * double(descs[i].num_values);
        b.in_total += in, b.out_total += out;
        b.in_max = std::max(b.in_max, in), b.out_max = std::max(b.out_max, out);
    }
    return b;
}
bool auto_takes_scout(const pq2_page_desc *descs, uint64_t n_pages) {
    if (n_pages > kRingSlots) return false;
    const BatchBytes bb = batch_bytes(descs, n_pages);
    const double in_total = bb.in_total, out_total = bb.out_total, in_max = bb.in_max, out_max = bb.out_max;
    if (n_pages < kScoutMaxPages) {
        // ... except where every stream is so short that one wave walks it, whatever it holds (a header every two
        // bytes, 0.06 us each), before the scout kernels' seven launches are through (70 us): 64 pages of 65 536
        // definition levels take 55 us in ring_kernel and 76 us the other way, 90 pages of 16 384: 34 and 74
        constexpr double kShortStream = 1024;
        if (in_max > kShortStream) return true;
        const double scout_us = 70.0 + in_total * 2.8e-6 + out_total / 3.5e6;
        const double ring_us = 30.0 + std::max(in_max * 0.03, out_max / 9.5e3);
        return scout_us < ring_us;
    }
    const double scout_us = 40.0 + in_total * 2.8e-6 + out_total / 3.5e6;
    const double ring_us = 30.0 + std::max(in_max * 0.57e-3, out_max / 9.5e3);
    return scout_us * 1.1 < ring_us;
}

// Validates the batch, builds the device descriptors in pinned memory, uploads
// them and enqueues parse + expand on `stream`.
int32_t enqueue_batch(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs, uint64_t n_pages,
                      uint32_t *d_out, uint64_t out_values, hipStream_t stream, const FrontJob *front = nullptr,
                      uint32_t min_max_bits = 0, const TileMask *mask = nullptr, const FusedDict *fused = nullptr,
                      const BitmapOut *bits = nullptr, const ExpandLookup *lookup = nullptr,
                      hipEvent_t before_kernels = nullptr) {
    ctx->expand_looked_up = false;
    if (n_pages > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many pages in one batch");
    if (n_pages > 0 && !descs) return fail(ctx, PQ2_INVALID_PARAMETER, "descs is NULL");
    // the device-side metadata and the Resume table are shared by consecutive batches: stream order
    // protects them on one stream, a change of stream needs the previous batch to have finished
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    // Until this batch is accepted and enqueued there is no batch whose statuses could be fetched: a call that
    // fails below leaves pq2_fetch_status nothing to report (n_pages = 0) instead of an earlier batch's array.
    ctx->last_n_pages = 0;
    ctx->last_scout = false;
    ctx->host_status.assign(n_pages, PQ2_OK);
    if (n_pages == 0) {
        ctx->last_stream = stream;
        return PQ2_OK;
    }

    // the page descriptors and, behind them, the tile mask of a selected-rows batch
    const size_t pages_bytes = n_pages * sizeof(PageDev);
    const size_t mask_bytes = mask ? mask->words.size() * sizeof(uint32_t) : 0;
    // ... and, for the scout kernels, where every page's stream tiles begin
    const size_t scout_at = (pages_bytes + mask_bytes + 7) & ~size_t(7);
    const size_t meta_bytes = scout_at + n_pages * sizeof(ScoutPage);
    int32_t st = reserve_pinned(ctx, meta_bytes);
    if (st) return st;
    if (ctx->meta_in_flight) {  // the previous batch's upload still reads the pinned block
        PQ2_HIP(ctx, hipEventSynchronize(ctx->meta_copied));
        ctx->meta_in_flight = false;
    }
    PageDev *pages = static_cast<PageDev *>(ctx->h_meta);
    if (mask_bytes) std::memcpy(static_cast<uint8_t *>(ctx->h_meta) + pages_bytes, mask->words.data(), mask_bytes);

    uint64_t tiles = 0, run_entries = 0;
    uint32_t max_bits = 0;  // over the pages that are decoded
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.in_len > PQ2_MAX_PAGE_BYTES) return fail(ctx, PQ2_INVALID_PARAMETER, "page stream too long");
        if (d.in_off > in_bytes || d.in_len > in_bytes - d.in_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page stream outside the input buffer");
        if (d.num_values > PQ2_MAX_PAGE_VALUES) return fail(ctx, PQ2_INVALID_PARAMETER, "page has too many values");
        if (bits) {  // out_off counts bytes of the bitmap buffer; a page's bitmap is written in whole 32-bit words
            if (d.num_bits != 1) return fail(ctx, PQ2_INVALID_PARAMETER, "a validity bitmap takes one-bit levels (max_def_level == 1)");
            const uint64_t page_bytes = (d.num_values + 31) / 32 * 4;
            if ((d.out_off & 3u) != 0 || d.out_off > bits->bitmap_bytes || page_bytes > bits->bitmap_bytes - d.out_off)
                return fail(ctx, PQ2_INVALID_PARAMETER, "page bitmap outside the bitmap buffer, or not at a multiple of 4 bytes");
        } else if (d.out_off > out_values || d.num_values > out_values - d.out_off) {
            return fail(ctx, PQ2_INVALID_PARAMETER, "page output outside the output buffer");
        }
        PageDev &p = pages[i];
        p.in_addr = reinterpret_cast<uint64_t>(d_in) + d.in_off;
        p.out_addr = reinterpret_cast<uint64_t>(d_out) + (bits ? d.out_off : d.out_off * sizeof(uint32_t));
        p.in_len = d.in_len;
        p.num_values = static_cast<uint32_t>(d.num_values);
        p.num_bits = d.num_bits;
        if (d.num_bits > 32) {  // unpack32 has no such width (src/encoding/bitpacked/unpack.rs:87)
            ctx->host_status[i] = PQ2_INVALID_PARAMETER;
            p.num_values = 0;
        }
        const uint64_t page_tiles = (uint64_t(p.num_values) + kTileValues - 1) / kTileValues;
        p.reserved = mask ? mask->at[i] : 0;
        if (p.num_values > 0) max_bits = std::max(max_bits, p.num_bits);
        // header-position list: a run that yields values takes >= 2 stream bytes and >= 1 value; + the closing
        // entry, + the 64 entries a wave may read past the last one, rounded to whole 64-entry stores
        p.run_base = run_entries;
        run_entries += (std::min<uint64_t>(p.num_values, d.in_len / 2) + 1 + 64 + 63) & ~uint64_t(63);
        tiles += page_tiles;
        if (tiles > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "batch too large for one launch");
    }
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_out misaligned");

    // enough pages to fill the GPU with one workgroup each: the per-page kernel has every walk running from the
    // start, nothing of a page waits for other pages, and the stream comes out of the cache the second time it is
    // read; fewer pages: the tiles of all pages share the whole GPU
    // few pages: no page's walk has another page's expansion to hide behind, so the walk itself is spread over
    // the GPU (not where the streams' addresses are only known on the device)
    if (fused && !takes_ring(ctx, n_pages)) return fail(ctx, PQ2_DEVICE_ERROR, "internal: fused lookup outside ring_kernel");
    const bool scout = !front && !fused && !bits &&
                       (ctx->mode == pq2_ctx::kScout || (ctx->mode == pq2_ctx::kAuto && auto_takes_scout(descs, n_pages)));
    // (a workgroup per page costs a few microseconds before the page's first value is written: pages of a few thousand
    // values, in batches that fill the GPU either way, go to parse_kernel + expand_kernel, one wave per page and one
    // per 4096 values -- 65 536 pages of 4096 values: 0.60 against 0.99 ms, 32 768 of 8192: 0.56 against 0.61,
    // 16 384 of 16 384: 0.54 against 0.46)
    bool small_pages = false;
    if (ctx->mode == pq2_ctx::kAuto && !fused && !bits && !scout && n_pages > 1024) {
        uint64_t values = 0;
        for (uint64_t i = 0; i < n_pages; ++i) values += descs[i].num_values;
        small_pages = values <= n_pages * 8192;
    }
    const bool ring = !scout && !small_pages &&
                      (bits != nullptr /* only ring_kernel writes bitmaps */ || ctx->mode == pq2_ctx::kRing ||
                       (ctx->mode == pq2_ctx::kAuto && (n_pages >= kFusedMinPages || (!front && !fused))));
    // (kScoutMaxPages >= kFusedMinPages: in kAuto a batch whose streams' addresses the host knows is either walked by
    // the scout kernels or takes ring_kernel)
    const bool per_page = ring || ctx->mode == pq2_ctx::kPerPage;
    // tiles are numbered round-major: all tiles of round 0 of all pages, then round 1, ...
    const bool split = !per_page && !scout && tiles >= ctx->min_rounded_tiles;
    // Nothing hides the parse of the first part, and the walk of a page takes the same time whatever the batch
    // holds while the expansion grows with it: the larger the batch, the smaller the first part.
    // the dictionary lookup inside expand_kernel: one round (the statuses are final before the expansion begins)
    static const bool expand_lookup_allowed = [] {
        const char *v = std::getenv("PQ2_EXPAND_LOOKUP");
        return !(v && std::atoi(v) == 0);
    }();
    const bool looks_up = lookup && expand_lookup_allowed && !per_page && !split && !front && !mask && !fused && !bits &&
                          max_bits <= pq2_expand_dict_max_bits();
    if (looks_up) {
        for (uint64_t i = 0; i < n_pages; ++i)
            pages[i].out_addr = reinterpret_cast<uint64_t>(lookup->d_values) + descs[i].out_off * lookup->value_words * 4ull;
    } else if (lookup) {
        // the indices go to the context's buffer (the caller passes no d_out), which exists only where it is needed
        if ((st = reserve(ctx, ctx->d_indices, std::max<uint64_t>(out_values, 1) * sizeof(uint32_t)))) return st;
        for (uint64_t i = 0; i < n_pages; ++i)
            pages[i].out_addr = reinterpret_cast<uint64_t>(ctx->d_indices.ptr) + descs[i].out_off * sizeof(uint32_t);
    }
    double round_split[kRounds - 1];
    std::copy(ctx->round_split, ctx->round_split + kRounds - 1, round_split);
    if (!ctx->split_given && tiles > (1ull << 30) / kTileValues) {
        const double f = 0.25 * std::sqrt(double((1ull << 30) / kTileValues) / double(tiles));
        round_split[0] = f < 0.08 ? 0.08 : f;
    }
    uint64_t round_start[kRounds + 1] = {0};
    for (uint64_t i = 0; i < n_pages; ++i) {
        PageDev &p = pages[i];
        const uint64_t t = (uint64_t(p.num_values) + kTileValues - 1) / kTileValues;
        p.round_tile[0] = 0;
        for (uint32_t r = 1; r < kRounds; ++r) {
            const uint64_t first = split ? static_cast<uint64_t>(round_split[r - 1] * double(t) + 0.5) : t;
            p.round_tile[r] = static_cast<uint32_t>(std::min(t, std::max<uint64_t>(first, p.round_tile[r - 1])));
        }
        for (uint32_t r = 0; r < kRounds; ++r)
            round_start[r + 1] += (r + 1 < kRounds ? p.round_tile[r + 1] : t) - p.round_tile[r];
    }
    uint32_t rounds = 1;
    for (uint32_t r = 0; r < kRounds; ++r) {
        if (round_start[r + 1] > 0) rounds = r + 1;
        round_start[r + 1] += round_start[r];
    }
    {
        uint64_t next[kRounds];
        for (uint32_t r = 0; r < kRounds; ++r) next[r] = round_start[r];
        for (uint64_t i = 0; i < n_pages; ++i) {
            PageDev &p = pages[i];
            const uint64_t t = (uint64_t(p.num_values) + kTileValues - 1) / kTileValues;
            for (uint32_t r = 0; r < kRounds; ++r) {
                p.tile_base[r] = static_cast<uint32_t>(next[r]);
                next[r] += (r + 1 < kRounds ? p.round_tile[r + 1] : t) - p.round_tile[r];
            }
        }
    }

    // stream tiles of the scout kernels: a-offsets [0, delta + in_len) of every page that has headers to walk
    uint64_t stiles = 0;
    if (scout) {
        ScoutPage *sp = reinterpret_cast<ScoutPage *>(static_cast<uint8_t *>(ctx->h_meta) + scout_at);
        const uint64_t tile_bytes = pq2_scout_tile_bytes();
        for (uint64_t i = 0; i < n_pages; ++i) {
            const PageDev &p = pages[i];
            const bool walk = p.num_values > 0 && p.num_bits > 0 && p.num_bits <= 32 && p.in_len > 0;
            const uint64_t t = walk ? ((p.in_addr & 15u) + uint64_t(p.in_len) + tile_bytes - 1) / tile_bytes : 0;
            sp[i].tile_base = static_cast<uint32_t>(stiles);
            sp[i].n_tiles = static_cast<uint32_t>(t);
            stiles += t;
            if (stiles > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "batch too large for one launch");
        }
        if ((st = reserve(ctx, ctx->d_scout_table, std::max<size_t>(pq2_scout_table_bytes(stiles), 16)))) return st;
        if ((st = reserve(ctx, ctx->d_scout_words, pq2_scout_words_bytes(stiles, n_pages)))) return st;
    }
    if ((st = reserve(ctx, ctx->d_meta, meta_bytes))) return st;
    if ((st = reserve(ctx, ctx->d_status, n_pages * sizeof(int32_t)))) return st;
    if (!ring) {  // what the header walk hands to the expansion through device memory
        if ((st = reserve(ctx, ctx->d_state, n_pages * kParseStateBytes))) return st;
        if ((st = reserve(ctx, ctx->d_resume, std::max<uint64_t>(tiles, 1) * kBlocksPerTile * 16))) return st;
        if (!per_page && (st = reserve(ctx, ctx->d_tile_page, std::max<uint64_t>(tiles, 1) * sizeof(uint32_t)))) return st;
        if ((st = reserve(ctx, ctx->d_run_pos, std::max<uint64_t>(run_entries, 1) * sizeof(uint32_t)))) return st;
    }

    ctx->last_stream = stream;  // from here on work of this batch may be running on `stream`
    ctx->busy = true;
    PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_meta.ptr, ctx->h_meta, meta_bytes, hipMemcpyHostToDevice, stream));
    PQ2_HIP(ctx, hipEventRecord(ctx->meta_copied, stream));
    ctx->meta_in_flight = true;
    if (front)  // the streams' addresses, lengths and index widths come out of the page buffers on the device
        PQ2_HIP(ctx, pq2_launch_front(ctx->d_meta.ptr, front->d_raw, front->n_pages, front->d_front_status, stream));
    max_bits = std::max(max_bits, min_max_bits);
    const uint32_t *d_tile_mask =
        mask_bytes ? reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(ctx->d_meta.ptr) + pages_bytes) : nullptr;

    const PageDev *d_pages = static_cast<const PageDev *>(ctx->d_meta.ptr);
    uint32_t *d_tile_page = static_cast<uint32_t *>(ctx->d_tile_page.ptr);
    uint32_t *d_run_pos = static_cast<uint32_t *>(ctx->d_run_pos.ptr);
    int32_t *d_status = static_cast<int32_t *>(ctx->d_status.ptr);

    // with one round both kernels go to the caller's stream; with more, to the two internal streams, which
    // fork from the caller's stream and join it again
    hipStream_t ps = stream, es[2] = {stream, stream};
    if (rounds > 1) {
        if (!ctx->parse_stream) {
            PQ2_HIP(ctx, hipStreamCreateWithFlags(&ctx->parse_stream, hipStreamNonBlocking));
            PQ2_HIP(ctx, hipStreamCreateWithFlags(&ctx->expand_stream[0], hipStreamNonBlocking));
            PQ2_HIP(ctx, hipStreamCreateWithFlags(&ctx->expand_stream[1], hipStreamNonBlocking));
            for (hipEvent_t *e : {&ctx->fork, &ctx->parsed[0], &ctx->parsed[1], &ctx->parsed[2], &ctx->parsed[3],
                                  &ctx->parse_joined, &ctx->expand_joined[0], &ctx->expand_joined[1]})
                PQ2_HIP(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        ps = ctx->parse_stream;
        es[0] = ctx->expand_stream[0];
        es[1] = ctx->expand_stream[ctx->one_expand_stream ? 0 : 1];
        PQ2_HIP(ctx, hipEventRecord(ctx->fork, stream));
        PQ2_HIP(ctx, hipStreamWaitEvent(ps, ctx->fork, 0));
        PQ2_HIP(ctx, hipStreamWaitEvent(es[0], ctx->fork, 0));
        PQ2_HIP(ctx, hipStreamWaitEvent(es[1], ctx->fork, 0));
    }
    // `kernel_events`: the launch is one kernel whose launcher takes the two events itself (the kernel's own start and
    // end: what rocprofv3 reports for it); else the events are recorded in the stream around whatever is launched.
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    auto stamped = [&](bool expand, hipStream_t on, auto &&launch, bool kernel_events = false) -> int32_t {
        pq2_ctx::Stamp stamp{nullptr, nullptr, expand};
        ev_begin = ev_end = nullptr;
        if (ctx->profiling) {
            PQ2_HIP(ctx, hipEventCreate(&stamp.begin));
            PQ2_HIP(ctx, hipEventCreate(&stamp.end));
            if (kernel_events)
                ev_begin = stamp.begin, ev_end = stamp.end;
            else
                PQ2_HIP(ctx, hipEventRecord(stamp.begin, on));
        }
        PQ2_HIP(ctx, launch());
        if (ctx->profiling) {
            if (!kernel_events) PQ2_HIP(ctx, hipEventRecord(stamp.end, on));
            ctx->stamps.push_back(stamp);
        }
        ev_begin = ev_end = nullptr;
        return PQ2_OK;
    };
    // The descriptors say ring_kernel -- but a page of short runs takes its one walking wave far longer than the
    // scout kernels would take for the whole batch, and only the streams tell (pq2_density.hip).  In the range of
    // batch sizes where that can turn the choice round, density_kernel looks at the first headers of every page and
    // leaves its verdict in device memory; the ring_kernel launch goes out behind it at once and returns
    // immediately if the verdict is "no", so that the usual case costs the few microseconds of the probe and no idle
    // time; the host reads the verdict when the probe has run and, on a "no", enqueues the scout kernels instead.
    if (per_page) {
        if (!ctx->d_tickets.ptr) {
            const size_t ticket_bytes = pq2_cu_ticket_bytes() + pq2_ring_ticket_bytes();
            if ((st = reserve(ctx, ctx->d_tickets, ticket_bytes))) return st;
            PQ2_HIP(ctx, hipMemsetAsync(ctx->d_tickets.ptr, 0, ticket_bytes, stream));
        }
        if (fused)  // the walking wave and the expanding waves merge what they report: from 0
            PQ2_HIP(ctx, hipMemsetAsync(d_status, 0, n_pages * sizeof(int32_t), stream));
        if (bits)  // the expanding waves add up a page's set bits
            PQ2_HIP(ctx, hipMemsetAsync(bits->d_meta, 0, n_pages * 2 * sizeof(uint32_t), stream));
    }
    // what the kernels read besides the descriptors (the streams, records of the caller's) is ready behind this
    // event: the uploads and fills above do not wait for it
    if (before_kernels) PQ2_HIP(ctx, hipStreamWaitEvent(stream, before_kernels, 0));
    uint32_t *d_go = nullptr;
    ctx->last_density_headers = 0;
    // (with the dictionary lookup inside ring_kernel as well: on a "no" the caller takes the indices through memory)
    const bool probe = ring && !(fused && fused->d_nulls) && !bits && !front && !mask && ctx->mode == pq2_ctx::kAuto &&
                       ctx->probe_density && n_pages >= kScoutMaxPages && n_pages <= 4 * kRingSlots;
    ctx->fused_declined = false;
    if (probe) {
        // (device: the verdict word, then two words per sampled page of what the pages showed)
        if ((st = reserve(ctx, ctx->d_density, 16 + 512 * 8))) return st;
        if (!ctx->h_density) {
            PQ2_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_density), 64, hipHostMallocMapped));
            ctx->h_density[0] = ctx->h_density[1] = 0;
            PQ2_HIP(ctx, hipEventCreateWithFlags(&ctx->density_done, hipEventDisableTiming));
        }
        uint32_t *d_host_words = nullptr;
        PQ2_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&d_host_words), ctx->h_density, 0));
        d_go = static_cast<uint32_t *>(ctx->d_density.ptr);
        void *const d_seen = d_go + 4;
        const BatchBytes bb = batch_bytes(descs, n_pages);
        // (microseconds, as in auto_takes_scout; a header costs a walking wave 0.1 us, and up to twice that where four
        // walks of dense headers share a compute unit)
        double scout_us = 40.0 + bb.in_total * 2.8e-6 + bb.out_total / 3.5e6;
        // (a batch with the lookup inside ring_kernel: the other way has gather_kernel behind the scout kernels, which
        // reads the indices and writes the values at about 4 TB/s)
        if (fused) scout_us += 20.0 + bb.out_total * (1.0 + fused->value_words) / 4.0e6;
        const double turns = double((n_pages + kRingSlots - 1) / kRingSlots);
        const double per_header_us = n_pages <= kRingSlots / 4 ? 0.1 : 0.16;
        ++ctx->density_seq;
        if ((st = stamped(false, stream, [&] {
                 return pq2_launch_density(d_pages, static_cast<uint32_t>(n_pages), d_seen, float(scout_us), 30.0f,
                                           float(per_header_us), float(bb.out_max / 9.5e3), float(turns), d_go,
                                           d_host_words, ctx->density_seq & 0x7fffffffu, stream);
             })))
            return st;
        PQ2_HIP(ctx, hipEventRecord(ctx->density_done, stream));
    }
    if (per_page) {
        if ((st = stamped(true, stream, [&] {
                 uint32_t *const ring_tickets = static_cast<uint32_t *>(ctx->d_tickets.ptr) + pq2_cu_ticket_bytes() / 4u;
                 if (fused)
                     return pq2_launch_ring_dict(d_pages, static_cast<uint32_t>(n_pages), d_status, ring_tickets,
                                                 ctx->ring_epoch++, d_tile_mask, fused->d_dicts, fused->value_words,
                                                 fused->max_dict_bytes, fused->d_nulls, d_go, stream);
                 if (bits)
                     return pq2_launch_ring_bits(d_pages, static_cast<uint32_t>(n_pages), d_status, ring_tickets,
                                                 ctx->ring_epoch++, bits->d_meta, stream);
                 if (ring)
                     return pq2_launch_ring(d_pages, static_cast<uint32_t>(n_pages), d_status, ring_tickets,
                                            ctx->ring_epoch++, max_bits, d_tile_mask, d_go, ev_begin, ev_end, stream);
                 return pq2_launch_decode(d_pages, static_cast<uint32_t>(n_pages), ctx->d_resume.ptr, d_run_pos, d_status,
                                          ctx->d_state.ptr, static_cast<uint32_t *>(ctx->d_tickets.ptr), max_bits,
                                          d_tile_mask, stream);
             }, ring && !fused && !bits)))
            return st;
        rounds = 0;
    }
    if (probe) {
        PQ2_HIP(ctx, hipEventSynchronize(ctx->density_done));
        const uint32_t word = reinterpret_cast<volatile uint32_t *>(ctx->h_density)[0];
        ctx->last_density_headers = reinterpret_cast<volatile uint32_t *>(ctx->h_density)[1];
        if ((word >> 1) != (ctx->density_seq & 0x7fffffffu)) return fail(ctx, PQ2_DEVICE_ERROR, "the density probe left no verdict");
        if ((word & 1u) == 0u) {
            // the ring_kernel launch above finds the same verdict and touches nothing: the scout kernels take the batch
            // (what was stamped for it counts as time of the header walk)
            if (ctx->profiling && !ctx->stamps.empty()) ctx->stamps.back().expand = false;
            if (fused) {  // the caller decodes to indices (through the scout kernels) and looks them up afterwards
                ctx->fused_declined = true;
                return PQ2_OK;
            }
            const pq2_ctx::Mode was = ctx->mode;
            const uint32_t expected = ctx->last_density_headers;
            ctx->mode = pq2_ctx::kScout;
            st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, d_out, out_values, stream, front, min_max_bits, mask,
                               fused, bits, lookup, nullptr /* (waited for above already) */);
            ctx->mode = was;
            ctx->last_density_headers = expected;
            return st;
        }
    }
    if (scout) {
        if ((st = stamped(false, stream, [&] {
                 return pq2_launch_scout(d_pages, static_cast<const uint8_t *>(ctx->d_meta.ptr) + scout_at,
                                         static_cast<uint32_t>(n_pages), static_cast<uint32_t>(stiles),
                                         ctx->d_scout_table.ptr, ctx->d_scout_words.ptr, ctx->d_resume.ptr, d_tile_page,
                                         d_run_pos, d_status, ctx->d_state.ptr, max_bits, stream);
             })))
            return st;
        if ((st = stamped(true, stream, [&] {
                 return pq2_launch_expand(d_pages, d_tile_page, 0u, static_cast<uint32_t>(tiles), ctx->d_resume.ptr,
                                          d_run_pos, 0u, max_bits, d_tile_mask, looks_up ? lookup->d_dicts : nullptr,
                                          looks_up ? lookup->value_words : 0u, d_status, stream);
             })))
            return st;
        rounds = 0;
    }
    for (uint32_t r = 0; r < rounds; ++r) {
        if ((st = stamped(false, ps, [&] {
                 return pq2_launch_parse(d_pages, static_cast<uint32_t>(n_pages), ctx->d_resume.ptr, d_tile_page,
                                         d_run_pos, d_status, ctx->d_state.ptr, r, max_bits, ps);
             })))
            return st;
        if (rounds > 1) {
            PQ2_HIP(ctx, hipEventRecord(ctx->parsed[r], ps));
            PQ2_HIP(ctx, hipStreamWaitEvent(es[r & 1u], ctx->parsed[r], 0));
        }
        if ((st = stamped(true, es[r & 1u], [&] {
                 return pq2_launch_expand(d_pages, d_tile_page, static_cast<uint32_t>(round_start[r]),
                                          static_cast<uint32_t>(round_start[r + 1] - round_start[r]),
                                          ctx->d_resume.ptr, d_run_pos, r, max_bits, d_tile_mask,
                                          looks_up ? lookup->d_dicts : nullptr, looks_up ? lookup->value_words : 0u,
                                          d_status, es[r & 1u]);
             })))
            return st;
    }
    if (rounds > 1) {
        PQ2_HIP(ctx, hipEventRecord(ctx->parse_joined, ps));
        PQ2_HIP(ctx, hipStreamWaitEvent(stream, ctx->parse_joined, 0));
        for (int i = 0; i < 2; ++i) {
            PQ2_HIP(ctx, hipEventRecord(ctx->expand_joined[i], es[i]));
            PQ2_HIP(ctx, hipStreamWaitEvent(stream, ctx->expand_joined[i], 0));
        }
    }
    if (ctx->profiling) ++ctx->stamped_batches;
    ctx->last_n_pages = n_pages;
    ctx->last_scout = scout;
    ctx->last_scout_tiles = stiles;
    ctx->last_schedule = fused && fused->d_nulls ? "ring_kernel+dictionary+rows" : fused ? "ring_kernel+dictionary" : bits ? "ring_kernel+bitmap" : ring ? "ring_kernel" : per_page ? "decode_kernel" : scout ? "scout+expand_kernel" : "expand_kernel";
    if (looks_up) {
        ctx->expand_looked_up = true;
        ctx->last_schedule = scout ? "scout+expand_kernel+dictionary" : "expand_kernel+dictionary";
    }
    return PQ2_OK;
}

int32_t fetch_status(pq2_ctx *ctx, int32_t *status, uint64_t n_pages) {
    if (n_pages != ctx->last_n_pages) return fail(ctx, PQ2_INVALID_PARAMETER, "n_pages does not match the last batch");
    if (n_pages == 0) return PQ2_OK;
    PQ2_HIP(ctx, hipMemcpyAsync(status, ctx->d_status.ptr, n_pages * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->last_stream));
    PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    for (uint64_t i = 0; i < n_pages; ++i)
        if (ctx->host_status[i] != PQ2_OK) status[i] = ctx->host_status[i];
    return PQ2_OK;
}

}  // namespace

extern "C" {

int32_t pq2_ctx_create(int32_t device_id, pq2_ctx **out) {
    if (!out) return PQ2_INVALID_PARAMETER;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PQ2_DEVICE_ERROR;  // no CPU fallback
    if (device_id < 0 || device_id >= count) return PQ2_INVALID_PARAMETER;
    if (hipSetDevice(device_id) != hipSuccess) return PQ2_DEVICE_ERROR;
    pq2_ctx *ctx = new (std::nothrow) pq2_ctx();
    if (!ctx) return PQ2_WOULD_OVER_ALLOCATE;
    ctx->device = device_id;
    if (const char *v = std::getenv("PQ2_MIN_ROUNDED_TILES")) {
        const long long t = std::atoll(v);
        if (t > 0) ctx->min_rounded_tiles = static_cast<uint64_t>(t);
    }
    if (const char *v = std::getenv("PQ2_EXPAND_STREAMS")) ctx->one_expand_stream = std::atoi(v) != 2;
    if (const char *v = std::getenv("PQ2_MODE")) (void)pq2_ctx_set_schedule(ctx, v);  // another word: as without
    if (const char *v = std::getenv("PQ2_DENSITY_PROBE")) ctx->probe_density = std::atoi(v) != 0;
    if (const char *v = std::getenv("PQ2_OPTIONAL_BITS")) ctx->optional_bits = std::atoi(v) != 0;
    if (const char *v = std::getenv("PQ2_OPTIONAL_FUSED")) ctx->optional_fused = std::atoi(v) != 0;
    if (const char *v = std::getenv("PQ2_ROUND_SPLIT")) {
        double a = 0, b = 0, c = 0;
        if (std::sscanf(v, "%lf,%lf,%lf", &a, &b, &c) == 3 && 0 <= a && a <= b && b <= c && c <= 1)
            ctx->round_split[0] = a, ctx->round_split[1] = b, ctx->round_split[2] = c, ctx->split_given = true;
    }
    if (hipEventCreateWithFlags(&ctx->meta_copied, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return PQ2_DEVICE_ERROR;
    }
    // where this GPU's copy of the fixed working memory of pq2_decode_batch_device_noalloc lies: looked up here so
    // that a process which creates a context first never does it inside a capture (a GPU number beyond that
    // table only means the entry point reports it)
    (void)pq2_noalloc_setup(device_id, nullptr);
    *out = ctx;
    return PQ2_OK;
}

int32_t pq2_ctx_set_schedule(pq2_ctx *ctx, const char *name) {
    if (!ctx || !name) return PQ2_INVALID_PARAMETER;
    pq2_ctx::Mode mode;
    if (std::strcmp(name, "auto") == 0) mode = pq2_ctx::kAuto;
    else if (std::strcmp(name, "page") == 0) mode = pq2_ctx::kPerPage;
    else if (std::strcmp(name, "tile") == 0) mode = pq2_ctx::kPerTile;
    else if (std::strcmp(name, "ring") == 0) mode = pq2_ctx::kRing;
    else if (std::strcmp(name, "scout") == 0) mode = pq2_ctx::kScout;
    else return PQ2_INVALID_PARAMETER;
    ctx->mode = mode;
    for (pq2_ctx *lane : ctx->lanes)
        if (lane) lane->mode = mode;
    return PQ2_OK;
}

int32_t pq2_ctx_destroy(pq2_ctx *ctx) {
    if (!ctx) return PQ2_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (DevBuf *b : {&ctx->d_meta, &ctx->d_status, &ctx->d_resume, &ctx->d_tile_page, &ctx->d_run_pos, &ctx->d_state,
                      &ctx->d_tickets, &ctx->d_scout_table, &ctx->d_scout_words, &ctx->d_seg_valid, &ctx->d_bit_meta, &ctx->d_rowdir, &ctx->d_density, &ctx->d_indices, &ctx->d_gather, &ctx->d_in, &ctx->d_out})
        if (b->ptr) (void)hipFree(b->ptr);
    for (hipStream_t s : {ctx->parse_stream, ctx->expand_stream[0], ctx->expand_stream[1]})
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : {ctx->fork, ctx->parsed[0], ctx->parsed[1], ctx->parsed[2], ctx->parsed[3], ctx->parse_joined,
                         ctx->expand_joined[0], ctx->expand_joined[1]})
        if (e) (void)hipEventDestroy(e);
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    if (ctx->h_gather) (void)hipHostFree(ctx->h_gather);
    if (ctx->h_density) (void)hipHostFree(ctx->h_density);
    if (ctx->density_done) (void)hipEventDestroy(ctx->density_done);
    if (ctx->gather_copied) (void)hipEventDestroy(ctx->gather_copied);
    if (ctx->meta_copied) (void)hipEventDestroy(ctx->meta_copied);
    for (auto &s : ctx->stamps) {
        (void)hipEventDestroy(s.begin);
        (void)hipEventDestroy(s.end);
    }
    for (pq2_ctx *lane : ctx->lanes)
        if (lane) (void)pq2_ctx_destroy(lane);
    if (ctx->bits_lane) (void)pq2_ctx_destroy(ctx->bits_lane);
    for (hipEvent_t e : {ctx->bits_fork, ctx->bits_joined})
        if (e) (void)hipEventDestroy(e);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
    delete ctx;
    return PQ2_OK;
}

const char *pq2_strerror(int32_t status) {
    switch (status) {
        case PQ2_OK: return "ok";
        case PQ2_OUT_OF_SPEC: return "out of spec";
        case PQ2_INVALID_PARAMETER: return "invalid parameter";
        case PQ2_FEATURE_NOT_SUPPORTED: return "feature not supported";
        case PQ2_WOULD_OVER_ALLOCATE: return "would over-allocate";
        case PQ2_REFERENCE_WOULD_PANIC: return "the reference decoder panics on this input";
        case PQ2_DEVICE_ERROR: return "device error";
        default: return "unknown status";
    }
}

const char *pq2_last_error(const pq2_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

const char *pq2_ctx_last_schedule(const pq2_ctx *ctx) { return ctx ? ctx->last_schedule : ""; }

uint64_t pq2_ctx_pages_total(const pq2_ctx *ctx) { return ctx ? ctx->pages_total : 0; }

uint32_t pq2_ctx_last_density(const pq2_ctx *ctx) { return ctx ? ctx->last_density_headers : 0; }

int32_t pq2_ctx_scout_rounds(pq2_ctx *ctx, uint32_t *rounds, uint64_t n_pages) {
    if (!ctx || (!rounds && n_pages)) return PQ2_INVALID_PARAMETER;
    if (!ctx->last_scout || n_pages != ctx->last_n_pages)
        return fail(ctx, PQ2_INVALID_PARAMETER, "the last batch did not go through the scout kernels with this many pages");
    if (n_pages == 0) return PQ2_OK;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    const uint8_t *at = static_cast<const uint8_t *>(ctx->d_scout_words.ptr) + pq2_scout_rounds_at(ctx->last_scout_tiles, n_pages);
    PQ2_HIP(ctx, hipMemcpyAsync(rounds, at, n_pages * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->last_stream));
    PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    return PQ2_OK;
}

int32_t pq2_ctx_enable_profiling(pq2_ctx *ctx, int32_t on) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    ctx->profiling = on != 0;
    return PQ2_OK;
}

int32_t pq2_ctx_kernel_times(pq2_ctx *ctx, double *parse_ms, double *expand_ms, uint64_t *batches,
                             uint64_t *expand_launches) {
    if (!ctx || !parse_ms || !expand_ms || !batches) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    *parse_ms = *expand_ms = 0.0;
    *batches = ctx->stamped_batches;
    uint64_t launches = 0;
    for (auto &s : ctx->stamps) {
        PQ2_HIP(ctx, hipEventSynchronize(s.end));
        float ms = 0.f;
        PQ2_HIP(ctx, hipEventElapsedTime(&ms, s.begin, s.end));
        if (s.expand) {
            *expand_ms += ms;
            ++launches;
        } else {
            *parse_ms += ms;
        }
        (void)hipEventDestroy(s.begin);
        (void)hipEventDestroy(s.end);
    }
    if (expand_launches) *expand_launches = launches;
    ctx->stamps.clear();
    ctx->stamped_batches = 0;
    return PQ2_OK;
}

int32_t pq2_decode_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                uint64_t n_pages, uint32_t *d_out, uint64_t out_values, int32_t *status,
                                void *stream) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    ctx->pages_total += n_pages;
    int32_t st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, d_out, out_values, static_cast<hipStream_t>(stream));
    if (st != PQ2_OK) return st;
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

// pq2_decode_validity_bitmap_batch_device; `before_kernels` (may be null): an event the kernels wait for on `stream`,
// but not the upload of the descriptors.
static int32_t validity_bitmaps(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                uint64_t n_pages, void *d_bitmap, uint64_t bitmap_bytes, uint32_t *rows,
                                uint32_t *set_counts, int32_t *status, hipStream_t stream, hipEvent_t before_kernels) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    if (n_pages > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many pages in one batch");
    int32_t st = reserve(ctx, ctx->d_bit_meta, std::max<uint64_t>(n_pages, 1) * 2 * sizeof(uint32_t));
    if (st != PQ2_OK) return st;
    const BitmapOut bits{bitmap_bytes, static_cast<uint32_t *>(ctx->d_bit_meta.ptr)};
    st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, static_cast<uint32_t *>(d_bitmap), 0, stream, nullptr, 0,
                       nullptr, nullptr, &bits, nullptr, before_kernels);
    if (st != PQ2_OK) return st;
    if ((rows || set_counts) && n_pages) {
        std::vector<uint32_t> meta(n_pages * 2);
        PQ2_HIP(ctx, hipMemcpyAsync(meta.data(), ctx->d_bit_meta.ptr, meta.size() * sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, stream));
        PQ2_HIP(ctx, hipStreamSynchronize(stream));
        for (uint64_t i = 0; i < n_pages; ++i) {
            if (rows) rows[i] = meta[2 * i];
            if (set_counts) set_counts[i] = meta[2 * i + 1];
        }
    }
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

int32_t pq2_decode_validity_bitmap_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                                const pq2_page_desc *descs, uint64_t n_pages, void *d_bitmap,
                                                uint64_t bitmap_bytes, uint32_t *rows, uint32_t *set_counts,
                                                int32_t *status, void *stream) {
    return validity_bitmaps(ctx, d_in, in_bytes, descs, n_pages, d_bitmap, bitmap_bytes, rows, set_counts, status,
                            static_cast<hipStream_t>(stream), nullptr);
}

int32_t pq2_decode_dict_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                     uint64_t n_pages, const void *d_dict, uint64_t dict_bytes,
                                     const pq2_dict_ref *dicts, uint32_t value_size, void *d_out, uint64_t out_values,
                                     int32_t *status, void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (value_size != 4 && value_size != 8 && value_size != 12)
        return fail(ctx, PQ2_INVALID_PARAMETER, "value_size must be 4, 8 or 12");
    if (n_pages > 0 && !dicts) return fail(ctx, PQ2_INVALID_PARAMETER, "dicts is NULL");
    if ((reinterpret_cast<uintptr_t>(d_out) & 7u) != 0 || (reinterpret_cast<uintptr_t>(d_dict) & 7u) != 0)
        return fail(ctx, PQ2_INVALID_PARAMETER, "d_out / d_dict misaligned");
    const uint64_t align = value_size == 8 ? 8 : 4;
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_dict_ref &r = dicts[i];
        if (r.dict_off % align != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary offset misaligned");
        if (r.dict_off > dict_bytes || uint64_t(r.dict_len) * value_size > dict_bytes - r.dict_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary outside the dictionary buffer");
    }
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    // Batches that ring_kernel takes, of pages narrow enough and with dictionaries small enough to lie in LDS beside
    // the ring, are looked up where they are decoded; everything else is decoded into a buffer of indices first
    // and looked up by a second kernel.  (PQ2_FUSED_GATHER=0: always the latter.)
    uint64_t max_dict_bytes = 0;
    uint32_t max_bits = 0;
    for (uint64_t i = 0; i < n_pages; ++i) {
        max_dict_bytes = std::max<uint64_t>(max_dict_bytes, uint64_t(dicts[i].dict_len) * value_size);
        if (descs[i].num_bits <= 32 && descs[i].num_values > 0) max_bits = std::max(max_bits, descs[i].num_bits);
    }
    static const bool fused_allowed = [] {
        const char *v = std::getenv("PQ2_FUSED_GATHER");
        return !(v && std::atoi(v) == 0);
    }();
    const bool fuse = fused_allowed && n_pages > 0 && takes_ring(ctx, n_pages) && max_bits <= pq2_ring_dict_max_bits() &&
                      max_dict_bytes <= pq2_ring_dict_max_bytes();
    uint32_t *d_idx = nullptr;
    if (!fuse && n_pages == 0 && (st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, d_idx, out_values, stream)))
        return st;
    if (n_pages > 0) {
        const size_t bytes = n_pages * sizeof(GatherPage);
        const uint64_t kGatherTileValues = pq2_gather_tile_values();
        uint64_t tiles = 0;
        GatherPage *g = nullptr;
        // (`decoded`: the batch has been through enqueue_batch, which has checked the descriptors)
        auto stage_dicts = [&](bool decoded) -> int32_t {
            if (int32_t e = stage_block(ctx, bytes, bytes)) return e;
            g = static_cast<GatherPage *>(ctx->h_gather);
            tiles = 0;
            for (uint64_t i = 0; i < n_pages; ++i) {
                const pq2_page_desc &d = descs[i];
                g[i].idx_addr = reinterpret_cast<uint64_t>(d_idx) + d.out_off * sizeof(uint32_t);
                g[i].out_addr = reinterpret_cast<uint64_t>(d_out) + d.out_off * value_size;
                g[i].dict_addr = reinterpret_cast<uint64_t>(d_dict) + dicts[i].dict_off;
                g[i].dict_len = dicts[i].dict_len;
                g[i].num_values = (decoded ? ctx->host_status[i] == PQ2_OK : d.num_bits <= 32) ? static_cast<uint32_t>(d.num_values) : 0u;
                g[i].first_tile = static_cast<uint32_t>(tiles);
                g[i].reserved = 0;
                tiles += (uint64_t(g[i].num_values) + kGatherTileValues - 1) / kGatherTileValues;
            }
            return upload_block(ctx, bytes, stream);
        };
        if ((st = stage_dicts(false))) return st;
        if (!fuse) {
            // Where expand_kernel carries the values it looks them up as it writes (neither kernel of the other way
            // then touches a buffer of indices); else -- one workgroup per page, several rounds, indices wider than
            // the lookup takes -- the batch is decoded into indices and gather_kernel looks them up.
            const ExpandLookup lookup{ctx->d_gather.ptr, value_size / 4, d_out};
            if ((st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, d_idx, out_values, stream, nullptr, 0, nullptr,
                                    nullptr, nullptr, &lookup)))
                return st;
            if (!ctx->expand_looked_up) {
                d_idx = static_cast<uint32_t *>(ctx->d_indices.ptr);
                if ((st = stage_dicts(true))) return st;
                PQ2_HIP(ctx, pq2_launch_gather(ctx->d_gather.ptr, static_cast<uint32_t>(n_pages), static_cast<uint32_t>(tiles),
                                               value_size / 4, max_dict_bytes, static_cast<int32_t *>(ctx->d_status.ptr),
                                               stream));
            }
        } else {
            // (the descriptors' output offsets are checked against the values; the kernel writes through `dicts`)
            const FusedDict fd{ctx->d_gather.ptr, value_size / 4, static_cast<uint32_t>(max_dict_bytes)};
            if ((st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, static_cast<uint32_t *>(d_out), out_values, stream,
                                    nullptr, 0, nullptr, &fd)))
                return st;
            if (ctx->fused_declined) {
                // The pages' headers are too dense for one walking wave per page (density_kernel): the scout kernels
                // decode the batch into indices and gather_kernel looks them up, as for a batch of few pages.
                const pq2_ctx::Mode was = ctx->mode;
                const uint32_t expected = ctx->last_density_headers;
                ctx->mode = pq2_ctx::kScout;
                const ExpandLookup lookup{ctx->d_gather.ptr, value_size / 4, d_out};  // (the descriptors are on the device)
                st = enqueue_batch(ctx, d_in, in_bytes, descs, n_pages, d_idx, out_values, stream, nullptr, 0, nullptr, nullptr,
                                   nullptr, &lookup);
                ctx->mode = was;
                ctx->last_density_headers = expected;
                if (st) return st;
                if (!ctx->expand_looked_up) {
                    d_idx = static_cast<uint32_t *>(ctx->d_indices.ptr);
                    if ((st = stage_dicts(true))) return st;
                    PQ2_HIP(ctx, pq2_launch_gather(ctx->d_gather.ptr, static_cast<uint32_t>(n_pages),
                                                   static_cast<uint32_t>(tiles), value_size / 4, max_dict_bytes,
                                                   static_cast<int32_t *>(ctx->d_status.ptr), stream));
                    ctx->last_schedule = "scout+expand_kernel+gather_kernel";
                }
            }
        }
    }
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

int32_t pq2_decode_optional_dict_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                              const pq2_page_desc *def_descs, const pq2_page_desc *idx_descs,
                                              const uint32_t *max_def_levels, uint64_t n_pages, const void *d_dict,
                                              uint64_t dict_bytes, const pq2_dict_ref *dicts, uint32_t value_size,
                                              void *d_out, uint8_t *d_validity, uint64_t out_rows, int32_t *status,
                                              void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (value_size != 4 && value_size != 8 && value_size != 12)
        return fail(ctx, PQ2_INVALID_PARAMETER, "value_size must be 4, 8 or 12");
    if (n_pages > 0x3fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many pages in one batch");
    if (n_pages > 0 && (!def_descs || !idx_descs || !max_def_levels || !dicts))
        return fail(ctx, PQ2_INVALID_PARAMETER, "a descriptor array is NULL");
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0 || (reinterpret_cast<uintptr_t>(d_dict) & 3u) != 0)
        return fail(ctx, PQ2_INVALID_PARAMETER, "d_out / d_dict misaligned");
    // Flat nullable columns (every maximum level 1, the streams one bit wide), enough pages for a workgroup each: the
    // levels are not expanded at all.  A child context turns the level streams into validity bitmaps (ring_kernel's
    // bitmap mode, which is bound by the header walks and moves next to nothing) on a stream of its own while this
    // context decodes the index streams, and the scatter reads one bit per row instead of a u32 that had to be written
    // first.  Same values, validity and statuses as the path below (a level is valid where it equals 1 either way).
    // (PQ2_OPTIONAL_BITS=0 when the context was created: levels as u32, for A/B runs)
    bool by_bits = ctx->optional_bits && n_pages >= kOptionalWholePages &&
                   (ctx->mode == pq2_ctx::kAuto || ctx->mode == pq2_ctx::kRing);
    for (uint64_t i = 0; by_bits && i < n_pages; ++i)
        by_bits = max_def_levels[i] == 1 && def_descs[i].num_bits == 1 && def_descs[i].num_values <= PQ2_MAX_PAGE_VALUES &&
                  // (optional_bits_kernel reads indices and dictionaries through 32-bit byte offsets)
                  idx_descs[i].num_values < (1u << 30) && uint64_t(dicts[i].dict_len) * value_size < (1ull << 32);
    if (by_bits) {
        int32_t st;
        for (uint64_t i = 0; i < n_pages; ++i) {
            const pq2_dict_ref &r = dicts[i];
            if (r.dict_off % 4 != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary offset misaligned");
            if (r.dict_off > dict_bytes || uint64_t(r.dict_len) * value_size > dict_bytes - r.dict_off)
                return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary outside the dictionary buffer");
            if (def_descs[i].out_off > out_rows || def_descs[i].num_values > out_rows - def_descs[i].out_off)
                return fail(ctx, PQ2_INVALID_PARAMETER, "page rows outside the output buffer");
        }
        if (!ctx->bits_lane) {
            if ((st = pq2_ctx_create(ctx->device, &ctx->bits_lane))) return fail(ctx, st, "cannot create the context of the level streams");
            ctx->bits_lane->profiling = false;
            PQ2_HIP(ctx, hipStreamCreateWithFlags(&ctx->bits_lane->lane_stream, hipStreamNonBlocking));
            PQ2_HIP(ctx, hipEventCreateWithFlags(&ctx->bits_fork, hipEventDisableTiming));
            PQ2_HIP(ctx, hipEventCreateWithFlags(&ctx->bits_joined, hipEventDisableTiming));
        }
        pq2_ctx *const lane = ctx->bits_lane;
        std::vector<pq2_page_desc> levels(def_descs, def_descs + n_pages), indices(idx_descs, idx_descs + n_pages);
        uint64_t bitmap_bytes = 0, idx_values = 0;
        for (uint64_t i = 0; i < n_pages; ++i) {
            levels[i].out_off = bitmap_bytes;  // (bytes)
            bitmap_bytes += (def_descs[i].num_values + 31) / 32 * 4;
            indices[i].out_off = idx_values;
            idx_values += (idx_descs[i].num_values + 3) & ~uint64_t(3);
        }
        if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
        if ((st = reserve(ctx, lane->d_out, std::max<uint64_t>(bitmap_bytes, 4)))) return st;
        // Where the indices are narrow enough and the dictionaries small enough for the lookup inside ring_kernel, the
        // indices are not written either: the bitmaps first (and from them, for every page, the rows at which its
        // spans of 2048 valid rows begin: rowdir_kernel), then ring_kernel over the index streams, whose expanding waves
        // keep a span's indices in LDS and write the span's ROWS -- values, nulls and validity bytes.  What moves
        // through device memory is the streams, the bitmaps (a bit per row, written and read) and the result.
        // (PQ2_OPTIONAL_FUSED=0 when the context was created: the indices go through memory, optional_bits_kernel places them.)
        uint32_t idx_max_bits = 0;
        uint64_t dict_max_bytes = 0;
        for (uint64_t i = 0; i < n_pages; ++i) {
            idx_max_bits = std::max(idx_max_bits, idx_descs[i].num_bits);  // (a width above 32 fails the page: not here)
            dict_max_bytes = std::max<uint64_t>(dict_max_bytes, uint64_t(dicts[i].dict_len) * value_size);
        }
        if (ctx->optional_fused && idx_max_bits <= pq2_ring_nullable_max_bits() && dict_max_bytes <= pq2_ring_nullable_max_dict_bytes()) {
            const uint64_t step = pq2_rowdir_step();
            uint64_t dir_words = 0;
            std::vector<uint64_t> dir_at(n_pages);
            for (uint64_t i = 0; i < n_pages; ++i) {
                dir_at[i] = dir_words;
                dir_words += idx_descs[i].num_values / step + 3;  // (a tile's first span asks for three entries)
            }
            if ((st = reserve(ctx, ctx->d_rowdir, dir_words * sizeof(uint32_t)))) return st;
            // The level streams on the child's stream, behind whatever the caller's stream held at the call -- the
            // kernels, that is: the uploads of descriptors and the fills go ahead of that (here and for the index
            // streams below), so that they run beside the kernel before them instead of between two kernels.
            PQ2_HIP(ctx, hipEventRecord(ctx->bits_fork, stream));
            st = validity_bitmaps(lane, d_in, in_bytes, levels.data(), n_pages, lane->d_out.ptr,
                                  std::max<uint64_t>(bitmap_bytes, 4), nullptr, nullptr, nullptr, lane->lane_stream,
                                  ctx->bits_fork);
            if (st != PQ2_OK) return fail(ctx, st, pq2_last_error(lane));
            // one block: the GatherPage records ring_kernel takes the dictionaries from, then the RingNullable records
            const size_t dict_recs = n_pages * sizeof(GatherPage);
            const size_t bytes = dict_recs + n_pages * sizeof(RingNullable);
            if ((st = stage_block(ctx, bytes, bytes))) return st;
            GatherPage *g = static_cast<GatherPage *>(ctx->h_gather);
            RingNullable *nl = reinterpret_cast<RingNullable *>(static_cast<uint8_t *>(ctx->h_gather) + dict_recs);
            for (uint64_t i = 0; i < n_pages; ++i) {
                const bool skip = lane->host_status[i] != PQ2_OK;  // (its rows are left alone, as on the other paths)
                if (skip) indices[i].num_values = 0;
                g[i].idx_addr = 0;
                g[i].out_addr = reinterpret_cast<uint64_t>(d_out) + def_descs[i].out_off * value_size;
                g[i].dict_addr = reinterpret_cast<uint64_t>(d_dict) + dicts[i].dict_off;
                g[i].dict_len = dicts[i].dict_len;
                g[i].num_values = static_cast<uint32_t>(indices[i].num_values);
                g[i].first_tile = 0;
                g[i].reserved = 0;
                nl[i].bitmap_addr = reinterpret_cast<uint64_t>(lane->d_out.ptr) + levels[i].out_off;
                nl[i].validity_addr = reinterpret_cast<uint64_t>(d_validity) + def_descs[i].out_off;
                nl[i].out_addr = g[i].out_addr;
                nl[i].rowdir_addr = reinterpret_cast<uint64_t>(ctx->d_rowdir.ptr) + dir_at[i] * sizeof(uint32_t);
                nl[i].rows = skip ? 0u : static_cast<uint32_t>(def_descs[i].num_values);
                nl[i].idx_values = static_cast<uint32_t>(indices[i].num_values);
            }
            // (the records go up on the caller's stream, beside the bitmaps' kernel; rowdir_kernel waits for them)
            if ((st = upload_block(ctx, bytes, stream))) return st;
            PQ2_HIP(ctx, hipStreamWaitEvent(lane->lane_stream, ctx->gather_copied, 0));
            const uint8_t *const d_nulls = static_cast<const uint8_t *>(ctx->d_gather.ptr) + dict_recs;
            PQ2_HIP(ctx, pq2_launch_rowdir(d_nulls, static_cast<uint32_t>(n_pages), lane->lane_stream));
            PQ2_HIP(ctx, hipEventRecord(ctx->bits_joined, lane->lane_stream));
            FusedDict fd{ctx->d_gather.ptr, value_size / 4, static_cast<uint32_t>(dict_max_bytes)};
            fd.d_nulls = d_nulls;
            // (the descriptors' output offsets only have to be consistent: the kernel writes through the records above)
            if ((st = enqueue_batch(ctx, d_in, in_bytes, indices.data(), n_pages, static_cast<uint32_t *>(d_out),
                                    std::max<uint64_t>(idx_values, 1), stream, nullptr, 0, nullptr, &fd, nullptr, nullptr,
                                    ctx->bits_joined)))
                return st;
            PQ2_HIP(ctx, pq2_launch_nullable_status(static_cast<int32_t *>(ctx->d_status.ptr),
                                                    static_cast<const int32_t *>(lane->d_status.ptr),
                                                    static_cast<uint32_t>(n_pages), stream));
            for (uint64_t i = 0; i < n_pages; ++i)
                if (lane->host_status[i] != PQ2_OK) ctx->host_status[i] = lane->host_status[i];
            ctx->last_schedule = "ring_kernel+bitmap (levels), ring_kernel+dictionary+rows (indices)";
            if (status) return fetch_status(ctx, status, n_pages);
            return PQ2_OK;
        }
        // (optional_bits_kernel reads 16 bytes at a time from any index of a page on: 12 bytes behind the last page's)
        if ((st = reserve(ctx, ctx->d_indices, std::max<uint64_t>(idx_values, 1) * sizeof(uint32_t) + 16))) return st;
        // The index streams first: where this context looks at the headers before it settles on a schedule, that look
        // should not have to wait for a compute unit behind the workgroups of the level streams.  Then the level
        // streams on the child's stream, behind whatever the caller's stream held at the call.
        PQ2_HIP(ctx, hipEventRecord(ctx->bits_fork, stream));
        uint32_t *d_idx = static_cast<uint32_t *>(ctx->d_indices.ptr);
        if ((st = enqueue_batch(ctx, d_in, in_bytes, indices.data(), n_pages, d_idx, idx_values, stream))) return st;
        st = validity_bitmaps(lane, d_in, in_bytes, levels.data(), n_pages, lane->d_out.ptr,
                              std::max<uint64_t>(bitmap_bytes, 4), nullptr, nullptr, nullptr, lane->lane_stream,
                              ctx->bits_fork);
        if (st != PQ2_OK) return fail(ctx, st, pq2_last_error(lane));
        PQ2_HIP(ctx, hipEventRecord(ctx->bits_joined, lane->lane_stream));
        const size_t bytes = n_pages * sizeof(OptionalPage);
        if ((st = stage_block(ctx, bytes, bytes))) return st;
        OptionalPage *g = static_cast<OptionalPage *>(ctx->h_gather);
        uint64_t largest_dict = 0;  // bytes
        for (uint64_t i = 0; i < n_pages; ++i) {
            const bool skip = ctx->host_status[i] != PQ2_OK || lane->host_status[i] != PQ2_OK;
            largest_dict = std::max<uint64_t>(largest_dict, uint64_t(dicts[i].dict_len) * value_size);
            g[i].level_addr = reinterpret_cast<uint64_t>(lane->d_out.ptr) + levels[i].out_off;
            g[i].idx_addr = reinterpret_cast<uint64_t>(d_idx + indices[i].out_off);
            g[i].out_addr = reinterpret_cast<uint64_t>(d_out) + def_descs[i].out_off * value_size;
            g[i].validity_addr = reinterpret_cast<uint64_t>(d_validity) + def_descs[i].out_off;
            g[i].dict_addr = reinterpret_cast<uint64_t>(d_dict) + dicts[i].dict_off;
            g[i].rows = skip ? 0u : static_cast<uint32_t>(def_descs[i].num_values);
            g[i].idx_values = static_cast<uint32_t>(idx_descs[i].num_values);
            g[i].dict_len = dicts[i].dict_len;
            g[i].max_level = 1;
            g[i].idx_status_at = static_cast<uint32_t>(i);
            g[i].first_seg = 0;
            // a page's own status (its level stream's), or else its index stream's
            if (lane->host_status[i] != PQ2_OK) ctx->host_status[i] = lane->host_status[i];
        }
        if ((st = upload_block(ctx, bytes, stream))) return st;
        PQ2_HIP(ctx, hipStreamWaitEvent(stream, ctx->bits_joined, 0));  // (behind the upload: that one need not wait)
        PQ2_HIP(ctx, pq2_launch_optional_bits(ctx->d_gather.ptr, static_cast<uint32_t>(n_pages), value_size / 4,
                                              static_cast<int32_t *>(ctx->d_status.ptr),
                                              static_cast<const int32_t *>(lane->d_status.ptr), largest_dict, stream));
        ctx->last_schedule = "ring_kernel (indices) + ring_kernel+bitmap (levels)";
        ctx->last_n_pages = n_pages;
        if (status) return fetch_status(ctx, status, n_pages);
        return PQ2_OK;
    }
    // the level and index streams of all pages as one batch of 2 * n_pages streams decoded into the context's buffer
    std::vector<pq2_page_desc> both(2 * n_pages);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_dict_ref &r = dicts[i];
        if (r.dict_off % 4 != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary offset misaligned");
        if (r.dict_off > dict_bytes || uint64_t(r.dict_len) * value_size > dict_bytes - r.dict_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "dictionary outside the dictionary buffer");
        if (def_descs[i].out_off > out_rows || def_descs[i].num_values > out_rows - def_descs[i].out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page rows outside the output buffer");
        both[i] = def_descs[i];
        both[i].out_off = at;
        at += (def_descs[i].num_values + 3) & ~uint64_t(3);  // every page's levels start 16-byte aligned
    }
    for (uint64_t i = 0; i < n_pages; ++i) {
        both[n_pages + i] = idx_descs[i];
        both[n_pages + i].out_off = at;
        at += (idx_descs[i].num_values + 3) & ~uint64_t(3);
    }
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    if ((st = reserve(ctx, ctx->d_indices, std::max<uint64_t>(at, 1) * sizeof(uint32_t)))) return st;
    uint32_t *d_scratch = static_cast<uint32_t *>(ctx->d_indices.ptr);
    if ((st = enqueue_batch(ctx, d_in, in_bytes, both.data(), 2 * n_pages, d_scratch, at, stream))) return st;
    if (n_pages > 0) {
        const size_t bytes = n_pages * sizeof(OptionalPage);
        if ((st = stage_block(ctx, bytes, bytes))) return st;
        OptionalPage *g = static_cast<OptionalPage *>(ctx->h_gather);
        // few pages: every page in segments, so that large pages are shared out over the GPU
        static const long long seg_knob = [] {  // PQ2_OPTIONAL_SEG_ROWS: 0 = always whole pages, n = segments of n * 4096 rows
            const char *v = std::getenv("PQ2_OPTIONAL_SEG_ROWS");
            return v ? std::atoll(v) : -1ll;
        }();
        const uint64_t seg_rows = seg_knob == 0 ? 0 : seg_knob > 0 ? uint64_t(seg_knob) * 4096
                                  : n_pages >= kOptionalWholePages ? 0 : kOptionalSegRows;
        uint64_t n_segs = 0;
        for (uint64_t i = 0; i < n_pages; ++i) {
            const bool skip = ctx->host_status[i] != PQ2_OK || ctx->host_status[n_pages + i] != PQ2_OK;
            g[i].level_addr = reinterpret_cast<uint64_t>(d_scratch + both[i].out_off);
            g[i].idx_addr = reinterpret_cast<uint64_t>(d_scratch + both[n_pages + i].out_off);
            g[i].out_addr = reinterpret_cast<uint64_t>(d_out) + def_descs[i].out_off * value_size;
            g[i].validity_addr = reinterpret_cast<uint64_t>(d_validity) + def_descs[i].out_off;
            g[i].dict_addr = reinterpret_cast<uint64_t>(d_dict) + dicts[i].dict_off;
            g[i].rows = skip ? 0u : static_cast<uint32_t>(def_descs[i].num_values);
            g[i].idx_values = static_cast<uint32_t>(idx_descs[i].num_values);
            g[i].dict_len = dicts[i].dict_len;
            g[i].max_level = max_def_levels[i];
            g[i].idx_status_at = static_cast<uint32_t>(n_pages + i);
            g[i].first_seg = static_cast<uint32_t>(n_segs);
            if (seg_rows) n_segs += std::max<uint64_t>(1, (uint64_t(g[i].rows) + seg_rows - 1) / seg_rows);
        }
        if (n_segs > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "batch too large for one launch");
        if (seg_rows && (st = reserve(ctx, ctx->d_seg_valid, n_segs * sizeof(uint32_t)))) return st;
        if ((st = upload_block(ctx, bytes, stream))) return st;
        PQ2_HIP(ctx, pq2_launch_optional(ctx->d_gather.ptr, static_cast<uint32_t>(n_pages), value_size / 4,
                                         static_cast<int32_t *>(ctx->d_status.ptr), static_cast<uint32_t>(seg_rows),
                                         static_cast<uint32_t>(n_segs), static_cast<uint32_t *>(ctx->d_seg_valid.ptr),
                                         stream));
        // from here on the batch has n_pages statuses: a page's own, or else its index stream's
        for (uint64_t i = 0; i < n_pages; ++i)
            if (ctx->host_status[i] == PQ2_OK) ctx->host_status[i] = ctx->host_status[n_pages + i];
        ctx->host_status.resize(n_pages);
    }
    ctx->last_n_pages = n_pages;
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

int32_t pq2_decode_selected_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                         uint64_t n_pages, const pq2_interval *intervals,
                                         const uint64_t *interval_offsets, uint32_t *d_out, uint64_t out_values,
                                         int32_t *status, void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_pages > 0 && (!descs || !interval_offsets)) return fail(ctx, PQ2_INVALID_PARAMETER, "an array is NULL");
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_out misaligned");
    // every page decoded in full into the context's buffer, 16-byte aligned
    std::vector<pq2_page_desc> full(descs, descs + n_pages);
    std::vector<SelectPiece> pieces;
    TileMask mask;  // the tiles no interval touches are not expanded
    mask.at.assign(n_pages, 0);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_pages; ++i) {
        full[i].out_off = at;
        if (descs[i].num_values <= PQ2_MAX_PAGE_VALUES && mask.words.size() < 0x7fffffffull) {
            mask.at[i] = static_cast<uint32_t>(mask.words.size()) + 1u;
            mask.words.resize(mask.words.size() + ((descs[i].num_values + kTileValues - 1) / kTileValues + 31) / 32, 0u);
        }
        if (interval_offsets[i] > interval_offsets[i + 1])
            return fail(ctx, PQ2_INVALID_PARAMETER, "interval_offsets must ascend");
        if (interval_offsets[i] < interval_offsets[i + 1] && !intervals)
            return fail(ctx, PQ2_INVALID_PARAMETER, "intervals is NULL");
        uint64_t current = 0, selected = 0;  // SliceFilteredIter::current (utils.rs:101, 128-131)
        for (uint64_t k = interval_offsets[i]; k < interval_offsets[i + 1]; ++k) {
            const pq2_interval &iv = intervals[k];
            if (iv.length == 0 || iv.start < current || iv.start > descs[i].num_values ||
                iv.length > descs[i].num_values - iv.start)
                return fail(ctx, PQ2_INVALID_PARAMETER,
                            "intervals must hold at least one row, ascend without overlap and end inside the page");
            current = iv.start + iv.length;
            selected += iv.length;
            if (mask.at[i] != 0)
                for (uint64_t t = iv.start / kTileValues; t <= (iv.start + iv.length - 1) / kTileValues; ++t)
                    mask.words[mask.at[i] - 1u + (t >> 5)] |= 1u << (t & 31u);
        }
        if (descs[i].out_off > out_values || selected > out_values - descs[i].out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "selected rows outside the output buffer");
        at += (descs[i].num_values + 3) & ~uint64_t(3);
    }
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    if ((st = reserve(ctx, ctx->d_indices, std::max<uint64_t>(at, 1) * sizeof(uint32_t)))) return st;
    uint32_t *d_full = static_cast<uint32_t *>(ctx->d_indices.ptr);
    if ((st = enqueue_batch(ctx, d_in, in_bytes, full.data(), n_pages, d_full, at, stream, nullptr, 0, &mask))) return st;
    for (uint64_t i = 0; i < n_pages; ++i) {
        if (ctx->host_status[i] != PQ2_OK) continue;
        uint64_t dst = descs[i].out_off;
        for (uint64_t k = interval_offsets[i]; k < interval_offsets[i + 1]; ++k) {
            for (uint64_t o = 0; o < intervals[k].length; o += kSelectPieceValues) {
                SelectPiece pc;
                pc.src_addr = reinterpret_cast<uint64_t>(d_full + full[i].out_off + intervals[k].start + o);
                pc.dst_addr = reinterpret_cast<uint64_t>(d_out + dst + o);
                pc.count = static_cast<uint32_t>(std::min<uint64_t>(kSelectPieceValues, intervals[k].length - o));
                pc.reserved = 0;
                pieces.push_back(pc);
            }
            dst += intervals[k].length;
        }
    }
    if (pieces.size() > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many intervals in one batch");
    if (!pieces.empty()) {
        const size_t bytes = pieces.size() * sizeof(SelectPiece);
        if ((st = stage_block(ctx, bytes, bytes))) return st;
        std::memcpy(ctx->h_gather, pieces.data(), bytes);
        if ((st = upload_block(ctx, bytes, stream))) return st;
        PQ2_HIP(ctx, pq2_launch_select(ctx->d_gather.ptr, static_cast<uint32_t>(pieces.size()), stream));
    }
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

int32_t pq2_filtered_validity_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes, const pq2_page_desc *descs,
                                           uint64_t n_pages, const pq2_interval *intervals,
                                           const uint64_t *interval_offsets, uint8_t *d_validity, uint64_t validity_rows,
                                           uint32_t *d_counts, int32_t *status, void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_pages > 0 && (!descs || !interval_offsets)) return fail(ctx, PQ2_INVALID_PARAMETER, "an array is NULL");
    if ((reinterpret_cast<uintptr_t>(d_counts) & 3u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_counts misaligned");
    // every page's levels decoded in full into the context's buffer, 16-byte aligned
    std::vector<pq2_page_desc> full(descs, descs + n_pages);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_pages; ++i) {
        if (descs[i].num_bits != 1) return fail(ctx, PQ2_INVALID_PARAMETER, "levels of a bitmap are one bit wide");
        full[i].out_off = at;
        if (interval_offsets[i] > interval_offsets[i + 1])
            return fail(ctx, PQ2_INVALID_PARAMETER, "interval_offsets must ascend");
        if (interval_offsets[i] < interval_offsets[i + 1] && (!intervals || !d_counts))
            return fail(ctx, PQ2_INVALID_PARAMETER, "intervals or d_counts is NULL");
        uint64_t current = 0, selected = 0;
        for (uint64_t k = interval_offsets[i]; k < interval_offsets[i + 1]; ++k) {
            const pq2_interval &iv = intervals[k];
            if (iv.length == 0 || iv.start < current || iv.start > descs[i].num_values ||
                iv.length > descs[i].num_values - iv.start)
                return fail(ctx, PQ2_INVALID_PARAMETER,
                            "intervals must hold at least one row, ascend without overlap and end inside the page");
            current = iv.start + iv.length;
            selected += iv.length;
        }
        if (descs[i].out_off > validity_rows || selected > validity_rows - descs[i].out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "selected rows outside the validity buffer");
        at += (descs[i].num_values + 3) & ~uint64_t(3);
    }
    const uint64_t n_intervals = n_pages ? interval_offsets[n_pages] : 0;
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    if ((st = reserve(ctx, ctx->d_indices, std::max<uint64_t>(at, 1) * sizeof(uint32_t)))) return st;
    uint32_t *d_full = static_cast<uint32_t *>(ctx->d_indices.ptr);
    if ((st = enqueue_batch(ctx, d_in, in_bytes, full.data(), n_pages, d_full, at, stream))) return st;
    std::vector<FilterPiece> pieces;
    for (uint64_t i = 0; i < n_pages; ++i) {
        if (ctx->host_status[i] != PQ2_OK) continue;
        uint64_t dst = descs[i].out_off, current = 0;
        auto add = [&](uint64_t first, uint64_t rows, uint8_t *to, uint64_t counter) {
            for (uint64_t o = 0; o < rows; o += kSelectPieceValues) {
                FilterPiece pc;
                pc.src_addr = reinterpret_cast<uint64_t>(d_full + full[i].out_off + first + o);
                pc.dst_addr = to ? reinterpret_cast<uint64_t>(to + o) : 0;
                pc.count = static_cast<uint32_t>(std::min<uint64_t>(kSelectPieceValues, rows - o));
                pc.counter = static_cast<uint32_t>(counter);
                pieces.push_back(pc);
            }
        };
        for (uint64_t k = interval_offsets[i]; k < interval_offsets[i + 1]; ++k) {
            add(current, intervals[k].start - current, nullptr, 2 * k);              // Skipped
            add(intervals[k].start, intervals[k].length, d_validity + dst, 2 * k + 1);  // Bitmap / Repeated
            current = intervals[k].start + intervals[k].length;
            dst += intervals[k].length;
        }
    }
    if (pieces.size() > 0x7fffffffull || n_intervals > 0x3fffffffull)
        return fail(ctx, PQ2_INVALID_PARAMETER, "too many intervals in one batch");
    if (n_intervals) PQ2_HIP(ctx, hipMemsetAsync(d_counts, 0, 2 * n_intervals * sizeof(uint32_t), stream));
    if (!pieces.empty()) {
        const size_t bytes = pieces.size() * sizeof(FilterPiece);
        if ((st = stage_block(ctx, bytes, bytes))) return st;
        std::memcpy(ctx->h_gather, pieces.data(), bytes);
        if ((st = upload_block(ctx, bytes, stream))) return st;
        PQ2_HIP(ctx, pq2_launch_filtered(ctx->d_gather.ptr, static_cast<uint32_t>(pieces.size()), d_counts, stream));
    }
    if (status) return fetch_status(ctx, status, n_pages);
    return PQ2_OK;
}

int32_t pq2_decode_raw_pages_v1_device(pq2_ctx *ctx, const void *d_buf, uint64_t buf_bytes, const pq2_raw_page *pages,
                                       uint64_t n_pages, uint32_t max_index_bits, uint32_t *d_rep, uint32_t *d_def,
                                       uint32_t *d_idx, uint64_t out_values, int32_t *status, void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_pages > 0x2aaaaaaaull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many pages in one batch");
    if (n_pages > 0 && !pages) return fail(ctx, PQ2_INVALID_PARAMETER, "pages is NULL");
    // the statuses of the page front end live in scratch that the next call reuses: they are delivered with this call
    if (n_pages > 0 && !status) return fail(ctx, PQ2_INVALID_PARAMETER, "status is NULL");
    if (max_index_bits == 0 || max_index_bits > 32) max_index_bits = 32;
    uint32_t *const outs[3] = {d_rep, d_def, d_idx};
    for (uint32_t *o : outs)
        if ((reinterpret_cast<uintptr_t>(o) & 3u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "an output is misaligned");
    // three streams per page; until front_kernel has run they span the whole page buffer (an upper bound of
    // every stream's length, which sizes the header-position list) and the indices have the largest width
    std::vector<pq2_page_desc> streams(3 * n_pages);
    std::vector<RawPageDev> raw(n_pages);
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_raw_page &p = pages[i];
        if (p.buf_off > buf_bytes || p.buf_len > buf_bytes - p.buf_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page buffer outside the buffer");
        if (p.out_off > out_values || p.num_values > out_values - p.out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page output outside the output buffers");
        const bool has[3] = {p.max_rep_level > 0, p.max_def_level > 0, (p.flags & PQ2_RAW_DICT_INDICES) != 0};
        const uint32_t bits[3] = {pq2_get_bit_width(static_cast<int16_t>(p.max_rep_level)),
                                  pq2_get_bit_width(static_cast<int16_t>(p.max_def_level)), max_index_bits};
        for (int k = 0; k < 3; ++k) {
            if (has[k] && p.num_values > 0 && !outs[k]) return fail(ctx, PQ2_INVALID_PARAMETER, "an output that pages need is NULL");
            pq2_page_desc &d = streams[3 * i + k];
            d.in_off = p.buf_off;
            d.in_len = p.buf_len;
            d.num_bits = bits[k];
            d.num_values = has[k] ? p.num_values : 0;
            // absolute: enqueue_batch is given a null base
            d.out_off = (reinterpret_cast<uintptr_t>(outs[k]) >> 2) + p.out_off;
        }
        raw[i].buf_addr = reinterpret_cast<uint64_t>(d_buf) + p.buf_off;
        raw[i].buf_len = p.buf_len;
        raw[i].what = (has[0] ? 1u : 0u) | (has[1] ? 2u : 0u) | (has[2] ? 4u : 0u) | (max_index_bits << 8);
    }
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    const size_t raw_bytes = n_pages * sizeof(RawPageDev), all_bytes = raw_bytes + n_pages * sizeof(int32_t);
    FrontJob job{nullptr, static_cast<uint32_t>(n_pages), nullptr};
    if (n_pages > 0) {
        if ((st = stage_block(ctx, raw_bytes, all_bytes))) return st;
        std::memcpy(ctx->h_gather, raw.data(), raw_bytes);
        if ((st = upload_block(ctx, raw_bytes, stream))) return st;
        job.d_raw = ctx->d_gather.ptr;
        job.d_front_status = reinterpret_cast<int32_t *>(static_cast<char *>(ctx->d_gather.ptr) + raw_bytes);
    }
    if ((st = enqueue_batch(ctx, nullptr, ~0ull, streams.data(), 3 * n_pages, nullptr, ~0ull >> 3, stream,
                            n_pages ? &job : nullptr, max_index_bits)))
        return st;
    if (n_pages == 0) return PQ2_OK;
    std::vector<int32_t> of_streams(3 * n_pages), of_front(n_pages);
    if ((st = fetch_status(ctx, of_streams.data(), 3 * n_pages))) return st;  // waits for the batch
    PQ2_HIP(ctx, hipMemcpy(of_front.data(), job.d_front_status, n_pages * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n_pages; ++i) {
        status[i] = of_front[i];
        for (int k = 0; k < 3 && status[i] == PQ2_OK; ++k) status[i] = of_streams[3 * i + k];
    }
    return PQ2_OK;
}

int32_t pq2_fetch_status(pq2_ctx *ctx, int32_t *status, uint64_t n_pages) {
    if (!ctx || (!status && n_pages)) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    return fetch_status(ctx, status, n_pages);
}

}  // extern "C"

namespace {

constexpr uint64_t kChunkValues = 16ull << 20;       // a chunk of the host path ends once it holds this many values
constexpr uint64_t kSingleShotBytes = 32ull << 20;   // batches below this go through in one piece

// One chunk of a host batch through one lane: upload of the byte range its pages span, decode, download of the
// value ranges its pages cover (adjacent pages in one copy), statuses.  Everything is enqueued on the lane's stream.
int32_t enqueue_host_chunk(pq2_ctx *ctx, pq2_ctx *lane, const uint8_t *h_in, const pq2_page_desc *descs, uint64_t first,
                           uint64_t last, uint32_t *h_out, int32_t *status, std::vector<pq2_page_desc> &scratch,
                           std::vector<std::pair<uint64_t, int32_t>> &overrides) {
    uint64_t in_lo = ~0ull, in_hi = 0, out_lo = ~0ull, out_hi = 0;
    for (uint64_t i = first; i < last; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.num_values == 0) continue;
        out_lo = std::min(out_lo, d.out_off);
        out_hi = std::max(out_hi, d.out_off + d.num_values);
        if (d.in_len == 0) continue;
        in_lo = std::min(in_lo, d.in_off);
        in_hi = std::max(in_hi, d.in_off + d.in_len);
    }
    if (in_hi <= in_lo) in_lo = in_hi = 0;
    if (out_hi <= out_lo) out_lo = out_hi = 0;
    scratch.assign(descs + first, descs + last);
    for (pq2_page_desc &d : scratch) {
        d.in_off = (d.in_len == 0 || d.num_values == 0) ? 0 : d.in_off - in_lo;
        if (d.num_values == 0) d.in_len = 0;
        d.out_off = d.num_values == 0 ? 0 : d.out_off - out_lo;
    }
    hipStream_t s = lane->lane_stream;
    // the lane's arenas and metadata are reused: its previous chunk has to be through
    PQ2_HIP(ctx, hipStreamSynchronize(s));
    int32_t st;
    // +16: the kernels read whole aligned dwords, never beyond the 16-byte line of the last stream byte
    if ((st = reserve(ctx, lane->d_in, in_hi - in_lo + 16))) return st;
    if ((st = reserve(ctx, lane->d_out, std::max<uint64_t>(out_hi - out_lo, 1) * sizeof(uint32_t)))) return st;
    if (in_hi > in_lo) PQ2_HIP(ctx, hipMemcpyAsync(lane->d_in.ptr, h_in + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, s));
    st = enqueue_batch(lane, lane->d_in.ptr, in_hi - in_lo, scratch.data(), last - first,
                       static_cast<uint32_t *>(lane->d_out.ptr), out_hi - out_lo, s);
    if (st != PQ2_OK) {
        ctx->last_error = lane->last_error;
        return st;
    }
    // values back: one copy per maximal range of pages that follow each other in the output
    uint64_t run_lo = 0, run_hi = 0;
    auto flush = [&]() -> int32_t {
        if (run_hi > run_lo)
            PQ2_HIP(ctx, hipMemcpyAsync(h_out + run_lo, static_cast<uint32_t *>(lane->d_out.ptr) + (run_lo - out_lo),
                                        (run_hi - run_lo) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        return PQ2_OK;
    };
    for (uint64_t i = first; i < last; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.num_values == 0) continue;
        if (d.out_off == run_hi && run_hi > run_lo) {
            run_hi += d.num_values;
        } else {
            if ((st = flush())) return st;
            run_lo = d.out_off;
            run_hi = d.out_off + d.num_values;
        }
    }
    if ((st = flush())) return st;
    if (status) {
        PQ2_HIP(ctx, hipMemcpyAsync(ctx->h_status + first, lane->d_status.ptr, (last - first) * sizeof(int32_t),
                                    hipMemcpyDeviceToHost, s));
        for (uint64_t i = first; i < last; ++i)
            if (lane->host_status[i - first] != PQ2_OK) overrides.emplace_back(i, lane->host_status[i - first]);
    }
    return PQ2_OK;
}

}  // namespace

extern "C" {

int32_t pq2_decode_batch_host(pq2_ctx *ctx, const void *h_in, uint64_t in_bytes, const pq2_page_desc *descs,
                              uint64_t n_pages, uint32_t *h_out, uint64_t out_values, int32_t *status) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    ctx->pages_total += n_pages;
    int32_t st;
    uint64_t total_values = 0;
    for (uint64_t i = 0; i < n_pages; ++i) total_values += descs ? descs[i].num_values : 0;
    if (in_bytes + total_values * sizeof(uint32_t) < kSingleShotBytes) {
        // +16: the kernels read whole aligned dwords, never beyond the 16-byte line of the last stream byte
        if ((st = reserve(ctx, ctx->d_in, in_bytes + 16))) return st;
        if ((st = reserve(ctx, ctx->d_out, std::max<uint64_t>(out_values, 1) * sizeof(uint32_t)))) return st;
        hipStream_t s = nullptr;
        PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_in.ptr, h_in, in_bytes, hipMemcpyHostToDevice, s));
        st = enqueue_batch(ctx, ctx->d_in.ptr, in_bytes, descs, n_pages, static_cast<uint32_t *>(ctx->d_out.ptr),
                           out_values, s);
        if (st != PQ2_OK) return st;
        // only the ranges the pages cover are defined; copy the covering span back
        uint64_t lo = out_values, hi = 0;
        for (uint64_t i = 0; i < n_pages; ++i) {
            if (descs[i].num_values == 0) continue;
            lo = std::min(lo, descs[i].out_off);
            hi = std::max(hi, descs[i].out_off + descs[i].num_values);
        }
        if (hi > lo)
            PQ2_HIP(ctx, hipMemcpyAsync(h_out + lo, static_cast<uint32_t *>(ctx->d_out.ptr) + lo,
                                        (hi - lo) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (status) return fetch_status(ctx, status, n_pages);
        PQ2_HIP(ctx, hipStreamSynchronize(s));
        return PQ2_OK;
    }

    // the same checks as the device entry point, against the caller's buffers, before any chunk is started
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.in_len > PQ2_MAX_PAGE_BYTES) return fail(ctx, PQ2_INVALID_PARAMETER, "page stream too long");
        if (d.in_off > in_bytes || d.in_len > in_bytes - d.in_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page stream outside the input buffer");
        if (d.num_values > PQ2_MAX_PAGE_VALUES) return fail(ctx, PQ2_INVALID_PARAMETER, "page has too many values");
        if (d.out_off > out_values || d.num_values > out_values - d.out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "page output outside the output buffer");
    }
    for (pq2_ctx *&lane : ctx->lanes) {
        if (lane) continue;
        if ((st = pq2_ctx_create(ctx->device, &lane))) return fail(ctx, st, "cannot create a lane of the host path");
        lane->profiling = false;
        lane->mode = ctx->mode;
        lane->min_rounded_tiles = ctx->min_rounded_tiles;
        std::copy(ctx->round_split, ctx->round_split + kRounds - 1, lane->round_split);
        PQ2_HIP(ctx, hipStreamCreateWithFlags(&lane->lane_stream, hipStreamNonBlocking));
    }
    if (status && ctx->h_status_cap < n_pages) {
        if (ctx->h_status) PQ2_HIP(ctx, hipHostFree(ctx->h_status));
        ctx->h_status = nullptr;
        ctx->h_status_cap = 0;
        PQ2_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_status), n_pages * sizeof(int32_t),
                                   hipHostMallocDefault));
        ctx->h_status_cap = n_pages;
    }
    std::vector<pq2_page_desc> scratch;
    std::vector<std::pair<uint64_t, int32_t>> overrides;
    uint64_t first = 0, chunk = 0;
    st = PQ2_OK;
    while (first < n_pages && st == PQ2_OK) {
        uint64_t last = first, values = 0;
        while (last < n_pages && (values < kChunkValues || last == first)) values += descs[last++].num_values;
        st = enqueue_host_chunk(ctx, ctx->lanes[chunk % 3], static_cast<const uint8_t *>(h_in), descs, first, last, h_out,
                                status, scratch, overrides);
        first = last;
        ++chunk;
    }
    for (pq2_ctx *lane : ctx->lanes) {
        const hipError_t e = hipStreamSynchronize(lane->lane_stream);
        if (e != hipSuccess && st == PQ2_OK) st = hip_fail(ctx, e, "hipStreamSynchronize");
    }
    if (st != PQ2_OK) return st;
    if (status) std::memcpy(status, ctx->h_status, n_pages * sizeof(int32_t));
    for (const auto &o : overrides) status[o.first] = o.second;
    ctx->last_n_pages = 0;  // the statuses were delivered with the call
    return PQ2_OK;
}

static int32_t check_bitunpack(pq2_ctx *ctx, uint64_t packed_len, uint32_t num_bits, uint64_t *length,
                               uint32_t max_bits = 32) {
    // src/encoding/bitpacked/decode.rs:30-50, in the reference's order
    if (num_bits == 0) return fail(ctx, PQ2_OUT_OF_SPEC, "Bitpacking requires num_bits > 0");
    if (static_cast<unsigned __int128>(packed_len) * 8 < static_cast<unsigned __int128>(*length) * num_bits)
        return fail(ctx, PQ2_OUT_OF_SPEC, "Unpacking " + std::to_string(*length) + " items with a number of bits " +
                                               std::to_string(num_bits) + " requires at least " +
                                               std::to_string(*length * num_bits / 8) + " bytes.");
    if (packed_len == 0) *length = 0;
    if (num_bits > max_bits && packed_len > 0) return fail(ctx, PQ2_REFERENCE_WOULD_PANIC, "invalid num_bits");
    if (packed_len > PQ2_MAX_PAGE_BYTES || *length > PQ2_MAX_PAGE_VALUES)
        return fail(ctx, PQ2_INVALID_PARAMETER, "buffer too large for one call");
    return PQ2_OK;
}

int32_t pq2_bitunpack_u32_device(pq2_ctx *ctx, const void *d_packed, uint64_t packed_len, uint32_t num_bits,
                                 uint64_t length, uint32_t *d_out, void *stream) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    int32_t st = check_bitunpack(ctx, packed_len, num_bits, &length);
    if (st != PQ2_OK) return st;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_out misaligned");
    PQ2_HIP(ctx, pq2_launch_bitunpack(d_packed, static_cast<uint32_t>(packed_len), num_bits,
                                      static_cast<uint32_t>(length), d_out, static_cast<hipStream_t>(stream)));
    return PQ2_OK;
}

int32_t pq2_bitunpack_u32_host(pq2_ctx *ctx, const void *packed, uint64_t packed_len, uint32_t num_bits,
                               uint64_t length, uint32_t *out) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    int32_t st = check_bitunpack(ctx, packed_len, num_bits, &length);
    if (st != PQ2_OK) return st;
    if (length == 0) return PQ2_OK;
    if ((st = reserve(ctx, ctx->d_in, packed_len + 16))) return st;
    if ((st = reserve(ctx, ctx->d_out, length * sizeof(uint32_t)))) return st;
    hipStream_t s = nullptr;
    PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_in.ptr, packed, packed_len, hipMemcpyHostToDevice, s));
    PQ2_HIP(ctx, pq2_launch_bitunpack(ctx->d_in.ptr, static_cast<uint32_t>(packed_len), num_bits,
                                      static_cast<uint32_t>(length), static_cast<uint32_t *>(ctx->d_out.ptr), s));
    PQ2_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, length * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PQ2_HIP(ctx, hipStreamSynchronize(s));
    return PQ2_OK;
}

int32_t pq2_bitunpack_u64_device(pq2_ctx *ctx, const void *d_packed, uint64_t packed_len, uint32_t num_bits,
                                 uint64_t length, uint64_t *d_out, void *stream) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    int32_t st = check_bitunpack(ctx, packed_len, num_bits, &length, 64);
    if (st != PQ2_OK) return st;
    if ((reinterpret_cast<uintptr_t>(d_out) & 7u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_out misaligned");
    PQ2_HIP(ctx, pq2_launch_bitunpack64(d_packed, static_cast<uint32_t>(packed_len), num_bits,
                                        static_cast<uint32_t>(length), d_out, static_cast<hipStream_t>(stream)));
    return PQ2_OK;
}

int32_t pq2_bitunpack_u64_host(pq2_ctx *ctx, const void *packed, uint64_t packed_len, uint32_t num_bits,
                               uint64_t length, uint64_t *out) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    int32_t st = check_bitunpack(ctx, packed_len, num_bits, &length, 64);
    if (st != PQ2_OK) return st;
    if (length == 0) return PQ2_OK;
    if ((st = reserve(ctx, ctx->d_in, packed_len + 16))) return st;
    if ((st = reserve(ctx, ctx->d_out, length * sizeof(uint64_t)))) return st;
    hipStream_t s = nullptr;
    PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_in.ptr, packed, packed_len, hipMemcpyHostToDevice, s));
    PQ2_HIP(ctx, pq2_launch_bitunpack64(ctx->d_in.ptr, static_cast<uint32_t>(packed_len), num_bits,
                                        static_cast<uint32_t>(length), static_cast<uint64_t *>(ctx->d_out.ptr), s));
    PQ2_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, length * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    PQ2_HIP(ctx, hipStreamSynchronize(s));
    return PQ2_OK;
}

int32_t pq2_delta_bitpacked_decode_batch_device(pq2_ctx *ctx, const void *d_in, uint64_t in_bytes,
                                                const pq2_page_desc *descs, uint64_t n_pages, int64_t *d_out,
                                                uint64_t out_values, uint64_t *num_values, uint64_t *consumed_bytes,
                                                int32_t *status, void *stream_) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_pages > 0x7fffffffull) return fail(ctx, PQ2_INVALID_PARAMETER, "too many streams in one batch");
    if (n_pages > 0 && (!descs || !status)) return fail(ctx, PQ2_INVALID_PARAMETER, "descs or status is NULL");
    if ((reinterpret_cast<uintptr_t>(d_out) & 7u) != 0) return fail(ctx, PQ2_INVALID_PARAMETER, "d_out misaligned");
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.in_len > PQ2_MAX_PAGE_BYTES || d.in_off > in_bytes || d.in_len > in_bytes - d.in_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "stream " + std::to_string(i) + " outside the input buffer");
        if (d.num_values > PQ2_MAX_PAGE_VALUES || d.out_off > out_values || d.num_values > out_values - d.out_off)
            return fail(ctx, PQ2_INVALID_PARAMETER, "stream " + std::to_string(i) + " outside the output buffer");
    }
    if (n_pages == 0) return PQ2_OK;
    int32_t st;
    if (ctx->busy && ctx->last_stream != stream) PQ2_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    // one block: the descriptors, then the results
    const size_t desc_bytes = (n_pages * sizeof(DeltaPage) + 15) & ~size_t(15);
    const size_t all_bytes = desc_bytes + n_pages * sizeof(DeltaResult);
    if ((st = stage_block(ctx, desc_bytes, all_bytes))) return st;
    DeltaPage *h = static_cast<DeltaPage *>(ctx->h_gather);
    for (uint64_t i = 0; i < n_pages; ++i) {
        h[i].in_addr = reinterpret_cast<uint64_t>(static_cast<const uint8_t *>(d_in) + descs[i].in_off);
        h[i].out_addr = reinterpret_cast<uint64_t>(d_out + descs[i].out_off);
        h[i].in_len = descs[i].in_len;
        h[i].cap = static_cast<uint32_t>(descs[i].num_values);
    }
    if ((st = upload_block(ctx, n_pages * sizeof(DeltaPage), stream))) return st;
    uint8_t *d_results = static_cast<uint8_t *>(ctx->d_gather.ptr) + desc_bytes;
    PQ2_HIP(ctx, pq2_launch_delta(ctx->d_gather.ptr, static_cast<uint32_t>(n_pages), d_results, stream));
    std::vector<DeltaResult> results(n_pages);
    PQ2_HIP(ctx, hipMemcpyAsync(results.data(), d_results, n_pages * sizeof(DeltaResult), hipMemcpyDeviceToHost,
                                stream));
    PQ2_HIP(ctx, hipStreamSynchronize(stream));
    for (uint64_t i = 0; i < n_pages; ++i) {
        status[i] = results[i].status;
        if (num_values) num_values[i] = results[i].num_values;
        if (consumed_bytes) consumed_bytes[i] = results[i].consumed;
    }
    return PQ2_OK;
}

int32_t pq2_delta_bitpacked_decode_host(pq2_ctx *ctx, const void *values, uint64_t len, int64_t *out, uint64_t cap,
                                        uint64_t *num_values, uint64_t *consumed_bytes) {
    if (!ctx) return PQ2_INVALID_PARAMETER;
    PQ2_HIP(ctx, hipSetDevice(ctx->device));
    if (len > PQ2_MAX_PAGE_BYTES || cap > PQ2_MAX_PAGE_VALUES)
        return fail(ctx, PQ2_INVALID_PARAMETER, "buffer too large for one call");
    int32_t st;
    if ((st = reserve(ctx, ctx->d_in, len + 16))) return st;
    if ((st = reserve(ctx, ctx->d_out, std::max<uint64_t>(cap, 1) * sizeof(int64_t)))) return st;
    hipStream_t s = nullptr;
    if (len) PQ2_HIP(ctx, hipMemcpyAsync(ctx->d_in.ptr, values, len, hipMemcpyHostToDevice, s));
    pq2_page_desc d = {};
    d.in_len = static_cast<uint32_t>(len);
    d.num_values = cap;
    int32_t page_status = PQ2_OK;
    uint64_t n = 0;
    if ((st = pq2_delta_bitpacked_decode_batch_device(ctx, ctx->d_in.ptr, len, &d, 1,
                                                      static_cast<int64_t *>(ctx->d_out.ptr), cap, &n, consumed_bytes,
                                                      &page_status, s)))
        return st;
    if (num_values) *num_values = n;
    if (page_status != PQ2_OK) return fail(ctx, page_status, "delta_bitpacked::Decoder");
    if (n) PQ2_HIP(ctx, hipMemcpy(out, ctx->d_out.ptr, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return PQ2_OK;
}

// What a page costs its GPU: the bytes it moves through HBM, stream in and values out.
static inline uint64_t page_cost(const pq2_page_desc &d) { return uint64_t(d.in_len) + d.num_values * sizeof(uint32_t); }

// A run header costs a decoder about what 256 bytes of traffic cost it (BASELINE's adversarial pages, a header per
// six values, take a GPU 7 to 13 times as long as index pages of the same size: 9 to 16 us a page against 1.2).
static constexpr uint64_t kHeaderCostBytes = 256;
// How many run headers a stream holds, from its first headers (uleb128.rs:3-23, decoder.rs:30-64): exact where the
// sample reaches the stream's end, else the sample's bytes per header carried over the whole stream.
static uint64_t estimate_headers(const uint8_t *stream, uint64_t len, uint32_t num_bits) {
    if (len == 0 || num_bits == 0 || num_bits > 32) return 0;
    const uint64_t rle_bytes = (num_bits + 7) / 8;
    uint64_t pos = 0, headers = 0;
    while (pos < len && headers < 256) {
        uint64_t indicator = 0;
        uint32_t shift = 0;
        bool ended = false;
        while (pos < len && shift < 64) {
            const uint8_t byte = stream[pos++];
            indicator |= uint64_t(byte & 0x7f) << shift;
            shift += 7;
            if (!(byte & 0x80)) {
                ended = true;
                break;
            }
        }
        if (!ended) break;
        ++headers;
        const uint64_t body = (indicator & 1) ? (indicator >> 1) * num_bits : rle_bytes;
        if (body > len - pos) {
            pos = len;
            break;
        }
        pos += body;
    }
    if (headers == 0) return 0;
    if (pos >= len) return headers;
    return static_cast<uint64_t>(double(headers) * double(len) / double(pos) + 0.5);
}

int32_t pq2_page_costs(const void *h_in, uint64_t in_bytes, const pq2_page_desc *descs, uint64_t n_pages, uint64_t *costs) {
    if (n_pages && (!descs || !costs)) return PQ2_INVALID_PARAMETER;
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_page_desc &d = descs[i];
        costs[i] = page_cost(d);
        if (!h_in) continue;
        if (d.in_off > in_bytes || d.in_len > in_bytes - d.in_off) return PQ2_INVALID_PARAMETER;
        if (d.num_values == 0) continue;
        const uint64_t headers = estimate_headers(static_cast<const uint8_t *>(h_in) + d.in_off, d.in_len, d.num_bits);
        costs[i] += std::min<uint64_t>(headers, d.num_values) * kHeaderCostBytes;  // (a run that counts holds a value)
    }
    return PQ2_OK;
}

int32_t pq2_partition_pages(const pq2_page_desc *descs, uint64_t n_pages, uint32_t n_parts, uint64_t *bounds) {
    if (n_parts == 0 || !bounds || (n_pages && !descs)) return PQ2_INVALID_PARAMETER;
    std::vector<uint64_t> costs(n_pages);
    for (uint64_t i = 0; i < n_pages; ++i) costs[i] = page_cost(descs[i]);
    return pq2_partition_by_cost(costs.data(), n_pages, n_parts, bounds);
}

int32_t pq2_partition_by_cost(const uint64_t *costs, uint64_t n_pages, uint32_t n_parts, uint64_t *bounds) {
    if (n_parts == 0 || !bounds || (n_pages && !costs)) return PQ2_INVALID_PARAMETER;
    unsigned __int128 total = 0;
    uint64_t largest = 0;
    for (uint64_t i = 0; i < n_pages; ++i) {
        total += costs[i];
        largest = std::max(largest, costs[i]);
    }
    // first cut: page i goes to the part in which the midpoint of its cost range falls
    std::vector<uint64_t> even(n_parts + 1, 0);
    {
        uint64_t page = 0;
        unsigned __int128 before = 0;
        for (uint32_t r = 1; r <= n_parts; ++r) {
            const unsigned __int128 target2 = total * 2 * r / n_parts;  // twice the cumulative target
            while (page < n_pages && before * 2 + costs[page] <= target2) {
                before += costs[page];
                ++page;
            }
            even[r] = (r == n_parts) ? n_pages : page;
        }
    }
    auto heaviest = [&](const std::vector<uint64_t> &b) {
        unsigned __int128 worst = 0;
        for (uint32_t r = 0; r < n_parts; ++r) {
            unsigned __int128 c = 0;
            for (uint64_t i = b[r]; i < b[r + 1]; ++i) c += costs[i];
            worst = std::max(worst, c);
        }
        return worst;
    };
    // second cut: the smallest limit under which a greedy fill needs no more than n_parts contiguous parts
    // (bisection on the limit); it is optimal for the heaviest part, the one the batch waits for
    auto fill = [&](unsigned __int128 limit, std::vector<uint64_t> *b) {
        uint32_t part = 1;
        unsigned __int128 c = 0;
        for (uint64_t i = 0; i < n_pages; ++i) {
            const uint64_t w = costs[i];
            if (c + w > limit && c > 0) {
                if (part == n_parts) return false;
                if (b) (*b)[part] = i;
                ++part;
                c = 0;
            }
            c += w;
        }
        if (b)
            for (; part <= n_parts; ++part) (*b)[part] = n_pages;
        return true;
    };
    unsigned __int128 lo = std::max<unsigned __int128>(largest, (total + n_parts - 1) / n_parts), hi = total;
    while (lo < hi) {
        const unsigned __int128 mid = lo + (hi - lo) / 2;
        if (fill(mid, nullptr)) hi = mid; else lo = mid + 1;
    }
    std::vector<uint64_t> tight(n_parts + 1, 0);
    fill(lo, &tight);
    const std::vector<uint64_t> &best = heaviest(even) <= heaviest(tight) ? even : tight;
    for (uint32_t r = 0; r <= n_parts; ++r) bounds[r] = best[r];
    return PQ2_OK;
}

int32_t pq2_multi_decode(pq2_ctx *const *ctxs, uint32_t n_ctx, const void *h_in, uint64_t in_bytes,
                         const pq2_page_desc *descs, uint64_t n_pages, uint32_t *h_out, uint64_t out_values,
                         int32_t *status) {
    if (!ctxs || n_ctx == 0 || (n_pages && (!descs || !status))) return PQ2_INVALID_PARAMETER;
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (!ctxs[r]) return PQ2_INVALID_PARAMETER;
    // a context is driven by one thread at a time: the same context twice would be driven by two
    for (uint32_t a = 0; a < n_ctx; ++a)
        for (uint32_t b = a + 1; b < n_ctx; ++b)
            if (ctxs[a] == ctxs[b]) return fail(ctxs[0], PQ2_INVALID_PARAMETER, "the same context is listed twice");
    for (uint64_t i = 0; i < n_pages; ++i) {
        const pq2_page_desc &d = descs[i];
        if (d.in_off > in_bytes || d.in_len > in_bytes - d.in_off)
            return fail(ctxs[0], PQ2_INVALID_PARAMETER, "page stream outside the input buffer");
        if (d.out_off > out_values || d.num_values > out_values - d.out_off)
            return fail(ctxs[0], PQ2_INVALID_PARAMETER, "page output outside the output buffer");
    }
    // (the streams are at hand: the parts are balanced on bytes and run headers, pq2_page_costs)
    std::vector<uint64_t> bounds(n_ctx + 1, 0), costs(n_pages);
    int32_t st = pq2_page_costs(h_in, in_bytes, descs, n_pages, costs.data());
    if (st == PQ2_OK) st = pq2_partition_by_cost(costs.data(), n_pages, n_ctx, bounds.data());
    if (st != PQ2_OK) return st;
    struct Span { uint64_t in_lo, in_hi, out_lo, out_hi; };
    std::vector<Span> spans(n_ctx, Span{in_bytes, 0, out_values, 0});
    for (uint32_t r = 0; r < n_ctx; ++r)
        for (uint64_t i = bounds[r]; i < bounds[r + 1]; ++i) {
            Span &s = spans[r];
            s.in_lo = std::min(s.in_lo, descs[i].in_off);
            s.in_hi = std::max(s.in_hi, descs[i].in_off + descs[i].in_len);
            if (descs[i].num_values == 0) continue;
            s.out_lo = std::min(s.out_lo, descs[i].out_off);
            s.out_hi = std::max(s.out_hi, descs[i].out_off + descs[i].num_values);
        }
    // every GPU copies the span its pages cover back in one piece: the spans must not interleave
    for (uint32_t a = 0; a < n_ctx; ++a)
        for (uint32_t b = a + 1; b < n_ctx; ++b)
            if (spans[a].out_lo < spans[a].out_hi && spans[b].out_lo < spans[b].out_hi &&
                spans[a].out_lo < spans[b].out_hi && spans[b].out_lo < spans[a].out_hi)
                return fail(ctxs[0], PQ2_INVALID_PARAMETER, "page outputs are not laid out in page order");
    std::vector<int32_t> result(n_ctx, PQ2_OK);
    // one host thread per context: each drives the three-lane host path of its own GPU over its page range,
    // on buffers rebased to the span of that range; nothing is exchanged between the GPUs
    auto work = [&](uint32_t r) {
        const uint64_t first = bounds[r], last = bounds[r + 1];
        if (first == last) return;
        Span s = spans[r];
        if (s.out_lo > s.out_hi) s.out_lo = s.out_hi = 0;  // only pages without values
        std::vector<pq2_page_desc> mine(descs + first, descs + last);
        for (pq2_page_desc &d : mine) {
            d.in_off -= s.in_lo;
            d.out_off = d.num_values ? d.out_off - s.out_lo : 0;
        }
        result[r] = pq2_decode_batch_host(ctxs[r], static_cast<const uint8_t *>(h_in) + s.in_lo, s.in_hi - s.in_lo,
                                          mine.data(), mine.size(), h_out + s.out_lo, s.out_hi - s.out_lo,
                                          status + first);
    };
    // (a thread that cannot be started must not take the process down through the C boundary: its part is done by
    // this thread instead)
    std::vector<std::thread> threads;
    std::vector<uint32_t> inline_parts{0};
    for (uint32_t r = 1; r < n_ctx; ++r) {
        try {
            threads.emplace_back(work, r);
        } catch (const std::exception &) {
            inline_parts.push_back(r);
        }
    }
    for (uint32_t r : inline_parts) work(r);
    for (std::thread &t : threads) t.join();
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (result[r] != PQ2_OK) return result[r];
    return PQ2_OK;
}

}  // extern "C"
