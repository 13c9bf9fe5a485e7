// pq2_decode_batch_device_noalloc: the batch decode for callers that cannot allocate.
//
// pq2_decode_batch_device keeps its per-batch working memory (device page descriptors, per-page parse state, the
// Resume table, the header-position list, the per-compute-unit tickets) in buffers that the context grows with
// hipMalloc, and uploads the descriptors through page-locked staging memory.  Code under HIP graph capture, tight
// serving loops and harnesses that forbid allocation cannot call that.  This entry point runs the same kernels
// with all of that working memory inside one fixed array of the library's code object, `scratch` below: it calls
// no hipMalloc / hipFree / hipHostMalloc and no routine that allocates.  The page descriptors come from device
// memory (the caller's own buffer), a small kernel checks them and lays out the working memory, and the statuses
// go to a device array of the caller.
//
// Every GPU has its own copy of `scratch` at its own address; `scratch` in host code is only the handle that
// hipGetSymbolAddress takes, never a device address.  Every call on a GPU uses that GPU's one array: calls that
// overlap on one GPU (two streams, two threads) race, calls on one stream follow each other safely.
#include <mutex>

#include "../../include/pq2hip.h"
#include "pq2_device.h"

extern "C" hipError_t pq2_launch_decode(const void *pages, uint32_t n_pages, void *resume, uint32_t *run_pos,
                                        int32_t *status, void *state, uint32_t *cu_tickets, uint32_t max_bits,
                                        const uint32_t *tile_mask, hipStream_t stream);
extern "C" uint32_t pq2_cu_ticket_bytes();

namespace {

// ---- limits (documented in include/pq2hip.h) --------------------------------------------------------------
constexpr uint32_t kMaxPages = PQ2_NOALLOC_MAX_PAGES;
constexpr uint64_t kMaxValues = PQ2_NOALLOC_MAX_VALUES;
constexpr uint64_t kMaxRunEntries = PQ2_NOALLOC_MAX_RUN_ENTRIES;
constexpr uint64_t kMaxTiles = kMaxValues / pq2::kTileValues + kMaxPages;  // every page may end in a partial tile
constexpr uint32_t kTicketBytes = 16384;                                   // pq2_cu_ticket_bytes(), checked at set-up

// ---- layout of the array (offsets are multiples of 256) ----------------------------------------------------
constexpr uint64_t align256(uint64_t v) { return (v + 255u) & ~uint64_t(255); }
constexpr uint64_t kOffPages = 0;                                                          // PageDev[kMaxPages]
constexpr uint64_t kOffStreamStatus = align256(kOffPages + kMaxPages * sizeof(pq2::PageDev));  // decode_kernel's
constexpr uint64_t kOffDescStatus = align256(kOffStreamStatus + kMaxPages * 4u);           // the descriptor check's
constexpr uint64_t kOffState = align256(kOffDescStatus + kMaxPages * 4u);                  // ParseState[kMaxPages]
constexpr uint64_t kOffTickets = align256(kOffState + kMaxPages * sizeof(pq2::ParseState));
constexpr uint64_t kOffResume = align256(kOffTickets + kTicketBytes);                      // Resume[4 * kMaxTiles]
constexpr uint64_t kOffRunPos = align256(kOffResume + kMaxTiles * pq2::kBlocksPerTile * sizeof(pq2::Resume));
constexpr uint64_t kScratchBytes = align256(kOffRunPos + kMaxRunEntries * 4u);
static_assert(kScratchBytes <= (40ull << 20), "the fixed working memory is meant to stay modest");

}  // namespace

// Zero when the code object is loaded.  Nothing in it has to be zero at the start of a call: the tickets only
// ever count up (their low bits are used), everything else is written by the call before it is read.
static __device__ __align__(256) uint8_t scratch[kScratchBytes];

namespace {

constexpr uint32_t kPrepThreads = 1024;
constexpr uint32_t kPagesPerThread = kMaxPages / kPrepThreads;
static_assert(kMaxPages % kPrepThreads == 0, "every thread of the descriptor check takes the same number of pages");

// One workgroup checks the page descriptors as pq2_decode_batch_device does on the host and lays the batch out in
// the working memory: tiles and header-list entries are prefix-summed over the pages (thread t takes pages
// [t * kPagesPerThread, ...)).  A descriptor that the host entry point would reject fails its own page
// (PQ2_INVALID_PARAMETER) and the page is skipped; a batch whose tiles or header lists do not fit the array fails
// every page (PQ2_WOULD_OVER_ALLOCATE) and nothing is decoded.
__global__ __launch_bounds__(kPrepThreads) void noalloc_prep_kernel(const pq2_page_desc *__restrict__ descs,
                                                                    uint32_t n_pages, uint64_t in_addr,
                                                                    uint64_t in_bytes, uint64_t out_addr,
                                                                    uint64_t out_values, uint32_t max_bits,
                                                                    pq2::PageDev *__restrict__ pages,
                                                                    int32_t *__restrict__ desc_status) {
    __shared__ uint64_t wave_tiles[kPrepThreads / 64], wave_runs[kPrepThreads / 64];
    __shared__ uint32_t over;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    pq2_page_desc d[kPagesPerThread];
    int32_t st[kPagesPerThread];
    uint64_t tiles[kPagesPerThread], runs[kPagesPerThread];
    uint64_t my_tiles = 0, my_runs = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPagesPerThread; ++k) {
        const uint32_t i = t * kPagesPerThread + k;
        st[k] = PQ2_OK;
        tiles[k] = runs[k] = 0;
        if (i >= n_pages) continue;
        d[k] = descs[i];
        const bool bad = d[k].in_len > PQ2_MAX_PAGE_BYTES || d[k].in_off > in_bytes ||
                         d[k].in_len > in_bytes - d[k].in_off || d[k].num_values > PQ2_MAX_PAGE_VALUES ||
                         d[k].out_off > out_values || d[k].num_values > out_values - d[k].out_off ||
                         d[k].num_bits > 32u /* unpack32 has no such width */ ||
                         (d[k].num_bits > max_bits && d[k].num_values > 0);
        if (bad) {
            st[k] = PQ2_INVALID_PARAMETER;
            d[k].num_values = 0;
        }
        tiles[k] = (d[k].num_values + pq2::kTileValues - 1u) / pq2::kTileValues;
        // as the host entry point sizes the header-position list: a run that yields values takes at least two
        // stream bytes and one value; the closing entry; the 64 entries a wave may read past it; whole 64-entry rows
        const uint64_t most = d[k].num_values < d[k].in_len / 2u ? d[k].num_values : d[k].in_len / 2u;
        runs[k] = (most + 1u + 64u + 63u) & ~uint64_t(63);
        my_tiles += tiles[k];
        my_runs += runs[k];
    }
    // exclusive prefix sums over the threads: inside the wave by shuffles, over the waves through LDS
    uint64_t incl_tiles = my_tiles, incl_runs = my_runs;
    for (uint32_t s = 1; s < 64u; s <<= 1) {
        const uint64_t a = __shfl_up(incl_tiles, s), b = __shfl_up(incl_runs, s);
        if (lane >= s) incl_tiles += a, incl_runs += b;
    }
    if (lane == 63u) wave_tiles[wave] = incl_tiles, wave_runs[wave] = incl_runs;
    __syncthreads();
    uint64_t tile_at = incl_tiles - my_tiles, run_at = incl_runs - my_runs, all_tiles = 0, all_runs = 0;
    for (uint32_t w = 0; w < kPrepThreads / 64u; ++w) {
        if (w < wave) tile_at += wave_tiles[w], run_at += wave_runs[w];
        all_tiles += wave_tiles[w], all_runs += wave_runs[w];
    }
    if (t == 0) over = (all_tiles > kMaxTiles || all_runs > kMaxRunEntries) ? 1u : 0u;
    __syncthreads();
    const bool too_large = over != 0u;
#pragma unroll
    for (uint32_t k = 0; k < kPagesPerThread; ++k) {
        const uint32_t i = t * kPagesPerThread + k;
        if (i >= n_pages) continue;
        pq2::PageDev p;
        p.in_addr = in_addr + d[k].in_off;
        p.out_addr = out_addr + d[k].out_off * 4u;
        p.run_base = too_large ? 0u : run_at;
        p.in_len = d[k].in_len;
        p.num_values = too_large ? 0u : (uint32_t)d[k].num_values;
        p.num_bits = d[k].num_bits;
        // one round: the page's tiles are [tile_at, tile_at + tiles) of the batch
        const uint32_t base = too_large ? 0u : (uint32_t)tile_at, n_t = too_large ? 0u : (uint32_t)tiles[k];
        p.round_tile[0] = 0, p.round_tile[1] = p.round_tile[2] = p.round_tile[3] = n_t;
        p.tile_base[0] = base, p.tile_base[1] = p.tile_base[2] = p.tile_base[3] = base + n_t;
        p.reserved = 0;
        pages[i] = p;
        desc_status[i] = too_large ? PQ2_WOULD_OVER_ALLOCATE : st[k];
        tile_at += tiles[k];
        run_at += runs[k];
    }
}

// The caller's status of a page: what the descriptor check found, else what the decode found.
__global__ __launch_bounds__(256) void noalloc_status_kernel(const int32_t *__restrict__ desc_status,
                                                             const int32_t *__restrict__ stream_status,
                                                             uint32_t n_pages, int32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_pages) out[i] = desc_status[i] != PQ2_OK ? desc_status[i] : stream_status[i];
}

constexpr int kMaxDevices = 64;
std::mutex g_mutex;
uint8_t *g_scratch[kMaxDevices] = {};  // the device address of `scratch` on every GPU that has been set up

}  // namespace

// Set-up for `device`, which must be the current device: asks the runtime where this GPU's copy of the array
// lies.  Called by pq2_ctx_create (so that a process that creates a context first never does this inside a
// capture or a serving loop) and by the entry point's first call on a device.
extern "C" int32_t pq2_noalloc_setup(int32_t device, uint8_t **out) {
    if (device < 0 || device >= kMaxDevices) return PQ2_INVALID_PARAMETER;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_scratch[device]) {
        if (pq2_cu_ticket_bytes() > kTicketBytes) return PQ2_DEVICE_ERROR;
        uint8_t *ptr = nullptr;
        if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return PQ2_DEVICE_ERROR;
        g_scratch[device] = ptr;
    }
    if (out) *out = g_scratch[device];
    return PQ2_OK;
}

extern "C" int32_t pq2_decode_batch_device_noalloc(int32_t device_id, const void *d_in, uint64_t in_bytes,
                                                   const pq2_page_desc *d_descs, uint64_t n_pages, uint32_t max_bits,
                                                   uint32_t *d_out, uint64_t out_values, int32_t *d_status,
                                                   void *stream_) {
    if (n_pages > kMaxPages || out_values > kMaxValues) return PQ2_WOULD_OVER_ALLOCATE;  // beyond the fixed array
    if (n_pages > 0 && (!d_descs || !d_status)) return PQ2_INVALID_PARAMETER;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0 || (reinterpret_cast<uintptr_t>(d_descs) & 7u) != 0)
        return PQ2_INVALID_PARAMETER;
    if (max_bits == 0 || max_bits > 32) max_bits = 32;
    if (hipSetDevice(device_id) != hipSuccess) return PQ2_DEVICE_ERROR;
    uint8_t *base = nullptr;  // this GPU's array: a device address, unlike `scratch` itself
    const int32_t st = pq2_noalloc_setup(device_id, &base);
    if (st != PQ2_OK) return st;
    if (n_pages == 0) return PQ2_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    pq2::PageDev *pages = reinterpret_cast<pq2::PageDev *>(base + kOffPages);
    int32_t *stream_status = reinterpret_cast<int32_t *>(base + kOffStreamStatus);
    int32_t *desc_status = reinterpret_cast<int32_t *>(base + kOffDescStatus);
    const uint32_t n = static_cast<uint32_t>(n_pages);
    hipLaunchKernelGGL(noalloc_prep_kernel, dim3(1), dim3(kPrepThreads), 0, stream, d_descs, n,
                       (uint64_t) reinterpret_cast<uintptr_t>(d_in), in_bytes,
                       (uint64_t) reinterpret_cast<uintptr_t>(d_out), out_values, max_bits, pages, desc_status);
    if (hipGetLastError() != hipSuccess) return PQ2_DEVICE_ERROR;
    if (pq2_launch_decode(pages, n, base + kOffResume, reinterpret_cast<uint32_t *>(base + kOffRunPos), stream_status,
                          base + kOffState, reinterpret_cast<uint32_t *>(base + kOffTickets), max_bits, nullptr,
                          stream) != hipSuccess)
        return PQ2_DEVICE_ERROR;
    hipLaunchKernelGGL(noalloc_status_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, desc_status,
                       stream_status, n, d_status);
    if (hipGetLastError() != hipSuccess) return PQ2_DEVICE_ERROR;
    return PQ2_OK;
}
