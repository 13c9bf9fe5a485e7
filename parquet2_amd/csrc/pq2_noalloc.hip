// pq2_decode_batch_device_noalloc: the batch decode for callers that cannot allocate.
//
// pq2_decode_batch_device keeps its per-batch working memory (device page descriptors, per-page statuses, the
// per-compute-unit tickets) in buffers that the context grows with hipMalloc, and uploads the descriptors through
// page-locked staging memory.  Code under HIP graph capture, tight serving loops and harnesses that forbid
// allocation cannot call that.  This entry point runs ring_kernel -- the kernel that decodes large batches in the
// main entry point: everything its walking and expanding waves hand to each other stays in LDS -- with the little
// device memory it needs inside one fixed array of the library's code object, `scratch` below: it calls no
// hipMalloc / hipFree / hipHostMalloc and no routine that allocates.  The page descriptors come from device memory
// (the caller's own buffer), a small kernel checks them and writes the kernel's descriptors, and the statuses go to
// a device array of the caller.
//
// Every GPU has its own copy of `scratch` at its own address; `scratch` in host code is only the handle that
// hipGetSymbolAddress takes, never a device address.  Every call on a GPU uses that GPU's one array: calls that
// overlap on one GPU (two streams, two threads) race, calls on one stream follow each other safely.
#include <mutex>

#include "../../include/pq2hip.h"
#include "pq2_device.h"

extern "C" hipError_t pq2_launch_ring(const void *pages, uint32_t n_pages, int32_t *status, uint32_t *cu_tickets,
                                      uint32_t epoch, uint32_t max_bits, const uint32_t *tile_mask, const uint32_t *go,
                                      hipEvent_t begin, hipEvent_t end, hipStream_t stream);
extern "C" uint32_t pq2_ring_ticket_bytes();

namespace {

// ---- limit (documented in include/pq2hip.h) ----------------------------------------------------------------
constexpr uint32_t kMaxPages = PQ2_NOALLOC_MAX_PAGES;
constexpr uint32_t kTicketBytes = 16384;  // pq2_ring_ticket_bytes(), checked at set-up

// ---- layout of the array (offsets are multiples of 256) ----------------------------------------------------
constexpr uint64_t align256(uint64_t v) { return (v + 255u) & ~uint64_t(255); }
constexpr uint64_t kOffPages = 0;                                                          // PageDev[kMaxPages]
constexpr uint64_t kOffStreamStatus = align256(kOffPages + kMaxPages * sizeof(pq2::PageDev));  // ring_kernel's
constexpr uint64_t kOffDescStatus = align256(kOffStreamStatus + kMaxPages * 4u);           // the descriptor check's
constexpr uint64_t kOffTickets = align256(kOffDescStatus + kMaxPages * 4u);
constexpr uint64_t kScratchBytes = align256(kOffTickets + kTicketBytes);
static_assert(kScratchBytes <= (1ull << 20), "the fixed working memory is meant to stay modest");

}  // namespace

// Zero when the code object is loaded.  The tickets are cleared by every call (ring_kernel counts a launch's
// workgroups per compute unit from 0); everything else is written by the call before it is read.
static __device__ __align__(256) uint8_t scratch[kScratchBytes];

namespace {

constexpr uint32_t kPrepThreads = 256;

// Checks the page descriptors as pq2_decode_batch_device does on the host and writes the kernel's descriptors, one
// thread per page.  A descriptor that the host entry point would reject fails its own page (PQ2_INVALID_PARAMETER)
// and the page is skipped (it is handed on with no values).
__global__ __launch_bounds__(kPrepThreads) void noalloc_prep_kernel(const pq2_page_desc *__restrict__ descs,
                                                                    uint32_t n_pages, uint64_t in_addr,
                                                                    uint64_t in_bytes, uint64_t out_addr,
                                                                    uint64_t out_values, uint32_t max_bits,
                                                                    pq2::PageDev *__restrict__ pages,
                                                                    int32_t *__restrict__ desc_status) {
    const uint32_t i = blockIdx.x * kPrepThreads + threadIdx.x;
    if (i >= n_pages) return;
    pq2_page_desc d = descs[i];
    const bool bad = d.in_len > PQ2_MAX_PAGE_BYTES || d.in_off > in_bytes || d.in_len > in_bytes - d.in_off ||
                     d.num_values > PQ2_MAX_PAGE_VALUES || d.out_off > out_values ||
                     d.num_values > out_values - d.out_off || d.num_bits > 32u /* unpack32 has no such width */ ||
                     (d.num_bits > max_bits && d.num_values > 0);
    if (bad) d.num_values = 0;
    pq2::PageDev p;
    p.in_addr = in_addr + d.in_off;
    p.out_addr = out_addr + d.out_off * 4u;
    p.run_base = 0;
    p.in_len = d.in_len;
    p.num_values = (uint32_t)d.num_values;
    p.num_bits = d.num_bits;
    for (int r = 0; r < 4; ++r) p.round_tile[r] = p.tile_base[r] = 0;  // (the kernels that work in rounds use these)
    p.reserved = 0;
    pages[i] = p;
    desc_status[i] = bad ? PQ2_INVALID_PARAMETER : PQ2_OK;
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
        if (pq2_ring_ticket_bytes() > kTicketBytes) return PQ2_DEVICE_ERROR;
        uint8_t *ptr = nullptr;
        if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return PQ2_DEVICE_ERROR;
        g_scratch[device] = ptr;
    }
    if (out) *out = g_scratch[device];
    return PQ2_OK;
}

namespace {
// Makes `device` the calling thread's current device and puts the former one back when it goes out of scope.
struct CurrentDevice {
    int was = -1;
    bool ok = false;
    explicit CurrentDevice(int device) {
        if (hipGetDevice(&was) != hipSuccess) was = -1;
        ok = was == device || hipSetDevice(device) == hipSuccess;
        if (was == device) was = -1;
    }
    ~CurrentDevice() {
        if (was >= 0) (void)hipSetDevice(was);
    }
};
}  // namespace

extern "C" int32_t pq2_decode_batch_device_noalloc(int32_t device_id, const void *d_in, uint64_t in_bytes,
                                                   const pq2_page_desc *d_descs, uint64_t n_pages, uint32_t max_bits,
                                                   uint32_t *d_out, uint64_t out_values, int32_t *d_status,
                                                   void *stream_) {
    if (n_pages > kMaxPages) return PQ2_WOULD_OVER_ALLOCATE;  // beyond the fixed array
    if (n_pages > 0 && (!d_descs || !d_status)) return PQ2_INVALID_PARAMETER;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) != 0 || (reinterpret_cast<uintptr_t>(d_descs) & 7u) != 0)
        return PQ2_INVALID_PARAMETER;
    if (max_bits == 0 || max_bits > 32) max_bits = 32;
    CurrentDevice current(device_id);  // the caller's current device is put back on every way out
    if (!current.ok) return PQ2_DEVICE_ERROR;
    uint8_t *base = nullptr;  // this GPU's array: a device address, unlike `scratch` itself
    const int32_t st = pq2_noalloc_setup(device_id, &base);
    if (st != PQ2_OK) return st;
    if (n_pages == 0) return PQ2_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    pq2::PageDev *pages = reinterpret_cast<pq2::PageDev *>(base + kOffPages);
    int32_t *stream_status = reinterpret_cast<int32_t *>(base + kOffStreamStatus);
    int32_t *desc_status = reinterpret_cast<int32_t *>(base + kOffDescStatus);
    const uint32_t n = static_cast<uint32_t>(n_pages);
    // (a memset node and three kernel nodes under capture; the launch counts in the first half of the tickets, which
    // a context's launches clear for each other in turns and a call that may be replayed has to clear itself)
    if (hipMemsetAsync(base + kOffTickets, 0, kTicketBytes, stream) != hipSuccess) return PQ2_DEVICE_ERROR;
    hipLaunchKernelGGL(noalloc_prep_kernel, dim3((n + kPrepThreads - 1u) / kPrepThreads), dim3(kPrepThreads), 0, stream,
                       d_descs, n, (uint64_t) reinterpret_cast<uintptr_t>(d_in), in_bytes,
                       (uint64_t) reinterpret_cast<uintptr_t>(d_out), out_values, max_bits, pages, desc_status);
    if (hipGetLastError() != hipSuccess) return PQ2_DEVICE_ERROR;
    if (pq2_launch_ring(pages, n, stream_status, reinterpret_cast<uint32_t *>(base + kOffTickets), 0u, max_bits, nullptr,
                        nullptr, nullptr, nullptr, stream) != hipSuccess)
        return PQ2_DEVICE_ERROR;
    hipLaunchKernelGGL(noalloc_status_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, desc_status,
                       stream_status, n, d_status);
    if (hipGetLastError() != hipSuccess) return PQ2_DEVICE_ERROR;
    return PQ2_OK;
}
