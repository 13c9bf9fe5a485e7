// density_kernel: how dense the run headers of a batch are, found out on the device before the batch's schedule is
// final.
//
// Which schedule is faster for a batch of a hundred to a few thousand pages depends on how many run headers a page
// holds: ring_kernel gives a page one walking wave, so a page takes at least its header count times the 0.1 us
// a header costs that wave, whatever else the GPU could do meanwhile; the scout kernels spread the walk over the
// GPU and take time in proportion to the bytes of the whole batch.  Pages of long runs (a header per hundred
// values) want ring_kernel from about a hundred pages on, pages of short runs (a header per six values) want the
// scout kernels at any number -- and the page descriptors do not tell the two apart: both kinds of stream have
// about four fifths of their bit-packed size (src/encoding/hybrid_rle/decoder.rs:30-64 is one decoder for both).
//
// So one wave per page (of up to 512 pages of the batch, evenly spaced) walks the first headers of its page out of
// registers: 512 bytes of the stream lie eight per lane, a header's bytes are fetched with lane reads; 32 headers are
// enough, and a stream that has not shown as many in its first 500 bytes gets up to four more windows.  A stream of
// long runs still shows a dozen headers at most, too few to say how many the page has; but the pages of a batch
// that share a bit width are pages of one column more often than not, so headers and the bytes they span are summed
// per bit width over the batch, and a page's header count is estimated as its stream length times its width's headers
// per byte (a page that shows 32 headers and more speaks for itself as well: one page of short runs among many of
// long ones is what ring_kernel would wait for).  A second launch of one workgroup does the sums, takes the largest
// estimate, compares what ring_kernel would take with what the host expects of the scout kernels, and writes the
// verdict where the ring_kernel launch behind it (which returns at once on a "no") and the host find it.
#include "pq2_device.h"

namespace pq2 {

constexpr uint32_t kDensityBytes = 512;   // bytes of a stream held in registers at a time
constexpr uint32_t kDensityHeaders = 64;  // headers walked at most
constexpr uint32_t kDensityWindows = 5;   // windows of kDensityBytes a page's walk is given at most
constexpr uint32_t kDensitySolid = 32;    // headers from which on a page's own estimate counts
constexpr uint32_t kDensitySamples = 512; // pages looked at, at most: evenly spaced ones of a larger batch

struct DensityModel {   // microseconds, as measured on an MI355X
    float scout_us;      // the scout kernels + expand_kernel for this batch (from its bytes)
    float ring_fixed_us; // ring_kernel: launch and first fill
    float per_header_us; // ... a header of a page's walk
    float values_us;     // ... the largest page's values, were it nothing but values
    float turns;         // ... pages a workgroup slot takes one after the other
};

// seen[page] = (headers walked, with bit 31 set if the walk reached the stream's end; the bytes they span)
// (sample s of the batch: page s * every)
__global__ __launch_bounds__(256) void density_kernel(const PageDev *__restrict__ pages, uint32_t n_samples, uint32_t every,
                                                      uint2 *__restrict__ seen) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t page = blockIdx.x * 4u + (threadIdx.x >> 6);  // of the samples
    if (page >= n_samples) return;
    const PageDev pg = pages[(size_t)page * every];
    const uint32_t w = pg.num_bits;
    uint32_t headers = 0, walked = 0;
    bool whole = false;  // the walk reached the stream's end: `headers` is the page's count
    if (pg.num_values > 0 && w > 0 && w <= 32 && pg.in_len > 0) {
        const uint32_t delta = (uint32_t)(pg.in_addr & 15u);
        const uint32_t end = delta + pg.in_len;  // a-offset one past the last stream byte
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(pg.in_addr - delta, (end + 3u) & ~3u);
        const uint32_t rb = (w + 7u) >> 3;
        uint32_t pos = delta;
        // A window of 512 bytes at a time: a-offsets [base + 8 lane, base + 8 lane + 8), bytes behind the stream read
        // as 0.  A stream whose runs are long leaves the window after a few headers; it gets a few more windows, each
        // where the walk has come to, so that its headers per byte rest on more than three or four of them.
        for (uint32_t window = 0; window < kDensityWindows; ++window) {
            const uint32_t base = pos & ~7u;
            const uint32_t x0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + lane * 8u, 0, 0);
            const uint32_t x1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + lane * 8u + 4u, 0, 0);
            const uint32_t limit = min((uint64_t)end, (uint64_t)base + kDensityBytes - 12u);
            while (pos < limit && headers < kDensityHeaders) {
                const uint32_t i = __builtin_amdgcn_readfirstlane((pos - base) >> 3);
                // (the lane reads return signed words)
                const uint64_t a = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(x0, i) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(x1, i) << 32);
                const uint64_t b = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(x0, i + 1u) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(x1, i + 1u) << 32);
                const uint32_t sh = (pos & 7u) * 8u;
                const uint64_t v = sh ? (a >> sh) | (b << (64u - sh)) : a;  // the eight bytes from pos on
                // the varint (uleb128.rs:3-23), as far as five bytes go: an indicator beyond 2^35 ends the walk
                uint64_t ind = 0;
                uint32_t c = 0;
                bool more = true;
                for (uint32_t k = 0; k < 5u && more; ++k) {
                    const uint32_t byte = (uint32_t)(v >> (8u * k)) & 0xffu;
                    ind |= (uint64_t)(byte & 0x7fu) << (7u * k);
                    more = (byte & 0x80u) != 0u;
                    ++c;
                }
                ++headers;
                if (more) {
                    window = kDensityWindows;
                    break;
                }
                const uint64_t bytes = (ind & 1u) ? (ind >> 1) * w : rb;
                const uint64_t next = (uint64_t)pos + c + bytes;
                pos = next > 0xffffffffull ? 0xffffffffu : (uint32_t)next;
            }
            if (pos >= end || headers >= kDensitySolid) break;
        }
        whole = pos >= end;
        walked = max(min(pos, end) - delta, 1u);
    }
    if (lane == 0) seen[page] = make_uint2(headers | (whole ? 0x80000000u : 0u), walked);
}

// One workgroup sums what the pages showed per bit width and gives the verdict.
__global__ __launch_bounds__(kDensitySamples) void density_verdict_kernel(const PageDev *__restrict__ pages, uint32_t n_samples,
                                                               uint32_t every, const uint2 *__restrict__ seen, DensityModel model,
                                                               uint32_t *__restrict__ go, uint32_t *__restrict__ host_words,
                                                               uint32_t seq) {
    __shared__ uint32_t headers_of[33], bytes_of[33], longest_of[33], solid;
    if (threadIdx.x < 33u) headers_of[threadIdx.x] = bytes_of[threadIdx.x] = longest_of[threadIdx.x] = 0u;
    if (threadIdx.x == 0) solid = 0u;
    __syncthreads();
    if (threadIdx.x < n_samples) {
        const uint32_t i = threadIdx.x;
        const uint2 s = seen[i];
        const uint32_t headers = s.x & 0x7fffffffu, walked = s.y;
        const bool whole = (s.x >> 31) != 0u;
        if (walked != 0u) {  // (0: a page without headers to walk)
            const uint32_t w = pages[(size_t)i * every].num_bits, in_len = pages[(size_t)i * every].in_len;
            atomicAdd(&headers_of[w], headers);
            atomicAdd(&bytes_of[w], min(walked, 1u << 22));  // (the sums of kDensitySamples pages fit 32 bits)
            atomicMax(&longest_of[w], whole ? walked : in_len);
            if (headers >= kDensitySolid || whole) {
                const uint64_t own = whole ? headers : (uint64_t)headers * in_len / walked;
                atomicMax(&solid, own > 0xffffffffull ? 0xffffffffu : (uint32_t)own);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64u) return;
    const uint32_t lane = threadIdx.x;
    uint64_t est = 0;
    if (lane <= 32u && bytes_of[lane] != 0u) est = (uint64_t)longest_of[lane] * headers_of[lane] / bytes_of[lane];
    if (lane == 33u) est = solid;
    uint32_t h_max = est > 0xffffffffull ? 0xffffffffu : (uint32_t)est;
    h_max = wave_inclusive_max(h_max);
    if (lane == 63u) {
        const float ring_us = model.ring_fixed_us + model.turns * fmaxf((float)h_max * model.per_header_us, model.values_us);
        // (what is known of the pages is a sample: the scout kernels only where they win by half)
        const uint32_t ring = ring_us <= 1.5f * model.scout_us ? 1u : 0u;
        *go = ring;
        host_words[1] = h_max;
        __threadfence_system();
        host_words[0] = (seq << 1) | ring;
        __threadfence_system();
    }
}

}  // namespace pq2

// `seen`: two words per sample of device memory (pq2_density_samples() at most); `go`: where the ring_kernel launch
// behind this one looks; `host_words`: two words of page-locked host memory the device can write: (seq << 1 | verdict)
// and the largest estimated header count of a page.
extern "C" uint32_t pq2_density_samples() { return pq2::kDensitySamples; }
extern "C" hipError_t pq2_launch_density(const void *pages, uint32_t n_pages, void *seen, float scout_us,
                                         float ring_fixed_us, float per_header_us, float values_us, float turns,
                                         uint32_t *go, uint32_t *host_words, uint32_t seq, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    const pq2::DensityModel model{scout_us, ring_fixed_us, per_header_us, values_us, turns};
    const uint32_t every = (n_pages + pq2::kDensitySamples - 1u) / pq2::kDensitySamples;
    const uint32_t n_samples = (n_pages + every - 1u) / every;
    hipLaunchKernelGGL(pq2::density_kernel, dim3((n_samples + 3u) / 4u), dim3(256), 0, stream,
                       (const pq2::PageDev *)pages, n_samples, every, (uint2 *)seen);
    hipLaunchKernelGGL(pq2::density_verdict_kernel, dim3(1), dim3(pq2::kDensitySamples), 0, stream,
                       (const pq2::PageDev *)pages, n_samples, every, (const uint2 *)seen, model, go, host_words, seq);
    return hipGetLastError();
}
