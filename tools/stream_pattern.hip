// What the memory system gives for the access pattern of the page-per-workgroup kernels, without any decoding:
// every workgroup reads 6 KiB and writes 16 KiB per step (the 3 : 8 of bit width 12 expanded to u32), either
//   pattern 0: steps dealt round robin over the whole grid (all workgroups write inside one moving window), or
//   pattern 1: every workgroup walks through regions of its own (4 MiB written, 1.5 MiB read: a page of 2^20 values),
//              as a kernel that holds one page per workgroup has to;
//   pattern 2: as 1, each workgroup beginning at another offset of its region (regions that are a power of two
//              apart, walked in step, meet in the same memory channels).
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_pattern.hip -o gpurun_out/stream_pattern
// Run:   gpurun_out/stream_pattern [regions = 2048] [workgroups = 1024] [threads = 256 | 128] [more: patterns 3 and 4]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));             \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kOutStep = 16384, kInStep = 6144;        // bytes per workgroup step
constexpr uint32_t kRegionSteps = 256;                      // steps per region: 4 MiB written
constexpr uint64_t kRegionOut = (uint64_t)kOutStep * kRegionSteps, kRegionIn = (uint64_t)kInStep * kRegionSteps;

// kPairs: a thread writes two pieces that lie side by side (32 bytes per lane in two stores) instead of pieces a whole
// workgroup apart (what 8 values per lane and step would mean for a decoder that stores 16 bytes at a time)
template <int kPattern, int kThreads, bool kNt = false, bool kPairs = false>
__global__ __launch_bounds__(kThreads) void stream_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          uint32_t n_regions) {
    const uint32_t tid = threadIdx.x;
    const uint64_t total_steps = (uint64_t)n_regions * kRegionSteps;
    constexpr uint32_t kPieces = kOutStep / 16u / kThreads;  // 16-byte stores per thread and step
    auto step_of = [&](uint64_t it) -> uint64_t {
        if (kPattern == 0) return it * gridDim.x + blockIdx.x;
        const uint64_t region = (it / kRegionSteps) * gridDim.x + blockIdx.x;
        // pattern 2: the workgroups do not all stand at the same offset of their regions
        const uint64_t turn = kPattern == 2 ? (uint64_t)blockIdx.x * 37u : 0u;
        return region * kRegionSteps + (it + turn) % kRegionSteps;
    };
    // 6 KiB: 384 pieces of 16 bytes
    auto load = [&](uint64_t step) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (step >= total_steps) return v;
        const uint8_t *const src = in + step * kInStep;
#pragma unroll
        for (uint32_t k = 0; k * kThreads < 384u; ++k)
            if (tid + k * kThreads < 384u) v ^= *(const u32x4 *)(src + 16u * (tid + k * kThreads));
        return v;
    };
    // (the loads of the next step are asked for before the stores of the one in hand)
    u32x4 v = load(step_of(0));
    for (uint64_t it = 0;; ++it) {
        const uint64_t step = step_of(it);
        if (step >= total_steps) break;
        const u32x4 next = load(step_of(it + 1));
        uint8_t *const dst = out + step * kOutStep;
#pragma unroll
        for (uint32_t k = 0; k < kPieces; ++k) {
            const u32x4 o = {v.x + k, v.y, v.z, v.w};
            const uint32_t piece = kPairs ? 2u * tid + (k & 1u) + 2u * kThreads * (k >> 1) : tid + kThreads * k;
            if (kNt)
                __builtin_nontemporal_store(o, (u32x4 *)(dst + 16u * piece));
            else
                *(u32x4 *)(dst + 16u * piece) = o;
        }
        v = next;
    }
}

template <int kPattern, int kThreads, bool kNt = false, bool kPairs = false>
static double run(const uint8_t *in, uint8_t *out, uint32_t regions, uint32_t grid, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_kernel<kPattern, kThreads, kNt, kPairs>), dim3(grid), dim3(kThreads), 0, 0, in, out, regions);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((stream_kernel<kPattern, kThreads, kNt, kPairs>), dim3(grid), dim3(kThreads), 0, 0, in, out, regions);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// ---- what tools/ubench/membw.hip measures (5.5-5.7 TB/s for 1 read : 4 written), in this file's terms ------
// pattern 3: no workgroup stays: a grid of total_steps / per workgroups, each doing `per` consecutive steps of kStep
//            bytes written (and 3/8 of that read) and ending -- the hardware hands the pieces out in order, so all that
//            is written at one time lies in one window that moves through memory once.
// pattern 4: a region per workgroup as pattern 1, with the step size as a parameter (4 ... 64 KiB written per step) and
//            the loads marked non-temporal or not.
template <int kThreads, uint32_t kStep, bool kNtLoad>
__device__ __forceinline__ void one_step(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t step) {
    constexpr uint32_t kIn = kStep * 3u / 8u;
    const uint32_t tid = threadIdx.x;
    const uint8_t *const src = in + step * kIn;
    uint8_t *const dst = out + step * kStep;
    u32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k * kThreads * 16u < kIn; ++k)
        if ((tid + k * kThreads) * 16u < kIn) {
            const u32x4 *const p = (const u32x4 *)(src + 16u * (tid + k * kThreads));
            v ^= kNtLoad ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
    for (uint32_t k = 0; k * kThreads * 16u < kStep; ++k) {
        const u32x4 o = {v.x + k, v.y, v.z, v.w};
        __builtin_nontemporal_store(o, (u32x4 *)(dst + 16u * (tid + kThreads * k)));
    }
}
template <int kThreads, uint32_t kStep, bool kNtLoad>
__global__ __launch_bounds__(kThreads) void once_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t per) {
    for (uint32_t i = 0; i < per; ++i) one_step<kThreads, kStep, kNtLoad>(in, out, (uint64_t)blockIdx.x * per + i);
}
template <int kThreads, uint32_t kStep, bool kNtLoad>
__global__ __launch_bounds__(kThreads) void region_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          uint32_t n_regions, uint32_t region_steps) {
    for (uint32_t region = blockIdx.x; region < n_regions; region += gridDim.x)
        for (uint32_t i = 0; i < region_steps; ++i)
            one_step<kThreads, kStep, kNtLoad>(in, out, (uint64_t)region * region_steps + i);
}
template <typename Launch>
static double timed(Launch launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
template <int kThreads, uint32_t kStep, bool kNtLoad>
static void more_patterns(const uint8_t *in, uint8_t *out, uint32_t regions, uint32_t grid) {
    const uint64_t out_bytes = (uint64_t)regions * kRegionOut;
    const uint64_t steps = out_bytes / kStep;
    const double bytes = (double)out_bytes * 11.0 / 8.0;
    for (uint32_t per : {1u, 4u, 16u, 64u}) {
        if (steps % per) continue;
        const double ms = timed([&] { hipLaunchKernelGGL((once_kernel<kThreads, kStep, kNtLoad>), dim3((uint32_t)(steps / per)), dim3(kThreads), 0, 0, in, out, per); }, 5);
        printf("pattern 3 (one pass, %u steps of %u KiB per workgroup of %d%s): %.3f ms, %.2f TB/s (%.3f of 8 TB/s)\n", per, kStep >> 10,
               kThreads, kNtLoad ? ", nt loads" : "", ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 8.0);
    }
    const uint32_t region_steps = (uint32_t)(kRegionOut / kStep);
    const double ms = timed([&] { hipLaunchKernelGGL((region_kernel<kThreads, kStep, kNtLoad>), dim3(grid), dim3(kThreads), 0, 0, in, out, regions, region_steps); }, 5);
    printf("pattern 4 (a region per workgroup, steps of %u KiB, %u workgroups of %d%s): %.3f ms, %.2f TB/s (%.3f of 8 TB/s)\n", kStep >> 10, grid,
           kThreads, kNtLoad ? ", nt loads" : "", ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 8.0);
}

int main(int argc, char **argv) {
    const uint32_t regions = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048u;
    const uint32_t grid = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024u;
    const int threads = argc > 3 ? atoi(argv[3]) : 256;
    uint8_t *in, *out;
    CHECK(hipMalloc(&in, regions * kRegionIn));
    CHECK(hipMalloc(&out, regions * kRegionOut));
    CHECK(hipMemset(in, 1, regions * kRegionIn));
    CHECK(hipMemset(out, 0, regions * kRegionOut));
    const double bytes = (double)regions * (kRegionIn + kRegionOut);
    for (int pattern = 0; pattern < 3; ++pattern) {
        double ms;
        if (threads == 256)
            ms = pattern == 0   ? run<0, 256>(in, out, regions, grid, 10)
                 : pattern == 1 ? run<1, 256>(in, out, regions, grid, 10)
                                : run<2, 256>(in, out, regions, grid, 10);
        else
            ms = pattern == 0   ? run<0, 128>(in, out, regions, grid, 10)
                 : pattern == 1 ? run<1, 128>(in, out, regions, grid, 10)
                                : run<2, 128>(in, out, regions, grid, 10);
        printf("pattern %d (%s): %u regions, %u workgroups of %d: %.3f ms, %.2f TB/s read + written (%.3f of 8 TB/s)\n", pattern,
               pattern == 0 ? "round robin" : pattern == 1 ? "a region per workgroup" : "a region per workgroup, out of step",
               regions, grid, threads, ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 8.0);
    }
    if (threads == 256) {  // the same with stores marked as not to be read again soon
        const double a = run<0, 256, true>(in, out, regions, grid, 10), b = run<1, 256, true>(in, out, regions, grid, 10);
        printf("nontemporal stores: round robin %.3f ms (%.3f of 8 TB/s), a region per workgroup %.3f ms (%.3f)\n", a,
               bytes / a * 1e-9 / 8.0, b, bytes / b * 1e-9 / 8.0);
    }
    if (threads == 256) {
        const double b = run<1, 256, false, true>(in, out, regions, grid, 10);
        printf("two pieces side by side per thread: a region per workgroup %.3f ms (%.3f of 8 TB/s)\n", b, bytes / b * 1e-9 / 8.0);
    }
    // what the two directions give alone: the same grid reading / writing 8 GiB
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const double gib8 = (double)regions * kRegionOut;
        for (int dir = 0; dir < 2; ++dir) {
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < 5; ++r) {
                if (dir == 0)
                    CHECK(hipMemsetAsync(out, r, regions * kRegionOut, 0));
                else
                    CHECK(hipMemcpyAsync(out, out + regions * kRegionOut / 2, regions * kRegionOut / 2, hipMemcpyDeviceToDevice, 0));
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            printf("%s of %.1f GB: %.3f ms, %.2f TB/s%s\n", dir == 0 ? "hipMemsetAsync" : "hipMemcpyAsync (half onto half)",
                   gib8 * 1e-9, ms, gib8 / ms * 1e-9, dir == 0 ? " written" : " read + written");
        }
    }
    if (argc > 4) {  // the further patterns
        more_patterns<256, 4096, false>(in, out, regions, grid);
        more_patterns<256, 16384, false>(in, out, regions, grid);
        more_patterns<256, 16384, true>(in, out, regions, grid);
        more_patterns<256, 65536, false>(in, out, regions, grid);
        more_patterns<128, 4096, false>(in, out, regions, grid);
        more_patterns<128, 16384, true>(in, out, regions, grid);
        more_patterns<128, 65536, true>(in, out, regions, grid);
    }
    CHECK(hipFree(in));
    CHECK(hipFree(out));
    return 0;
}
