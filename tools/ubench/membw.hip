// Practical HBM rates of gfx950 for streaming kernels: read only, write only, and a read:write mix like the
// decoder's (1 byte read per ~3.3 bytes written).  dwordx4 per lane, grid-stride-free (one pass).
// Build: hipcc --offload-arch=gfx950 -O3 -o membw membw.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each workgroup of 256 threads handles `per` consecutive 4 KiB pieces
__global__ __launch_bounds__(256) void k_write(u32x4 *out, uint32_t per, uint32_t v) {
    u32x4 x = {v, v + 1, v + 2, v + 3};
    size_t base = (size_t)blockIdx.x * per * 256 + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) out[base + (size_t)i * 256] = x;
}
__global__ __launch_bounds__(256) void k_read(const u32x4 *in, uint32_t per, uint32_t *sink) {
    u32x4 acc = {0, 0, 0, 0};
    size_t base = (size_t)blockIdx.x * per * 256 + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) acc += in[base + (size_t)i * 256];
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) *sink = 1;
}
// reads one piece per `ratio` pieces written
__global__ __launch_bounds__(256) void k_mix(const u32x4 *in, u32x4 *out, uint32_t per, uint32_t ratio) {
    size_t base = (size_t)blockIdx.x * per * 256 + threadIdx.x;
    size_t rbase = (size_t)blockIdx.x * (per / ratio) * 256 + threadIdx.x;
    for (uint32_t i = 0; i < per / ratio; ++i) {
        u32x4 x = in[rbase + (size_t)i * 256];
        for (uint32_t k = 0; k < ratio; ++k) out[base + (size_t)(i * ratio + k) * 256] = x + k;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t bytes = 4ull << 30;
    void *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t per : {4u, 16u, 64u}) {
        const uint32_t grid = (uint32_t)(bytes / 4096 / per);
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) k_write<<<grid, 256>>>((u32x4 *)b, per, i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("per %2u write 4 GiB: %.3f ms %.0f GB/s\n", per, ms / 5, bytes / (ms / 5) / 1e6);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) k_read<<<grid, 256>>>((const u32x4 *)a, per, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("per %2u read  4 GiB: %.3f ms %.0f GB/s\n", per, ms / 5, bytes / (ms / 5) / 1e6);
        for (uint32_t ratio : {1u, 2u, 4u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) k_mix<<<grid, 256>>>((const u32x4 *)a, (u32x4 *)b, per, ratio); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double total = (double)bytes + (double)bytes / ratio;
            printf("per %2u mix 1:%u (4 GiB written, %.1f GiB read): %.3f ms %.0f GB/s total\n", per, ratio, 4.0 / ratio, ms / 5, total / (ms / 5) / 1e6);
        }
    }
    return 0;
}
