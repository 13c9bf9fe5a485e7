// Latency microbenchmarks for the header chase of parse_kernel (one wave, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run: ./lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048

// A dependent chain of LDS reads: the address of each read comes from the value of the previous one.
// `stride` bytes apart, starting `mis` bytes off an 8-byte boundary.
template <int BYTES>
__global__ void lds_chain(uint64_t *out, uint32_t mis) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192 + 64];
    for (uint32_t i = threadIdx.x; i < 8192 + 64; i += 64) lds[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
    uint32_t addr = mis;
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        if (BYTES == 4)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(addr + base) : "memory");
        else if (BYTES == 8)
            asm volatile("ds_read_b64 v[2:3], %1\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v2" : "=v"(v0) : "v"(addr + base) : "memory", "v2", "v3");
        else if (BYTES == 16)
            asm volatile("ds_read_b128 v[4:7], %1\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v4" : "=v"(v0) : "v"(addr + base) : "memory", "v4", "v5", "v6", "v7");
        else
            asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(addr + base) : "memory");
        addr = ((addr + 24u + v0) & 8191u);
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = addr + v0 + v1 + v2 + v3 + lds[mis];
}

// the same with the value moved to a scalar register and the address computed there (as the chase does)
__global__ void lds_chain_scalar(uint64_t *out, uint32_t mis) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192 + 64];
    for (uint32_t i = threadIdx.x; i < 8192 + 64; i += 64) lds[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
    uint32_t pos = mis;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        uint32_t a, lo;
        asm volatile(
            "v_add_u32 %[a], %[pos], %[base]\n\t"
            "ds_read_b64 v[2:3], %[a]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_readfirstlane_b32 %[lo], v2\n\t"
            "s_add_u32 %[pos], %[pos], %[lo]\n\t"
            "s_add_u32 %[pos], %[pos], 24\n\t"
            "s_and_b32 %[pos], %[pos], 8191"
            : [pos] "+s"(pos), [a] "=&v"(a), [lo] "=&s"(lo)
            : [base] "v"(base)
            : "memory", "v2", "v3", "scc");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = pos + lds[mis];
}

// lds_chain_scalar with only lane 0 enabled while the reads are issued
template <int BYTES>
__global__ void lds_chain_lane0(uint64_t *out, uint32_t mis) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192 + 64];
    for (uint32_t i = threadIdx.x; i < 8192 + 64; i += 64) lds[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
    uint32_t pos = mis;
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1" : "=s"(saved));
    for (int i = 0; i < REP; ++i) {
        uint32_t a, lo;
        if (BYTES == 8)
            asm volatile(
                "v_add_u32 %[a], %[pos], %[base]\n\t"
                "ds_read_b64 v[2:3], %[a]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_readfirstlane_b32 %[lo], v2\n\t"
                "s_add_u32 %[pos], %[pos], %[lo]\n\t"
                "s_add_u32 %[pos], %[pos], 24\n\t"
                "s_and_b32 %[pos], %[pos], 8191"
                : [pos] "+s"(pos), [a] "=&v"(a), [lo] "=&s"(lo)
                : [base] "v"(base)
                : "memory", "v2", "v3", "scc");
        else
            asm volatile(
                "v_add_u32 %[a], %[pos], %[base]\n\t"
                "ds_read_b128 v[4:7], %[a]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_readfirstlane_b32 %[lo], v4\n\t"
                "s_add_u32 %[pos], %[pos], %[lo]\n\t"
                "s_add_u32 %[pos], %[pos], 24\n\t"
                "s_and_b32 %[pos], %[pos], 8191"
                : [pos] "+s"(pos), [a] "=&v"(a), [lo] "=&s"(lo)
                : [base] "v"(base)
                : "memory", "v4", "v5", "v6", "v7", "scc");
    }
    asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = pos + lds[mis];
}

// N dependent scalar adds per iteration
__global__ void salu_chain(uint64_t *out, uint32_t x) {
    uint32_t s = x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        asm volatile(
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3"
            : "+s"(s)::"scc");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = s;
}

// 16 independent scalar adds per iteration (4 registers)
__global__ void salu_indep(uint64_t *out, uint32_t x) {
    uint32_t a = x, b = x + 1, c = x + 2, d = x + 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        asm volatile(
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3"
            : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"scc");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    // the slowest wave counts: the issue arbiter favours the oldest wave
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[0], (unsigned long long)(t1 - t0));
    if (threadIdx.x == 0) out[1] = a + b + c + d;
}

// 16 dependent vector adds per iteration
__global__ void valu_chain(uint64_t *out, uint32_t x) {
    uint32_t v = x + threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        asm volatile(
            "v_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\t"
            "v_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\t"
            "v_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\t"
            "v_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3\n\tv_add_u32 %0, %0, 3"
            : "+v"(v));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = v;
}

// scalar -> v_readlane -> scalar chain, 8 per iteration
__global__ void readlane_chain(uint64_t *out, uint32_t x) {
    uint32_t v = (threadIdx.x * 7u + x) & 63u;
    uint32_t s = x & 63u;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        asm volatile(
            "v_readlane_b32 %0, %1, %0\n\tv_readlane_b32 %0, %1, %0\n\tv_readlane_b32 %0, %1, %0\n\t"
            "v_readlane_b32 %0, %1, %0\n\tv_readlane_b32 %0, %1, %0\n\tv_readlane_b32 %0, %1, %0\n\t"
            "v_readlane_b32 %0, %1, %0\n\tv_readlane_b32 %0, %1, %0"
            : "+s"(s)
            : "v"(v));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = s;
}

// a loop of 4 scalar adds closed by a taken branch, against the same 4 adds x 8 per taken branch
__global__ void branch_cost(uint64_t *out, uint32_t x, int unrolled) {
    uint32_t s = x, n = REP * 8;
    uint64_t t0 = __builtin_readcyclecounter();
    if (!unrolled) {
        asm volatile(
            "1:\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_sub_u32 %1, %1, 1\n\ts_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1b"
            : "+s"(s), "+s"(n)::"scc");
    } else {
        asm volatile(
            "1:\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_sub_u32 %1, %1, 1\n\ts_cmp_lg_u32 %1, 0\n\ts_cbranch_scc0 2f\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\t"
            "s_sub_u32 %1, %1, 1\n\ts_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1b\n\t"
            "2:"
            : "+s"(s), "+s"(n)::"scc");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = s;
}

// integer VALU throughput of one SIMD: `waves_per_simd` waves per SIMD (blockDim = 256 * waves_per_simd), each
// with 4 independent chains of the given instruction
template <int KIND>
__global__ void valu_tput(uint64_t *out, uint32_t x) {
    uint32_t a = x + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        if (KIND == 0)
            asm volatile(
                "v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\t"
                "v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\t"
                "v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\t"
                "v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 1)
            asm volatile(
                "v_alignbit_b32 %0, %1, %0, 3\n\tv_alignbit_b32 %1, %2, %1, 3\n\tv_alignbit_b32 %2, %3, %2, 3\n\tv_alignbit_b32 %3, %0, %3, 3\n\t"
                "v_alignbit_b32 %0, %1, %0, 3\n\tv_alignbit_b32 %1, %2, %1, 3\n\tv_alignbit_b32 %2, %3, %2, 3\n\tv_alignbit_b32 %3, %0, %3, 3\n\t"
                "v_alignbit_b32 %0, %1, %0, 3\n\tv_alignbit_b32 %1, %2, %1, 3\n\tv_alignbit_b32 %2, %3, %2, 3\n\tv_alignbit_b32 %3, %0, %3, 3\n\t"
                "v_alignbit_b32 %0, %1, %0, 3\n\tv_alignbit_b32 %1, %2, %1, 3\n\tv_alignbit_b32 %2, %3, %2, 3\n\tv_alignbit_b32 %3, %0, %3, 3"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else
            asm volatile(
                "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    // the slowest wave counts: the issue arbiter favours the oldest wave
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[0], (unsigned long long)(t1 - t0));
    if (threadIdx.x == 0) out[1] = a + b + c + d;
}

__global__ void clocks(uint64_t *out) {
    uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    uint32_t v = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));
    uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = c1 - c0, out[1] = w1 - w0, out[2] = v;
}

#define RUN(name, per, ...)                                                          \
    do {                                                                             \
        hipLaunchKernelGGL(__VA_ARGS__);                                             \
        hipDeviceSynchronize();                                                      \
        hipMemset(d, 0, 64);                                                         \
        hipLaunchKernelGGL(__VA_ARGS__);                                             \
        hipDeviceSynchronize();                                                      \
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);                                  \
        printf("%-44s %8.2f ticks each\n", name, (double)h[0] / ((double)REP * per)); \
    } while (0)

int main() {
    uint64_t *d, h[4];
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(clocks, dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("cycle counter ticks %llu, wall clock ticks %llu (ratio %.3f); 200000 dependent v_add\n",
           (unsigned long long)h[0], (unsigned long long)h[1], (double)h[0] / (double)h[1]);
    RUN("ds_read_u8 chain", 1, lds_chain<1>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("ds_read_b32 chain aligned", 1, lds_chain<4>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("ds_read_b32 chain +1", 1, lds_chain<4>, dim3(1), dim3(64), 0, 0, d, 1u);
    RUN("ds_read_b64 chain aligned", 1, lds_chain<8>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("ds_read_b64 chain +4", 1, lds_chain<8>, dim3(1), dim3(64), 0, 0, d, 4u);
    RUN("ds_read_b64 chain +3", 1, lds_chain<8>, dim3(1), dim3(64), 0, 0, d, 3u);
    RUN("ds_read_b128 chain aligned", 1, lds_chain<16>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("ds_read_b128 chain +8", 1, lds_chain<16>, dim3(1), dim3(64), 0, 0, d, 8u);
    RUN("ds_read_b128 chain +3", 1, lds_chain<16>, dim3(1), dim3(64), 0, 0, d, 3u);
    RUN("ds_read_b64 +3 -> readfirstlane -> 3 salu", 1, lds_chain_scalar, dim3(1), dim3(64), 0, 0, d, 3u);
    RUN("ds_read_b64 +0 -> readfirstlane -> 3 salu", 1, lds_chain_scalar, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("lane 0 only: ds_read_b64 +3 -> rfl -> 3 salu", 1, lds_chain_lane0<8>, dim3(1), dim3(64), 0, 0, d, 3u);
    RUN("lane 0 only: ds_read_b64 +0 -> rfl -> 3 salu", 1, lds_chain_lane0<8>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("lane 0 only: ds_read_b128 +3 -> rfl -> 3 salu", 1, lds_chain_lane0<16>, dim3(1), dim3(64), 0, 0, d, 3u);
    RUN("lane 0 only: ds_read_b128 +0 -> rfl -> 3 salu", 1, lds_chain_lane0<16>, dim3(1), dim3(64), 0, 0, d, 0u);
    RUN("lane 0 only: ds_read_b64 +3, 1024 waves", 1, lds_chain_lane0<8>, dim3(1024), dim3(64), 0, 0, d, 3u);
    RUN("lane 0 only: ds_read_b128 +3, 1024 waves", 1, lds_chain_lane0<16>, dim3(1024), dim3(64), 0, 0, d, 3u);
    RUN("all lanes: ds_read_b64 +3 scalar, 1024 waves", 1, lds_chain_scalar, dim3(1024), dim3(64), 0, 0, d, 3u);
    RUN("all lanes: ds_read_b64 +0 scalar, 1024 waves", 1, lds_chain_scalar, dim3(1024), dim3(64), 0, 0, d, 0u);
    RUN("s_add_u32 dependent", 16, salu_chain, dim3(1), dim3(64), 0, 0, d, 1u);
    RUN("s_add_u32 independent", 16, salu_indep, dim3(1), dim3(64), 0, 0, d, 1u);
    RUN("v_add_u32 dependent", 16, valu_chain, dim3(1), dim3(64), 0, 0, d, 1u);
    RUN("v_readlane dependent", 8, readlane_chain, dim3(1), dim3(64), 0, 0, d, 1u);
    RUN("7-instruction loop, taken branch each", 8, branch_cost, dim3(1), dim3(64), 0, 0, d, 1u, 0);
    RUN("7-instruction loop, taken branch every 2nd", 8, branch_cost, dim3(1), dim3(64), 0, 0, d, 1u, 1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        char name[96];
        snprintf(name, sizeof name, "v_add_u32, %d waves per SIMD (per wave)", wps);
        RUN(name, 16, valu_tput<0>, dim3(1), dim3(256 * wps), 0, 0, d, 1u);
        snprintf(name, sizeof name, "v_alignbit_b32, %d waves per SIMD (per wave)", wps);
        RUN(name, 16, valu_tput<1>, dim3(1), dim3(256 * wps), 0, 0, d, 1u);
        snprintf(name, sizeof name, "v_fma_f32, %d waves per SIMD (per wave)", wps);
        RUN(name, 16, valu_tput<2>, dim3(1), dim3(256 * wps), 0, 0, d, 1u);
    }
    // the same with every SIMD of the chip busy (1024 waves)
    RUN("ds_read_b64 +3 chain, 1024 waves", 1, lds_chain<8>, dim3(1024), dim3(64), 0, 0, d, 3u);
    RUN("s_add_u32 dependent, 1024 waves", 16, salu_chain, dim3(1024), dim3(64), 0, 0, d, 1u);
    return 0;
}
