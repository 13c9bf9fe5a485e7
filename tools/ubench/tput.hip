// Issue rate of vector instructions on one SIMD of gfx950: 4 waves per SIMD, 4 independent chains each.
// Build: hipcc --offload-arch=gfx950 -O3 -o tput tput.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048
#define I4(op) op "\n\t" op "\n\t" op "\n\t" op "\n\t"
#define KERNEL(name, A, B, C, D)                                                                     \
    __global__ void name(uint64_t *out, uint32_t x) {                                                \
        uint32_t a = x + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;                               \
        uint64_t wide = a;                                                                           \
        __syncthreads();                                                                             \
        uint64_t t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < REP; ++i)                                                                \
            asm volatile(A "\n\t" B "\n\t" C "\n\t" D "\n\t" A "\n\t" B "\n\t" C "\n\t" D "\n\t" A "\n\t" B "\n\t" C \
                           "\n\t" D "\n\t" A "\n\t" B "\n\t" C "\n\t" D                            \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(wide));                          \
        uint64_t t1 = __builtin_readcyclecounter();                                                  \
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[0], (unsigned long long)(t1 - t0)); \
        if (threadIdx.x == 0) out[1] = a + b + c + d + wide;                                         \
    }

KERNEL(k_add, "v_add_u32 %0, %0, 3", "v_add_u32 %1, %1, 3", "v_add_u32 %2, %2, 3", "v_add_u32 %3, %3, 3")
KERNEL(k_and, "v_and_b32 %0, %0, %1", "v_and_b32 %1, %1, %2", "v_and_b32 %2, %2, %3", "v_and_b32 %3, %3, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0", "v_lshrrev_b32 %1, 3, %1", "v_lshrrev_b32 %2, 3, %2", "v_lshrrev_b32 %3, 3, %3")
KERNEL(k_lshrv, "v_lshrrev_b32 %0, %1, %0", "v_lshrrev_b32 %1, %2, %1", "v_lshrrev_b32 %2, %3, %2", "v_lshrrev_b32 %3, %0, %3")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 12", "v_bfe_u32 %1, %1, 3, 12", "v_bfe_u32 %2, %2, 3, 12", "v_bfe_u32 %3, %3, 3, 12")
KERNEL(k_bfev, "v_bfe_u32 %0, %0, %1, 12", "v_bfe_u32 %1, %1, %2, 12", "v_bfe_u32 %2, %2, %3, 12", "v_bfe_u32 %3, %3, %0, 12")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %0, 3", "v_alignbit_b32 %1, %2, %1, 3", "v_alignbit_b32 %2, %3, %2, 3", "v_alignbit_b32 %3, %0, %3, 3")
KERNEL(k_alignbitv, "v_alignbit_b32 %0, %1, %0, %2", "v_alignbit_b32 %1, %2, %1, %3", "v_alignbit_b32 %2, %3, %2, %0", "v_alignbit_b32 %3, %0, %3, %1")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %1, %0, 1", "v_alignbyte_b32 %1, %2, %1, 1", "v_alignbyte_b32 %2, %3, %2, 1", "v_alignbyte_b32 %3, %0, %3, 1")
KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %2", "v_perm_b32 %1, %2, %1, %3", "v_perm_b32 %2, %3, %2, %0", "v_perm_b32 %3, %0, %3, %1")
KERNEL(k_lshr64, "v_lshrrev_b64 %4, 3, %4", "v_lshrrev_b64 %4, 5, %4", "v_lshrrev_b64 %4, 7, %4", "v_lshrrev_b64 %4, 1, %4")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %1", "v_lshl_or_b32 %1, %1, 3, %2", "v_lshl_or_b32 %2, %2, 3, %3", "v_lshl_or_b32 %3, %3, 3, %0")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2", "v_and_or_b32 %1, %1, %2, %3", "v_and_or_b32 %2, %2, %3, %0", "v_and_or_b32 %3, %3, %0, %1")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %1", "v_lshl_add_u32 %1, %1, 3, %2", "v_lshl_add_u32 %2, %2, 3, %3", "v_lshl_add_u32 %3, %3, 3, %0")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", "v_add3_u32 %1, %1, %2, %3", "v_add3_u32 %2, %2, %3, %0", "v_add3_u32 %3, %3, %0, %1")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1", "v_mul_u32_u24 %1, %1, %2", "v_mul_u32_u24 %2, %2, %3", "v_mul_u32_u24 %3, %3, %0")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "v_mad_u32_u24 %1, %1, %2, %3", "v_mad_u32_u24 %2, %2, %3, %0", "v_mad_u32_u24 %3, %3, %0, %1")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1", "v_mul_lo_u32 %1, %1, %2", "v_mul_lo_u32 %2, %2, %3", "v_mul_lo_u32 %3, %3, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32 %1, %1, %2, vcc", "v_cndmask_b32 %2, %2, %3, vcc", "v_cndmask_b32 %3, %3, %0, vcc")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1", "v_cmp_lt_u32 vcc, %1, %2", "v_cmp_lt_u32 vcc, %2, %3", "v_cmp_lt_u32 vcc, %3, %0")
KERNEL(k_min, "v_min_u32 %0, %0, %1", "v_min_u32 %1, %1, %2", "v_min_u32 %2, %2, %3", "v_min_u32 %3, %3, %0")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_and_b32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")

// scalar unit: `blockDim / 64` waves on one CU, each with 4 independent chains of s_add_u32 (or s_and / s_cmp+cselect)
__global__ void k_salu(uint64_t *out, uint32_t x) {
    uint32_t a = x, b = x + 1, c = x + 2, d = x + 3;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i)
        asm volatile(
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3"
            : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"scc");
    uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[0], (unsigned long long)(t1 - t0));
    if (threadIdx.x == 0) out[1] = a + b + c + d;
}
// half scalar, half vector
__global__ void k_mix(uint64_t *out, uint32_t x) {
    uint32_t a = x, b = x + 1, c = x + threadIdx.x, d = x + 3 + threadIdx.x;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i)
        asm volatile(
            "s_add_u32 %0, %0, 3\n\tv_add_u32 %2, %2, 3\n\ts_add_u32 %1, %1, 3\n\tv_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\tv_add_u32 %2, %2, 3\n\ts_add_u32 %1, %1, 3\n\tv_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\tv_add_u32 %2, %2, 3\n\ts_add_u32 %1, %1, 3\n\tv_add_u32 %3, %3, 3\n\t"
            "s_add_u32 %0, %0, 3\n\tv_add_u32 %2, %2, 3\n\ts_add_u32 %1, %1, 3\n\tv_add_u32 %3, %3, 3"
            : "+s"(a), "+s"(b), "+v"(c), "+v"(d)::"scc");
    uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[0], (unsigned long long)(t1 - t0));
    if (threadIdx.x == 0) out[1] = a + b + c + d;
}

#define RUN(k)                                                                                        \
    do {                                                                                              \
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 1u);                                      \
        hipDeviceSynchronize();                                                                       \
        hipMemset(d, 0, 64);                                                                          \
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 1u);                                      \
        hipDeviceSynchronize();                                                                       \
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);                                                   \
        printf("%-14s %6.2f cycles per wave instruction on one SIMD\n", #k, (double)h[0] / (REP * 16.0 * 4.0)); \
    } while (0)

int main() {
    uint64_t *d, h[4];
    hipMalloc(&d, 64);
    RUN(k_add); RUN(k_and); RUN(k_lshr); RUN(k_lshrv); RUN(k_bfe); RUN(k_bfev); RUN(k_alignbit); RUN(k_alignbitv);
    RUN(k_alignbyte); RUN(k_perm); RUN(k_lshr64); RUN(k_lshlor); RUN(k_andor); RUN(k_lshladd); RUN(k_add3);
    RUN(k_mul24); RUN(k_mad24); RUN(k_mullo); RUN(k_cndmask); RUN(k_cmp); RUN(k_min); RUN(k_mov_dpp); RUN(k_sdwa);
    for (int waves = 4; waves <= 16; waves *= 2) {
        for (int which = 0; which < 2; ++which) {
            hipMemset(d, 0, 64);
            if (which == 0) hipLaunchKernelGGL(k_salu, dim3(1), dim3(64 * waves), 0, 0, d, 1u);
            else hipLaunchKernelGGL(k_mix, dim3(1), dim3(64 * waves), 0, 0, d, 1u);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("%s, %2d waves on one CU: %6.2f cycles per instruction per wave, %5.2f instructions per cycle on the CU\n",
                   which == 0 ? "s_add_u32        " : "s_add + v_add mix", waves, (double)h[0] / (REP * 16.0),
                   waves * REP * 16.0 / (double)h[0]);
        }
    }
    return 0;
}
