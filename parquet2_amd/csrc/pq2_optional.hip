// optional_bits_kernel: validity bitmaps + decoded dictionary indices -> nullable values, for flat nullable columns
// (maximum definition level 1).
//
// What it computes is what optional_kernel (pq2_kernels.hip) computes from u32 levels: row r is valid where its
// level is 1, and the k-th valid row takes the dictionary value of the page's k-th index (deserialize_levels,
// tests/it/read/utils.rs:52-66, over an OptionalDictionary page, tests/it/read/primitive.rs:104-110).  The levels come
// as the bitmaps ring_kernel writes for one-bit level streams (row r at bit r % 32 of word r / 32, bits behind the
// page's rows 0), one bit per row to read instead of four bytes that had to be written first.
//
// One workgroup per page walks the page's rows 4096 at a time: quad q (of four) of thread t are the rows
// q * 1024 + 4 t ... + 3 of the step.  The valid rows of a step are counted (a prefix sum over the lanes of a wave, the
// wave totals through LDS, one barrier per step), which says where in the index stream the step begins; the indices a
// step consumes are the at most 4096 from there on, so they are read with whole 16-byte loads as soon as the step
// before has counted its rows, and handed round through LDS (index p of the step at slot (p % 4) * 1024 + p / 4:
// both the threads that bring the indices in and the rows that take them out walk consecutive slots).  The
// dictionary lies in LDS as well where it fits: 64 lanes looking up 64 different values are 64 cache lines for the
// vector cache, but spread over the LDS banks.
#include "pq2_device.h"

namespace pq2 {

constexpr uint32_t kOptQuads = 4;                       // quads of rows per thread and step
constexpr uint32_t kOptStepRows = 1024u * kOptQuads;    // rows of a page per step
constexpr uint32_t kOptStageBytes = 2u * kOptStepRows * 4u;  // two steps' indices
constexpr uint32_t kOptDictLdsBytes = 32u << 10;        // the largest dictionary kept in LDS

template <int kWords, bool kDictLds>
__global__ __launch_bounds__(256) void optional_bits_kernel(const OptionalPage *__restrict__ pages, uint32_t n_pages,
                                                            int32_t *__restrict__ status,
                                                            const int32_t *__restrict__ level_status) {
    extern __shared__ __align__(16) uint32_t opt_lds[];
    __shared__ uint32_t wave_total[2][kOptQuads][4];
    uint32_t *const stage = opt_lds;                       // [2][kOptStepRows]
    uint32_t *const dict_lds = opt_lds + 2u * kOptStepRows;
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const OptionalPage pg = pages[page];
    typedef __attribute__((address_space(1))) const uint32_t gu32;
    out_u32 *const out = (out_u32 *)pg.out_addr;
    __attribute__((address_space(1))) uint8_t *const validity =
        (__attribute__((address_space(1))) uint8_t *)pg.validity_addr;
    // Reads go through buffer descriptors, which answer 0 behind their ends: no read is under a branch.
    //   bitmap: the words of the page's rows (rows behind the page: not valid);
    //   indices: a 16-byte read that begins at an index of the page may reach 12 bytes behind the page's last (the
    //   host leaves as much room behind the indices of every page), one that begins further on is never looked at;
    //   dictionary: an index that is not below dict_len reads 0 (and is reported).
    const __amdgpu_buffer_rsrc_t bits_rsrc = make_rsrc(pg.level_addr, ((pg.rows + 31u) >> 5) * 4u);  // (rows < 2^31)
    const __amdgpu_buffer_rsrc_t idx_rsrc = make_rsrc(pg.idx_addr, (uint32_t)min((uint64_t)pg.idx_values * 4u + 12u, (uint64_t)0xfffffffcu));
    const uint64_t dict_words = (uint64_t)pg.dict_len * kWords;
    const __amdgpu_buffer_rsrc_t dict_rsrc = make_rsrc(pg.dict_addr, (uint32_t)min(dict_words * 4u, (uint64_t)0xfffffffcu));
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if constexpr (kDictLds) {  // (read after the first step's barrier)
        if (pg.rows != 0u)
            for (uint32_t i = threadIdx.x; i < (uint32_t)dict_words; i += 256u) dict_lds[i] = ((gu32 *)pg.dict_addr)[i];
    }
    // the step's indices from `from` on: quad q of this thread brings in those at from + q * 1024 + 4 t ... + 3
    auto load_idx = [&](const uint32_t from, u32x4 (&x)[kOptQuads]) {
#pragma unroll
        for (uint32_t q = 0; q < kOptQuads; ++q) {
            const uint64_t pos = (uint64_t)from + q * 1024u + 4u * threadIdx.x;
            x[q] = __builtin_amdgcn_raw_buffer_load_b128(idx_rsrc, (uint32_t)min(pos * 4u, (uint64_t)0xfffffffcu), 0, 0);
        }
    };
    // the validity bits of rows r0 .. r0 + 3 (r0 a multiple of 4: they lie in one word)
    auto load_bits = [&](const uint32_t r0) -> uint32_t {
        return (__builtin_amdgcn_raw_buffer_load_b32(bits_rsrc, (r0 >> 5) * 4u, 0, 0) >> (r0 & 31u)) & 15u;
    };
    // whole quads of rows are written with one store each where the page's values and validity bytes are aligned
    const bool aligned = ((pg.out_addr & 15u) | (pg.validity_addr & 3u)) == 0u;
    uint32_t used = 0;  // valid rows before this step = indices consumed
    bool bad = false;
    uint32_t flip = 0;
    uint32_t nib_next[kOptQuads];
    u32x4 x[kOptQuads];
#pragma unroll
    for (uint32_t q = 0; q < kOptQuads; ++q) nib_next[q] = load_bits(q * 1024u + 4u * threadIdx.x);
    load_idx(0u, x);
    for (uint32_t base = 0; base < pg.rows; base += kOptStepRows, flip ^= 1u) {  // (rows < 2^31: nothing wraps)
        const bool more = base + kOptStepRows < pg.rows;
        uint32_t *const st = stage + flip * kOptStepRows;
        uint32_t r0[kOptQuads], nib[kOptQuads], c[kOptQuads], incl[kOptQuads];
#pragma unroll
        for (uint32_t q = 0; q < kOptQuads; ++q) {
            st[0u * 1024u + q * 256u + threadIdx.x] = x[q].x;
            st[1u * 1024u + q * 256u + threadIdx.x] = x[q].y;
            st[2u * 1024u + q * 256u + threadIdx.x] = x[q].z;
            st[3u * 1024u + q * 256u + threadIdx.x] = x[q].w;
            r0[q] = base + q * 1024u + 4u * threadIdx.x;
            nib[q] = nib_next[q];
            c[q] = __builtin_popcount(nib[q]);
            incl[q] = wave_inclusive_sum(c[q]);
            if (lane == 63u) wave_total[flip][q][wave] = incl[q];
        }
        if (more) {
#pragma unroll
            for (uint32_t q = 0; q < kOptQuads; ++q) nib_next[q] = load_bits(r0[q] + kOptStepRows);
        }
        __syncthreads();  // (the other buffers are written in the next step, these again only after that step's barrier)
        uint32_t rank[kOptQuads];  // of the quad's first row among the step's valid rows
        uint32_t step_valid = 0;
#pragma unroll
        for (uint32_t q = 0; q < kOptQuads; ++q) {
            uint32_t below = 0, all = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t t = wave_total[flip][q][k];
                below += k < wave ? t : 0u;
                all += t;
            }
            rank[q] = step_valid + below + incl[q] - c[q];
            step_valid += all;
        }
        // indices left for this step's valid rows: behind the last index values.next() is None
        const uint32_t avail = pg.idx_values > used ? pg.idx_values - used : 0u;
        used += step_valid;
        if (more) load_idx(used, x);  // on their way while this step works
        uint32_t pres[kOptQuads];  // bit k: row k of the quad has a value
        uint32_t j[kOptQuads][4];
#pragma unroll
        for (uint32_t q = 0; q < kOptQuads; ++q) {
            uint32_t rl = rank[q];
            pres[q] = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t v = (nib[q] >> k) & 1u;
                pres[q] |= (v != 0u && rl < avail ? 1u : 0u) << k;
                j[q][k] = st[((rl & 3u) << 10) + ((rl >> 2) & 1023u)];
                rl += v;
            }
        }
        uint32_t val[kOptQuads][4][kWords];
#pragma unroll
        for (uint32_t q = 0; q < kOptQuads; ++q)
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const bool present = ((pres[q] >> k) & 1u) != 0u;
                const bool ok = j[q][k] < pg.dict_len;
                bad |= present && !ok;
                const uint32_t at = present && ok ? j[q][k] * (uint32_t)kWords : 0u;
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    const uint32_t got = kDictLds ? dict_lds[at + w]
                                                  : __builtin_amdgcn_raw_buffer_load_b32(dict_rsrc, (at + w) * 4u, 0, 0);
                    val[q][k][w] = present && ok ? got : 0u;
                }
            }
        if (aligned && base + kOptStepRows <= pg.rows) {
#pragma unroll
            for (uint32_t q = 0; q < kOptQuads; ++q) {
                uint32_t flat[4 * kWords];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
                    for (int w = 0; w < kWords; ++w) flat[k * kWords + w] = val[q][k][w];
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    const u32x4 o = {flat[4 * w], flat[4 * w + 1], flat[4 * w + 2], flat[4 * w + 3]};
                    *((out_u32x4 *)out + (size_t)(r0[q] >> 2) * kWords + w) = o;
                }
                // bits 0..3 -> bytes 0..3
                *(__attribute__((address_space(1))) uint32_t *)(validity + r0[q]) = (pres[q] * 0x00204081u) & 0x01010101u;
            }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < kOptQuads; ++q)
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t r = r0[q] + k;
                    if (r < pg.rows) {
#pragma unroll
                        for (int w = 0; w < kWords; ++w) out[(size_t)r * kWords + w] = val[q][k][w];
                        validity[r] = (uint8_t)((pres[q] >> k) & 1u);
                    }
                }
        }
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) {  // (`status` holds the index stream's)
        const int32_t of_levels = level_status[page];
        if (of_levels != 0)
            status[page] = of_levels;
        else if (status[page] == 0 && any_bad)
            status[page] = 1;  // PQ2_OUT_OF_SPEC
    }
}

// ---------------------------------------------------------------------------
// rowdir_kernel: the row directories of flat nullable pages (RingNullable, pq2_device.h): for every page, the row of
// its valid row number kRowDirStep * m for m = 1, 2, ... -- what an expanding wave of ring_kernel<., ., true> needs to
// know where the rows of its span of indices begin and end.  One workgroup per page walks the page's validity bitmap
// 1024 words at a time (four words per thread), carrying the count of valid rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdir_kernel(const RingNullable *__restrict__ nulls, uint32_t n_pages) {
    __shared__ uint32_t wave_total[2][4];
    const uint32_t page = blockIdx.x;
    if (page >= n_pages) return;
    const RingNullable nl = nulls[page];
    uint32_t *const dir = (uint32_t *)nl.rowdir_addr;
    const uint32_t n_spans = (nl.idx_values + kRowDirStep - 1u) / kRowDirStep;  // entries 0 .. n_spans are written
    const uint32_t n_words = (nl.rows + 31u) >> 5;
    const __amdgpu_buffer_rsrc_t bits = make_rsrc(nl.bitmap_addr, n_words * 4u);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t carry = 0, flip = 0;
    for (uint32_t w0 = 0; w0 < n_words; w0 += 1024u, flip ^= 1u) {
        const uint32_t wi = w0 + 4u * threadIdx.x;
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(bits, wi * 4u, 0, 0);  // (behind the bitmap: 0)
        const uint32_t word[4] = {x.x, x.y, x.z, x.w};
        uint32_t pc[4], s = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) pc[i] = (uint32_t)__builtin_popcount(word[i]), s += pc[i];
        const uint32_t incl = wave_inclusive_sum(s);
        if (lane == 63u) wave_total[flip][wave] = incl;
        __syncthreads();  // (the other buffer is written in the next step, this one again only after that step's barrier)
        uint32_t before = carry, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t t = wave_total[flip][k];
            before += k < wave ? t : 0u;
            all += t;
        }
        uint32_t c = before + incl - s;  // valid rows before this thread's first word
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            // the step of the directory that lies in this word, if one does: the smallest m with kRowDirStep * m >= c
            const uint32_t m = (c + kRowDirStep - 1u) / kRowDirStep;
            if (pc[i] != 0u && m >= 1u && m <= n_spans && m * kRowDirStep < c + pc[i]) {
                uint32_t left = word[i];
                for (uint32_t t = m * kRowDirStep - c; t != 0u; --t) left &= left - 1u;  // (fewer than 32 times)
                dir[m] = (wi + i) * 32u + (uint32_t)__builtin_ctz(left);
            }
            c += pc[i];
        }
        carry += all;
    }
    // the steps the page's valid rows do not reach
    for (uint32_t m = (carry + kRowDirStep - 1u) / kRowDirStep + threadIdx.x; m <= n_spans; m += 256u)
        if (m >= 1u) dir[m] = nl.rows;
    if (threadIdx.x == 0) dir[0] = 0u;
}

// status[i] = the level stream's where it has one, else what ring_kernel left of the index stream and the lookup
__global__ __launch_bounds__(256) void nullable_status_kernel(int32_t *__restrict__ status, const int32_t *__restrict__ level_status,
                                                              uint32_t n_pages) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_pages && level_status[i] != 0) status[i] = level_status[i];
}

template <int kWords>
static void launch_optional_bits(const OptionalPage *p, uint32_t n_pages, int32_t *status, const int32_t *level_status,
                                 uint32_t dict_bytes, hipStream_t stream) {
    if (dict_bytes <= kOptDictLdsBytes)
        hipLaunchKernelGGL((optional_bits_kernel<kWords, true>), dim3(n_pages), dim3(256), kOptStageBytes + ((dict_bytes + 15u) & ~15u),
                           stream, p, n_pages, status, level_status);
    else
        hipLaunchKernelGGL((optional_bits_kernel<kWords, false>), dim3(n_pages), dim3(256), kOptStageBytes, stream, p, n_pages,
                           status, level_status);
}

}  // namespace pq2

// One workgroup per page.  `status` holds the statuses of the index streams, page i's at i, and receives the pages';
// `level_status`: those of the level streams.  `dict_bytes`: the largest dictionary of the batch (dict_len values of
// value_words dwords), which decides whether the workgroups keep their dictionaries in LDS.
extern "C" hipError_t pq2_launch_optional_bits(const void *pages, uint32_t n_pages, uint32_t value_words, int32_t *status,
                                               const int32_t *level_status, uint64_t dict_bytes, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    const pq2::OptionalPage *p = (const pq2::OptionalPage *)pages;
    const uint32_t db = dict_bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)dict_bytes;
    if (value_words == 1)
        pq2::launch_optional_bits<1>(p, n_pages, status, level_status, db, stream);
    else if (value_words == 2)
        pq2::launch_optional_bits<2>(p, n_pages, status, level_status, db, stream);
    else
        pq2::launch_optional_bits<3>(p, n_pages, status, level_status, db, stream);
    return hipGetLastError();
}

// `nulls`: one pq2::RingNullable per page (device), the pages' bitmaps complete on `stream`.
extern "C" hipError_t pq2_launch_rowdir(const void *nulls, uint32_t n_pages, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::rowdir_kernel, dim3(n_pages), dim3(256), 0, stream, (const pq2::RingNullable *)nulls, n_pages);
    return hipGetLastError();
}
extern "C" uint32_t pq2_rowdir_step() { return pq2::kRowDirStep; }
extern "C" hipError_t pq2_launch_nullable_status(int32_t *status, const int32_t *level_status, uint32_t n_pages, hipStream_t stream) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(pq2::nullable_status_kernel, dim3((n_pages + 255u) / 256u), dim3(256), 0, stream, status, level_status, n_pages);
    return hipGetLastError();
}
