/*
 * CPU oracle: plain-C restatement of parquet2's hybrid-RLE / bit-unpack path.
 * TEST INFRASTRUCTURE ONLY -- see pq2_oracle.h.  Every function cites the
 * reference file:line it follows (paths relative to the reference root).
 */
#include "pq2_oracle.h"

#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* src/encoding/mod.rs                                                 */
/* ------------------------------------------------------------------ */

/* src/encoding/mod.rs:23-27 */
size_t pq2o_ceil8(size_t value) { return value / 8 + ((value % 8 != 0) ? 1 : 0); }

/* src/encoding/mod.rs:14-21 -- returns None (here -1) when fewer than 4 bytes */
int pq2o_get_length(const uint8_t *values, size_t len, size_t *out) {
    if (len < 4) return -1;
    *out = (size_t)values[0] | ((size_t)values[1] << 8) | ((size_t)values[2] << 16) | ((size_t)values[3] << 24);
    return 0;
}

/* src/read/levels.rs:3-5: 16 - (max_level as i16).leading_zeros() */
uint32_t pq2o_get_bit_width(int16_t max_level) {
    uint16_t v = (uint16_t)max_level;
    uint32_t lz = 0;
    for (int bit = 15; bit >= 0; --bit) {
        if (v & (1u << bit)) break;
        ++lz;
    }
    return 16 - lz;
}

/* ------------------------------------------------------------------ */
/* src/encoding/uleb128.rs                                             */
/* ------------------------------------------------------------------ */

/* uleb128.rs:3-23.  Never errs; stops at the first byte without the
 * continuation bit or at the end of the input; panics (:10-12) when the byte
 * at shift 63 is larger than 1. */
int32_t pq2o_uleb128_decode(const uint8_t *values, size_t len, uint64_t *value, size_t *consumed) {
    uint64_t result = 0;
    unsigned shift = 0;
    size_t used = 0;
    for (size_t i = 0; i < len; ++i) {
        uint8_t byte = values[i];
        used += 1;
        if (shift == 63 && byte > 1) return PQ2O_REFERENCE_WOULD_PANIC;
        result |= (uint64_t)(byte & 0x7f) << shift;
        if ((byte & 0x80) == 0) break;
        shift += 7;
    }
    *value = result;
    *consumed = used;
    return PQ2O_OK;
}

/* uleb128.rs:30-46 */
size_t pq2o_uleb128_encode(uint64_t value, uint8_t container[10]) {
    size_t consumed = 0;
    for (;;) {
        uint8_t byte = (uint8_t)value & (uint8_t)~128u;
        value >>= 7;
        if (value != 0) byte |= 128;
        container[consumed] = byte;
        consumed += 1;
        if (value == 0) break;
    }
    return consumed;
}

/* ------------------------------------------------------------------ */
/* src/encoding/bitpacked/unpack.rs and pack.rs                        */
/* ------------------------------------------------------------------ */

#define PQ2O_DEFINE_UNPACK(NAME, T, BYTES, BITS)                                                      \
    /* unpack.rs:23-68 for a width known to the caller */                                             \
    static inline __attribute__((always_inline)) void NAME##_width(const uint8_t *input, T *output,  \
                                                                   const unsigned W) {               \
        if (W == 0) { /* unpack.rs:24-29 */                                                           \
            for (unsigned i = 0; i < BITS; ++i) output[i] = 0;                                        \
            return;                                                                                   \
        }                                                                                             \
        const T mask = (W == BITS) ? (T) ~(T)0 : (T)((((T)1) << (W % BITS)) - 1); /* :33-36 */        \
        _Pragma("GCC unroll 64") for (unsigned i = 0; i < BITS; ++i) {                                \
            const unsigned start_bit = i * W;                                                         \
            const unsigned end_bit = start_bit + W;                                                   \
            const unsigned start_bit_offset = start_bit % BITS;                                       \
            const unsigned end_bit_offset = end_bit % BITS;                                           \
            const unsigned start_byte = start_bit / BITS;                                             \
            const unsigned end_byte = end_bit / BITS;                                                 \
            T val;                                                                                    \
            memcpy(&val, input + start_byte * BYTES, BYTES); /* little-endian host */                 \
            if (start_byte != end_byte && end_bit_offset != 0) { /* :56-62 */                         \
                T a = (T)(val >> start_bit_offset);                                                   \
                T val2;                                                                               \
                memcpy(&val2, input + end_byte * BYTES, BYTES);                                       \
                T b = (T)(val2 << (W - end_bit_offset));                                              \
                output[i] = (T)(a | (b & mask));                                                      \
            } else { /* :63-66 */                                                                     \
                output[i] = (T)((val >> start_bit_offset) & mask);                                    \
            }                                                                                         \
        }                                                                                             \
    }

#define PQ2O_DEFINE_PACK(NAME, T, BYTES, BITS)                                                        \
    /* pack.rs:4-48 */                                                                                \
    static void NAME##_width(const T *input, uint8_t *output, const unsigned W) {                     \
        if (W == 0) return; /* pack.rs:5-10 zeroes the output; callers pass zeroed buffers */         \
        const T mask = (W == BITS) ? (T) ~(T)0 : (T)((((T)1) << (W % BITS)) - 1);                     \
        for (unsigned i = 0; i < BITS; ++i) {                                                         \
            const unsigned start_bit = i * W;                                                         \
            const unsigned end_bit = start_bit + W;                                                   \
            const unsigned start_bit_offset = start_bit % BITS;                                       \
            const unsigned end_bit_offset = end_bit % BITS;                                           \
            const unsigned start_byte = start_bit / BITS;                                             \
            const unsigned end_byte = end_bit / BITS;                                                 \
            if (start_byte != end_byte && end_bit_offset != 0) { /* :27-38 */                         \
                T a = (T)(input[i] << start_bit_offset);                                              \
                uint8_t le[BYTES];                                                                    \
                memcpy(le, &a, BYTES);                                                                \
                for (unsigned k = 0; k < BYTES; ++k) output[start_byte * BYTES + k] |= le[k];         \
                T b = (T)((T)(input[i] >> (W - end_bit_offset)) & mask);                              \
                memcpy(le, &b, BYTES);                                                                \
                for (unsigned k = 0; k < BYTES; ++k) output[end_byte * BYTES + k] |= le[k];           \
            } else { /* :39-46 */                                                                     \
                T val = (T)((T)(input[i] & mask) << start_bit_offset);                                \
                uint8_t le[BYTES];                                                                    \
                memcpy(le, &val, BYTES);                                                              \
                for (unsigned k = 0; k < BYTES; ++k) output[start_byte * BYTES + k] |= le[k];         \
            }                                                                                         \
        }                                                                                             \
    }

PQ2O_DEFINE_UNPACK(unpack8, uint8_t, 1, 8)
PQ2O_DEFINE_UNPACK(unpack16, uint16_t, 2, 16)
PQ2O_DEFINE_UNPACK(unpack32, uint32_t, 4, 32)
PQ2O_DEFINE_UNPACK(unpack64, uint64_t, 8, 64)
PQ2O_DEFINE_PACK(pack8, uint8_t, 1, 8)
PQ2O_DEFINE_PACK(pack16, uint16_t, 2, 16)
PQ2O_DEFINE_PACK(pack32, uint32_t, 4, 32)
PQ2O_DEFINE_PACK(pack64, uint64_t, 8, 64)

/* unpack.rs:80-88 -- the reference dispatches through a compile-time
 * unrolled `if i == num_bits` chain ("optimised into a jump table"); the u32
 * dispatcher below does the same with a switch so every width gets its own
 * fully unrolled body, as in the reference. */
#define PQ2O_CASE(N) \
    case N:          \
        unpack32_width(input, output, N); \
        return PQ2O_OK;
int32_t pq2o_unpack32(const uint8_t *input, uint32_t output[32], size_t num_bits) {
    switch (num_bits) {
        PQ2O_CASE(0) PQ2O_CASE(1) PQ2O_CASE(2) PQ2O_CASE(3) PQ2O_CASE(4) PQ2O_CASE(5) PQ2O_CASE(6) PQ2O_CASE(7)
        PQ2O_CASE(8) PQ2O_CASE(9) PQ2O_CASE(10) PQ2O_CASE(11) PQ2O_CASE(12) PQ2O_CASE(13) PQ2O_CASE(14)
        PQ2O_CASE(15) PQ2O_CASE(16) PQ2O_CASE(17) PQ2O_CASE(18) PQ2O_CASE(19) PQ2O_CASE(20) PQ2O_CASE(21)
        PQ2O_CASE(22) PQ2O_CASE(23) PQ2O_CASE(24) PQ2O_CASE(25) PQ2O_CASE(26) PQ2O_CASE(27) PQ2O_CASE(28)
        PQ2O_CASE(29) PQ2O_CASE(30) PQ2O_CASE(31) PQ2O_CASE(32)
        default:
            return PQ2O_REFERENCE_WOULD_PANIC; /* unreachable!() unpack.rs:87 */
    }
}
#undef PQ2O_CASE

int32_t pq2o_unpack8(const uint8_t *input, uint8_t output[8], size_t num_bits) {
    if (num_bits > 8) return PQ2O_REFERENCE_WOULD_PANIC;
    unpack8_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_unpack16(const uint8_t *input, uint16_t output[16], size_t num_bits) {
    if (num_bits > 16) return PQ2O_REFERENCE_WOULD_PANIC;
    unpack16_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_unpack64(const uint8_t *input, uint64_t output[64], size_t num_bits) {
    if (num_bits > 64) return PQ2O_REFERENCE_WOULD_PANIC;
    unpack64_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_pack8(const uint8_t input[8], uint8_t *output, size_t num_bits) {
    if (num_bits > 8) return PQ2O_REFERENCE_WOULD_PANIC;
    pack8_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_pack16(const uint16_t input[16], uint8_t *output, size_t num_bits) {
    if (num_bits > 16) return PQ2O_REFERENCE_WOULD_PANIC;
    pack16_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_pack32(const uint32_t input[32], uint8_t *output, size_t num_bits) {
    if (num_bits > 32) return PQ2O_REFERENCE_WOULD_PANIC;
    pack32_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}
int32_t pq2o_pack64(const uint64_t input[64], uint8_t *output, size_t num_bits) {
    if (num_bits > 64) return PQ2O_REFERENCE_WOULD_PANIC;
    pack64_width(input, output, (unsigned)num_bits);
    return PQ2O_OK;
}

/* src/encoding/bitpacked/encode.rs:11-36 (T = u32) */
void pq2o_bitpacked_encode_u32(const uint32_t *unpacked, size_t len, size_t num_bits, uint8_t *packed,
                               size_t packed_len) {
    const size_t packed_size = (32 * num_bits + 7) / 8;
    size_t chunks = len / 32;
    size_t remainder = len % 32;
    size_t off = 0;
    for (size_t c = 0; c < chunks && off + packed_size <= packed_len; ++c, off += packed_size)
        pack32_width(unpacked + 32 * c, packed + off, (unsigned)num_bits);
    if (remainder != 0 && off + packed_size <= packed_len) {
        uint32_t last_chunk[32] = {0};
        for (size_t i = 0; i < remainder; ++i) last_chunk[i] = unpacked[32 * chunks + i];
        pack32_width(last_chunk, packed + off, (unsigned)num_bits);
    }
}

/* ------------------------------------------------------------------ */
/* src/encoding/bitpacked/decode.rs                                    */
/* ------------------------------------------------------------------ */

/* decode.rs:17-26, T = u32 */
static inline void decode_pack_u32(const uint8_t *packed, size_t packed_len, size_t num_bits, uint32_t *unpacked) {
    if (packed_len < 32 * num_bits / 8) {
        uint8_t buf[32 * 4];
        memset(buf, 0, sizeof(buf));
        memcpy(buf, packed, packed_len);
        (void)pq2o_unpack32(buf, unpacked, num_bits);
    } else {
        (void)pq2o_unpack32(packed, unpacked, num_bits);
    }
}

/* decode.rs:30-59 */
int32_t pq2o_bitpacked_u32_try_new(pq2o_bitpacked_u32 *d, const uint8_t *packed, size_t packed_len,
                                   size_t num_bits, size_t length) {
    const size_t block_size = sizeof(uint32_t) * num_bits;
    if (num_bits == 0) return PQ2O_OUT_OF_SPEC;                                      /* :33-35 */
    if ((unsigned __int128)packed_len * 8 < (unsigned __int128)length * num_bits)   /* :37-42 */
        return PQ2O_OUT_OF_SPEC;
    if (num_bits > 32 && packed_len > 0) return PQ2O_REFERENCE_WOULD_PANIC; /* unpack.rs:87 on the first chunk */
    memset(d->unpacked, 0, sizeof(d->unpacked));
    d->packed = packed;
    d->packed_len = packed_len;
    d->block_size = block_size;
    d->num_bits = num_bits;
    d->current_pack_index = 0;
    if (packed_len > 0) { /* packed.chunks(block_size).next() :44-50 */
        size_t take = packed_len < block_size ? packed_len : block_size;
        decode_pack_u32(packed, take, num_bits, d->unpacked);
        d->packed += take;
        d->packed_len -= take;
    } else {
        length = 0;
    }
    d->remaining = length;
    return PQ2O_OK;
}

/* decode.rs:66-80 */
int pq2o_bitpacked_u32_next(pq2o_bitpacked_u32 *d, uint32_t *out) {
    if (d->remaining == 0) return 0;
    uint32_t result = d->unpacked[d->current_pack_index];
    d->current_pack_index += 1;
    d->remaining -= 1;
    if (d->current_pack_index == 32) {
        if (d->packed_len > 0) {
            size_t take = d->packed_len < d->block_size ? d->packed_len : d->block_size;
            decode_pack_u32(d->packed, take, d->num_bits, d->unpacked);
            d->packed += take;
            d->packed_len -= take;
            d->current_pack_index = 0;
        }
    }
    *out = result;
    return 1;
}

static inline void decode_pack_u64(const uint8_t *packed, size_t packed_len, size_t num_bits, uint64_t *unpacked) {
    if (packed_len < 64 * num_bits / 8) {
        uint8_t buf[64 * 8]; /* the reference's Packed for u64 is 64*64 bytes (mod.rs:46-52); 512 suffice */
        memset(buf, 0, sizeof(buf));
        memcpy(buf, packed, packed_len);
        (void)pq2o_unpack64(buf, unpacked, num_bits);
    } else {
        (void)pq2o_unpack64(packed, unpacked, num_bits);
    }
}

int32_t pq2o_bitpacked_u64_try_new(pq2o_bitpacked_u64 *d, const uint8_t *packed, size_t packed_len,
                                   size_t num_bits, size_t length) {
    const size_t block_size = sizeof(uint64_t) * num_bits;
    if (num_bits == 0) return PQ2O_OUT_OF_SPEC;
    if ((unsigned __int128)packed_len * 8 < (unsigned __int128)length * num_bits) return PQ2O_OUT_OF_SPEC;
    if (num_bits > 64 && packed_len > 0) return PQ2O_REFERENCE_WOULD_PANIC;
    memset(d->unpacked, 0, sizeof(d->unpacked));
    d->packed = packed;
    d->packed_len = packed_len;
    d->block_size = block_size;
    d->num_bits = num_bits;
    d->current_pack_index = 0;
    if (packed_len > 0) {
        size_t take = packed_len < block_size ? packed_len : block_size;
        decode_pack_u64(packed, take, num_bits, d->unpacked);
        d->packed += take;
        d->packed_len -= take;
    } else {
        length = 0;
    }
    d->remaining = length;
    return PQ2O_OK;
}

int pq2o_bitpacked_u64_next(pq2o_bitpacked_u64 *d, uint64_t *out) {
    if (d->remaining == 0) return 0;
    uint64_t result = d->unpacked[d->current_pack_index];
    d->current_pack_index += 1;
    d->remaining -= 1;
    if (d->current_pack_index == 64) {
        if (d->packed_len > 0) {
            size_t take = d->packed_len < d->block_size ? d->packed_len : d->block_size;
            decode_pack_u64(d->packed, take, d->num_bits, d->unpacked);
            d->packed += take;
            d->packed_len -= take;
            d->current_pack_index = 0;
        }
    }
    *out = result;
    return 1;
}

int32_t pq2o_bitunpack_u32(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                           uint32_t *out) {
    pq2o_bitpacked_u32 d;
    int32_t st = pq2o_bitpacked_u32_try_new(&d, packed, packed_len, num_bits, length);
    if (st != PQ2O_OK) return st;
    size_t i = 0;
    uint32_t v;
    while (pq2o_bitpacked_u32_next(&d, &v)) out[i++] = v;
    return PQ2O_OK;
}

int32_t pq2o_bitunpack_u64(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                           uint64_t *out) {
    pq2o_bitpacked_u64 d;
    int32_t st = pq2o_bitpacked_u64_try_new(&d, packed, packed_len, num_bits, length);
    if (st != PQ2O_OK) return st;
    size_t i = 0;
    uint64_t v;
    while (pq2o_bitpacked_u64_next(&d, &v)) out[i++] = v;
    return PQ2O_OK;
}

/* ------------------------------------------------------------------ */
/* src/encoding/hybrid_rle/decoder.rs                                  */
/* ------------------------------------------------------------------ */

void pq2o_run_decoder_new(pq2o_run_decoder *d, const uint8_t *values, size_t len, size_t num_bits) {
    d->values = values;
    d->len = len;
    d->num_bits = num_bits;
}

/* decoder.rs:30-64 */
int pq2o_run_decoder_next(pq2o_run_decoder *d, pq2o_hybrid_encoded *run) {
    if (d->num_bits == 0) return 0; /* :31-33 */
    if (d->len == 0) return 0;      /* :35-37 */
    uint64_t indicator;
    size_t consumed;
    int32_t st = pq2o_uleb128_decode(d->values, d->len, &indicator, &consumed); /* :39-42 */
    if (st != PQ2O_OK) return -st;
    d->values += consumed; /* :43 */
    d->len -= consumed;
    if (d->len == 0) return 0; /* :44-46 header at the very end: run dropped */
    if ((indicator & 1) == 1) {
        /* :48-54 bit-packed.  `(indicator as usize >> 1) * self.num_bits` overflows usize only
         * for absurd headers; the reference panics there in debug builds. */
        unsigned __int128 wide = (unsigned __int128)(indicator >> 1) * d->num_bits;
        if (wide > (unsigned __int128)SIZE_MAX) return -PQ2O_REFERENCE_WOULD_PANIC;
        size_t bytes = (size_t)wide;
        if (bytes > d->len) bytes = d->len; /* :51 */
        run->kind = PQ2O_RUN_BITPACKED;
        run->data = d->values;
        run->data_len = bytes;
        run->run_length = 0;
        d->values += bytes;
        d->len -= bytes;
        return 1;
    } else {
        /* :55-63 rle; split_at panics when fewer than rle_bytes remain */
        size_t run_length = (size_t)(indicator >> 1);
        size_t rle_bytes = pq2o_ceil8(d->num_bits);
        if (rle_bytes > d->len) return -PQ2O_REFERENCE_WOULD_PANIC;
        run->kind = PQ2O_RUN_RLE;
        run->data = d->values;
        run->data_len = rle_bytes;
        run->run_length = run_length;
        d->values += rle_bytes;
        d->len -= rle_bytes;
        return 1;
    }
}

/* ------------------------------------------------------------------ */
/* src/encoding/hybrid_rle/mod.rs                                      */
/* ------------------------------------------------------------------ */

enum { ST_NONE = 0, ST_BITPACKED = 1, ST_RLE = 2, ST_SINGLE = 3 };

/* mod.rs:41-64 read_next */
static int32_t read_next(pq2o_hybrid_rle_decoder *d, size_t remaining) {
    pq2o_hybrid_encoded run;
    int r = pq2o_run_decoder_next(&d->decoder, &run);
    if (r < 0) return -r;
    if (r == 0) {
        d->state = ST_NONE; /* :62 */
        return PQ2O_OK;
    }
    if (run.kind == PQ2O_RUN_BITPACKED) {
        size_t num_bits = d->decoder.num_bits; /* :45-48 */
        size_t length = run.data_len * 8 / num_bits;
        if (length > remaining) length = remaining;
        int32_t st = pq2o_bitpacked_u32_try_new(&d->bitpacked, run.data, run.data_len, num_bits, length);
        if (st != PQ2O_OK) return st;
        d->state = ST_BITPACKED;
    } else {
        uint8_t bytes[4] = {0, 0, 0, 0}; /* :51-55 zip: at most 4 bytes, value not masked */
        for (size_t i = 0; i < run.data_len && i < 4; ++i) bytes[i] = run.data[i];
        uint32_t value = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) |
                         ((uint32_t)bytes[3] << 24);
        d->rle_value = value;
        if (run.run_length == 1) { /* :56-57 */
            d->state = ST_SINGLE;
            d->single_some = 1;
        } else { /* :58-60 */
            d->state = ST_RLE;
            d->rle_left = run.run_length;
        }
    }
    return PQ2O_OK;
}

/* mod.rs:68-77 */
int32_t pq2o_hybrid_rle_try_new(pq2o_hybrid_rle_decoder *d, const uint8_t *data, size_t len, uint32_t num_bits,
                                size_t num_values) {
    pq2o_run_decoder_new(&d->decoder, data, len, (size_t)num_bits);
    d->state = ST_NONE;
    d->remaining = num_values;
    return read_next(d, num_values);
}

/* mod.rs:83-109 */
int pq2o_hybrid_rle_next(pq2o_hybrid_rle_decoder *d, uint32_t *out) {
    for (;;) {
        if (d->remaining == 0) return 0; /* :84-86 */
        int have = 0;
        uint32_t result = 0;
        switch (d->state) {
            case ST_SINGLE: /* :88-92 */
                if (d->single_some) {
                    d->single_some = 0;
                    result = d->rle_value;
                    have = 1;
                }
                break;
            case ST_BITPACKED: /* :93 */
                have = pq2o_bitpacked_u32_next(&d->bitpacked, &result);
                break;
            case ST_RLE: /* :94 */
                if (d->rle_left > 0) {
                    d->rle_left -= 1;
                    result = d->rle_value;
                    have = 1;
                }
                break;
            default: /* :95 State::None => Some(0) */
                result = 0;
                have = 1;
                break;
        }
        if (have) { /* :97-99 */
            d->remaining -= 1;
            *out = result;
            return 1;
        }
        int32_t st = read_next(d, d->remaining); /* :101-107 then self.next() */
        if (st != PQ2O_OK) return -st;
    }
}

int32_t pq2o_hybrid_rle_decode_u32(const uint8_t *data, size_t len, uint32_t num_bits, size_t num_values,
                                   uint32_t *out) {
    pq2o_hybrid_rle_decoder d;
    int32_t st = pq2o_hybrid_rle_try_new(&d, data, len, num_bits, num_values);
    if (st != PQ2O_OK) return st;
    size_t i = 0;
    for (;;) {
        uint32_t v;
        int r = pq2o_hybrid_rle_next(&d, &v);
        if (r == 0) break;
        if (r < 0) return -r;
        out[i++] = v;
    }
    return PQ2O_OK;
}

/* ------------------------------------------------------------------ */
/* src/encoding/hybrid_rle/bitmap.rs                                   */
/* ------------------------------------------------------------------ */

static inline uint8_t rotl8(uint8_t v, unsigned n) {
    n &= 7;
    return (uint8_t)((v << n) | (v >> ((8 - n) & 7)));
}

/* bitmap.rs:26-39 */
int32_t pq2o_bitmap_iter_new(pq2o_bitmap_iter *it, const uint8_t *slice, size_t slice_len, size_t offset,
                             size_t len) {
    if (offset / 8 > slice_len) return PQ2O_REFERENCE_WOULD_PANIC; /* &slice[offset / 8..] */
    const uint8_t *bytes = slice + offset / 8;
    const uint8_t *end = slice + slice_len;
    if (bytes < end) {
        it->current_byte = *bytes;
        it->iter = bytes + 1;
    } else {
        it->current_byte = 0; /* unwrap_or(&0) */
        it->iter = end;
    }
    it->end = end;
    it->mask = rotl8(1, (unsigned)(offset & 7)); /* 1u8.rotate_left(offset as u32) */
    it->remaining = len;
    return PQ2O_OK;
}

/* bitmap.rs:46-62 */
int pq2o_bitmap_iter_next(pq2o_bitmap_iter *it, uint8_t *out) {
    if (it->remaining == 0) return 0;
    it->remaining -= 1;
    uint8_t value = (it->current_byte & it->mask) != 0;
    it->mask = rotl8(it->mask, 1);
    if (it->mask == 1) {
        if (it->iter < it->end) { /* keeps the last byte when the slice is exhausted */
            it->current_byte = *it->iter;
            it->iter += 1;
        }
    }
    *out = value;
    return 1;
}

/* bitmap.rs:71-102 */
size_t pq2o_bitmap_encode_bool(const uint8_t *bools, size_t length, uint8_t *out) {
    size_t chunks = length / 8;
    size_t reminder = length % 8;
    size_t w = 0;
    for (size_t c = 0; c < chunks; ++c) {
        uint8_t byte = 0;
        for (unsigned i = 0; i < 8; ++i)
            if (bools[8 * c + i]) byte |= (uint8_t)(1u << i);
        out[w++] = byte;
    }
    if (reminder != 0) {
        uint8_t last = 0;
        for (unsigned i = 0; i < reminder; ++i)
            if (bools[8 * chunks + i]) last |= (uint8_t)(1u << i);
        out[w++] = last;
    }
    return w;
}

/* ------------------------------------------------------------------ */
/* src/encoding/hybrid_rle/encoder.rs                                  */
/* ------------------------------------------------------------------ */

/* encoder.rs:33-72 */
static size_t bitpacked_encode_u32(const uint32_t *values, size_t length, size_t num_bits, uint8_t *out) {
    const size_t chunks = length / 32;
    const size_t remainder = length - chunks * 32;
    uint32_t buffer[32];
    memset(buffer, 0, sizeof(buffer));
    const size_t compressed_chunk_size = pq2o_ceil8(32 * num_bits);
    size_t w = 0;
    for (size_t c = 0; c < chunks; ++c) {
        for (unsigned i = 0; i < 32; ++i) buffer[i] = values[32 * c + i];
        uint8_t packed[4 * 32];
        memset(packed, 0, sizeof(packed));
        pack32_width(buffer, packed, (unsigned)num_bits); /* encode_pack, full block */
        memcpy(out + w, packed, compressed_chunk_size);
        w += compressed_chunk_size;
    }
    if (remainder != 0) {
        const size_t compressed_remainder_size = pq2o_ceil8(remainder * num_bits);
        /* :61-65 overwrites only the first `remainder` slots: the tail of `buffer`
         * keeps the values of the previous full block, exactly as in the reference */
        for (size_t i = 0; i < remainder; ++i) buffer[i] = values[32 * chunks + i];
        uint8_t packed[4 * 32];
        memset(packed, 0, sizeof(packed));
        pack32_width(buffer, packed, (unsigned)num_bits);
        memcpy(out + w, packed, compressed_remainder_size);
        w += compressed_remainder_size;
    }
    return w;
}

/* encoder.rs:9-29 */
size_t pq2o_hybrid_rle_encode_u32(const uint32_t *values, size_t length, uint32_t num_bits, uint8_t *out) {
    uint8_t nb = (uint8_t)num_bits; /* :14 */
    uint64_t header = (uint64_t)pq2o_ceil8(length);
    header <<= 1;
    header |= 1;
    uint8_t container[10];
    size_t used = pq2o_uleb128_encode(header, container);
    memcpy(out, container, used);
    return used + bitpacked_encode_u32(values, length, (size_t)nb, out + used);
}

/* encoder.rs:75-93 */
size_t pq2o_hybrid_rle_encode_bool(const uint8_t *bools, size_t length, uint8_t *out) {
    uint64_t header = (uint64_t)pq2o_ceil8(length);
    header <<= 1;
    header |= 1;
    uint8_t container[10];
    size_t used = pq2o_uleb128_encode(header, container);
    memcpy(out, container, used);
    return used + pq2o_bitmap_encode_bool(bools, length, out + used);
}

/* ------------------------------------------------------------------ */
/* src/page/mod.rs:350-411                                             */
/* ------------------------------------------------------------------ */

int32_t pq2o_split_buffer_v1(const uint8_t *buffer, size_t len, int has_rep, int has_def, pq2o_split *out) {
    size_t base = 0;
    memset(out, 0, sizeof(*out));
    if (has_rep) { /* :357-375 */
        size_t l;
        if (pq2o_get_length(buffer + base, len - base, &l) != 0) return PQ2O_OUT_OF_SPEC;
        if (4 + l > len - base) return PQ2O_OUT_OF_SPEC; /* buffer.get(4..4 + l) */
        out->rep_off = base + 4;
        out->rep_len = l;
        base += 4 + l;
    } else {
        out->rep_off = base;
        out->rep_len = 0;
    }
    if (has_def) { /* :377-395 */
        size_t l;
        if (pq2o_get_length(buffer + base, len - base, &l) != 0) return PQ2O_OUT_OF_SPEC;
        if (4 + l > len - base) return PQ2O_OUT_OF_SPEC;
        out->def_off = base + 4;
        out->def_len = l;
        base += 4 + l;
    } else {
        out->def_off = base;
        out->def_len = 0;
    }
    out->val_off = base;
    out->val_len = len - base;
    return PQ2O_OK;
}

int32_t pq2o_split_buffer_v2(size_t len, size_t rep_len, size_t def_len, pq2o_split *out) {
    /* :406-410 plain slicing: panics when the lengths exceed the buffer */
    if (rep_len > len || def_len > len - rep_len) return PQ2O_REFERENCE_WOULD_PANIC;
    out->rep_off = 0;
    out->rep_len = rep_len;
    out->def_off = rep_len;
    out->def_len = def_len;
    out->val_off = rep_len + def_len;
    out->val_len = len - rep_len - def_len;
    return PQ2O_OK;
}

/* src/deserialize/utils.rs:13-27 */
int32_t pq2o_dict_indices_locate(const uint8_t *values, size_t len, uint32_t *bit_width, size_t *stream_off) {
    if (len == 0) return PQ2O_REFERENCE_WOULD_PANIC; /* indices_buffer[0] */
    uint8_t bw = values[0];
    if (bw > 32) return PQ2O_OUT_OF_SPEC;
    *bit_width = bw;
    *stream_off = 1;
    return PQ2O_OK;
}

/* ------------------------------------------------------------------ */
/* src/deserialize/hybrid_rle.rs                                       */
/* ------------------------------------------------------------------ */

/* hybrid_rle.rs:83-120 over hybrid_rle::Decoder::new(data, 1) (utils.rs:46-47) */
int32_t pq2o_hybrid_rle_bitmap_runs(const uint8_t *data, size_t len, size_t num_values, pq2o_bitmap_run *runs,
                                    size_t cap, size_t *n_runs) {
    pq2o_run_decoder dec;
    pq2o_run_decoder_new(&dec, data, len, 1);
    size_t consumed = 0, n = 0;
    for (;;) {
        if (consumed == num_values) break; /* :91-93 */
        pq2o_hybrid_encoded run;
        int r = pq2o_run_decoder_next(&dec, &run);
        if (r == 0) break; /* :94 */
        if (r < 0) {
            *n_runs = n;
            return -r;
        }
        pq2o_bitmap_run o;
        memset(&o, 0, sizeof(o));
        size_t left = num_values - consumed;
        if (run.kind == PQ2O_RUN_BITPACKED) { /* :97-105 */
            size_t pack_size = run.data_len * 8;
            size_t additional = pack_size < left ? pack_size : left;
            consumed += additional;
            o.kind = PQ2O_BITMAP;
            o.offset = (uint64_t)(run.data - data);
            o.bytes = run.data_len;
            o.length = additional;
        } else { /* :106-113 */
            size_t additional = run.run_length < left ? run.run_length : left;
            consumed += additional;
            o.kind = PQ2O_REPEATED;
            o.is_set = run.data[0] == 1;
            o.length = additional;
        }
        if (n < cap) runs[n] = o;
        n += 1;
    }
    *n_runs = n;
    return PQ2O_OK;
}

/* hybrid_rle.rs:157-202.  `current_run` is set once and never cleared, so
 * the iterator ends with the first run (:165-175). */
int32_t pq2o_hybrid_rle_boolean_iter(const uint8_t *data, size_t len, size_t num_values, uint8_t *out, size_t cap,
                                     size_t *n) {
    pq2o_bitmap_run first;
    memset(&first, 0, sizeof(first));
    *n = 0;
    if (num_values == 0) return PQ2O_OK; /* HybridRleIter::next: consumed == length */
    {
        pq2o_run_decoder dec;
        pq2o_hybrid_encoded run;
        pq2o_run_decoder_new(&dec, data, len, 1);
        int r = pq2o_run_decoder_next(&dec, &run);
        if (r == 0) return PQ2O_OK;
        if (r < 0) return -r; /* Some(Err(e)) */
        if (run.kind == PQ2O_RUN_BITPACKED) {
            size_t pack_size = run.data_len * 8;
            first.kind = PQ2O_BITMAP;
            first.offset = (uint64_t)(run.data - data);
            first.bytes = run.data_len;
            first.length = pack_size < num_values ? pack_size : num_values;
        } else {
            first.kind = PQ2O_REPEATED;
            first.is_set = run.data[0] == 1;
            first.length = run.run_length < num_values ? run.run_length : num_values;
        }
    }
    size_t k = 0;
    if (first.kind == PQ2O_BITMAP) {
        pq2o_bitmap_iter it;
        int32_t s2 = pq2o_bitmap_iter_new(&it, data + first.offset, (size_t)first.bytes, 0, (size_t)first.length);
        if (s2 != PQ2O_OK) return s2;
        uint8_t v;
        while (pq2o_bitmap_iter_next(&it, &v)) {
            if (k < cap) out[k] = v;
            k++;
        }
    } else {
        for (uint64_t i = 0; i < first.length; ++i) {
            if (k < cap) out[k] = (uint8_t)first.is_set;
            k++;
        }
    }
    *n = k;
    return PQ2O_OK;
}

/* ------------------------------------------------------------------ */
/* src/deserialize/filtered_rle.rs                                     */
/* ------------------------------------------------------------------ */

/* filtered_rle.rs:30-34 */
static size_t is_set_count(const uint8_t *values, size_t values_len, size_t offset, size_t length) {
    pq2o_bitmap_iter it;
    if (pq2o_bitmap_iter_new(&it, values, values_len, offset, length) != PQ2O_OK) return 0;
    size_t c = 0;
    uint8_t v;
    while (pq2o_bitmap_iter_next(&it, &v)) c += v;
    return c;
}

/* filtered_rle.rs:102-273, with the tail recursion of :113 and :127 turned into a loop */
int32_t pq2o_filtered_bitmap_runs(const pq2o_bitmap_run *runs, size_t n_runs, const uint8_t *stream,
                                  const pq2o_interval *intervals, size_t n_intervals, pq2o_filtered_run *out,
                                  size_t cap, size_t *n_out) {
    size_t next_run = 0, next_interval = 0, n = 0;
    int have_current = 0, have_interval = 0;
    pq2o_bitmap_run run;
    memset(&run, 0, sizeof(run));
    uint64_t offset = 0;
    pq2o_interval interval = {0, 0};
    uint64_t current_items_in_runs = 0;

    for (;;) {
        if (!have_interval) { /* :108-114 */
            if (next_interval == n_intervals) break;
            interval = intervals[next_interval++];
            have_interval = 1;
            continue;
        }
        if (!have_current) { /* :116-128 */
            if (next_run == n_runs) break;
            run = runs[next_run++];
            offset = 0;
            have_current = 1;
            continue;
        }
        const uint64_t full_run_length = run.length;
        const uint64_t run_length = full_run_length - offset;
        if (interval.start < current_items_in_runs) {
            *n_out = n;
            return PQ2O_REFERENCE_WOULD_PANIC; /* usize underflow at :140 / :206 */
        }
        const uint64_t interval_start = interval.start - current_items_in_runs;
        pq2o_filtered_run o;
        memset(&o, 0, sizeof(o));
        const uint8_t *values = run.kind == PQ2O_BITMAP ? stream + run.offset : NULL;

        if (interval_start > 0) { /* :142-161 / :208-227 */
            uint64_t max_skip = full_run_length - offset;
            uint64_t to_skip = interval_start < max_skip ? interval_start : max_skip;
            uint64_t set;
            if (run.kind == PQ2O_REPEATED)
                set = run.is_set ? to_skip : 0;
            else
                set = is_set_count(values, (size_t)run.bytes, (size_t)offset, (size_t)to_skip);
            current_items_in_runs += to_skip;
            if (to_skip == max_skip)
                have_current = 0;
            else
                offset += to_skip;
            o.kind = PQ2O_F_SKIPPED;
            o.set_count = set;
        } else {
            const uint64_t new_offset = offset + interval_start; /* :165 / :231 */
            /* `interval_start > run_length` (:167 / :233) cannot hold here: interval_start == 0 */
            uint64_t length;
            if (run_length > interval.length) { /* :175-185 / :241-251 */
                current_items_in_runs += interval.length;
                length = interval.length;
                offset += interval.length;
                if (next_interval < n_intervals) {
                    interval = intervals[next_interval++];
                } else {
                    have_interval = 0;
                }
            } else { /* :186-198 / :252-264 */
                current_items_in_runs += run_length;
                uint64_t alt = full_run_length - new_offset;
                length = run_length < alt ? run_length : alt;
                interval.start += length; /* advance_current_interval :82-88 */
                interval.length -= length;
                have_current = 0;
            }
            if (run.kind == PQ2O_REPEATED) {
                o.kind = PQ2O_F_REPEATED;
                o.is_set = run.is_set;
                o.length = length;
            } else {
                o.kind = PQ2O_F_BITMAP;
                o.values_off = run.offset;
                o.values_len = run.bytes;
                o.offset = new_offset;
                o.length = length;
            }
        }
        if (n < cap) out[n] = o;
        n += 1;
    }
    *n_out = n;
    return PQ2O_OK;
}

/* ------------------------------------------------------------------ */
/* timing helpers for the cpu_baseline leg of bench.py                 */
/* ------------------------------------------------------------------ */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double pq2o_time_hybrid_rle_decode(const uint8_t *data, size_t len, uint32_t num_bits, size_t num_values,
                                   uint32_t *out, int reps) {
    double t0 = now_s();
    for (int r = 0; r < reps; ++r) {
        (void)pq2o_hybrid_rle_decode_u32(data, len, num_bits, num_values, out);
        __asm__ volatile("" ::"r"(out) : "memory");
    }
    return now_s() - t0;
}

double pq2o_time_bitunpack_u32(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                               uint32_t *out, int reps) {
    double t0 = now_s();
    for (int r = 0; r < reps; ++r) {
        (void)pq2o_bitunpack_u32(packed, packed_len, num_bits, length, out);
        __asm__ volatile("" ::"r"(out) : "memory");
    }
    return now_s() - t0;
}

/* ---- zigzag_leb128 and delta_bitpacked ------------------------------------------------------------------ */

/* src/encoding/zigzag_leb128.rs:5-8 */
int32_t pq2o_zigzag_leb128_decode(const uint8_t *values, size_t len, int64_t *value, size_t *consumed) {
    uint64_t u;
    int32_t st = pq2o_uleb128_decode(values, len, &u, consumed);
    if (st != PQ2O_OK) return st;
    *value = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
    return PQ2O_OK;
}

/* delta_bitpacked::Block (decoder.rs:9-118) */
typedef struct {
    int64_t min_delta;
    size_t values_per_mini_block;
    const uint8_t *bitwidths, *bitwidths_end; /* std::slice::Iter<u8> */
    const uint8_t *values;
    size_t values_len;
    size_t remaining;
    size_t current_index;
    int has_miniblock; /* Option<bitpacked::Decoder<u64>> */
    pq2o_bitpacked_u64 miniblock;
    size_t consumed_bytes;
} delta_block;

/* Block::advance_miniblock, decoder.rs:66-91 */
static int32_t delta_advance_miniblock(delta_block *b) {
    if (b->bitwidths == b->bitwidths_end) return PQ2O_REFERENCE_WOULD_PANIC; /* .next().unwrap() */
    size_t num_bits = *b->bitwidths++;
    if (num_bits > 0) {
        size_t length = b->remaining < b->values_per_mini_block ? b->remaining : b->values_per_mini_block;
        size_t miniblock_length = (b->values_per_mini_block * num_bits + 7) / 8; /* ceil8 */
        if (miniblock_length > b->values_len) return PQ2O_OUT_OF_SPEC;
        const uint8_t *miniblock = b->values;
        b->values += miniblock_length;
        b->values_len -= miniblock_length;
        b->consumed_bytes += miniblock_length;
        /* bitpacked::Decoder::try_new(miniblock, num_bits, length).unwrap(): unpack64 has no width above 64 */
        int32_t st = pq2o_bitpacked_u64_try_new(&b->miniblock, miniblock, miniblock_length, num_bits, length);
        if (st != PQ2O_OK) return PQ2O_REFERENCE_WOULD_PANIC;
        b->has_miniblock = 1;
    } else {
        b->has_miniblock = 0;
    }
    b->current_index = 0;
    return PQ2O_OK;
}

/* Block::try_new, decoder.rs:26-64 */
static int32_t delta_block_try_new(delta_block *b, const uint8_t *values, size_t len, size_t num_mini_blocks,
                                   size_t values_per_mini_block, size_t length) {
    size_t most = num_mini_blocks * values_per_mini_block;
    if (length > most) length = most;
    size_t consumed;
    int32_t st = pq2o_zigzag_leb128_decode(values, len, &b->min_delta, &consumed);
    if (st != PQ2O_OK) return st;
    b->consumed_bytes = consumed;
    values += consumed;
    len -= consumed;
    if (num_mini_blocks > len) return PQ2O_OUT_OF_SPEC;
    b->bitwidths = values;
    b->bitwidths_end = values + num_mini_blocks;
    b->consumed_bytes += num_mini_blocks;
    b->values = values + num_mini_blocks;
    b->values_len = len - num_mini_blocks;
    b->values_per_mini_block = values_per_mini_block;
    b->remaining = length;
    b->current_index = 0;
    b->has_miniblock = 0;
    return delta_advance_miniblock(b);
}

/* Iterator for Block, decoder.rs:93-118: 1 = a value, 0 = None, negative = -(error status) */
static int delta_block_next(delta_block *b, int64_t *out) {
    if (b->remaining == 0) return 0;
    uint64_t v = 0;
    if (b->has_miniblock) {
        if (!pq2o_bitpacked_u64_next(&b->miniblock, &v)) v = 0; /* unwrap_or_default */
    }
    *out = (int64_t)((uint64_t)b->min_delta + v);
    b->current_index += 1;
    b->remaining -= 1;
    if (b->remaining > 0 && b->current_index == b->values_per_mini_block) {
        int32_t st = delta_advance_miniblock(b);
        if (st != PQ2O_OK) return -st; /* Some(Err(e)): the value is lost */
    }
    return 1;
}

int32_t pq2o_delta_bitpacked_decode(const uint8_t *values, size_t len, int64_t *out, size_t cap, size_t *n_values,
                                    size_t *consumed_bytes) {
    /* Decoder::try_new, decoder.rs:136-180 */
    size_t consumed_total = 0, consumed;
    uint64_t block_size, num_mini_blocks, total_count;
    int64_t first_value;
    int32_t st;
    if ((st = pq2o_uleb128_decode(values, len, &block_size, &consumed)) != PQ2O_OK) return st;
    consumed_total += consumed;
    if (block_size % 128 != 0) return PQ2O_REFERENCE_WOULD_PANIC; /* assert_eq! */
    values += consumed, len -= consumed;
    if ((st = pq2o_uleb128_decode(values, len, &num_mini_blocks, &consumed)) != PQ2O_OK) return st;
    consumed_total += consumed;
    values += consumed, len -= consumed;
    if ((st = pq2o_uleb128_decode(values, len, &total_count, &consumed)) != PQ2O_OK) return st;
    consumed_total += consumed;
    values += consumed, len -= consumed;
    if ((st = pq2o_zigzag_leb128_decode(values, len, &first_value, &consumed)) != PQ2O_OK) return st;
    consumed_total += consumed;
    values += consumed, len -= consumed;
    if (num_mini_blocks == 0) return PQ2O_REFERENCE_WOULD_PANIC; /* division by zero */
    size_t values_per_mini_block = (size_t)(block_size / num_mini_blocks);
    if (values_per_mini_block % 8 != 0) return PQ2O_REFERENCE_WOULD_PANIC; /* assert_eq! */
    *n_values = (size_t)total_count;
    if (total_count > cap) return PQ2O_INVALID_PARAMETER;

    delta_block block;
    int has_block = 0;
    if (total_count > 1) {
        st = delta_block_try_new(&block, values, len, (size_t)num_mini_blocks, values_per_mini_block,
                                 (size_t)total_count - 1);
        if (st != PQ2O_OK) return st;
        has_block = 1;
    }
    /* Iterator for Decoder, decoder.rs:216-243 */
    size_t values_remaining = (size_t)total_count, produced = 0;
    int64_t next_value = first_value;
    while (values_remaining != 0) {
        out[produced++] = next_value;
        values_remaining -= 1;
        if (values_remaining == 0) break;
        /* load_delta, decoder.rs:187-213 */
        int64_t delta;
        int r = delta_block_next(&block, &delta);
        if (r < 0) return -r;
        if (r == 0) {
            values += block.consumed_bytes, len -= block.consumed_bytes;
            consumed_total += block.consumed_bytes;
            st = delta_block_try_new(&block, values, len, (size_t)num_mini_blocks, values_per_mini_block,
                                     values_remaining);
            if (st != PQ2O_OK) return st;
            r = delta_block_next(&block, &delta);
            if (r < 0) return -r;
            if (r == 0) return PQ2O_OUT_OF_SPEC; /* "Missing block" */
        }
        next_value = (int64_t)((uint64_t)next_value + (uint64_t)delta);
    }
    *consumed_bytes = consumed_total + (has_block ? block.consumed_bytes : 0);
    return PQ2O_OK;
}

/* src/encoding/zigzag_leb128.rs:10-15 */
size_t pq2o_zigzag_leb128_encode(int64_t value, uint8_t container[10]) {
    uint64_t u = ((uint64_t)value << 1) ^ (uint64_t)(value >> 63);
    return pq2o_uleb128_encode(u, container);
}

/* encoder.rs:66-78 write_miniblock: bitpacked::encode::<u64> of all 128 slots (encode.rs:11-36 with two full
 * chunks of 64, each pack64 into 8 * num_bits bytes of a zeroed region) */
static size_t delta_write_miniblock(uint8_t *buffer, size_t num_bits, const uint64_t deltas[128]) {
    if (num_bits == 0) return 0;
    const size_t bytes = (128 * num_bits + 7) / 8;
    memset(buffer, 0, bytes);
    pack64_width(deltas, buffer, (unsigned)num_bits);
    pack64_width(deltas + 64, buffer + 8 * num_bits, (unsigned)num_bits);
    return bytes;
}

/* src/encoding/delta_bitpacked/encoder.rs:11-64.  The subtractions wrap (the reference's release build; its
 * debug build panics where they overflow).  `buffer` must hold 32 + ceil(len / 128) * 1035 bytes. */
size_t pq2o_delta_bitpacked_encode(const int64_t *iterator, size_t len, uint8_t *buffer) {
    uint8_t container[10];
    size_t at = 0, next = 0;
    size_t used = pq2o_uleb128_encode(128, container); /* block_size */
    memcpy(buffer + at, container, used), at += used;
    used = pq2o_uleb128_encode(1, container); /* mini_blocks */
    memcpy(buffer + at, container, used), at += used;
    used = pq2o_uleb128_encode((uint64_t)len, container);
    memcpy(buffer + at, container, used), at += used;

    int64_t values[128] = {0};
    uint64_t deltas[128] = {0};
    int64_t first_value = next < len ? iterator[next++] : 0; /* unwrap_or_default */
    used = pq2o_zigzag_leb128_encode(first_value, container);
    memcpy(buffer + at, container, used), at += used;

    int64_t prev = first_value;
    size_t length = len - next;
    while (length != 0) {
        int64_t min_delta = INT64_MAX, max_delta = INT64_MIN;
        unsigned num_bits = 0;
        size_t consumed = 0;
        for (size_t i = 0; i < 128 && next < len; ++i) { /* (0..128).zip(&mut iterator) */
            int64_t integer = iterator[next++];
            int64_t delta = (int64_t)((uint64_t)integer - (uint64_t)prev);
            if (delta < min_delta) min_delta = delta;
            if (delta > max_delta) max_delta = delta;
            uint64_t range = (uint64_t)max_delta - (uint64_t)min_delta;
            num_bits = range == 0 ? 0 : 64 - (unsigned)__builtin_clzll(range);
            values[i] = delta;
            prev = integer;
            consumed = i + 1;
        }
        length = len - next;
        /* the slots behind `consumed` keep the deltas of the block before (:27, :52-54) */
        for (size_t i = 0; i < consumed; ++i) deltas[i] = (uint64_t)values[i] - (uint64_t)min_delta;
        used = pq2o_zigzag_leb128_encode(min_delta, container);
        memcpy(buffer + at, container, used), at += used;
        buffer[at++] = (uint8_t)num_bits;
        at += delta_write_miniblock(buffer + at, num_bits, deltas);
    }
    return at;
}
