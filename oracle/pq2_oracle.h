/*
 * CPU oracle for the hybrid-RLE / bit-unpack path of parquet2 (v0.17.2).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm, kept structurally close to the Rust sources so that it can act
 * as (a) the bit-exact checker of the HIP path and (b) the CPU baseline that
 * bench.py times.  Nothing in the product library links, loads or calls it.
 *
 * Parity pinning: the functions below are checked against every literal
 * known-answer vector that the reference's own unit tests hold for this path
 * (tests/golden/reference_kats.json, transcribed from the `#[cfg(test)]`
 * modules cited next to each function).  Behaviours that no reference test
 * pins are listed in tests/test_oracle_golden.py as "parity unpinned"; they
 * are fixed by reading the cited code only.
 *
 * All paths are relative to the reference root (jorgecarleitao/parquet2).
 */
#ifndef PQ2_ORACLE_H
#define PQ2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes (same numbering as include/pq2hip.h). */
#define PQ2O_OK 0
#define PQ2O_OUT_OF_SPEC 1           /* src/error.rs:23-34 Error::OutOfSpec */
#define PQ2O_INVALID_PARAMETER 2
#define PQ2O_FEATURE_NOT_SUPPORTED 3
#define PQ2O_WOULD_OVER_ALLOCATE 4
#define PQ2O_REFERENCE_WOULD_PANIC 5 /* input on which the Rust code panics */

/* src/encoding/mod.rs:23-27 */
size_t pq2o_ceil8(size_t value);
/* src/encoding/mod.rs:14-21; returns 0 and sets *out, or -1 for None */
int pq2o_get_length(const uint8_t *values, size_t len, size_t *out);
/* src/read/levels.rs:3-5 */
uint32_t pq2o_get_bit_width(int16_t max_level);

/* src/encoding/uleb128.rs:3-23 / :30-46 */
int32_t pq2o_uleb128_decode(const uint8_t *values, size_t len, uint64_t *value, size_t *consumed);
size_t pq2o_uleb128_encode(uint64_t value, uint8_t container[10]);

/* src/encoding/bitpacked/unpack.rs:21-95 (one full block; input must hold
 * num_bits*sizeof(T) bytes).  Returns PQ2O_REFERENCE_WOULD_PANIC when
 * num_bits exceeds the type width (unreachable!() at unpack.rs:87). */
int32_t pq2o_unpack8(const uint8_t *input, uint8_t output[8], size_t num_bits);
int32_t pq2o_unpack16(const uint8_t *input, uint16_t output[16], size_t num_bits);
int32_t pq2o_unpack32(const uint8_t *input, uint32_t output[32], size_t num_bits);
int32_t pq2o_unpack64(const uint8_t *input, uint64_t output[64], size_t num_bits);
/* src/encoding/bitpacked/pack.rs:2-75 (output must hold num_bits*sizeof(T)
 * bytes and is OR-ed into, as in the reference) */
int32_t pq2o_pack8(const uint8_t input[8], uint8_t *output, size_t num_bits);
int32_t pq2o_pack16(const uint16_t input[16], uint8_t *output, size_t num_bits);
int32_t pq2o_pack32(const uint32_t input[32], uint8_t *output, size_t num_bits);
int32_t pq2o_pack64(const uint64_t input[64], uint8_t *output, size_t num_bits);

/* src/encoding/bitpacked/encode.rs:11-36: `packed` must be zeroed and hold
 * ceil(len/BLOCK) blocks of ceil8(BLOCK*num_bits) bytes. */
void pq2o_bitpacked_encode_u32(const uint32_t *unpacked, size_t len, size_t num_bits, uint8_t *packed,
                               size_t packed_len);

/* src/encoding/bitpacked/decode.rs:9-86, T = u32 */
typedef struct {
    const uint8_t *packed; /* unread chunks */
    size_t packed_len;
    size_t block_size; /* 4 * num_bits */
    size_t num_bits;
    size_t remaining;
    size_t current_pack_index;
    uint32_t unpacked[32];
} pq2o_bitpacked_u32;
int32_t pq2o_bitpacked_u32_try_new(pq2o_bitpacked_u32 *d, const uint8_t *packed, size_t packed_len,
                                   size_t num_bits, size_t length);
/* returns 1 and sets *out, or 0 for None */
int pq2o_bitpacked_u32_next(pq2o_bitpacked_u32 *d, uint32_t *out);

/* same, T = u64 (decode.rs:9-86 with Unpackable for u64, mod.rs:153-166) */
typedef struct {
    const uint8_t *packed;
    size_t packed_len;
    size_t block_size; /* 8 * num_bits */
    size_t num_bits;
    size_t remaining;
    size_t current_pack_index;
    uint64_t unpacked[64];
} pq2o_bitpacked_u64;
int32_t pq2o_bitpacked_u64_try_new(pq2o_bitpacked_u64 *d, const uint8_t *packed, size_t packed_len,
                                   size_t num_bits, size_t length);
int pq2o_bitpacked_u64_next(pq2o_bitpacked_u64 *d, uint64_t *out);

/* Convenience: try_new + drain into out[length]. */
int32_t pq2o_bitunpack_u32(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                           uint32_t *out);
int32_t pq2o_bitunpack_u64(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                           uint64_t *out);

/* src/encoding/hybrid_rle/mod.rs:15-21 */
#define PQ2O_RUN_BITPACKED 0
#define PQ2O_RUN_RLE 1
typedef struct {
    int32_t kind;
    const uint8_t *data; /* slice start */
    size_t data_len;     /* slice length in bytes */
    size_t run_length;   /* Rle only */
} pq2o_hybrid_encoded;

/* src/encoding/hybrid_rle/decoder.rs:8-65 */
typedef struct {
    const uint8_t *values;
    size_t len;
    size_t num_bits;
} pq2o_run_decoder;
void pq2o_run_decoder_new(pq2o_run_decoder *d, const uint8_t *values, size_t len, size_t num_bits);
/* returns 1 = Some(Ok(run)), 0 = None, <0 = -(status) for a panic path */
int pq2o_run_decoder_next(pq2o_run_decoder *d, pq2o_hybrid_encoded *run);

/* src/encoding/hybrid_rle/mod.rs:23-116 */
typedef struct {
    pq2o_run_decoder decoder;
    int32_t state; /* 0 None, 1 Bitpacked, 2 Rle, 3 Single */
    pq2o_bitpacked_u32 bitpacked;
    uint32_t rle_value;
    size_t rle_left;
    int single_some;
    size_t remaining;
} pq2o_hybrid_rle_decoder;
int32_t pq2o_hybrid_rle_try_new(pq2o_hybrid_rle_decoder *d, const uint8_t *data, size_t len, uint32_t num_bits,
                                size_t num_values);
/* returns 1 = Some(Ok(v)), 0 = None, <0 = -(status) */
int pq2o_hybrid_rle_next(pq2o_hybrid_rle_decoder *d, uint32_t *out);
/* try_new + collect::<Result<Vec<_>,_>>() into out[num_values] */
int32_t pq2o_hybrid_rle_decode_u32(const uint8_t *data, size_t len, uint32_t num_bits, size_t num_values,
                                   uint32_t *out);

/* src/encoding/hybrid_rle/bitmap.rs:14-68 */
typedef struct {
    const uint8_t *iter;
    const uint8_t *end;
    uint8_t current_byte;
    size_t remaining;
    uint8_t mask;
} pq2o_bitmap_iter;
int32_t pq2o_bitmap_iter_new(pq2o_bitmap_iter *it, const uint8_t *slice, size_t slice_len, size_t offset,
                             size_t len);
int pq2o_bitmap_iter_next(pq2o_bitmap_iter *it, uint8_t *out);

/* src/encoding/hybrid_rle/bitmap.rs:71-102 ; returns bytes written */
size_t pq2o_bitmap_encode_bool(const uint8_t *bools, size_t length, uint8_t *out);
/* src/encoding/hybrid_rle/encoder.rs:9-93 ; return bytes written.  `out`
 * must hold 10 + ceil8(length*num_bits) + 128 bytes. */
size_t pq2o_hybrid_rle_encode_u32(const uint32_t *values, size_t length, uint32_t num_bits, uint8_t *out);
size_t pq2o_hybrid_rle_encode_bool(const uint8_t *bools, size_t length, uint8_t *out);

/* src/page/mod.rs:350-411 ; offsets are relative to `buffer` */
typedef struct {
    size_t rep_off, rep_len, def_off, def_len, val_off, val_len;
} pq2o_split;
int32_t pq2o_split_buffer_v1(const uint8_t *buffer, size_t len, int has_rep, int has_def, pq2o_split *out);
int32_t pq2o_split_buffer_v2(size_t len, size_t rep_len, size_t def_len, pq2o_split *out);

/* src/deserialize/utils.rs:13-27 */
int32_t pq2o_dict_indices_locate(const uint8_t *values, size_t len, uint32_t *bit_width, size_t *stream_off);

/* src/deserialize/hybrid_rle.rs:5-123 (HybridRleIter over hybrid_rle::Decoder) */
#define PQ2O_BITMAP 0
#define PQ2O_REPEATED 1
typedef struct {
    int32_t kind;
    uint64_t offset; /* Bitmap: byte offset of the pack inside the stream */
    uint64_t bytes;  /* Bitmap: pack length in bytes */
    uint64_t length; /* number of items */
    uint32_t is_set; /* Repeated */
} pq2o_bitmap_run;
/* Collects HybridRleIter::new(Decoder::new(data, 1), num_values).  Returns
 * status; *n_runs is the number of items the iterator yields (runs beyond
 * `cap` are counted but not stored). */
int32_t pq2o_hybrid_rle_bitmap_runs(const uint8_t *data, size_t len, size_t num_values, pq2o_bitmap_run *runs,
                                    size_t cap, size_t *n_runs);
/* src/deserialize/hybrid_rle.rs:125-205 (HybridRleBooleanIter, including its
 * first-run-only behaviour).  Writes at most cap bools, returns count via n. */
int32_t pq2o_hybrid_rle_boolean_iter(const uint8_t *data, size_t len, size_t num_values, uint8_t *out, size_t cap,
                                     size_t *n);

/* src/indexes/intervals.rs:11-16 */
typedef struct {
    uint64_t start, length;
} pq2o_interval;
/* src/deserialize/filtered_rle.rs:8-274 */
#define PQ2O_F_BITMAP 0
#define PQ2O_F_REPEATED 1
#define PQ2O_F_SKIPPED 2
typedef struct {
    int32_t kind;
    uint64_t values_off; /* Bitmap: byte offset of the pack in the stream */
    uint64_t values_len;
    uint64_t offset; /* Bitmap: bit offset */
    uint64_t length;
    uint32_t is_set;    /* Repeated */
    uint64_t set_count; /* Skipped */
} pq2o_filtered_run;
int32_t pq2o_filtered_bitmap_runs(const pq2o_bitmap_run *runs, size_t n_runs, const uint8_t *stream,
                                  const pq2o_interval *intervals, size_t n_intervals, pq2o_filtered_run *out,
                                  size_t cap, size_t *n_out);

/* src/encoding/zigzag_leb128.rs:5-8 */
int32_t pq2o_zigzag_leb128_decode(const uint8_t *values, size_t len, int64_t *value, size_t *consumed);

/* delta_bitpacked::Decoder::try_new(values)?.collect::<Result<Vec<i64>>>() and consumed_bytes()
 * (src/encoding/delta_bitpacked/decoder.rs:26-244).  Writes at most `cap` values, reports the header's
 * total count in *n_values (PQ2O_INVALID_PARAMETER, with nothing else decided, if it exceeds cap).
 * PQ2O_OUT_OF_SPEC where the reference returns Error::OutOfSpec (:40-44, :75-79),
 * PQ2O_REFERENCE_WOULD_PANIC where it panics (the asserts :140, :156, a division by zero :155, slices
 * beyond the data, a 10th varint byte above 1, a mini-block wider than 64 bits).  Sums wrap (the
 * release build; the debug build panics on overflow, :235). */
int32_t pq2o_delta_bitpacked_decode(const uint8_t *values, size_t len, int64_t *out, size_t cap, size_t *n_values,
                                    size_t *consumed_bytes);

/* Timing helpers for bench.py's cpu_baseline leg: decode `reps` times and
 * return elapsed seconds (CLOCK_MONOTONIC). */
double pq2o_time_hybrid_rle_decode(const uint8_t *data, size_t len, uint32_t num_bits, size_t num_values,
                                   uint32_t *out, int reps);
double pq2o_time_bitunpack_u32(const uint8_t *packed, size_t packed_len, size_t num_bits, size_t length,
                               uint32_t *out, int reps);

/* src/encoding/zigzag_leb128.rs:10-15 */
size_t pq2o_zigzag_leb128_encode(int64_t value, uint8_t container[10]);
/* delta_bitpacked::encode (src/encoding/delta_bitpacked/encoder.rs:11-78): returns the bytes written; `buffer`
 * must hold 32 + ceil(len / 128) * 1035 bytes. */
size_t pq2o_delta_bitpacked_encode(const int64_t *iterator, size_t len, uint8_t *buffer);

#ifdef __cplusplus
}
#endif
#endif
