/*
 * pq2synth -- synthetic hybrid-RLE pages for tests and benchmarks (host only).
 *
 * The reference's own encoder only ever emits one bit-packed run
 * (src/encoding/hybrid_rle/encoder.rs:8), so streams with genuine RLE runs,
 * as other Parquet writers produce them, need an encoder of their own.  All
 * generators are deterministic: a splitmix64-seeded xoshiro256** drives them,
 * so every machine generates identical bytes for a given seed.
 */
#ifndef PQ2SYNTH_H
#define PQ2SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Upper bound of the bytes pq2s_encode_hybrid can write for n values. */
uint64_t pq2s_encode_bound(uint64_t n, uint32_t num_bits);

/* RLE / bit-packed hybrid encoder (Parquet Encodings.md, "RLE"): a value
 * repeated >= 8 times from a group boundary becomes an RLE run, everything
 * else goes to bit-packed groups of 8 values with at most `max_groups` groups
 * per header (63 keeps headers to one byte, as parquet-mr and pyarrow do).
 * Returns the number of bytes written (0 if `cap` is too small). */
uint64_t pq2s_encode_hybrid(const uint32_t *values, uint64_t n, uint32_t num_bits, uint32_t max_groups, uint8_t *out,
                            uint64_t cap);

/* Dictionary-index-shaped values: segments of one repeated value (length
 * geometric with mean `mean_repeat`) alternate with segments of distinct
 * uniform values (length geometric with mean `mean_distinct`, rounded up to a
 * multiple of 8); all values < 2^num_bits. */
void pq2s_gen_dict_values(uint64_t seed, uint64_t n, uint32_t num_bits, double mean_repeat, double mean_distinct,
                          uint32_t *values);

/* Definition-level-shaped values (0/1): runs of 1 (geometric, mean
 * `mean_valid`) alternate with runs of 0 (geometric, mean `mean_null`). */
void pq2s_gen_def_values(uint64_t seed, uint64_t n, double mean_valid, double mean_null, uint32_t *values);

/* Uniform values < 2^num_bits. */
void pq2s_gen_uniform_values(uint64_t seed, uint64_t n, uint32_t num_bits, uint32_t *values);

/* Adversarial stream: strictly alternating RLE runs of 1..8 values (value
 * bytes random, i.e. with bits above num_bits set) and bit-packed headers with
 * a single group.  Writes the stream to `out` and the values a spec decoder
 * that does not mask RLE values yields to `expected` (n of them).  Returns the
 * stream length (0 if `cap` is too small). */
uint64_t pq2s_gen_adversarial(uint64_t seed, uint64_t n, uint32_t num_bits, uint8_t *out, uint64_t cap,
                              uint32_t *expected);

#ifdef __cplusplus
}
#endif
#endif
