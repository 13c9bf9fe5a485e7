// Host-side page glue of libpq2hip: the small, sequential pieces of the
// reference that sit in front of the decoder (slicing a page buffer, picking
// the bit width, run-header varints).  No device work happens here.
#include <cstdint>
#include <cstring>

#include "../../include/pq2hip.h"

extern "C" {

// src/encoding/mod.rs:23-27
uint64_t pq2_ceil8(uint64_t value) { return value / 8 + (value % 8 != 0 ? 1 : 0); }

// src/encoding/mod.rs:14-21: little-endian u32 length prefix, None when fewer than 4 bytes
int32_t pq2_get_length(const uint8_t *values, uint64_t len, uint64_t *out) {
    if (len < 4 || !values) return PQ2_OUT_OF_SPEC;
    uint32_t v;
    std::memcpy(&v, values, 4);  // little-endian host
    *out = v;
    return PQ2_OK;
}

// src/read/levels.rs:3-5: bits needed for levels 0..=max_level
uint32_t pq2_get_bit_width(int16_t max_level) {
    const uint16_t v = static_cast<uint16_t>(max_level);
    return v == 0 ? 0u : 32u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(v)));
}

// src/encoding/uleb128.rs:3-23
int32_t pq2_uleb128_decode(const uint8_t *values, uint64_t len, uint64_t *value, uint64_t *consumed) {
    uint64_t result = 0, used = 0;
    for (unsigned shift = 0; used < len; shift += 7) {
        const uint8_t byte = values[used++];
        if (shift == 63 && byte > 1) return PQ2_REFERENCE_WOULD_PANIC;
        result |= static_cast<uint64_t>(byte & 0x7f) << shift;
        if (!(byte & 0x80)) break;
    }
    *value = result;
    *consumed = used;
    return PQ2_OK;
}

// src/encoding/uleb128.rs:30-46
uint64_t pq2_uleb128_encode(uint64_t value, uint8_t container[10]) {
    uint64_t n = 0;
    do {
        uint8_t byte = static_cast<uint8_t>(value & 0x7f);
        value >>= 7;
        if (value) byte |= 0x80;
        container[n++] = byte;
    } while (value);
    return n;
}

// src/page/mod.rs:350-398
int32_t pq2_split_buffer_v1(const uint8_t *buffer, uint64_t len, int32_t has_rep, int32_t has_def, pq2_split *out) {
    uint64_t cursor = 0;
    uint64_t off[2] = {0, 0}, lens[2] = {0, 0};
    const int32_t present[2] = {has_rep, has_def};
    for (int level = 0; level < 2; ++level) {
        off[level] = cursor;
        if (!present[level]) continue;
        uint64_t l;
        if (pq2_get_length(buffer + cursor, len - cursor, &l) != PQ2_OK) return PQ2_OUT_OF_SPEC;
        if (l > len - cursor - 4) return PQ2_OUT_OF_SPEC;
        off[level] = cursor + 4;
        lens[level] = l;
        cursor += 4 + l;
    }
    out->rep_off = off[0];
    out->rep_len = lens[0];
    out->def_off = off[1];
    out->def_len = lens[1];
    out->val_off = cursor;
    out->val_len = len - cursor;
    return PQ2_OK;
}

// src/page/mod.rs:400-411 (the reference slices without checking and panics)
int32_t pq2_split_buffer_v2(uint64_t len, uint64_t rep_len, uint64_t def_len, pq2_split *out) {
    if (rep_len > len || def_len > len - rep_len) return PQ2_REFERENCE_WOULD_PANIC;
    out->rep_off = 0;
    out->rep_len = rep_len;
    out->def_off = rep_len;
    out->def_len = def_len;
    out->val_off = rep_len + def_len;
    out->val_len = len - rep_len - def_len;
    return PQ2_OK;
}

// src/deserialize/utils.rs:13-27
int32_t pq2_dict_indices_locate(const uint8_t *values, uint64_t len, uint32_t *bit_width, uint64_t *stream_off) {
    if (len == 0) return PQ2_REFERENCE_WOULD_PANIC;  // indices_buffer[0] on an empty buffer
    if (values[0] > 32) return PQ2_OUT_OF_SPEC;
    *bit_width = values[0];
    *stream_off = 1;
    return PQ2_OK;
}

// hybrid_rle::Decoder collected: src/encoding/hybrid_rle/decoder.rs:30-64
int32_t pq2_hybrid_rle_runs(const uint8_t *data, uint64_t len, uint32_t num_bits, pq2_run *runs, uint64_t cap,
                            uint64_t *n_runs) {
    uint64_t n = 0, pos = 0;
    *n_runs = 0;
    if (num_bits == 0) return PQ2_OK;
    while (pos < len) {
        uint64_t indicator, used;
        const int32_t st = pq2_uleb128_decode(data + pos, len - pos, &indicator, &used);
        if (st != PQ2_OK) return st;
        pos += used;
        if (pos == len) break;  // a header that ends the stream is dropped
        pq2_run r;
        r.reserved = 0;
        r.offset = pos;
        const uint64_t count = indicator >> 1;
        if (indicator & 1) {
            if (count > UINT64_MAX / num_bits) return PQ2_REFERENCE_WOULD_PANIC;
            r.kind = PQ2_RUN_BITPACKED;
            r.len = count * num_bits < len - pos ? count * num_bits : len - pos;
            r.run_length = 0;
        } else {
            r.kind = PQ2_RUN_RLE;
            r.len = pq2_ceil8(num_bits);
            r.run_length = count;
            if (r.len > len - pos) return PQ2_REFERENCE_WOULD_PANIC;  // split_at past the end
        }
        pos += r.len;
        if (n < cap) runs[n] = r;
        *n_runs = ++n;
    }
    return PQ2_OK;
}

// ---- write side ----------------------------------------------------------------------

namespace {

// One 32-value block at `width` bits into `width` little-endian dwords (OR-ed
// in).  A field that crosses a dword boundary is split exactly as
// src/encoding/bitpacked/pack.rs:27-38 does: the low part is not masked, the
// high part is masked to `width` bits; a field inside one dword is masked first.
void pack_block_u32(const uint32_t *in, uint32_t width, uint8_t *out) {
    if (width == 0) return;
    const uint32_t mask = width == 32 ? 0xffffffffu : (1u << width) - 1u;
    uint32_t words[32] = {0};
    for (uint32_t i = 0; i < 32; ++i) {
        const uint32_t first = i * width, last = first + width;
        const uint32_t w0 = first / 32, w1 = last / 32, lo = first % 32, hi = last % 32;
        if (w0 != w1 && hi != 0) {
            words[w0] |= in[i] << lo;
            words[w1] |= (in[i] >> (width - hi)) & mask;
        } else {
            words[w0] |= (in[i] & mask) << lo;
        }
    }
    for (uint32_t k = 0; k < width; ++k) {
        uint8_t le[4];
        std::memcpy(le, &words[k], 4);
        for (int b = 0; b < 4; ++b) out[4 * k + b] |= le[b];
    }
}

}  // namespace

int32_t pq2_bitpacked_encode_u32(const uint32_t *unpacked, uint64_t len, uint32_t num_bits, uint8_t *packed,
                                 uint64_t packed_len) {
    if (num_bits > 32) return PQ2_REFERENCE_WOULD_PANIC;
    const uint64_t block_bytes = 4ull * num_bits;
    const uint64_t blocks = (len + 31) / 32;
    if (packed_len < blocks * block_bytes) return PQ2_INVALID_PARAMETER;
    for (uint64_t b = 0; b < blocks; ++b) {
        uint32_t block[32] = {0};
        const uint64_t n = len - 32 * b < 32 ? len - 32 * b : 32;
        std::memcpy(block, unpacked + 32 * b, n * sizeof(uint32_t));
        pack_block_u32(block, num_bits, packed + b * block_bytes);
    }
    return PQ2_OK;
}

int32_t pq2_hybrid_rle_encode_u32(const uint32_t *values, uint64_t len, uint32_t num_bits, uint8_t *out,
                                  uint64_t out_cap, uint64_t *written) {
    const uint32_t width = num_bits & 0xffu;  // `num_bits as u8`, encoder.rs:14
    if (width > 32) return PQ2_REFERENCE_WOULD_PANIC;
    uint8_t header[10];
    const uint64_t used = pq2_uleb128_encode((pq2_ceil8(len) << 1) | 1, header);
    const uint64_t full = len / 32, rest = len % 32;
    const uint64_t need = used + full * 4 * width + pq2_ceil8(rest * width);
    if (out_cap < need) return PQ2_INVALID_PARAMETER;
    std::memcpy(out, header, used);
    uint8_t *dst = out + used;
    // the reference reuses one 32-slot buffer, so the slots past a short last
    // block still hold the previous block's values (encoder.rs:43, :61-65)
    uint32_t block[32] = {0};
    for (uint64_t b = 0; b <= full; ++b) {
        const uint64_t n = b < full ? 32 : rest;
        if (n == 0) break;
        std::memcpy(block, values + 32 * b, n * sizeof(uint32_t));
        uint8_t packed[128] = {0};
        pack_block_u32(block, width, packed);
        const uint64_t bytes = b < full ? 4ull * width : pq2_ceil8(rest * width);
        std::memcpy(dst, packed, bytes);
        dst += bytes;
    }
    *written = static_cast<uint64_t>(dst - out);
    return PQ2_OK;
}

int32_t pq2_bitmap_encode_bool(const uint8_t *bools, uint64_t len, uint8_t *out, uint64_t out_cap,
                               uint64_t *written) {
    const uint64_t bytes = pq2_ceil8(len);
    if (out_cap < bytes) return PQ2_INVALID_PARAMETER;
    std::memset(out, 0, bytes);
    for (uint64_t i = 0; i < len; ++i)
        if (bools[i]) out[i >> 3] |= static_cast<uint8_t>(1u << (i & 7));
    *written = bytes;
    return PQ2_OK;
}

int32_t pq2_hybrid_rle_encode_bool(const uint8_t *bools, uint64_t len, uint8_t *out, uint64_t out_cap,
                                   uint64_t *written) {
    uint8_t header[10];
    const uint64_t used = pq2_uleb128_encode((pq2_ceil8(len) << 1) | 1, header);
    if (out_cap < used) return PQ2_INVALID_PARAMETER;
    std::memcpy(out, header, used);
    uint64_t body = 0;
    const int32_t st = pq2_bitmap_encode_bool(bools, len, out + used, out_cap - used, &body);
    if (st != PQ2_OK) return st;
    *written = used + body;
    return PQ2_OK;
}

}  // extern "C"

// ---- DELTA_BINARY_PACKED, write side ------------------------------------------------------

namespace {

// 64 values at `width` bits OR-ed into `width` little-endian qwords at `out`, with the split rule of
// src/encoding/bitpacked/pack.rs:27-46 (see pack_block_u32).
void pack_block_u64(const uint64_t *in, uint32_t width, uint8_t *out) {
    if (width == 0) return;
    const uint64_t mask = width == 64 ? ~0ull : (1ull << width) - 1ull;
    uint64_t words[64] = {0};
    for (uint32_t i = 0; i < 64; ++i) {
        const uint32_t first = i * width, last = first + width;
        const uint32_t w0 = first / 64, w1 = last / 64, lo = first % 64, hi = last % 64;
        if (w0 != w1 && hi != 0) {
            words[w0] |= in[i] << lo;
            words[w1] |= (in[i] >> (width - hi)) & mask;
        } else {
            words[w0] |= (in[i] & mask) << lo;
        }
    }
    for (uint32_t k = 0; k < width; ++k) {
        uint8_t le[8];
        std::memcpy(le, &words[k], 8);
        for (int b = 0; b < 8; ++b) out[8 * k + b] |= le[b];
    }
}

}  // namespace

extern "C" {

int32_t pq2_zigzag_leb128_decode(const uint8_t *values, uint64_t len, int64_t *value, uint64_t *consumed) {
    uint64_t u = 0;
    const int32_t st = pq2_uleb128_decode(values, len, &u, consumed);
    if (st != PQ2_OK) return st;
    *value = static_cast<int64_t>(u >> 1) ^ -static_cast<int64_t>(u & 1);
    return PQ2_OK;
}

uint64_t pq2_zigzag_leb128_encode(int64_t value, uint8_t container[10]) {
    return pq2_uleb128_encode((static_cast<uint64_t>(value) << 1) ^ static_cast<uint64_t>(value >> 63), container);
}

uint64_t pq2_delta_bitpacked_encode_bound(uint64_t len) { return 32 + ((len + 127) / 128) * (10 + 1 + 16 * 64); }

int32_t pq2_delta_bitpacked_encode(const int64_t *values, uint64_t len, uint8_t *out, uint64_t out_cap,
                                   uint64_t *written) {
    if (out_cap < pq2_delta_bitpacked_encode_bound(len)) return PQ2_INVALID_PARAMETER;
    uint8_t *dst = out;
    dst += pq2_uleb128_encode(128, dst);  // block size
    dst += pq2_uleb128_encode(1, dst);    // one mini-block per block
    dst += pq2_uleb128_encode(len, dst);
    const int64_t first = len ? values[0] : 0;
    dst += pq2_zigzag_leb128_encode(first, dst);
    // One array of 128 slots serves every block, as in the reference (encoder.rs:27): the slots behind a short
    // last block still hold the block before, and all 128 are packed.
    uint64_t rel[128] = {0};
    uint64_t prev = static_cast<uint64_t>(first);
    for (uint64_t at = 1; at < len; at += 128) {
        const uint64_t n = len - at < 128 ? len - at : 128;
        int64_t delta[128];
        int64_t low = INT64_MAX, high = INT64_MIN;
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t v = static_cast<uint64_t>(values[at + i]);
            delta[i] = static_cast<int64_t>(v - prev);  // wraps, as the reference's release build
            prev = v;
            low = delta[i] < low ? delta[i] : low;
            high = delta[i] > high ? delta[i] : high;
        }
        const uint64_t range = static_cast<uint64_t>(high) - static_cast<uint64_t>(low);
        const uint32_t width = range ? 64u - static_cast<uint32_t>(__builtin_clzll(range)) : 0u;
        for (uint64_t i = 0; i < n; ++i) rel[i] = static_cast<uint64_t>(delta[i]) - static_cast<uint64_t>(low);
        dst += pq2_zigzag_leb128_encode(low, dst);
        *dst++ = static_cast<uint8_t>(width);
        std::memset(dst, 0, 16ull * width);
        pack_block_u64(rel, width, dst);
        pack_block_u64(rel + 64, width, dst + 8ull * width);
        dst += 16ull * width;
    }
    *written = static_cast<uint64_t>(dst - out);
    return PQ2_OK;
}

}  // extern "C"
