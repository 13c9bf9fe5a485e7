// Synthetic hybrid-RLE page generator (host only); see include/pq2synth.h.
#include "../../include/pq2synth.h"

#include <cmath>
#include <cstring>

namespace {

struct Rng {  // xoshiro256** seeded through splitmix64
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        for (auto &x : s) {
            seed += 0x9e3779b97f4a7c15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            x = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
    uint64_t geometric(double mean) {  // >= 1, mean `mean`
        if (mean <= 1.0) return 1;
        const double p = 1.0 / mean;
        const double u = 1.0 - unit();  // (0, 1]
        return 1 + (uint64_t)std::floor(std::log(u) / std::log1p(-p));
    }
    uint32_t bits(uint32_t w) { return w == 0 ? 0u : (uint32_t)(next() >> (64 - w)); }
};

struct Writer {
    uint8_t *out;
    uint64_t cap, pos = 0;
    bool ok = true;
    void byte(uint8_t b) {
        if (pos < cap)
            out[pos++] = b;
        else
            ok = false;
    }
    void varint(uint64_t v) {
        do {
            uint8_t b = v & 0x7f;
            v >>= 7;
            if (v) b |= 0x80;
            byte(b);
        } while (v);
    }
    // LSB-first bit writer for bit-packed groups
    uint64_t acc = 0;
    uint32_t nacc = 0;
    void put(uint32_t v, uint32_t w) {
        acc |= (uint64_t)v << nacc;
        nacc += w;
        while (nacc >= 8) {
            byte((uint8_t)acc);
            acc >>= 8;
            nacc -= 8;
        }
    }
};

}  // namespace

extern "C" {

uint64_t pq2s_encode_bound(uint64_t n, uint32_t num_bits) {
    // worst case: every group of 8 carries its own header byte
    return 16 + ((n + 7) / 8) * (num_bits + 2) + 16;
}

uint64_t pq2s_encode_hybrid(const uint32_t *values, uint64_t n, uint32_t num_bits, uint32_t max_groups, uint8_t *out,
                            uint64_t cap) {
    Writer w{out, cap};
    if (max_groups == 0) max_groups = 63;
    const uint32_t mask = num_bits >= 32 ? 0xffffffffu : ((1u << num_bits) - 1u);
    const uint32_t value_bytes = (num_bits + 7) / 8;
    uint64_t i = 0;
    while (i < n) {
        // length of the repeat starting here
        uint64_t r = 1;
        while (i + r < n && values[i + r] == values[i]) ++r;
        if (r >= 8) {
            w.varint(r << 1);
            for (uint32_t b = 0; b < value_bytes; ++b) w.byte((uint8_t)((values[i] & mask) >> (8 * b)));
            i += r;
            continue;
        }
        // bit-packed run: groups of 8 until a long repeat starts at a group boundary
        uint64_t j = i;
        uint32_t groups = 0;
        while (j < n && groups < max_groups) {
            uint64_t rr = 1;
            while (rr < 8 && j + rr < n && values[j + rr] == values[j]) ++rr;
            if (rr >= 8 && groups > 0) break;
            if (rr >= 8 && groups == 0) break;  // cannot happen: handled above
            j += 8;
            ++groups;
        }
        w.varint(((uint64_t)groups << 1) | 1);
        for (uint64_t k = i; k < i + 8ull * groups; ++k) w.put(k < n ? (values[k] & mask) : 0u, num_bits);
        i += 8ull * groups;
    }
    return w.ok ? w.pos : 0;
}

void pq2s_gen_dict_values(uint64_t seed, uint64_t n, uint32_t num_bits, double mean_repeat, double mean_distinct,
                          uint32_t *values) {
    Rng rng(seed);
    uint64_t i = 0;
    while (i < n) {
        uint64_t len = rng.geometric(mean_repeat);
        const uint32_t v = rng.bits(num_bits);
        for (; len > 0 && i < n; --len) values[i++] = v;
        len = (rng.geometric(mean_distinct) + 7) & ~7ull;
        uint32_t prev = v;
        for (; len > 0 && i < n; --len) {
            uint32_t x = rng.bits(num_bits);
            if (x == prev && num_bits > 0) x ^= 1u;  // keep the segment free of accidental repeats
            values[i++] = prev = x;
        }
    }
}

void pq2s_gen_def_values(uint64_t seed, uint64_t n, double mean_valid, double mean_null, uint32_t *values) {
    Rng rng(seed);
    uint64_t i = 0;
    while (i < n) {
        uint64_t len = rng.geometric(mean_valid);
        for (; len > 0 && i < n; --len) values[i++] = 1;
        len = rng.geometric(mean_null);
        for (; len > 0 && i < n; --len) values[i++] = 0;
    }
}

void pq2s_gen_uniform_values(uint64_t seed, uint64_t n, uint32_t num_bits, uint32_t *values) {
    Rng rng(seed);
    for (uint64_t i = 0; i < n; ++i) values[i] = rng.bits(num_bits);
}

uint64_t pq2s_gen_adversarial(uint64_t seed, uint64_t n, uint32_t num_bits, uint8_t *out, uint64_t cap,
                              uint32_t *expected) {
    Rng rng(seed);
    Writer w{out, cap};
    const uint32_t value_bytes = (num_bits + 7) / 8;
    uint64_t i = 0;
    while (i < n) {
        // RLE run of 1..8 values whose value bytes are fully random
        const uint64_t run = 1 + (rng.next() & 7);
        uint32_t v = 0;
        w.varint(run << 1);
        for (uint32_t b = 0; b < value_bytes; ++b) {
            const uint8_t byte = (uint8_t)rng.next();
            w.byte(byte);
            if (b < 4) v |= (uint32_t)byte << (8 * b);
        }
        for (uint64_t k = 0; k < run && i < n; ++k) expected[i++] = v;
        if (i >= n) break;
        // one bit-packed group
        w.varint((1u << 1) | 1);
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = rng.bits(num_bits);
            w.put(x, num_bits);
            if (i < n) expected[i++] = x;
        }
    }
    return w.ok ? w.pos : 0;
}

}  // extern "C"
