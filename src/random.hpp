// Drop-in for KowerKoint/TFHE's src/random.hpp: the sources of randomness used by key
// generation and encryption (never by evaluation).
//
// The reference feeds every distribution straight from std::random_device and cannot be
// seeded (src/random.hpp:10,35-41).  Here one std::mt19937_64 is seeded once, from
// std::random_device by default or from an explicit seed, so that key material can be
// reproduced; the distributions are the same.
#pragma once

#include <algorithm>
#include <cstdint>
#include <random>

#include "types.hpp"

namespace TFHE {
class Random {
    std::mt19937_64 engine_;
    std::normal_distribution<> normal_{};
    std::uniform_real_distribution<> unit_{};

    static uint64_t entropy() {
        std::random_device rd;
        return (static_cast<uint64_t>(rd()) << 32) ^ rd();
    }

    template <int N>
    void fill_bits(bool* out) {
        for (int base = 0; base < N; base += 64) {
            const uint64_t word = engine_();
            for (int i = base; i < N && i < base + 64; i++) out[i] = (word >> (i - base)) & 1U;
        }
    }

public:
    Random() : engine_(entropy()) {}
    explicit Random(uint64_t seed) : engine_(seed) {}
    void seed(uint64_t s) { engine_.seed(s); normal_.reset(); }

    double normal() { return normal_(engine_); }              // N(0, 1)
    double uniform_real() { return unit_(engine_); }          // [0, 1)
    TorusValue uniform_torus() { return TorusValue::from_raw_value(static_cast<uint32_t>(engine_() >> 32)); }

    template <int N>
    Polynomial<bool, N> bit_polynomial() {
        Polynomial<bool, N> out;
        fill_bits<N>(out.begin());
        return out;
    }
    template <int N>
    Vector<bool, N> bit_vector() {
        Vector<bool, N> out;
        fill_bits<N>(out.begin());
        return out;
    }
};
}  // namespace TFHE
