// Drop-in for KowerKoint/TFHE's src/random.hpp: the sources of randomness used by key
// generation and encryption (never by evaluation).
//
// The reference feeds every distribution straight from std::random_device and cannot be seeded
// (src/random.hpp:10,35-41).  Here the words come from ChaCha20 (the RFC 8439 block function) keyed
// with 256 bits: drawn from std::random_device by default, which keeps the reference's security
// (key bits, masks and noise are unpredictable, and observing masks says nothing about the secret
// key, which comes from a generator with a key of its own); or expanded from an explicit 64-bit
// seed, which makes key material reproducible for tests and is NOT for real keys (64 bits of
// entropy).  The distributions are the reference's.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <limits>
#include <random>

#include "types.hpp"

namespace TFHE {

// ChaCha20 as a UniformRandomBitGenerator of 64-bit words.
class ChaCha20Engine {
    std::array<uint32_t, 8> key_{};
    uint32_t stream_ = 0;
    uint64_t block_ = 0;
    std::array<uint32_t, 16> buf_{};
    int next_ = 16;  // next unread word of buf_

    static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    static void quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
        a += b; d ^= a; d = rotl(d, 16);
        c += d; b ^= c; b = rotl(b, 12);
        a += b; d ^= a; d = rotl(d, 8);
        c += d; b ^= c; b = rotl(b, 7);
    }
    void refill() {
        const std::array<uint32_t, 16> in = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                                             key_[0], key_[1], key_[2], key_[3], key_[4], key_[5], key_[6], key_[7],
                                             static_cast<uint32_t>(block_), static_cast<uint32_t>(block_ >> 32), stream_, 0u};
        std::array<uint32_t, 16> x = in;
        for (int r = 0; r < 10; r++) {
            quarter(x[0], x[4], x[8], x[12]);
            quarter(x[1], x[5], x[9], x[13]);
            quarter(x[2], x[6], x[10], x[14]);
            quarter(x[3], x[7], x[11], x[15]);
            quarter(x[0], x[5], x[10], x[15]);
            quarter(x[1], x[6], x[11], x[12]);
            quarter(x[2], x[7], x[8], x[13]);
            quarter(x[3], x[4], x[9], x[14]);
        }
        for (int i = 0; i < 16; i++) buf_[i] = x[i] + in[i];
        block_++;
        next_ = 0;
    }

public:
    using result_type = uint64_t;
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return std::numeric_limits<uint64_t>::max(); }

    ChaCha20Engine() = default;
    // `stream` separates generators that share a key
    ChaCha20Engine(const std::array<uint32_t, 8>& key, uint32_t stream) : key_(key), stream_(stream) {}

    result_type operator()() {
        if (next_ >= 16) refill();
        const uint64_t lo = buf_[next_], hi = buf_[next_ + 1];
        next_ += 2;
        return lo | (hi << 32);
    }
};

class Random {
    ChaCha20Engine secret_;  // secret-key bits
    ChaCha20Engine public_;  // masks (published in every ciphertext) and noise
    std::normal_distribution<> normal_{};
    std::uniform_real_distribution<> unit_{};

    static std::array<uint32_t, 8> os_key() {
        std::random_device rd;
        std::array<uint32_t, 8> k;
        for (uint32_t& w : k) w = rd();
        return k;
    }
    static std::array<uint32_t, 8> test_key(uint64_t seed) {
        return {static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), 0x54464845u, 0x74657374u, 0, 0, 0, 0};
    }

    template <int N>
    void fill_bits(bool* out) {
        for (int base = 0; base < N; base += 64) {
            const uint64_t word = secret_();
            for (int i = base; i < N && i < base + 64; i++) out[i] = (word >> (i - base)) & 1U;
        }
    }

public:
    // two independent 256-bit keys from the operating system
    Random() : secret_(os_key(), 0), public_(os_key(), 1) {}
    // reproducible, for tests only: everything follows from 64 bits
    explicit Random(uint64_t seed) : secret_(test_key(seed), 0), public_(test_key(seed), 1) {}
    void seed(uint64_t s) {
        secret_ = ChaCha20Engine(test_key(s), 0);
        public_ = ChaCha20Engine(test_key(s), 1);
        normal_.reset();
    }

    double normal() { return normal_(public_); }              // N(0, 1)
    double uniform_real() { return unit_(public_); }          // [0, 1)
    TorusValue uniform_torus() { return TorusValue::from_raw_value(static_cast<uint32_t>(public_() >> 32)); }

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
