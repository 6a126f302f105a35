// Drop-in for KowerKoint/TFHE's src/tlwe.hpp: scalar LWE encryption over the torus,
// ciphertext layout [b, a_0 .. a_{N-1}].  Client side only; runs on the host.
#pragma once

#include <array>
#include <cstdint>
#include <utility>

#include "random.hpp"
#include "types.hpp"

namespace TFHE {

struct TLWEParameter128BitSecurity {
    constexpr static int N = 636;
    constexpr static double ALPHA = 0.0000925119974676756;
};

template <typename TLWEParameter = TLWEParameter128BitSecurity>
class TLWE {
public:
    constexpr static int N = TLWEParameter::N;
    constexpr static double ALPHA = TLWEParameter::ALPHA;

    using Sample = Vector<TorusValue, N + 1>;  // [b, a_0 .. a_{N-1}]
    using Key = Vector<bool, N>;

private:
    Random random;

public:
    TLWE() : random{} {}
    explicit TLWE(uint64_t seed) : random{seed} {}

    Key generate_s() { return random.template bit_vector<N>(); }

    // b = <a, s> + m + e with a uniform and e ~ N(0, ALPHA^2) (reference src/tlwe.hpp:55-61)
    Sample encrypt(const TorusValue& m, const Key& s) {
        Sample ct;
        TorusValue b = m + TorusValue{random.normal() * ALPHA};
        for (int i = 0; i < N; i++) {
            const TorusValue a = random.uniform_torus();
            ct[1 + i] = a;
            if (s[i]) b += a;
        }
        ct[0] = b;
        return ct;
    }
    Sample encrypt_single_binary(bool m, const Key& s) {
        return encrypt(TorusValue(m), s);
    }

    // phase b - <a, s> (reference src/tlwe.hpp:67-71)
    TorusValue decrypt(const Sample& ba, const Key& s) {
        TorusValue phase = ba[0];
        for (int i = 0; i < N; i++)
            if (s[i]) phase -= ba[1 + i];
        return phase;
    }
    bool decrypt_single_binary(const Sample& ba, const Key& s) {
        return static_cast<bool>(decrypt(ba, s));
    }
};
}  // namespace TFHE
