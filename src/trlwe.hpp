// Drop-in for KowerKoint/TFHE's src/trlwe.hpp: ring-LWE samples [b(X), a_1(X) .. a_K(X)] modulo
// X^N + 1.  Encryption and decryption are client side (host); sample extraction is an index
// permutation that the GPU blind-rotation kernel also performs when it writes its result.
#pragma once

#include <algorithm>
#include <cstdint>

#include "random.hpp"
#include "types.hpp"

namespace TFHE {
struct TRLWEParameterDefault {
    constexpr static int N = 512;
    constexpr static int K = 2;
    constexpr static double ALPHA = 2.9802322387695312e-08;  // 2^-25
};
struct TLWELv1ParameterDefault {
    constexpr static int N = TRLWEParameterDefault::N * TRLWEParameterDefault::K;
    constexpr static double ALPHA = 2.9802322387695312e-08;  // 2^-25
};

template <typename TRLWEParameter = TRLWEParameterDefault>
class TRLWE {
public:
    constexpr static int N = TRLWEParameter::N;
    constexpr static int K = TRLWEParameter::K;
    constexpr static double ALPHA = TRLWEParameter::ALPHA;

    using Plain = Polynomial<TorusValue, N>;        // message or phase
    using Bits = Polynomial<bool, N>;
    using Sample = Vector<Plain, K + 1>;            // [b, a_1 .. a_K]
    using Key = Vector<Bits, K>;

private:
    Random random;

    // acc +/-= a(X) * s(X) for a binary s, negacyclic
    static void mask_times_key(Plain& acc, const Plain& a, const Bits& s, bool subtract) {
        for (int i = 0; i < N; i++) {
            if (!s[i]) continue;
            for (int j = 0; j < N; j++) {
                const int k = i + j;
                const bool neg = (k >= N) != subtract;
                TorusValue& dst = acc[k >= N ? k - N : k];
                if (neg) dst -= a[j];
                else     dst += a[j];
            }
        }
    }

public:
    TRLWE() : random{} {}
    explicit TRLWE(uint64_t seed) : random{seed} {}

    Key generate_s() {
        Key s;
        for (int k = 0; k < K; k++) s[k] = random.template bit_polynomial<N>();
        return s;
    }

    // b = sum_k a_k * s_k + m + e (reference src/trlwe.hpp:75-82); the products are exact
    Sample encrypt(const Plain& m, const Key& s) {
        Sample ct;
        for (int k = 0; k < K; k++)
            std::generate(ct[1 + k].begin(), ct[1 + k].end(), [this] { return random.uniform_torus(); });
        for (int j = 0; j < N; j++) ct[0][j] = m[j] + TorusValue{random.normal() * ALPHA};
        for (int k = 0; k < K; k++) mask_times_key(ct[0], ct[1 + k], s[k], false);
        return ct;
    }
    Sample encrypt_binary_polynomial(const Bits& m, const Key& s) {
        Plain encoded;
        for (int j = 0; j < N; j++) encoded[j] = TorusValue(static_cast<bool>(m[j]));
        return encrypt(encoded, s);
    }

    // phase b - sum_k a_k * s_k (reference src/trlwe.hpp:91-97)
    Plain decrypt(const Sample& ba, const Key& s) const {
        Plain phase = ba[0];
        for (int k = 0; k < K; k++) mask_times_key(phase, ba[1 + k], s[k], true);
        return phase;
    }
    Bits decrypt_binary_polynomial(const Sample& ba, const Key& s) const {
        const Plain phase = decrypt(ba, s);
        Bits bits;
        for (int j = 0; j < N; j++) bits[j] = static_cast<bool>(phase[j]);
        return bits;
    }

    // LWE sample of coefficient x of the message (reference src/trlwe.hpp:109-122)
    Vector<TorusValue, N * K + 1> sample_extract_index(const Sample& trlwe_ba, int x) const {
        Vector<TorusValue, N * K + 1> lwe;
        lwe[0] = trlwe_ba[0][x];
        for (int k = 0; k < K; k++) {
            const Plain& a = trlwe_ba[1 + k];
            TorusValue* dst = lwe.begin() + 1 + k * N;
            for (int i = 0; i < N; i++) dst[i] = i <= x ? a[x - i] : -a[N + x - i];
        }
        return lwe;
    }

    // the ring key read as an LWE key of dimension N*K (reference src/trlwe.hpp:124-131)
    Vector<bool, N * K> extract_tlwe_lv0_key(const Key& s) {
        Vector<bool, N * K> flat;
        for (int k = 0; k < K; k++) std::copy(s[k].begin(), s[k].end(), flat.begin() + k * N);
        return flat;
    }
};
}  // namespace TFHE
