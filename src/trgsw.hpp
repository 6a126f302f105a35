// Drop-in for KowerKoint/TFHE's src/trgsw.hpp: TRGSW encryption (host, key generation) and the
// bootstrapping machinery -- external product, CMUX, blind rotation, gate bootstrapping -- which
// here is evaluated on the GPU through libtfhe_amd.  Same class, same member signatures.
#pragma once

#include <algorithm>
#include <vector>

#include "gpu_dispatch.hpp"
#include "trlwe.hpp"

namespace TFHE {
struct TRGSWParameterDefault {
    constexpr static int K = 2;
    constexpr static int BG_BIT = 8;
    constexpr static int L = 2;
};

template <typename TRGSWParameter = TRGSWParameterDefault, typename TRLWEParameter = TRLWEParameterDefault,
          std::enable_if_t<TRGSWParameter::K == TRLWEParameter::K>* = nullptr>
class TRGSW {
public:
    constexpr static int BG_BIT = TRGSWParameter::BG_BIT;
    constexpr static int N = TRLWEParameter::N;
    constexpr static int K = TRGSWParameter::K;
    constexpr static int L = TRGSWParameter::L;
    using Int = SignedInt<BG_BIT>;
    using Sample = Vector<Polynomial<TorusValue, N>, K + 1>;
    using Ciphertext = Matrix<Polynomial<TorusValue, N>, (K + 1) * L, K + 1>;

private:
    TRLWE<TRLWEParameter>& trlwe;

    // Signed gadget digits, most significant level first (reference src/trgsw.hpp:26-50), in
    // closed form: adding half a base at every level's position turns the reference's
    // "digit >= half: subtract the base and carry" chain into ordinary binary carries.
    // Host-side helper; evaluation decomposes on the GPU.
    Vector<Polynomial<Int, N>, L> decomposition(const Polynomial<TorusValue, N>& a) {
        uint32_t offset = 1U << (32 - L * BG_BIT - 1);
        for (int l = 0; l < L; l++) offset += 1U << (32 - BG_BIT * (l + 1) + BG_BIT - 1);
        Vector<Polynomial<Int, N>, L> digits;
        for (int j = 0; j < N; j++) {
            const uint32_t t = a[j].get_raw_value() + offset;
            for (int l = 0; l < L; l++)
                digits[l][j] = static_cast<Int>(static_cast<int>((t >> (32 - BG_BIT * (l + 1))) & ((1U << BG_BIT) - 1)) -
                                                (1 << (BG_BIT - 1)));
        }
        return digits;
    }

    static tfhe_amd_params shape(int tlwe_n) {
        tfhe_amd_params p;
        p.n = tlwe_n; p.N = N; p.K = K; p.bg_bit = BG_BIT; p.L = L;
        p.T = 1; p.base_bit = 1;  // unused by the calls made from this class
        return p;
    }

    template <int TLWE_N>
    std::shared_ptr<gpu::DeviceKey> context_for(const Vector<Ciphertext, TLWE_N>& bk) {
        return gpu::cache().get(
            shape(TLWE_N), &bk, nullptr, gpu::fingerprint(bk), [&] { return gpu::flatten_bk(bk); },
            [] { return std::vector<uint32_t>(); });
    }

public:
    TRGSW(TRLWE<TRLWEParameter>& trlwe) : trlwe(trlwe) {}

    // Row comp*L + level is a fresh TRLWE encryption of zero with mu * 2^(32 - BG_BIT (level+1))
    // added to component `comp` (reference src/trgsw.hpp:61-82).
    Ciphertext encrypt_integer_polynomial(const Polynomial<Int, N>& mu, const Vector<Polynomial<bool, N>, K>& s) {
        Ciphertext out;
        const Polynomial<TorusValue, N> zero;
        for (int comp = 0; comp <= K; comp++)
            for (int level = 0; level < L; level++) {
                Sample& row = out[comp * L + level];
                row = trlwe.encrypt(zero, s);
                const int shift = 32 - BG_BIT * (level + 1);
                for (int j = 0; j < N; j++)
                    row[comp][j] += TorusValue::from_raw_value(static_cast<uint32_t>(mu[j]) << shift);
            }
        return out;
    }
    Ciphertext encrypt_binary_polynomial(const Polynomial<bool, N>& mu, const Vector<Polynomial<bool, N>, K>& s) {
        Polynomial<Int, N> as_int;
        for (int j = 0; j < N; j++) as_int[j] = mu[j] ? 1 : 0;
        return encrypt_integer_polynomial(as_int, s);
    }
    Ciphertext encrypt_single_binary(bool m, const Vector<Polynomial<bool, N>, K>& s) {
        Polynomial<Int, N> as_int;
        as_int[0] = m ? 1 : 0;
        return encrypt_integer_polynomial(as_int, s);
    }

    // c (x) ba, evaluated on the GPU (reference src/trgsw.hpp:100-110)
    Sample external_product(const Ciphertext& c, const Sample& ba) {
        std::vector<uint32_t> g, x, out((size_t)(K + 1) * N);
        gpu::put(g, c);
        gpu::put(x, ba);
        const tfhe_amd_params p = shape(1);
        gpu::check(tfhe_amd_external_product_batch(&p, g.data(), x.data(), out.data(), 1, gpu::device()),
                   "external_product");
        return gpu::take_trlwe<N, K + 1>(out.data());
    }

    // c (x) (ba_1 - ba_0) + ba_0, evaluated on the GPU (reference src/trgsw.hpp:112-117)
    Sample cmux(const Ciphertext& c, const Sample& ba_0, const Sample& ba_1) {
        std::vector<uint32_t> g, d0, d1, out((size_t)(K + 1) * N);
        gpu::put(g, c);
        gpu::put(d0, ba_0);
        gpu::put(d1, ba_1);
        const tfhe_amd_params p = shape(1);
        gpu::check(tfhe_amd_cmux_batch(&p, g.data(), d0.data(), d1.data(), out.data(), 1, gpu::device()), "cmux");
        return gpu::take_trlwe<N, K + 1>(out.data());
    }

    // bk[i] = TRGSW encryption of the i-th lv0 key bit (reference src/trgsw.hpp:119-129)
    template <int TLWE_N>
    Vector<Ciphertext, TLWE_N> make_bk(const Vector<bool, TLWE_N>& tlwe_s, const Vector<Polynomial<bool, N>, K>& trlwe_s) {
        Vector<Ciphertext, TLWE_N> bk;
        for (int i = 0; i < TLWE_N; i++) bk[i] = encrypt_single_binary(tlwe_s[i], trlwe_s);
        return bk;
    }

    // CMUX chain over the lv0 ciphertext, on the GPU (reference src/trgsw.hpp:131-170)
    template <int TLWE_N>
    Sample blind_rotate(const Vector<TorusValue, TLWE_N + 1>& tlwe_lv0, const Vector<Ciphertext, TLWE_N>& bk,
                        const Sample& test_vector) {
        std::vector<uint32_t> c, tv, acc((size_t)(K + 1) * N);
        gpu::put(c, tlwe_lv0);
        gpu::put(tv, test_vector);
        gpu::check(tfhe_amd_blind_rotate_batch(context_for(bk)->ctx(), c.data(), tv.data(), acc.data(), 1), "blind_rotate");
        return gpu::take_trlwe<N, K + 1>(acc.data());
    }

    // blind rotation of the constant-1/8 test vector, then sample extraction at 0
    // (reference src/trgsw.hpp:172-182)
    template <int TLWE_N>
    Vector<TorusValue, N * K + 1> gate_bootstrapping_tlwe_to_tlwe(const Vector<TorusValue, TLWE_N + 1>& tlwe_lv0,
                                                                  const Vector<Ciphertext, TLWE_N>& bk) {
        std::vector<uint32_t> c, lv1((size_t)N * K + 1);
        gpu::put(c, tlwe_lv0);
        gpu::check(tfhe_amd_gate_bootstrapping_batch(context_for(bk)->ctx(), c.data(), lv1.data(), 1),
                   "gate_bootstrapping_tlwe_to_tlwe");
        return gpu::take_vector<N * K + 1>(lv1.data());
    }
};
}  // namespace TFHE
