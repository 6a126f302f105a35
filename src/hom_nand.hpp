// Drop-in for KowerKoint/TFHE's src/hom_nand.hpp: the gate API.  Key generation, encryption and
// decryption run on the host; nand() is evaluated on the GPU through libtfhe_amd, and
// nand_batch() (not in the reference) evaluates many independent gates in one call.  gate(),
// gate_batch(), hom_not() and mux() (not in the reference either) are the other gates: the same
// bootstrap behind another linear combination of the operands.
//
// Unlike the reference (src/hom_nand.hpp:26,49) the TRGSW member honours TRLWEParameter, so the
// class can be instantiated with a non-default ring.
#pragma once

#include <array>
#include <random>
#include <vector>

#include "gpu_dispatch.hpp"
#include "tlwe_key_switching.hpp"
#include "trgsw.hpp"

namespace TFHE {
template <typename TLWELv0Parameter = TLWEParameter128BitSecurity, typename TLWELv1Parameter = TLWELv1ParameterDefault,
          typename TRLWEParameter = TRLWEParameterDefault, typename TRGSWParameter = TRGSWParameterDefault,
          typename TLWEKeySwitchingParameter = TLWEKeySwitchingParameterDefault>
class HomNAND {
    constexpr static int N_LV0 = TLWELv0Parameter::N;
    constexpr static int N_LV1 = TLWELv1Parameter::N;
    constexpr static int N = TRLWEParameter::N;
    constexpr static int K = TRGSWParameter::K;
    constexpr static int BG_BIT = TRGSWParameter::BG_BIT;
    constexpr static int L = TRGSWParameter::L;
    constexpr static int BASE_BIT = TLWEKeySwitchingParameter::BASE_BIT;
    constexpr static int T = TLWEKeySwitchingParameter::T;
    static_assert(N_LV1 == N * K, "the lv1 dimension is the flattened ring key");

private:
    TLWE<TLWELv0Parameter> tlwe_lv0;
    TLWE<TLWELv1Parameter> tlwe_lv1;
    TRLWE<TRLWEParameter> trlwe;
    TRGSW<TRGSWParameter, TRLWEParameter> trgsw;
    TLWEKeySwitching<decltype(tlwe_lv0), decltype(tlwe_lv1), TLWEKeySwitchingParameter> key_switching;

public:
    struct SecretKey {
        Vector<bool, N_LV0> tlwe_lv0_s;
        Vector<Polynomial<bool, N>, K> trlwe_s;
    };
    struct EvaluateKey {
        Vector<Matrix<Polynomial<TorusValue, N>, (K + 1) * L, K + 1>, N_LV0> bk;
        Vector<Vector<Vector<Vector<TorusValue, N_LV0 + 1>, (1 << BASE_BIT)>, T>, N_LV1> ks;
    };
    using Cipher = Vector<TorusValue, N_LV0 + 1>;

    HomNAND() : tlwe_lv0{}, tlwe_lv1{}, trlwe{}, trgsw{trlwe}, key_switching{tlwe_lv0} {}
    // reproducible key material and encryption noise
    explicit HomNAND(uint64_t seed)
        : tlwe_lv0{seed}, tlwe_lv1{seed + 1}, trlwe{seed + 2}, trgsw{trlwe}, key_switching{tlwe_lv0} {}

    SecretKey generate_secret_key() { return {tlwe_lv0.generate_s(), trlwe.generate_s()}; }

    EvaluateKey make_evaluate_key(const SecretKey& secret_key) {
        EvaluateKey evaluate_key;
        evaluate_key.bk = trgsw.make_bk(secret_key.tlwe_lv0_s, secret_key.trlwe_s);
        evaluate_key.ks =
            key_switching.make_ks(secret_key.tlwe_lv0_s, trlwe.extract_tlwe_lv0_key(secret_key.trlwe_s));
        return evaluate_key;
    }

    // make_evaluate_key with the encryptions done on the GPU (tfhe_amd_keygen): same key structure,
    // masks and noise from ChaCha20 under a 256-bit seed.  Not in the reference, whose
    // make_evaluate_key takes seconds of host time per key set.  Without a seed argument the seed
    // comes from the operating system (std::random_device, the reference's source of every word,
    // src/random.hpp:10): that is the form for real keys.  The 64-bit form is for reproducible tests.
    EvaluateKey make_evaluate_key_gpu(const SecretKey& secret_key) {
        std::random_device rd;
        std::array<uint32_t, 8> seed;
        for (uint32_t& w : seed) w = rd();
        return make_evaluate_key_gpu(secret_key, seed);
    }
    EvaluateKey make_evaluate_key_gpu(const SecretKey& secret_key, uint64_t test_seed) {
        return make_evaluate_key_gpu(secret_key, std::array<uint32_t, 8>{static_cast<uint32_t>(test_seed),
                                                                        static_cast<uint32_t>(test_seed >> 32)});
    }
    EvaluateKey make_evaluate_key_gpu(const SecretKey& secret_key, const std::array<uint32_t, 8>& seed) {
        std::vector<uint8_t> s0((size_t)N_LV0), sr((size_t)K * N);
        for (int i = 0; i < N_LV0; i++) s0[i] = secret_key.tlwe_lv0_s[i] ? 1 : 0;
        for (int k = 0; k < K; k++)
            for (int j = 0; j < N; j++) sr[(size_t)k * N + j] = secret_key.trlwe_s[k][j] ? 1 : 0;
        const tfhe_amd_params p = params();
        std::vector<uint32_t> bk(tfhe_amd_bk_words(&p)), ks(tfhe_amd_ks_words(&p));
        gpu::check(tfhe_amd_keygen(&p, s0.data(), sr.data(), TLWELv0Parameter::ALPHA, TRLWEParameter::ALPHA, seed.data(),
                                   gpu::device(), bk.data(), ks.data()),
                   "make_evaluate_key_gpu");
        EvaluateKey ek;
        const uint32_t* src = bk.data();
        for (int i = 0; i < N_LV0; i++)
            for (int r = 0; r < (K + 1) * L; r++)
                for (int c = 0; c <= K; c++)
                    for (int j = 0; j < N; j++) ek.bk[i][r][c][j] = TorusValue::from_raw_value(*src++);
        src = ks.data();
        for (int i = 0; i < N_LV1; i++)
            for (int m = 0; m < T; m++)
                for (int o = 0; o < (1 << BASE_BIT); o++)
                    for (int w = 0; w <= N_LV0; w++) ek.ks[i][m][o][w] = TorusValue::from_raw_value(*src++);
        return ek;
    }

    Cipher encrypt(bool m, const SecretKey& secret_key) {
        return tlwe_lv0.encrypt_single_binary(m, secret_key.tlwe_lv0_s);
    }
    bool decrypt(const Cipher& c, const SecretKey& secret_key) {
        return tlwe_lv0.decrypt_single_binary(c, secret_key.tlwe_lv0_s);
    }

    static tfhe_amd_params params() {
        tfhe_amd_params p;
        p.n = N_LV0; p.N = N; p.K = K; p.bg_bit = BG_BIT; p.L = L; p.T = T; p.base_bit = BASE_BIT;
        return p;
    }

    // Forget the device copy of a key that was edited in place (the cache only samples its contents).
    void invalidate(const EvaluateKey& evaluate_key) { gpu::cache().invalidate(&evaluate_key.bk, &evaluate_key.ks); }

    // The device copy of an evaluation key; built on first use and cached on the key's identity.
    // The returned handle keeps it alive while the caller evaluates on it.
    std::shared_ptr<gpu::DeviceKey> context(const EvaluateKey& evaluate_key) {
        return gpu::cache().get(
            params(), &evaluate_key.bk, &evaluate_key.ks,
            gpu::fingerprint(evaluate_key.bk) ^ gpu::fingerprint(evaluate_key.ks),
            [&] { return gpu::flatten_bk(evaluate_key.bk); }, [&] { return gpu::flatten_ks(evaluate_key.ks); });
    }

    // (1/8, 0..0) - a - b, bootstrapped to lv1 and key-switched back (reference src/hom_nand.hpp:74-81)
    Cipher nand(const Cipher& a, const Cipher& b, const EvaluateKey& evaluate_key) {
        std::vector<uint32_t> fa, fb, out((size_t)N_LV0 + 1);
        gpu::put(fa, a);
        gpu::put(fb, b);
        gpu::check(tfhe_amd_nand_batch(context(evaluate_key)->ctx(), fa.data(), fb.data(), out.data(), 1), "nand");
        return gpu::take_vector<N_LV0 + 1>(out.data());
    }

    // out[g] = nand(a[g], b[g]) for independent gates, one GPU call
    std::vector<Cipher> nand_batch(const std::vector<Cipher>& a, const std::vector<Cipher>& b,
                                   const EvaluateKey& evaluate_key) {
        if (a.size() != b.size()) throw std::invalid_argument("nand_batch: a and b differ in length");
        const size_t G = a.size(), W = (size_t)N_LV0 + 1;
        std::vector<uint32_t> fa, fb, out(G * W);
        fa.reserve(G * W);
        fb.reserve(G * W);
        for (size_t g = 0; g < G; g++) {
            gpu::put(fa, a[g]);
            gpu::put(fb, b[g]);
        }
        // over every GPU of gpu::devices() when that lists more than one: contiguous blocks of gates
        const auto key = context(evaluate_key);
        if (key->multi)
            gpu::check(tfhe_amd_multi_nand_batch(key->multi, fa.data(), fb.data(), out.data(), G), "nand_batch");
        else
            gpu::check(tfhe_amd_nand_batch(key->single, fa.data(), fb.data(), out.data(), G), "nand_batch");
        std::vector<Cipher> res;
        res.reserve(G);
        for (size_t g = 0; g < G; g++) res.push_back(gpu::take_vector<N_LV0 + 1>(out.data() + g * W));
        return res;
    }

    // Gate codes of the C ABI (TFHE_AMD_GATE_*): ANDNY = (not a) and b, ANDYN = a and (not b).
    enum class Gate { NAND = 0, AND, OR, NOR, XOR, XNOR, ANDNY, ANDYN, ORNY, ORYN };

    Cipher gate(Gate op, const Cipher& a, const Cipher& b, const EvaluateKey& evaluate_key) {
        std::vector<uint32_t> fa, fb, out((size_t)N_LV0 + 1);
        gpu::put(fa, a);
        gpu::put(fb, b);
        gpu::check(tfhe_amd_gate_batch(context(evaluate_key)->ctx(), (int)op, fa.data(), fb.data(), out.data(), 1),
                   "gate");
        return gpu::take_vector<N_LV0 + 1>(out.data());
    }

    std::vector<Cipher> gate_batch(Gate op, const std::vector<Cipher>& a, const std::vector<Cipher>& b,
                                   const EvaluateKey& evaluate_key) {
        if (a.size() != b.size()) throw std::invalid_argument("gate_batch: a and b differ in length");
        const size_t G = a.size(), W = (size_t)N_LV0 + 1;
        std::vector<uint32_t> fa, fb, out(G * W);
        for (size_t g = 0; g < G; g++) {
            gpu::put(fa, a[g]);
            gpu::put(fb, b[g]);
        }
        const auto key = context(evaluate_key);
        if (key->multi)
            gpu::check(tfhe_amd_multi_gate_batch(key->multi, (int)op, fa.data(), fb.data(), out.data(), G), "gate_batch");
        else
            gpu::check(tfhe_amd_gate_batch(key->single, (int)op, fa.data(), fb.data(), out.data(), G), "gate_batch");
        std::vector<Cipher> res;
        res.reserve(G);
        for (size_t g = 0; g < G; g++) res.push_back(gpu::take_vector<N_LV0 + 1>(out.data() + g * W));
        return res;
    }

    // -a: no bootstrap, host arithmetic
    Cipher hom_not(const Cipher& a) { return -a; }

    // sel ? x1 : x0 (two blind rotations, one key switch)
    Cipher mux(const Cipher& sel, const Cipher& x1, const Cipher& x0, const EvaluateKey& evaluate_key) {
        std::vector<uint32_t> fs, f1, f0, out((size_t)N_LV0 + 1);
        gpu::put(fs, sel);
        gpu::put(f1, x1);
        gpu::put(f0, x0);
        gpu::check(tfhe_amd_mux_batch(context(evaluate_key)->ctx(), fs.data(), f1.data(), f0.data(), out.data(), 1),
                   "mux");
        return gpu::take_vector<N_LV0 + 1>(out.data());
    }
};
}  // namespace TFHE
