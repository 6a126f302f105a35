// Drop-in for KowerKoint/TFHE's src/tlwe_key_switching.hpp: key-switching key generation (host)
// and the identity key switch lv1 -> lv0, evaluated on the GPU through libtfhe_amd.
#pragma once

#include <vector>

#include "gpu_dispatch.hpp"
#include "tlwe.hpp"

namespace TFHE {
struct TLWEKeySwitchingParameterDefault {
    constexpr static int T = 5;
    constexpr static int BASE_BIT = 2;
};

template <typename TLWELv0, typename TLWELv1, typename TLWEKeySwitchingParameter = TLWEKeySwitchingParameterDefault>
class TLWEKeySwitching {
public:
    constexpr static int N_LV0 = TLWELv0::N;
    constexpr static int N_LV1 = TLWELv1::N;
    constexpr static int T = TLWEKeySwitchingParameter::T;
    constexpr static int BASE_BIT = TLWEKeySwitchingParameter::BASE_BIT;
    using SampleLv0 = Vector<TorusValue, N_LV0 + 1>;
    using SampleLv1 = Vector<TorusValue, N_LV1 + 1>;
    using Key = Vector<Vector<Vector<SampleLv0, (1 << BASE_BIT)>, T>, N_LV1>;  // [coefficient][level][digit]

private:
    TLWELv0& tlwe_lv0;

public:
    TLWEKeySwitching(TLWELv0& tlwe_lv0) : tlwe_lv0(tlwe_lv0) {}

    // ks[i][m][o] encrypts o * s1_i * 2^(32 - (m+1) BASE_BIT) under the lv0 key; slot o = 0 stays
    // all zero and is never read (reference src/tlwe_key_switching.hpp:26-45)
    Key make_ks(const Vector<bool, N_LV0>& s_lv0, const Vector<bool, N_LV1>& s_lv1) {
        Key ks;
        for (int i = 0; i < N_LV1; i++)
            for (int m = 0; m < T; m++)
                for (int o = 1; o < (1 << BASE_BIT); o++) {
                    const uint32_t msg = s_lv1[i] ? static_cast<uint32_t>(o) << (32 - (m + 1) * BASE_BIT) : 0U;
                    ks[i][m][o] = tlwe_lv0.encrypt(TorusValue::from_raw_value(msg), s_lv0);
                }
        return ks;
    }

    // (b, 0..0) - sum over i, m of ks[i][m][digit_m(a_i)], on the GPU
    // (reference src/tlwe_key_switching.hpp:47-65)
    SampleLv0 identity_key_switch(const SampleLv1& ba_lv1, const Key& ks) {
        tfhe_amd_params p;
        p.n = N_LV0;
        p.N = N_LV1 & -N_LV1;  // the key switch only sees N*K = N_LV1
        p.K = N_LV1 / p.N;
        p.bg_bit = 8; p.L = 1;  // unused by this call
        p.T = T; p.base_bit = BASE_BIT;
        const auto key = gpu::cache().get(
            p, nullptr, &ks, gpu::fingerprint(ks), [] { return std::vector<uint32_t>(); },
            [&] { return gpu::flatten_ks(ks); });
        tfhe_amd_ctx* ctx = key->ctx();  // `key` keeps the device copy alive for the call
        std::vector<uint32_t> in, out((size_t)N_LV0 + 1);
        gpu::put(in, ba_lv1);
        gpu::check(tfhe_amd_identity_key_switch_batch(ctx, in.data(), out.data(), 1), "identity_key_switch");
        return gpu::take_vector<N_LV0 + 1>(out.data());
    }
};
}  // namespace TFHE
