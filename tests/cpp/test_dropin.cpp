// Behavioural tests of the drop-in headers (src/*.hpp), modelled on what the reference's
// test driver pins (reference test/main.cpp:13-224): randomised encrypt -> operate -> decrypt
// round trips, plus exact-arithmetic checks the reference cannot pass (its FFT product is not
// exact).  Usage: dropin_test <group>   with group = host | gpu | all.
// "host" needs no GPU; "gpu" evaluates CMUX, bootstrapping, key switching and NAND on device 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "hom_nand.hpp"
#include "tlwe.hpp"
#include "tlwe_key_switching.hpp"
#include "trgsw.hpp"
#include "trlwe.hpp"
#include "types.hpp"

int second_translation_unit_value();  // defined in test_dropin_tu2.cpp, which includes the headers too

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)

using namespace TFHE;

// ---------------------------------------------------------------- host group
static void test_torus_value() {
    CHECK(TorusValue(true).get_raw_value() == (1u << 29));
    CHECK(TorusValue(false).get_raw_value() == 0u - (1u << 29));
    CHECK(static_cast<bool>(TorusValue(true)) && !static_cast<bool>(TorusValue(false)));
    CHECK(TorusValue(0.25).get_raw_value() == (1u << 30));
    CHECK(TorusValue(-0.25).get_raw_value() == 3u << 30);      // fmod, +1 when negative
    CHECK(TorusValue(1.75).get_raw_value() == 3u << 30);
    CHECK(TorusValue(0.1).get_raw_value() == 429496729u);      // truncation, not rounding
    const TorusValue a = TorusValue::from_raw_value(0xFFFFFFF0u), b = TorusValue::from_raw_value(0x20u);
    CHECK((a + b).get_raw_value() == 0x10u && (b - a).get_raw_value() == 0x30u && (-b).get_raw_value() == 0xFFFFFFE0u);
    CHECK((a * 3).get_raw_value() == 0xFFFFFFD0u && (int8_t(-2) * b).get_raw_value() == 0xFFFFFFC0u);
    CHECK(static_cast<double>(TorusValue::from_raw_value(1u << 31)) == 0.5);
}

static void test_polynomial_x_exp() {  // reference test/main.cpp:92-109, with an exact product
    std::mt19937 mt(0);
    for (int round = 0; round < 100; round++) {
        Polynomial<int, 512> poly;
        for (int i = 0; i < 512; i++) poly[i] = mt() % 10000;
        int n = mt() % 1024;
        Polynomial<int, 512> x_n;
        if (n >= 512) { x_n[n - 512] = -1; n -= 1024; }
        else          { x_n[n] = 1; }
        CHECK(poly * x_n == poly.multiply_x_exp(n));
    }
    Polynomial<TorusValue, 16> q;
    for (int i = 0; i < 16; i++) q[i] = TorusValue::from_raw_value(i + 1);
    CHECK(q.multiply_x_exp(32) == q && q.multiply_x_exp(-32) == q && q.multiply_x_exp(16) == -q);
    CHECK(q.multiply_x_exp(1)[0].get_raw_value() == 0u - 16u && q.multiply_x_exp(-1)[15].get_raw_value() == 0u - 1u);
}

static void test_tlwe() {  // reference test/main.cpp:13-22
    TLWE<> tlwe;
    std::mt19937 mt(0);
    for (int i = 0; i < 100; i++) {
        auto s = tlwe.generate_s();
        bool msg = mt() & 1;
        CHECK(tlwe.decrypt_single_binary(tlwe.encrypt_single_binary(msg, s), s) == msg);
    }
}

static void test_trlwe() {  // reference test/main.cpp:24-35
    TRLWE<> trlwe;
    constexpr int N = TRLWE<>::N;
    std::mt19937 mt(0);
    for (int i = 0; i < 20; i++) {
        auto s = trlwe.generate_s();
        Polynomial<bool, N> msg;
        for (int j = 0; j < N; j++) msg[j] = mt() & 1;
        CHECK(trlwe.decrypt_binary_polynomial(trlwe.encrypt_binary_polynomial(msg, s), s) == msg);
    }
}

static void test_sample_extract() {  // phase of the extracted sample = coefficient x of the ring phase
    TRLWE<> trlwe(7);
    TLWE<TLWELv1ParameterDefault> tlwe1;
    constexpr int N = TRLWE<>::N;
    auto s = trlwe.generate_s();
    auto s1 = trlwe.extract_tlwe_lv0_key(s);
    Polynomial<TorusValue, N> m;
    for (int j = 0; j < N; j++) m[j] = TorusValue::from_raw_value(j * 0x00800000u);
    auto ct = trlwe.encrypt(m, s);
    auto phase = trlwe.decrypt(ct, s);
    for (int x : {0, 1, 17, N - 1})
        CHECK(tlwe1.decrypt(trlwe.sample_extract_index(ct, x), s1) == phase[x]);
}

static void test_seeded_keys_reproduce() {
    HomNAND<> h1(42), h2(42), h3(43);
    auto k1 = h1.generate_secret_key(), k2 = h2.generate_secret_key(), k3 = h3.generate_secret_key();
    CHECK(k1.tlwe_lv0_s == k2.tlwe_lv0_s && k1.trlwe_s == k2.trlwe_s);
    CHECK(k1.tlwe_lv0_s != k3.tlwe_lv0_s);
    CHECK(h1.encrypt(true, k1) == h2.encrypt(true, k2));
}

static void run_host() {
    CHECK(second_translation_unit_value() == 637);  // two translation units include the headers
    test_torus_value();
    test_polynomial_x_exp();
    test_tlwe();
    test_trlwe();
    test_sample_extract();
    test_seeded_keys_reproduce();
    std::puts("host group ok");
}

// ---------------------------------------------------------------- gpu group
static void test_cmux() {  // reference test/main.cpp:66-90
    TRLWE<> trlwe;
    constexpr int N = TRLWE<>::N, K = TRLWE<>::K;
    TRGSW<> trgsw(trlwe);
    std::mt19937 mt(0);
    for (int i = 0; i < 10; i++) {
        auto s = trlwe.generate_s();
        Polynomial<bool, N> msgs[2];
        Vector<Polynomial<TorusValue, N>, K + 1> enc[2];
        for (int j = 0; j < 2; j++) {
            for (int t = 0; t < N; t++) msgs[j][t] = mt() & 1;
            enc[j] = trlwe.encrypt_binary_polynomial(msgs[j], s);
        }
        const int answer = mt() & 1;
        Polynomial<bool, N> sel;
        sel[0] = answer;
        auto c = trgsw.encrypt_binary_polynomial(sel, s);
        CHECK(msgs[answer] == trlwe.decrypt_binary_polynomial(trgsw.cmux(c, enc[0], enc[1]), s));
    }
}

static void test_external_product_by_one() {  // reference test/main.cpp:37-64, with an assertion
    TRLWE<> trlwe;
    constexpr int N = TRLWE<>::N;
    TRGSW<> trgsw(trlwe);
    auto s = trlwe.generate_s();
    Polynomial<bool, N> msg;
    for (int i = 0; i < N; i++) msg[i] = (i % 3) == 0;
    auto ba = trlwe.encrypt_binary_polynomial(msg, s);
    Polynomial<int8_t, N> one;
    one[0] = 1;
    auto e = trgsw.external_product(trgsw.encrypt_integer_polynomial(one, s), ba);
    CHECK(trlwe.decrypt_binary_polynomial(e, s) == msg);
}

static void test_bootstrapping() {  // reference test/main.cpp:111-148
    TLWE<> tlwe_lv0;
    TLWE<TLWELv1ParameterDefault> tlwe_lv1;
    TRLWE<> trlwe;
    TRGSW<> trgsw(trlwe);
    std::mt19937 mt(0);
    std::uniform_real_distribution<> rd[2] = {std::uniform_real_distribution<>{0.55, 0.95},
                                              std::uniform_real_distribution<>{0.05, 0.45}};
    for (int key = 0; key < 2; key++) {
        auto s0 = tlwe_lv0.generate_s();
        auto sr = trlwe.generate_s();
        auto s1 = trlwe.extract_tlwe_lv0_key(sr);
        auto bk = trgsw.make_bk(s0, sr);
        for (int i = 0; i < 10; i++) {
            const bool input = mt() & 1;
            auto lv0 = tlwe_lv0.encrypt(TorusValue{rd[input](mt)}, s0);  // true: [0.05,0.45), false: [0.55,0.95)
            CHECK(input == tlwe_lv1.decrypt_single_binary(trgsw.gate_bootstrapping_tlwe_to_tlwe(lv0, bk), s1));
        }
    }
}

static void test_key_switching() {  // reference test/main.cpp:150-178
    TLWE<> tlwe_lv0;
    TLWE<TLWELv1ParameterDefault> tlwe_lv1;
    TLWEKeySwitching<decltype(tlwe_lv0), decltype(tlwe_lv1)> key_switching(tlwe_lv0);
    std::mt19937 mt(0);
    std::uniform_real_distribution<> rd[2] = {std::uniform_real_distribution<>{0.55, 0.95},
                                              std::uniform_real_distribution<>{0.05, 0.45}};
    for (int key = 0; key < 2; key++) {
        auto s0 = tlwe_lv0.generate_s();
        auto s1 = tlwe_lv1.generate_s();
        auto ks = key_switching.make_ks(s0, s1);
        for (int i = 0; i < 10; i++) {
            const bool input = mt() & 1;
            auto lv1 = tlwe_lv1.encrypt(TorusValue{rd[input](mt)}, s1);
            CHECK(input == static_cast<bool>(tlwe_lv0.decrypt(key_switching.identity_key_switch(lv1, ks), s0)));
        }
    }
}

static void test_hom_nand() {  // reference test/main.cpp:180-224: 50 chained gates on 10 wires
    HomNAND<> hom_nand;
    std::mt19937 mt;
    auto secret_key = hom_nand.generate_secret_key();
    auto evaluate_key = hom_nand.make_evaluate_key(secret_key);
    bool a[10];
    HomNAND<>::Cipher ca[10];
    for (int i = 0; i < 10; i++) {
        a[i] = mt() & 1;
        ca[i] = hom_nand.encrypt(a[i], secret_key);
    }
    for (int i = 0; i < 50; i++) {
        int x = mt() % 10, y = mt() % 9, z = mt() % 8;
        if (y == x) y = 9;
        if (z == x) z = 8;
        if (z == y) z = 9;
        ca[z] = hom_nand.nand(ca[x], ca[y], evaluate_key);
        a[z] = !(a[x] && a[y]);
    }
    for (int i = 0; i < 10; i++) CHECK(a[i] == hom_nand.decrypt(ca[i], secret_key));

    // the batch entry point gives the same ciphertexts as the gate-at-a-time one
    std::vector<HomNAND<>::Cipher> va, vb;
    for (int i = 0; i < 9; i++) { va.push_back(ca[i]); vb.push_back(ca[i + 1]); }
    auto batch = hom_nand.nand_batch(va, vb, evaluate_key);
    for (int i = 0; i < 9; i++) {
        CHECK(batch[i] == hom_nand.nand(va[i], vb[i], evaluate_key));
        CHECK(hom_nand.decrypt(batch[i], secret_key) == !(a[i] && a[i + 1]));
    }
}

static void test_other_gates() {  // not in the reference: truth tables of the other gates and MUX
    using H = HomNAND<>;
    H h(77);
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key(sk);
    struct { H::Gate g; bool tt[4]; } table[] = {
        {H::Gate::NAND, {1, 1, 1, 0}}, {H::Gate::AND, {0, 0, 0, 1}},  {H::Gate::OR, {0, 1, 1, 1}},
        {H::Gate::NOR, {1, 0, 0, 0}},  {H::Gate::XOR, {0, 1, 1, 0}},  {H::Gate::XNOR, {1, 0, 0, 1}},
        {H::Gate::ANDNY, {0, 1, 0, 0}}, {H::Gate::ANDYN, {0, 0, 1, 0}}, {H::Gate::ORNY, {1, 1, 0, 1}},
        {H::Gate::ORYN, {1, 0, 1, 1}}};
    for (auto& row : table) {
        std::vector<H::Cipher> va, vb;
        for (int t = 0; t < 4; t++) {  // t = 2a + b
            va.push_back(h.encrypt(t >> 1, sk));
            vb.push_back(h.encrypt(t & 1, sk));
        }
        auto out = h.gate_batch(row.g, va, vb, ek);
        for (int t = 0; t < 4; t++) CHECK(h.decrypt(out[t], sk) == row.tt[t]);
        CHECK(h.gate(row.g, va[2], vb[2], ek) == out[2]);
    }
    for (int t = 0; t < 8; t++) {
        const bool s = t & 4, x1 = t & 2, x0 = t & 1;
        CHECK(h.decrypt(h.mux(h.encrypt(s, sk), h.encrypt(x1, sk), h.encrypt(x0, sk), ek), sk) == (s ? x1 : x0));
        CHECK(h.decrypt(h.hom_not(h.encrypt(s, sk)), sk) == !s);
    }
}

static void test_reference_fft_mode() {  // same gates, products through the reference's FFT replayed
    HomNAND<> h(21);
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key_gpu(sk, 3);
    auto a = h.encrypt(true, sk), b = h.encrypt(false, sk);
    auto exact = h.nand(a, b, ek);
    gpu::context_flags() = TFHE_AMD_CTX_REFERENCE_FFT;
    auto replay = h.nand(a, b, ek);
    gpu::context_flags() = TFHE_AMD_CTX_DEFAULT;
    CHECK(h.decrypt(exact, sk) && h.decrypt(replay, sk));
    CHECK(!(exact == replay));            // the reference's product is one unit low here and there
    CHECK(h.nand(a, b, ek) == exact);     // back on the exact path
}

static void test_gpu_keygen() {  // make_evaluate_key_gpu: not in the reference
    HomNAND<> h(5);
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key_gpu(sk, 99);
    for (int t = 0; t < 4; t++) {
        const bool x = t & 1, y = t & 2;
        CHECK(h.decrypt(h.nand(h.encrypt(x, sk), h.encrypt(y, sk), ek), sk) == !(x && y));
    }
    auto ek2 = h.make_evaluate_key_gpu(sk, 99);
    CHECK(ek2.bk[3][1][2] == ek.bk[3][1][2]);
    CHECK(ek2.ks[7][2][1] == ek.ks[7][2][1]);
}

struct RingN1024 { constexpr static int N = 1024; constexpr static int K = 1; constexpr static double ALPHA = 2.9802322387695312e-08; };
struct Lv1N1024 { constexpr static int N = 1024; constexpr static double ALPHA = 2.9802322387695312e-08; };
struct Gsw1024 { constexpr static int K = 1; constexpr static int BG_BIT = 8; constexpr static int L = 3; };
struct Lv0N64 { constexpr static int N = 64; constexpr static double ALPHA = 9.5367431640625e-07; };

static void test_hom_nand_other_ring() {  // a ring the reference's HomNAND cannot be instantiated with
    using H = HomNAND<Lv0N64, Lv1N1024, RingN1024, Gsw1024, TLWEKeySwitchingParameterDefault>;
    H h;
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key(sk);
    for (int t = 0; t < 4; t++) {
        const bool x = t & 1, y = t & 2;
        CHECK(h.decrypt(h.nand(h.encrypt(x, sk), h.encrypt(y, sk), ek), sk) == !(x && y));
    }
}

static void test_concurrent_nand_on_shared_key() {
    // Eight host threads call nand() on one shared EvaluateKey, as the reference allows (its nand reads
    // only its arguments, reference src/hom_nand.hpp:74-81).  The first calls race to build the device
    // copy of the key; every output must equal the one a single thread computes.
    HomNAND<> h(31);
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key_gpu(sk, 8);
    constexpr int THREADS = 8, PER = 6;
    std::vector<HomNAND<>::Cipher> a, b;
    std::vector<bool> pa, pb;
    std::mt19937 mt(9);
    for (int i = 0; i < THREADS * PER; i++) {
        pa.push_back(mt() & 1);
        pb.push_back(mt() & 1);
        a.push_back(h.encrypt(pa.back(), sk));
        b.push_back(h.encrypt(pb.back(), sk));
    }
    std::vector<HomNAND<>::Cipher> out(a.size());
    std::vector<std::thread> ts;
    for (int t = 0; t < THREADS; t++)
        ts.emplace_back([&, t] {
            for (int i = t * PER; i < (t + 1) * PER; i++) out[i] = h.nand(a[i], b[i], ek);
        });
    for (auto& t : ts) t.join();
    auto serial = h.nand_batch(a, b, ek);
    for (size_t i = 0; i < a.size(); i++) {
        CHECK(out[i] == serial[i]);
        CHECK(h.decrypt(out[i], sk) == !(pa[i] && pb[i]));
    }
}

static void test_multi_device_batch() {
    // nand_batch / gate_batch over a device list: GPU 0 three times when the box has one GPU, so the
    // contiguous-block split (ragged: 10 gates over 3 entries) is exercised either way.
    HomNAND<> h(32);
    auto sk = h.generate_secret_key();
    auto ek = h.make_evaluate_key_gpu(sk, 9);
    std::vector<HomNAND<>::Cipher> a, b;
    std::vector<bool> pa, pb;
    for (int i = 0; i < 10; i++) {
        pa.push_back(i & 1);
        pb.push_back(i & 2);
        a.push_back(h.encrypt(pa.back(), sk));
        b.push_back(h.encrypt(pb.back(), sk));
    }
    auto one = h.nand_batch(a, b, ek);
    gpu::devices() = {0, 0, 0};
    auto many = h.nand_batch(a, b, ek);
    auto many_or = h.gate_batch(HomNAND<>::Gate::OR, a, b, ek);
    auto single = h.nand(a[7], b[7], ek);  // single gates run on the first device of the list
    gpu::devices().clear();
    for (int i = 0; i < 10; i++) {
        CHECK(many[i] == one[i]);
        CHECK(h.decrypt(many[i], sk) == !(pa[i] && pb[i]));
        CHECK(h.decrypt(many_or[i], sk) == (pa[i] || pb[i]));
    }
    CHECK(single == one[7]);
    // a key edited in place: invalidate() drops the cached device copy
    ek.bk[0][0][0][0] += TorusValue::from_raw_value(0x01234567u);
    h.invalidate(ek);
    CHECK(!(h.nand(a[7], b[7], ek) == one[7]));
}

static void run_gpu() {
    test_external_product_by_one();
    test_cmux();
    test_bootstrapping();
    test_key_switching();
    test_hom_nand();
    test_other_gates();
    test_gpu_keygen();
    test_reference_fft_mode();
    test_hom_nand_other_ring();
    test_concurrent_nand_on_shared_key();
    test_multi_device_batch();
    std::puts("gpu group ok");
}

int main(int argc, char** argv) {
    const char* group = argc > 1 ? argv[1] : "host";
    if (!std::strcmp(group, "host") || !std::strcmp(group, "all")) run_host();
    if (!std::strcmp(group, "gpu") || !std::strcmp(group, "all")) run_gpu();
    return 0;
}
