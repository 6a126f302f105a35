// TEST INFRASTRUCTURE ONLY -- golden-vector generator.
//
// Compiled in the development container against the reference headers where they lie
// (-I /root/reference/src); it is never built or run on the GPU box.  It drives the
// reference's own classes and dumps inputs and outputs as .npy files, which
// oracle/make_golden.py packs into tests/golden/*.npz.  No reference source is copied:
// this file only calls the reference's public (and, for `decomposition`, private) members.
//
// Everything dumped here comes from code paths where the reference is exact integer
// arithmetic: ring degree < 32 for anything that multiplies polynomials
// (reference src/types.hpp:225-236), plus integer-only primitives at the default sizes.
// The one exception is `extprod512_*`: the reference's floating-point multiply at N = 512,
// recorded to pin the documented {-6..0} ulp relation.
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#define private public
#include "hom_nand.hpp"
#undef private

using namespace TFHE;

static std::string g_dir;

static void write_npy(const std::string& name, const char* descr, const std::vector<size_t>& shape,
                      const void* data, size_t nbytes) {
    std::string path = g_dir + "/" + name + ".npy";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    std::ostringstream hd;
    hd << "{'descr': '" << descr << "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); i++) hd << shape[i] << (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    hd << "), }";
    std::string h = hd.str();
    size_t total = 10 + h.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    h.append(pad, ' ');
    h.push_back('\n');
    unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(h.size() & 0xff),
                             (unsigned char)(h.size() >> 8)};
    fwrite(pre, 1, 10, f);
    fwrite(h.data(), 1, h.size(), f);
    fwrite(data, 1, nbytes, f);
    fclose(f);
}
static void dump_u32(const std::string& n, const std::vector<size_t>& s, const std::vector<uint32_t>& v) {
    write_npy(n, "<u4", s, v.data(), v.size() * 4);
}
static void dump_i32(const std::string& n, const std::vector<size_t>& s, const std::vector<int32_t>& v) {
    write_npy(n, "<i4", s, v.data(), v.size() * 4);
}
static void dump_u8(const std::string& n, const std::vector<size_t>& s, const std::vector<uint8_t>& v) {
    write_npy(n, "|u1", s, v.data(), v.size());
}

template <int N>
static void put(std::vector<uint32_t>& out, const Vector<TorusValue, N>& v) {
    for (int i = 0; i < N; i++) out.push_back(v[i].get_raw_value());
}
template <int N>
static void put(std::vector<uint32_t>& out, const Polynomial<TorusValue, N>& v) {
    for (int i = 0; i < N; i++) out.push_back(v[i].get_raw_value());
}
template <int N, int K1>
static void put(std::vector<uint32_t>& out, const Vector<Polynomial<TorusValue, N>, K1>& v) {
    for (int q = 0; q < K1; q++) put(out, v[q]);
}

static std::mt19937 g_mt(20261111);
static uint32_t rnd32() { return (uint32_t)g_mt(); }

// ---------------------------------------------------------------- toy-parameter pipeline
template <typename P0, typename PR, typename P1, typename PG, typename PK>
static void gen_pipeline(const std::string& tag, int n_gates, int n_traced) {
    constexpr int n = P0::N, N = PR::N, K = PR::K, L = PG::L, BG = PG::BG_BIT, T = PK::T, BB = PK::BASE_BIT;
    static_assert(N < 32, "exact branch only");
    static_assert(P1::N == N * K, "lv1 size");
    TLWE<P0> tlwe0;
    TLWE<P1> tlwe1;
    TRLWE<PR> trlwe;
    TRGSW<PG, PR> trgsw(trlwe);
    TLWEKeySwitching<TLWE<P0>, TLWE<P1>, PK> ksw(tlwe0);

    auto s0 = tlwe0.generate_s();
    auto sr = trlwe.generate_s();
    auto s1 = trlwe.extract_tlwe_lv0_key(sr);
    auto bk = trgsw.make_bk(s0, sr);
    auto ks = ksw.make_ks(s0, s1);

    std::vector<int32_t> params = {n, N, K, BG, L, T, BB};
    dump_i32(tag + "_params", {7}, params);
    std::vector<uint8_t> s0v, srv;
    for (int i = 0; i < n; i++) s0v.push_back(s0[i]);
    for (int k = 0; k < K; k++) for (int i = 0; i < N; i++) srv.push_back(sr[k][i]);
    dump_u8(tag + "_s_lv0", {(size_t)n}, s0v);
    dump_u8(tag + "_s_ring", {(size_t)K, (size_t)N}, srv);

    std::vector<uint32_t> bkv, ksv;
    for (int i = 0; i < n; i++)
        for (int r = 0; r < (K + 1) * L; r++)
            for (int c = 0; c < K + 1; c++) put(bkv, bk[i][r][c]);
    dump_u32(tag + "_bk", {(size_t)n, (size_t)(K + 1) * L, (size_t)K + 1, (size_t)N}, bkv);
    for (int i = 0; i < N * K; i++)
        for (int m = 0; m < T; m++)
            for (int o = 0; o < (1 << BB); o++) put(ksv, ks[i][m][o]);
    dump_u32(tag + "_ks", {(size_t)N * K, (size_t)T, (size_t)1 << BB, (size_t)n + 1}, ksv);

    std::vector<uint32_t> av, bv, cv, lv1v, outv, accv, trace;
    std::vector<uint8_t> bits_a, bits_b, bits_out;
    for (int g = 0; g < n_gates; g++) {
        bool pa = g_mt() & 1, pb = g_mt() & 1;
        auto ca = tlwe0.encrypt_single_binary(pa, s0);
        auto cb = tlwe0.encrypt_single_binary(pb, s0);
        // the linear step of HomNAND::nand, written with the reference's own operators
        Vector<TorusValue, n + 1> c;
        c[0] = TorusValue(true);
        c -= ca + cb;
        Vector<Polynomial<TorusValue, N>, K + 1> tv;
        std::fill(tv[0].begin(), tv[0].end(), TorusValue(true));
        auto acc = trgsw.blind_rotate(c, bk, tv);
        auto lv1 = trgsw.gate_bootstrapping_tlwe_to_tlwe(c, bk);
        auto out = ksw.identity_key_switch(lv1, ks);
        bool dec = tlwe0.decrypt_single_binary(out, s0);
        put(av, ca); put(bv, cb); put(cv, c); put(accv, acc); put(lv1v, lv1); put(outv, out);
        bits_a.push_back(pa); bits_b.push_back(pb); bits_out.push_back(dec);
        if (g < n_traced) {
            // the loop of blind_rotate replayed through the reference's public members
            int nbit = 0; while ((1 << nbit) < N) nbit++;
            int b2n = c[0].get_raw_value() >> (32 - (nbit + 1));
            Vector<Polynomial<TorusValue, N>, K + 1> ret;
            for (int q = 0; q < K + 1; q++) ret[q] = tv[q].multiply_x_exp(-b2n);
            put(trace, ret);
            for (int i = 0; i < n; i++) {
                int a2n = (c[1 + i].get_raw_value() + (1U << (32 - (nbit + 1) - 1))) >> (32 - (nbit + 1));
                Vector<Polynomial<TorusValue, N>, K + 1> rot;
                for (int q = 0; q < K + 1; q++) rot[q] = ret[q].multiply_x_exp(a2n);
                auto sel = trgsw.cmux(bk[i], ret, rot);
                ret = sel;
                put(trace, ret);
            }
            if (!(ret == acc)) { fprintf(stderr, "trace replay mismatch\n"); exit(2); }
        }
    }
    size_t G = n_gates;
    dump_u32(tag + "_a", {G, (size_t)n + 1}, av);
    dump_u32(tag + "_b", {G, (size_t)n + 1}, bv);
    dump_u32(tag + "_c", {G, (size_t)n + 1}, cv);
    dump_u32(tag + "_acc", {G, (size_t)K + 1, (size_t)N}, accv);
    dump_u32(tag + "_lv1", {G, (size_t)N * K + 1}, lv1v);
    dump_u32(tag + "_out", {G, (size_t)n + 1}, outv);
    dump_u32(tag + "_trace", {(size_t)n_traced, (size_t)n + 1, (size_t)K + 1, (size_t)N}, trace);
    dump_u8(tag + "_bit_a", {G}, bits_a);
    dump_u8(tag + "_bit_b", {G}, bits_b);
    dump_u8(tag + "_bit_out", {G}, bits_out);

    // a few stand-alone external products / cmux on random (not encrypted) data
    std::vector<uint32_t> ep_in, ep_gsw, ep_out, cm_d0, cm_d1, cm_out;
    for (int t = 0; t < 8; t++) {
        Matrix<Polynomial<TorusValue, N>, (K + 1) * L, K + 1> C;
        Vector<Polynomial<TorusValue, N>, K + 1> x, d0, d1;
        for (int r = 0; r < (K + 1) * L; r++)
            for (int c = 0; c < K + 1; c++)
                for (int j = 0; j < N; j++) C[r][c][j] = TorusValue::from_raw_value(rnd32());
        for (int q = 0; q < K + 1; q++)
            for (int j = 0; j < N; j++) {
                x[q][j] = TorusValue::from_raw_value(rnd32());
                d0[q][j] = TorusValue::from_raw_value(rnd32());
                d1[q][j] = TorusValue::from_raw_value(rnd32());
            }
        auto e = trgsw.external_product(C, x);
        auto s = trgsw.cmux(C, d0, d1);
        for (int r = 0; r < (K + 1) * L; r++) for (int c = 0; c < K + 1; c++) put(ep_gsw, C[r][c]);
        put(ep_in, x); put(ep_out, e); put(cm_d0, d0); put(cm_d1, d1); put(cm_out, s);
    }
    dump_u32(tag + "_ep_trgsw", {8, (size_t)(K + 1) * L, (size_t)K + 1, (size_t)N}, ep_gsw);
    dump_u32(tag + "_ep_in", {8, (size_t)K + 1, (size_t)N}, ep_in);
    dump_u32(tag + "_ep_out", {8, (size_t)K + 1, (size_t)N}, ep_out);
    dump_u32(tag + "_cmux_d0", {8, (size_t)K + 1, (size_t)N}, cm_d0);
    dump_u32(tag + "_cmux_d1", {8, (size_t)K + 1, (size_t)N}, cm_d1);
    dump_u32(tag + "_cmux_out", {8, (size_t)K + 1, (size_t)N}, cm_out);
}

// ---------------------------------------------------------------- integer primitives, real sizes
static const uint32_t kEdge[] = {0u, 1u, 0x7FFFu, 0x8000u, 0x8001u, 0xFFFFu, 0x10000u, 0x7F7FFFu, 0x7F8000u,
    0x7FFFFFu, 0x800000u, 0x807FFFu, 0x808000u, 0xFF7FFFu, 0xFF8000u, 0xFFFFFFu, 0x1000000u, 0x7F7F8000u,
    0x7FFF7FFFu, 0x7FFF8000u, 0x7FFFFFFFu, 0x80000000u, 0x80007FFFu, 0x80008000u, 0xFF7F7FFFu, 0xFF7F8000u,
    0xFFFF7FFFu, 0xFFFF8000u, 0xFFFFFFFFu, 0x00FF8000u, 0x7FFFFFu + 0x8000u, 0xFFFFFF00u, 0x3Fu, 0x40u, 0x7Fu,
    0x80u, 0xBFu, 0xC0u, 0xFFu, 0x100u};

template <typename PG, typename PR>
static void gen_decomposition(const std::string& tag, int n_polys) {
    constexpr int N = PR::N, L = PG::L;
    TRLWE<PR> trlwe;
    TRGSW<PG, PR> trgsw(trlwe);
    std::vector<uint32_t> in;
    std::vector<int32_t> out;
    for (int t = 0; t < n_polys; t++) {
        Polynomial<TorusValue, N> a;
        for (int j = 0; j < N; j++) {
            uint32_t v = rnd32();
            if (t == 0) v = kEdge[j % (sizeof(kEdge) / sizeof(kEdge[0]))] + (j / 40) * 0x01010101u * (j % 3 == 0);
            a[j] = TorusValue::from_raw_value(v);
        }
        auto d = trgsw.decomposition(a);
        put(in, a);
        for (int l = 0; l < L; l++) for (int j = 0; j < N; j++) out.push_back((int32_t)d[l][j]);
    }
    std::vector<int32_t> params = {0, N, PG::K, PG::BG_BIT, L, 0, 0};
    dump_i32(tag + "_params", {7}, params);
    dump_u32(tag + "_in", {(size_t)n_polys, (size_t)N}, in);
    dump_i32(tag + "_digits", {(size_t)n_polys, (size_t)L, (size_t)N}, out);
}

template <int N>
static void gen_mul_x_exp(const std::string& tag) {
    std::vector<int32_t> shifts = {0, 1, -1, N - 1, N, N + 1, 2 * N - 1, 2 * N, 2 * N + 1, -N, -N + 1, -N - 1,
                                   -2 * N + 1, -2 * N, -2 * N - 1, 4 * N + 3, -4 * N - 3, 7, -7, N / 2, -N / 2};
    for (int t = 0; t < 20; t++) shifts.push_back((int32_t)(g_mt() % (8 * N)) - 4 * N);
    Polynomial<TorusValue, N> a;
    for (int j = 0; j < N; j++) a[j] = TorusValue::from_raw_value(rnd32());
    std::vector<uint32_t> in, out;
    put(in, a);
    for (int s : shifts) put(out, a.multiply_x_exp(s));
    dump_u32(tag + "_in", {(size_t)N}, in);
    dump_i32(tag + "_shifts", {shifts.size()}, shifts);
    dump_u32(tag + "_out", {shifts.size(), (size_t)N}, out);
}

template <typename PR>
static void gen_sample_extract(const std::string& tag) {
    constexpr int N = PR::N, K = PR::K;
    TRLWE<PR> trlwe;
    Vector<Polynomial<TorusValue, N>, K + 1> x;
    for (int q = 0; q < K + 1; q++) for (int j = 0; j < N; j++) x[q][j] = TorusValue::from_raw_value(rnd32());
    std::vector<int32_t> idx = {0, 1, 5, N / 2, N - 2, N - 1};
    std::vector<uint32_t> in, out;
    put(in, x);
    for (int i : idx) put(out, trlwe.sample_extract_index(x, i));
    std::vector<int32_t> params = {0, N, K, 0, 0, 0, 0};
    dump_i32(tag + "_params", {7}, params);
    dump_u32(tag + "_in", {(size_t)K + 1, (size_t)N}, in);
    dump_i32(tag + "_index", {idx.size()}, idx);
    dump_u32(tag + "_out", {idx.size(), (size_t)N * K + 1}, out);
}

template <typename P0, typename P1, typename PK>
static void gen_key_switch(const std::string& tag, int n_in) {
    constexpr int n = P0::N, n1 = P1::N, T = PK::T, BB = PK::BASE_BIT;
    TLWE<P0> tlwe0;
    TLWEKeySwitching<TLWE<P0>, TLWE<P1>, PK> ksw(tlwe0);
    Vector<Vector<Vector<Vector<TorusValue, n + 1>, (1 << BB)>, T>, n1> ks;
    std::vector<uint32_t> ksv, in, out;
    for (int i = 0; i < n1; i++)
        for (int m = 0; m < T; m++)
            for (int o = 0; o < (1 << BB); o++) {
                // slot 0 is filled with junk on purpose: the reference never reads it
                for (int w = 0; w < n + 1; w++) ks[i][m][o][w] = TorusValue::from_raw_value(rnd32());
                put(ksv, ks[i][m][o]);
            }
    const int n_edge = sizeof(kEdge) / sizeof(kEdge[0]);
    for (int t = 0; t < n_in; t++) {
        Vector<TorusValue, n1 + 1> x;
        for (int j = 0; j < n1 + 1; j++) {
            uint32_t v = rnd32();
            if (t == 0) v = kEdge[j % n_edge] << (j % 5 == 0 ? 6 : 0);
            if (t == 1) v = (v & 0xFFC00000u) + (j % 3 == 0 ? 0x1FFFFFu : j % 3 == 1 ? 0x200000u : 0x3FFFFFu);
            x[j] = TorusValue::from_raw_value(v);
        }
        put(in, x);
        put(out, ksw.identity_key_switch(x, ks));
    }
    std::vector<int32_t> params = {n, n1, 1, 0, 0, T, BB};
    dump_i32(tag + "_params", {7}, params);
    dump_u32(tag + "_ks", {(size_t)n1, (size_t)T, (size_t)1 << BB, (size_t)n + 1}, ksv);
    dump_u32(tag + "_in", {(size_t)n_in, (size_t)n1 + 1}, in);
    dump_u32(tag + "_out", {(size_t)n_in, (size_t)n + 1}, out);
}

// the linear step of HomNAND::nand at the default lv0 size
static void gen_nand_combine(const std::string& tag) {
    constexpr int n = TLWEParameter128BitSecurity::N;
    std::vector<uint32_t> av, bv, cv;
    for (int t = 0; t < 8; t++) {
        Vector<TorusValue, n + 1> a, b, c;
        for (int j = 0; j < n + 1; j++) {
            a[j] = TorusValue::from_raw_value(rnd32());
            b[j] = TorusValue::from_raw_value(rnd32());
        }
        c[0] = TorusValue(true);
        c -= a + b;
        put(av, a); put(bv, b); put(cv, c);
    }
    dump_u32(tag + "_a", {8, (size_t)n + 1}, av);
    dump_u32(tag + "_b", {8, (size_t)n + 1}, bv);
    dump_u32(tag + "_c", {8, (size_t)n + 1}, cv);
}

// rotation amounts of blind_rotate at N = 512: read back from a rotation of a marker test vector
template <typename PR, typename PG>
static void gen_rotation_amounts(const std::string& tag) {
    constexpr int N = PR::N, K = PR::K, L = PG::L;
    TRLWE<PR> trlwe;
    TRGSW<PG, PR> trgsw(trlwe);
    // with TLWE_N = 0 blind_rotate is only the initial rotation by -floor(b*2N)
    std::vector<uint32_t> bs, outs;
    for (int t = 0; t < 48; t++) {
        uint32_t b = t < 40 ? kEdge[t] << (t % 2 ? 16 : 0) : rnd32();
        Vector<TorusValue, 1> c;
        c[0] = TorusValue::from_raw_value(b);
        Vector<Matrix<Polynomial<TorusValue, N>, (K + 1) * L, K + 1>, 0> bk0;
        Vector<Polynomial<TorusValue, N>, K + 1> tv;
        for (int j = 0; j < N; j++) tv[0][j] = TorusValue::from_raw_value(j + 1);
        auto r = trgsw.template blind_rotate<0>(c, bk0, tv);
        bs.push_back(b);
        put(outs, r[0]);
    }
    dump_u32(tag + "_b", {bs.size()}, bs);
    dump_u32(tag + "_out", {bs.size(), (size_t)N}, outs);
}

// reference floating-point external product at N = 512 (inexact by design of the reference)
static void gen_extprod512(const std::string& tag) {
    constexpr int N = 512, K = 2, L = 2;
    TRLWE<> trlwe;
    TRGSW<> trgsw(trlwe);
    std::vector<uint32_t> gsw, in, out;
    for (int t = 0; t < 2; t++) {
        Matrix<Polynomial<TorusValue, N>, (K + 1) * L, K + 1> C;
        Vector<Polynomial<TorusValue, N>, K + 1> x;
        for (int r = 0; r < (K + 1) * L; r++)
            for (int c = 0; c < K + 1; c++) {
                for (int j = 0; j < N; j++) C[r][c][j] = TorusValue::from_raw_value(rnd32());
                put(gsw, C[r][c]);
            }
        for (int q = 0; q < K + 1; q++) for (int j = 0; j < N; j++) x[q][j] = TorusValue::from_raw_value(rnd32());
        put(in, x);
        put(out, trgsw.external_product(C, x));
    }
    dump_u32(tag + "_trgsw", {2, (size_t)(K + 1) * L, (size_t)K + 1, (size_t)N}, gsw);
    dump_u32(tag + "_in", {2, (size_t)K + 1, (size_t)N}, in);
    dump_u32(tag + "_out", {2, (size_t)K + 1, (size_t)N}, out);
}

#define PARAM_TLWE(NAME, NN, AL) struct NAME { constexpr static int N = NN; constexpr static double ALPHA = AL; }
#define PARAM_RING(NAME, NN, KK, AL) \
    struct NAME { constexpr static int N = NN; constexpr static int K = KK; constexpr static double ALPHA = AL; }
#define PARAM_GSW(NAME, KK, BG, LL) \
    struct NAME { constexpr static int K = KK; constexpr static int BG_BIT = BG; constexpr static int L = LL; }
#define PARAM_KS(NAME, TT, BB) struct NAME { constexpr static int T = TT; constexpr static int BASE_BIT = BB; }

PARAM_TLWE(A0, 10, 9.5367431640625e-07);
PARAM_RING(AR, 16, 2, 2.9802322387695312e-08);
PARAM_TLWE(A1, 32, 2.9802322387695312e-08);
PARAM_GSW(AG, 2, 8, 2);
PARAM_KS(AK, 5, 2);

PARAM_TLWE(B0, 12, 9.5367431640625e-07);
PARAM_RING(BR, 16, 1, 2.9802322387695312e-08);
PARAM_TLWE(B1, 16, 2.9802322387695312e-08);
PARAM_GSW(BG_, 1, 8, 3);
PARAM_KS(BK_, 8, 2);

PARAM_TLWE(C0, 9, 9.5367431640625e-07);
PARAM_RING(CR, 8, 3, 2.9802322387695312e-08);
PARAM_TLWE(C1, 24, 2.9802322387695312e-08);
PARAM_GSW(CG, 3, 16, 1);
PARAM_KS(CK, 4, 3);

PARAM_RING(R2048, 2048, 1, 2.9802322387695312e-08);
PARAM_GSW(G2048, 1, 8, 3);
PARAM_GSW(G512x16, 2, 16, 1);
PARAM_TLWE(KS0, 8, 9.5367431640625e-07);
PARAM_TLWE(KS1, 64, 2.9802322387695312e-08);
PARAM_TLWE(KS1b, 48, 2.9802322387695312e-08);
PARAM_KS(KS8, 8, 2);
PARAM_KS(KS3x3, 3, 3);

PARAM_TLWE(KS1c, 24, 2.9802322387695312e-08);
PARAM_TLWE(KS1d, 12, 2.9802322387695312e-08);
PARAM_KS(KS4x4, 4, 4);
PARAM_KS(KS6x5, 6, 5);
PARAM_KS(KS3x7, 3, 7);
PARAM_KS(KS1x8, 1, 8);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <out_dir> [family]\n", argv[0]); return 1; }
    g_dir = argv[1];
    if (argc > 2 && std::string(argv[2]) == "kswide") {
        // identity key switch at the wider bases the reference accepts (T * BASE_BIT <= 31), a family of
        // its own so that it can be added without regenerating the others
        gen_key_switch<KS0, KS1c, KS4x4>("kswide_t4b4", 8);
        gen_key_switch<KS0, KS1c, KS6x5>("kswide_t6b5", 8);
        gen_key_switch<KS0, KS1d, KS3x7>("kswide_t3b7", 8);
        gen_key_switch<KS0, KS1d, KS1x8>("kswide_t1b8", 8);
        return 0;
    }
    gen_pipeline<A0, AR, A1, AG, AK>("toyA", 64, 4);
    gen_pipeline<B0, BR, B1, BG_, BK_>("toyB", 64, 4);
    gen_pipeline<C0, CR, C1, CG, CK>("toyC", 32, 2);
    gen_decomposition<TRGSWParameterDefault, TRLWEParameterDefault>("dec512", 6);
    gen_decomposition<G2048, R2048>("dec2048", 2);
    gen_decomposition<G512x16, TRLWEParameterDefault>("dec512x16", 2);
    gen_mul_x_exp<512>("xexp512");
    gen_mul_x_exp<2048>("xexp2048");
    gen_mul_x_exp<16>("xexp16");
    gen_sample_extract<TRLWEParameterDefault>("extract512");
    gen_sample_extract<R2048>("extract2048");
    gen_key_switch<KS0, KS1, TLWEKeySwitchingParameterDefault>("ks_t5", 8);
    gen_key_switch<KS0, KS1b, KS8>("ks_t8", 8);
    gen_key_switch<KS0, KS1b, KS3x3>("ks_t3b3", 8);
    gen_nand_combine("combine636");
    gen_rotation_amounts<TRLWEParameterDefault, TRGSWParameterDefault>("rot512");
    gen_extprod512("extprod512");
    return 0;
}
