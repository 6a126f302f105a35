// Host-side exact arithmetic in the prime field the device NTT works over, and the
// twiddle tables derived from it.
//
// The field is F_p with p = 6432^4 + 1 = 1711528530149377 (a 51-bit generalised Fermat
// prime, p - 1 = 2^20 * 3^4 * 67^4).  Why this prime:
//   * p > 2 * (K+1)*L*N * 2^(bg_bit-1) * 2^31 = 3 * 2^49 for the default parameter set, so the
//     centred residue of every negacyclic accumulation is the true integer and the result
//     mod 2^32 is exact;
//   * p < 2^53 / 5.26, so elements and a few lazy additions of them are exact integers in an
//     IEEE double and one v_fma_f64 recovers the low half of a product;
//   * zeta = 6432 has zeta^4 = -1: the first forward butterfly stages on 8-bit gadget digits
//     multiply by zeta^2 (26 bits) or zeta (13 bits) without any reduction;
//   * 2^20 | p - 1, so 2N-th roots of unity exist up to N = 2^19.
#pragma once

#include <cstdint>
#include <vector>

namespace tfhe_amd {

constexpr uint64_t kP = 1711528530149377ULL;
constexpr uint64_t kZeta = 6432ULL;

struct Field {
    using u128 = unsigned __int128;
    static uint64_t mul(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % kP); }
    static uint64_t add(uint64_t a, uint64_t b) { uint64_t s = a + b; return s >= kP ? s - kP : s; }
    static uint64_t sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + kP - b; }
    static uint64_t pow(uint64_t a, uint64_t e) {
        uint64_t r = 1;
        while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
        return r;
    }
    static uint64_t inv(uint64_t a) { return pow(a, kP - 2); }
    // centred representative in (-p/2, p/2) as a double (exact: |v| < 2^50)
    static double centred(uint64_t v) { return v > kP / 2 ? -(double)(kP - v) : (double)v; }
    static uint64_t from_i64(int64_t v) { return v >= 0 ? (uint64_t)v % kP : kP - ((uint64_t)(-v) % kP); }

    // primitive 2N-th root of unity psi with psi^(N/4) = zeta (so psi^(N/2) = zeta^2 = sqrt(-1))
    static uint64_t psi(int N) {
        const uint64_t g = 5;  // generates F_p^*: 5^((p-1)/q) != 1 for q in {2, 3, 67}
        const uint64_t psi0 = pow(g, (kP - 1) / (2 * (uint64_t)N));
        for (uint64_t j = 1; j < 8; j += 2) {
            const uint64_t c = pow(psi0, j);
            if (pow(c, (uint64_t)N / 4) == kZeta) return c;
        }
        return 0;
    }
};

inline int bit_reverse(int v, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++) if (v & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}
inline int log2i(int v) { int b = 0; while ((1 << b) < v) b++; return b; }

// Twiddle pair as the device consumes it: w = centred residue, wp = RN(w / p).
struct TwPair { double w, wp; };
inline TwPair tw_pair(uint64_t v) {
    const double w = Field::centred(v);
    return {w, w / (double)kP};
}

// Merged negacyclic twiddles in Cooley-Tukey order: fwd[m + i] = psi^bitrev(m + i) for stage
// m = 1, 2, 4, .., N/2 and group i < m; inv[m + i] = psi^-bitrev(m + i).
struct TwiddleSet {
    int N;
    std::vector<uint64_t> fwd, inv;  // [N], entry 0 unused
    uint64_t n_inv;
};
inline TwiddleSet make_twiddles(int N) {
    TwiddleSet t;
    t.N = N;
    const int lg = log2i(N);
    const uint64_t psi = Field::psi(N), psi_inv = Field::inv(psi);
    t.fwd.assign(N, 0);
    t.inv.assign(N, 0);
    for (int i = 1; i < N; i++) {
        const int r = bit_reverse(i, lg);
        t.fwd[i] = Field::pow(psi, (uint64_t)r);
        t.inv[i] = Field::pow(psi_inv, (uint64_t)r);
    }
    t.n_inv = Field::inv((uint64_t)N);
    return t;
}

}  // namespace tfhe_amd
