// Drop-in for KowerKoint/TFHE's src/types.hpp: the torus scalar and the fixed-size
// containers, same names and signatures, written from scratch.
//
// Differences from the reference that a caller can observe:
//   * Polynomial::operator* is the exact negacyclic convolution for every ring degree.  The
//     reference takes that route only for N < 32 (src/types.hpp:225-236) and otherwise goes
//     through a double-precision FFT whose result is the exact one or one unit lower
//     (src/types.hpp:237-292).  Here there is one definition, in wrap-around arithmetic of the
//     result type.
//   * the stray namespace-scope `int p` (src/types.hpp:10) is `inline`, so the headers can be
//     included from several translation units.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <iterator>
#include <memory>
#include <type_traits>
#include <utility>

namespace TFHE {

inline int p = 0;

// ---- integer width helpers (reference src/types.hpp:11-19) ---------------------------------
template <int N>
struct is_valid_int_bits : std::integral_constant<bool, N == 8 || N == 16 || N == 32> {};
template <int N>
constexpr bool is_valid_int_bits_v = is_valid_int_bits<N>::value;

namespace detail {
template <int N> struct signed_int_of;
template <> struct signed_int_of<8> { using type = int8_t; };
template <> struct signed_int_of<16> { using type = int16_t; };
template <> struct signed_int_of<32> { using type = int32_t; };
}  // namespace detail
template <int N, std::enable_if_t<is_valid_int_bits_v<N>>* = nullptr>
using SignedInt = typename detail::signed_int_of<N>::type;

namespace detail {
// element type of a product of containers
template <typename A, typename B>
using product_t = decltype(std::declval<A>() * std::declval<B>());
}  // namespace detail

// ---- torus scalar (reference src/types.hpp:21-92) -------------------------------------------
// An element of R/Z stored as raw / 2^32; all arithmetic wraps modulo 2^32.
class TorusValue {
    uint32_t raw_ = 0;

public:
    constexpr static uint32_t MU = 1U << 29;  // 1/8

    constexpr TorusValue() = default;
    // fractional part of d, scaled by 2^32 and truncated (reference src/types.hpp:28-33)
    explicit TorusValue(double d) {
        double frac = std::fmod(d, 1.0);
        if (frac < 0.0) frac += 1.0;
        raw_ = static_cast<uint32_t>(frac * 4294967296.0);
    }
    // true -> +1/8, false -> -1/8 (reference src/types.hpp:34)
    constexpr explicit TorusValue(bool b) : raw_(b ? MU : 0U - MU) {}
    constexpr static TorusValue from_raw_value(uint32_t v) {
        TorusValue t;
        t.raw_ = v;
        return t;
    }

    constexpr explicit operator double() const { return static_cast<double>(raw_) / 4294967296.0; }
    // decode: the half of the torus around +1/8 is "true" (reference src/types.hpp:46)
    constexpr explicit operator bool() const { return (raw_ >> 31) == 0U; }
    constexpr const uint32_t& get_raw_value() const { return raw_; }
    constexpr uint32_t& get_raw_value() { return raw_; }

    constexpr TorusValue operator+() const { return *this; }
    constexpr TorusValue operator-() const { return from_raw_value(0U - raw_); }
    constexpr TorusValue& operator+=(const TorusValue& o) { raw_ += o.raw_; return *this; }
    constexpr TorusValue& operator-=(const TorusValue& o) { raw_ -= o.raw_; return *this; }
    constexpr TorusValue operator+(const TorusValue& o) const { return from_raw_value(raw_ + o.raw_); }
    constexpr TorusValue operator-(const TorusValue& o) const { return from_raw_value(raw_ - o.raw_); }

    template <typename Scalar, std::enable_if_t<std::is_scalar_v<Scalar>>* = nullptr>
    constexpr TorusValue& operator*=(Scalar k) { raw_ *= k; return *this; }
    template <typename Scalar, std::enable_if_t<std::is_scalar_v<Scalar>>* = nullptr>
    constexpr TorusValue operator*(Scalar k) const { TorusValue t(*this); t *= k; return t; }
    template <typename Scalar, std::enable_if_t<std::is_scalar_v<Scalar>>* = nullptr>
    friend constexpr TorusValue operator*(Scalar k, const TorusValue& t) { return from_raw_value(t.raw_ * k); }

    constexpr bool operator==(const TorusValue& o) const { return raw_ == o.raw_; }
    constexpr bool operator!=(const TorusValue& o) const { return raw_ != o.raw_; }
};

// ---- fixed-size heap block shared by Vector and Polynomial (reference src/types.hpp:94-147) --
// Value semantics: copying copies the N elements; elements start value-initialised.
template <typename Value, int N, typename Self>
class _VectorOrPolynomial {
protected:
    std::unique_ptr<Value[]> _val;

    Self& self() { return static_cast<Self&>(*this); }
    const Self& self() const { return static_cast<const Self&>(*this); }

    static Value* fresh() { return new Value[N > 0 ? N : 1](); }

public:
    _VectorOrPolynomial() : _val(fresh()) {}
    _VectorOrPolynomial(const std::unique_ptr<Value[]>& src) : _val(fresh()) { std::copy_n(src.get(), N, _val.get()); }
    _VectorOrPolynomial(std::unique_ptr<Value[]>&& src) : _val(std::move(src)) {}
    // the containers built on this block take their copy and move operations from here
    _VectorOrPolynomial(const _VectorOrPolynomial& o) : _VectorOrPolynomial(o._val) {}
    _VectorOrPolynomial(_VectorOrPolynomial&&) = default;
    _VectorOrPolynomial& operator=(const _VectorOrPolynomial& o) {
        if (this != &o) std::copy_n(o._val.get(), N, _val.get());
        return *this;
    }
    _VectorOrPolynomial& operator=(_VectorOrPolynomial&&) = default;

    Value* begin() { return _val.get(); }
    const Value* begin() const { return _val.get(); }
    Value* end() { return _val.get() + N; }
    const Value* end() const { return _val.get() + N; }
    Value& operator[](int i) { return _val[i]; }
    const Value& operator[](int i) const { return _val[i]; }

    Self operator+() const { return Self(self()); }
    Self operator-() const {
        Self r(self());
        for (int i = 0; i < N; i++) r[i] = -_val[i];
        return r;
    }
    Self& operator+=(const Self& o) {
        for (int i = 0; i < N; i++) _val[i] += o[i];
        return self();
    }
    Self& operator-=(const Self& o) {
        for (int i = 0; i < N; i++) _val[i] -= o[i];
        return self();
    }
    Self operator+(const Self& o) const { Self r(self()); r += o; return r; }
    Self operator-(const Self& o) const { Self r(self()); r -= o; return r; }
    bool operator==(const Self& o) const { return std::equal(begin(), end(), o.begin()); }
    bool operator!=(const Self& o) const { return !(*this == o); }
};

template <typename Value, int N>
class Polynomial;

// Discrete Fourier transform X[k] = sum_j a[j] * omega_pow[1]^(jk), with the caller's table
// omega_pow[i] = omega^i (reference src/types.hpp:152-197; only its own test_FFT uses it, the
// polynomial product here does not).  Iterative radix-2, N a power of two.
template <int N>
Polynomial<std::complex<double>, N> fft(const Polynomial<std::complex<double>, N>& a,
                                        const std::array<std::complex<double>, (size_t)N>& omega_pow) {
    static_assert(N > 0 && (N & (N - 1)) == 0, "N must be a power of two");
    Polynomial<std::complex<double>, N> x;
    int bits = 0;
    while ((1 << bits) < N) bits++;
    for (int i = 0; i < N; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
        x[r] = a[i];
    }
    for (int len = 2; len <= N; len <<= 1) {
        const int half = len / 2, stride = N / len;
        for (int base = 0; base < N; base += len)
            for (int k = 0; k < half; k++) {
                const std::complex<double> t = x[base + half + k] * omega_pow[(size_t)k * stride];
                x[base + half + k] = x[base + k] - t;
                x[base + k] += t;
            }
    }
    return x;
}

// ---- polynomial modulo X^N + 1 (reference src/types.hpp:199-313) -----------------------------
template <typename Value, int N>
class Polynomial : public _VectorOrPolynomial<Value, N, Polynomial<Value, N>> {
    using Base = _VectorOrPolynomial<Value, N, Polynomial>;
    friend Base;

public:
    using Base::Base;  // default, from an owned block, from a copied block
    Polynomial() = default;
    Polynomial(const Polynomial&) = default;
    Polynomial(Polynomial&&) = default;
    Polynomial& operator=(const Polynomial&) = default;
    Polynomial& operator=(Polynomial&&) = default;

    // Negacyclic product: r_k = sum_{i+j=k} a_i b_j - sum_{i+j=k+N} a_i b_j, accumulated in the
    // result type (wrap-around for integers and TorusValue).
    template <typename RHSValue>
    Polynomial<detail::product_t<Value, RHSValue>, N> operator*(const Polynomial<RHSValue, N>& rhs) const {
        Polynomial<detail::product_t<Value, RHSValue>, N> out;
        for (int i = 0; i < N; i++) {
            const Value& ai = (*this)[i];
            if (ai == Value{}) continue;
            int k = i;
            for (int j = 0; j < N - i; j++, k++) out[k] += ai * rhs[j];
            k = 0;
            for (int j = N - i; j < N; j++, k++) out[k] -= ai * rhs[j];
        }
        return out;
    }
    template <typename RHSValue>
    Polynomial& operator*=(const Polynomial<RHSValue, N>& rhs) {
        *this = *this * rhs;
        return *this;
    }

    // this * X^n mod X^N + 1 for any integer n (reference src/types.hpp:299-312)
    Polynomial multiply_x_exp(int n) const {
        int s = n % (2 * N);
        if (s < 0) s += 2 * N;
        Polynomial out;
        for (int j = 0; j < N; j++) {
            int src = j - s;           // coefficient that lands on X^j, up to sign
            bool negate = false;
            while (src < 0) { src += N; negate = !negate; }
            out[j] = negate ? -(*this)[src] : (*this)[src];
        }
        return out;
    }
};

template <typename Value, int M, int N>
class Matrix;

// ---- vector (reference src/types.hpp:318-357) -------------------------------------------------
template <typename Value, int N>
class Vector : public _VectorOrPolynomial<Value, N, Vector<Value, N>> {
    using Base = _VectorOrPolynomial<Value, N, Vector>;
    friend Base;
    friend Matrix<Value, N, N>;

public:
    using Base::Base;
    Vector() = default;
    Vector(const Vector&) = default;
    Vector(Vector&&) = default;
    Vector& operator=(const Vector&) = default;
    Vector& operator=(Vector&&) = default;

    template <typename RHSValue>
    detail::product_t<Value, RHSValue> dot(const Vector<RHSValue, N>& rhs) const {
        detail::product_t<Value, RHSValue> acc{};
        for (int i = 0; i < N; i++) acc += (*this)[i] * rhs[i];
        return acc;
    }

    Vector operator*=(const Matrix<Value, N, N>& rhs) {
        *this = *this * rhs;
        return *this;
    }
};

// ---- matrix = M rows of N (reference src/types.hpp:359-454) -----------------------------------
template <typename Value, int M, int N>
class Matrix {
    friend Vector<Value, M>;
    friend Vector<Value, N>;

    Vector<Vector<Value, N>, M> _val;

public:
    // value semantics come from the rows
    Matrix() = default;
    Matrix(const Matrix&) = default;
    Matrix(Matrix&&) = default;
    Matrix& operator=(const Matrix&) = default;
    Matrix& operator=(Matrix&&) = default;

    auto begin() { return _val.begin(); }
    auto begin() const { return _val.begin(); }
    auto end() { return _val.end(); }
    auto end() const { return _val.end(); }
    Vector<Value, N>& operator[](int r) { return _val[r]; }
    const Vector<Value, N>& operator[](int r) const { return _val[r]; }

    Matrix operator+() const { return Matrix(*this); }
    Matrix operator-() const {
        Matrix r(*this);
        for (int i = 0; i < M; i++) r._val[i] = -r._val[i];
        return r;
    }
    Matrix operator+=(const Matrix& o) { _val += o._val; return *this; }
    Matrix operator-=(const Matrix& o) { _val -= o._val; return *this; }
    Matrix operator+(const Matrix& o) const { Matrix r(*this); r += o; return r; }
    Matrix operator-(const Matrix& o) const { Matrix r(*this); r -= o; return r; }

    template <typename RHSValue, int O>
    Matrix<detail::product_t<Value, RHSValue>, M, O> operator*(const Matrix<RHSValue, N, O>& rhs) const {
        Matrix<detail::product_t<Value, RHSValue>, M, O> out;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++)
                for (int k = 0; k < O; k++) out[i][k] += (*this)[i][j] * rhs[j][k];
        return out;
    }
    template <typename RHSValue>
    Vector<detail::product_t<Value, RHSValue>, M> operator*(const Vector<RHSValue, N>& rhs) const {
        Vector<detail::product_t<Value, RHSValue>, M> out;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) out[i] += (*this)[i][j] * rhs[j];
        return out;
    }
    // row vector times matrix: out[j] = sum_i lhs[i] * rhs[i][j]
    template <typename LHSValue>
    friend Vector<detail::product_t<LHSValue, Value>, N> operator*(const Vector<LHSValue, M>& lhs,
                                                                   const Matrix<Value, M, N>& rhs) {
        Vector<detail::product_t<LHSValue, Value>, N> out;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) out[j] += lhs[i] * rhs[i][j];
        return out;
    }
    template <typename RHSValue>
    Matrix& operator*=(const Matrix<RHSValue, N, N>& rhs) {
        *this = *this * rhs;
        return *this;
    }

    bool operator==(const Matrix& o) const { return _val == o._val; }
    bool operator!=(const Matrix& o) const { return _val != o._val; }
};

}  // namespace TFHE
