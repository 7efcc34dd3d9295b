// ORACLE — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// CPU restatement (C++17, header-only) of RustFFT 6.4.1's *scalar* code path, the path
// `FftPlannerScalar` builds and `Fft<T>::process*()` runs.  Every function cites the
// reference file:line it follows (paths relative to the reference checkout).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code,
// and only as the checker / the reported CPU baseline.  The shipped library (libmi355fft.so)
// never links, includes or calls anything in oracle/.
//
// Parity status: pinned to the reference's own known-answer tests (dft.rs:283-398,
// math_utils.rs:495-522 and :590-682, plan.rs:700-830) and cross-checked against numpy.fft in
// complex128; the reference cannot be compiled in this environment (no rustc/cargo), so
// *bit-level* parity with the Rust binary is unpinned — see DESIGN.md §Oracle.
//
// Must be compiled with -ffp-contract=off: scalar Rust never fuses a*b+c.
//
// Third-party crates the reference leans on and that are restated here from their published
// semantics: num-complex 0.4 (Complex<T> + - * conj), transpose 0.2 (plain out-of-place
// transpose), strength_reduce 0.2 (exact % and /), num-integer (gcd, extended_gcd),
// primal-check (miller_rabin -> here: deterministic trial division primality).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rustfft_oracle {

// ------------------------------------------------------------------------------------------
// num_complex::Complex<T> (num-complex 0.4): #[repr(C)] {re, im}
// ------------------------------------------------------------------------------------------
template <class T> struct cx {
    T re, im;
};
template <class T> inline cx<T> operator+(cx<T> a, cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <class T> inline cx<T> operator-(cx<T> a, cx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <class T> inline cx<T> operator-(cx<T> a) { return {-a.re, -a.im}; }
// num-complex Mul: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
template <class T> inline cx<T> operator*(cx<T> a, cx<T> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class T> inline cx<T> operator*(cx<T> a, T s) { return {a.re * s, a.im * s}; }
template <class T> inline cx<T> conj(cx<T> a) { return {a.re, -a.im}; }

enum class Direction { Forward = 0, Inverse = 1 };
inline Direction opposite(Direction d) { return d == Direction::Forward ? Direction::Inverse : Direction::Forward; }
inline const char* dir_name(Direction d) { return d == Direction::Forward ? "Forward" : "Inverse"; }

// Rust panics are mapped to this exception; the message text is the reference's.
struct FftPanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------------------------------
// src/twiddles.rs
// ------------------------------------------------------------------------------------------
// twiddles.rs:6-23  compute_twiddle: angle in f64, cos/sin in f64, cast to T, inverse = conj
template <class T> inline cx<T> compute_twiddle(size_t index, size_t fft_len, Direction direction) {
    const double constant = -2.0 * 3.14159265358979323846264338327950288 / (double)fft_len;
    const double angle = constant * (double)index;
    cx<T> result{(T)std::cos(angle), (T)std::sin(angle)};
    return direction == Direction::Forward ? result : conj(result);
}
// twiddles.rs:25-57  fill_bluesteins_twiddles: w[i] = twiddle(i^2 mod 2n, 2n)
template <class T> inline void fill_bluesteins_twiddles(cx<T>* dst, size_t n, Direction direction) {
    const size_t twice_len = n * 2;
    for (size_t i = 0; i < n; ++i) {
        unsigned __int128 sq = (unsigned __int128)i * (unsigned __int128)i;  // u64 / u128 branches give the same value
        size_t i_mod = (size_t)(sq % (unsigned __int128)twice_len);
        dst[i] = compute_twiddle<T>(i_mod, twice_len, direction);
    }
}
// twiddles.rs:59-70  rotate_90
template <class T> inline cx<T> rotate_90(cx<T> v, Direction direction) {
    return direction == Direction::Forward ? cx<T>{v.im, -v.re} : cx<T>{-v.im, v.re};
}

// ------------------------------------------------------------------------------------------
// src/math_utils.rs
// ------------------------------------------------------------------------------------------
// math_utils.rs:23-37
inline uint64_t modular_exponent(uint64_t base, uint64_t exponent, uint64_t modulo) {
    uint64_t result = 1;
    while (exponent > 0) {
        if (exponent & 1) result = result * base % modulo;
        exponent >>= 1;
        base = (base * base) % modulo;
    }
    return result;
}
// math_utils.rs:40-74 (note the f32 sqrt limit, :51, :63)
inline std::vector<uint64_t> distinct_prime_factors(uint64_t n) {
    std::vector<uint64_t> result;
    if (n % 2 == 0) {
        while (n % 2 == 0) n /= 2;
        result.push_back(2);
    }
    if (n > 1) {
        uint64_t divisor = 3;
        uint64_t limit = (uint64_t)std::sqrt((float)n) + 1;
        while (divisor < limit) {
            if (n % divisor == 0) {
                while (n % divisor == 0) n /= divisor;
                result.push_back(divisor);
                limit = (uint64_t)std::sqrt((float)n) + 1;
            }
            divisor += 2;
        }
        if (n > 1) result.push_back(n);
    }
    return result;
}
// math_utils.rs:3-20
inline bool primitive_root(uint64_t prime, uint64_t* out) {
    std::vector<uint64_t> test_exponents;
    for (uint64_t f : distinct_prime_factors(prime - 1)) test_exponents.push_back((prime - 1) / f);
    for (uint64_t potential_root = 2; potential_root < prime; ++potential_root) {
        bool rejected = false;
        for (uint64_t e : test_exponents) {
            if (modular_exponent(potential_root, e, prime) == 1) {
                rejected = true;
                break;
            }
        }
        if (!rejected) {
            *out = potential_root;
            return true;
        }
    }
    return false;
}

struct PrimeFactor {
    size_t value;
    uint32_t count;
};
inline size_t upow(size_t b, uint32_t e) {
    size_t r = 1;
    for (uint32_t i = 0; i < e; ++i) r *= b;
    return r;
}
// math_utils.rs:83-369
struct PrimeFactors {
    std::vector<PrimeFactor> other_factors;
    size_t n = 0;
    uint32_t power_two = 0, power_three = 0, total_factor_count = 0, distinct_factor_count = 0;

    // math_utils.rs:92-160
    static PrimeFactors compute(size_t n) {
        PrimeFactors r;
        r.n = n;
        r.power_two = n == 0 ? (uint32_t)(8 * sizeof(size_t)) : (uint32_t)__builtin_ctzll((unsigned long long)n);
        r.total_factor_count += r.power_two;
        n = r.power_two >= 8 * sizeof(size_t) ? 0 : (n >> r.power_two);
        if (r.power_two > 0) r.distinct_factor_count += 1;
        while (n != 0 && n % 3 == 0) {
            r.power_three += 1;
            n /= 3;
        }
        r.total_factor_count += r.power_three;
        if (r.power_three > 0) r.distinct_factor_count += 1;
        if (n > 1) {
            size_t divisor = 5;
            size_t limit = (size_t)std::sqrt((float)n) + 1;
            while (divisor < limit) {
                uint32_t count = 0;
                while (n % divisor == 0) {
                    n /= divisor;
                    count += 1;
                }
                if (count > 0) {
                    r.other_factors.push_back({divisor, count});
                    r.total_factor_count += count;
                    r.distinct_factor_count += 1;
                    limit = (size_t)std::sqrt((float)n) + 1;
                }
                divisor += 2;
            }
            if (n > 1) {
                r.other_factors.push_back({n, 1});
                r.total_factor_count += 1;
                r.distinct_factor_count += 1;
            }
        }
        return r;
    }
    bool is_prime() const { return total_factor_count == 1; }             // :162
    size_t get_product() const { return n; }
    // :240-247
    bool has_factors_leq(size_t factor) const {
        return power_two > 0 || power_three > 0 || (!other_factors.empty() && other_factors.front().value <= factor);
    }
    // :249-256
    bool has_factors_gt(size_t factor) const {
        return (factor < 2 && power_two > 0) || (factor < 3 && power_three > 0) ||
               (!other_factors.empty() && other_factors.back().value > factor);
    }
    // :258-266
    size_t product_above(size_t min_factor) const {
        size_t p = 1;
        bool skipping = true;
        for (auto& f : other_factors) {
            if (skipping && f.value <= min_factor) continue;
            skipping = false;
            p *= upow(f.value, f.count);
        }
        return p;
    }
    // :269-368
    std::pair<PrimeFactors, PrimeFactors> partition_factors() const {
        PrimeFactors self = *this;
        if (self.is_prime()) throw FftPanic("assertion failed: !self.is_prime()");
        bool all_even = self.power_two % 2 == 0 && self.power_three % 2 == 0;
        for (auto& f : self.other_factors) all_even = all_even && (f.count % 2 == 0);
        if (all_even) {
            size_t new_product = 1;
            self.power_two /= 2;
            new_product <<= self.power_two;
            self.power_three /= 2;
            new_product *= upow(3, self.power_three);
            for (auto& f : self.other_factors) {
                f.count /= 2;
                new_product *= upow(f.value, f.count);
            }
            self.total_factor_count /= 2;
            self.n = new_product;
            return {self, self};
        } else if (self.distinct_factor_count == 1) {
            PrimeFactors half;
            half.n = self.n;
            half.power_two = self.power_two / 2;
            half.power_three = self.power_three / 2;
            half.total_factor_count = self.total_factor_count / 2;
            half.distinct_factor_count = 1;
            self.power_two -= half.power_two;
            self.power_three -= half.power_three;
            self.total_factor_count -= half.total_factor_count;
            if (!self.other_factors.empty()) {
                PrimeFactor& first = self.other_factors.front();
                PrimeFactor half_factor{first.value, first.count / 2};
                first.count -= half_factor.count;
                half.other_factors.push_back(half_factor);
                self.n = upow(first.value, first.count);
                half.n = upow(half_factor.value, half_factor.count);
            } else if (half.power_two > 0) {
                half.n = (size_t)1 << half.power_two;
                self.n = (size_t)1 << self.power_two;
            } else if (half.power_three > 0) {
                half.n = upow(3, half.power_three);
                self.n = upow(3, self.power_three);
            }
            return {self, half};
        } else {
            size_t left_product = 1, right_product = 1;
            for (auto& f : self.other_factors) {
                size_t fp = upow(f.value, f.count);
                if (left_product <= right_product)
                    left_product *= fp;
                else
                    right_product *= fp;
            }
            if (left_product <= right_product)
                left_product <<= self.power_two;
            else
                right_product <<= self.power_two;
            if (self.power_three > 0 && left_product <= right_product)
                left_product *= upow(3, self.power_three);
            else
                right_product *= upow(3, self.power_three);
            return {compute(left_product), compute(right_product)};
        }
    }
};

// num-integer: gcd and extended_gcd (Bezout coefficients as num-integer's ExtendedGcd computes them)
inline size_t gcd(size_t a, size_t b) {
    while (b) {
        size_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
struct ExtGcd {
    int64_t gcd, x, y;
};
inline ExtGcd extended_gcd(int64_t a, int64_t b) {
    // num-integer Integer::extended_gcd: iterative, (s, t) with a*s + b*t = gcd
    int64_t s0 = 0, s1 = 1, t0 = 1, t1 = 0, r0 = b, r1 = a;
    while (r0 != 0) {
        int64_t q = r1 / r0;
        int64_t tmp = r0;
        r0 = r1 - q * r0;
        r1 = tmp;
        tmp = s0;
        s0 = s1 - q * s0;
        s1 = tmp;
        tmp = t0;
        t0 = t1 - q * t0;
        t1 = tmp;
    }
    return {r1, s1, t1};
}
// primal-check::miller_rabin is exact for u64; trial division gives the same predicate.
inline bool is_prime_u64(uint64_t n) {
    if (n < 2) return false;
    if (n % 2 == 0) return n == 2;
    for (uint64_t d = 3; d * d <= n; d += 2)
        if (n % d == 0) return false;
    return true;
}

// ------------------------------------------------------------------------------------------
// src/common.rs:13-104 — the three panic helpers (messages verbatim)
// ------------------------------------------------------------------------------------------
inline void fft_error_inplace(size_t expected_len, size_t actual_len, size_t expected_scratch, size_t actual_scratch) {
    std::ostringstream m;
    if (!(actual_len >= expected_len)) {
        m << "Provided FFT buffer was too small. Expected len = " << expected_len << ", got len = " << actual_len;
        throw FftPanic(m.str());
    }
    if (actual_len % expected_len != 0) {
        m << "Input FFT buffer must be a multiple of FFT length. Expected multiple of " << expected_len
          << ", got len = " << actual_len;
        throw FftPanic(m.str());
    }
    if (!(actual_scratch >= expected_scratch)) {
        m << "Not enough scratch space was provided. Expected scratch len >= " << expected_scratch
          << ", got scratch len = " << actual_scratch;
        throw FftPanic(m.str());
    }
}
inline void fft_error_outofplace(size_t expected_len, size_t actual_input, size_t actual_output,
                                 size_t expected_scratch, size_t actual_scratch) {
    if (actual_input != actual_output) {
        std::ostringstream m;
        m << "Provided FFT input buffer and output buffer must have the same length. Got input.len() = " << actual_input
          << ", output.len() = " << actual_output;
        throw FftPanic(m.str());
    }
    fft_error_inplace(expected_len, actual_input, expected_scratch, actual_scratch);
}

// ------------------------------------------------------------------------------------------
// src/lib.rs:140-278 — the trait surface (Length + Direction + Fft)
// Slices are (pointer, length) pairs.
// ------------------------------------------------------------------------------------------
template <class T> struct Fft {
    typedef cx<T> C;
    virtual ~Fft() {}
    virtual size_t len() const = 0;
    virtual Direction fft_direction() const = 0;
    virtual size_t get_inplace_scratch_len() const = 0;
    virtual size_t get_outofplace_scratch_len() const = 0;
    virtual size_t get_immutable_scratch_len() const = 0;
    virtual const char* name() const = 0;

    // per-chunk kernels (the `perform_fft_*` of each algorithm)
    virtual void perform_fft_inplace(C* buffer, C* scratch, size_t scratch_len) const = 0;
    virtual void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const = 0;
    virtual void perform_fft_immut(const C* input, C* output, C* scratch, size_t scratch_len) const = 0;

    // lib.rs:195-198
    void process(C* buffer, size_t n) const {
        std::vector<C> scratch(get_inplace_scratch_len(), C{0, 0});
        process_with_scratch(buffer, n, scratch.data(), scratch.size());
    }
    // fft_helper.rs:9-28 + array_utils.rs:151-177
    void process_with_scratch(C* buffer, size_t n, C* scratch, size_t scratch_n) const {
        const size_t chunk = len(), req = get_inplace_scratch_len();
        if (chunk == 0) return;
        bool ok = scratch_n >= req;
        if (ok) {
            size_t remaining = n;
            C* p = buffer;
            while (remaining >= chunk) {
                perform_fft_inplace(p, scratch, req);
                p += chunk;
                remaining -= chunk;
            }
            ok = remaining == 0;
        }
        if (!ok) fft_error_inplace(chunk, n, req, scratch_n);
    }
    // fft_helper.rs:118-150 + array_utils.rs:293-327
    void process_outofplace_with_scratch(C* input, size_t n_in, C* output, size_t n_out, C* scratch, size_t scratch_n) const {
        const size_t chunk = len(), req = get_outofplace_scratch_len();
        if (chunk == 0) return;
        bool ok = scratch_n >= req && n_in == n_out;
        if (ok) {
            size_t remaining = n_in;
            C *pi = input, *po = output;
            while (remaining >= chunk) {
                perform_fft_out_of_place(pi, po, scratch, req);
                pi += chunk;
                po += chunk;
                remaining -= chunk;
            }
            ok = remaining == 0;
        }
        if (!ok) fft_error_outofplace(chunk, n_in, n_out, req, scratch_n);
    }
    // fft_helper.rs:50-85 + array_utils.rs:212-246
    void process_immutable_with_scratch(const C* input, size_t n_in, C* output, size_t n_out, C* scratch, size_t scratch_n) const {
        const size_t chunk = len(), req = get_immutable_scratch_len();
        if (chunk == 0) return;
        bool ok = scratch_n >= req && n_in == n_out;
        if (ok) {
            size_t remaining = n_in;
            const C* pi = input;
            C* po = output;
            while (remaining >= chunk) {
                perform_fft_immut(pi, po, scratch, req);
                pi += chunk;
                po += chunk;
                remaining -= chunk;
            }
            ok = remaining == 0;
        }
        if (!ok) fft_error_outofplace(chunk, n_in, n_out, req, scratch_n);
    }
};
template <class T> using FftPtr = std::shared_ptr<const Fft<T>>;

// common.rs:106-182 boilerplate_fft_oop!: in-place = OOP into scratch[..len] + copy back
template <class T> struct FftOop : Fft<T> {
    typedef cx<T> C;
    void perform_fft_inplace(C* chunk, C* scratch, size_t scratch_len) const override {
        const size_t n = this->len();
        this->perform_fft_out_of_place(chunk, scratch, scratch + n, scratch_len - n);
        std::memcpy(chunk, scratch, n * sizeof(C));
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/dft.rs:22-81
// ------------------------------------------------------------------------------------------
template <class T> struct Dft : FftOop<T> {
    typedef cx<T> C;
    std::vector<C> twiddles;
    Direction direction;
    Dft(size_t len, Direction d) : direction(d) {
        twiddles.resize(len);
        for (size_t i = 0; i < len; ++i) twiddles[i] = compute_twiddle<T>(i, len, d);
    }
    size_t len() const override { return twiddles.size(); }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return len(); }
    size_t get_outofplace_scratch_len() const override { return 0; }
    size_t get_immutable_scratch_len() const override { return 0; }
    const char* name() const override { return "Dft"; }
    // dft.rs:49-71
    void perform_fft_immut(const C* signal, C* spectrum, C*, size_t) const override {
        const size_t n = twiddles.size();
        for (size_t k = 0; k < n; ++k) {
            C acc{0, 0};
            size_t twiddle_index = 0;
            for (size_t j = 0; j < n; ++j) {
                C tw = twiddles[twiddle_index];
                acc = acc + tw * signal[j];
                twiddle_index += k;
                if (twiddle_index >= n) twiddle_index -= n;
            }
            spectrum[k] = acc;
        }
    }
    void perform_fft_out_of_place(C* in, C* out, C* s, size_t sl) const override { perform_fft_immut(in, out, s, sl); }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/butterflies.rs — hard-coded small DFTs.
// `LS` mirrors LoadStore/DoubleBuf (array_utils.rs:33-108): loads come from `in`, stores go to
// `out`; every butterfly performs all loads before its first store.
// ------------------------------------------------------------------------------------------
template <class T> struct LS {
    const cx<T>* in;
    cx<T>* out;
    cx<T> load(size_t i) const { return in[i]; }
    void store(cx<T> v, size_t i) { out[i] = v; }
};
template <class T> inline LS<T> ls_inplace(cx<T>* p) { return LS<T>{p, p}; }

// butterflies.rs:154-181
template <class T> inline void bf2_strided(cx<T>& left, cx<T>& right) {
    cx<T> temp = left + right;
    right = left - right;
    left = temp;
}
template <class T> inline void bf2_contiguous(LS<T> b) {
    cx<T> v0 = b.load(0), v1 = b.load(1);
    b.store(v0 + v1, 0);
    b.store(v0 - v1, 1);
}
// butterflies.rs:183-249
template <class T> struct B3 {
    cx<T> twiddle;
    Direction direction;
    explicit B3(Direction d) : twiddle(compute_twiddle<T>(1, 3, d)), direction(d) {}
    void strided(cx<T>& val0, cx<T>& val1, cx<T>& val2) const {
        cx<T> xp = val1 + val2, xn = val1 - val2, sum = val0 + xp;
        cx<T> temp_a = val0 + cx<T>{twiddle.re * xp.re, twiddle.re * xp.im};
        cx<T> temp_b = cx<T>{-twiddle.im * xn.im, twiddle.im * xn.re};
        val0 = sum;
        val1 = temp_a + temp_b;
        val2 = temp_a - temp_b;
    }
    void contiguous(LS<T> b) const {
        cx<T> xp = b.load(1) + b.load(2), xn = b.load(1) - b.load(2), sum = b.load(0) + xp;
        cx<T> temp_a = b.load(0) + cx<T>{twiddle.re * xp.re, twiddle.re * xp.im};
        cx<T> temp_b = cx<T>{-twiddle.im * xn.im, twiddle.im * xn.re};
        b.store(sum, 0);
        b.store(temp_a + temp_b, 1);
        b.store(temp_a - temp_b, 2);
    }
};
// butterflies.rs:251-321
template <class T> struct B4 {
    Direction direction;
    explicit B4(Direction d) : direction(d) {}
    void strided(cx<T>& v0, cx<T>& v1, cx<T>& v2, cx<T>& v3) const {
        bf2_strided(v0, v2);
        bf2_strided(v1, v3);
        v3 = rotate_90(v3, direction);
        bf2_strided(v0, v1);
        bf2_strided(v2, v3);
        cx<T> t = v1;
        v1 = v2;
        v2 = t;
    }
    void contiguous(LS<T> b) const {
        cx<T> v0 = b.load(0), v1 = b.load(1), v2 = b.load(2), v3 = b.load(3);
        bf2_strided(v0, v2);
        bf2_strided(v1, v3);
        v3 = rotate_90(v3, direction);
        bf2_strided(v0, v1);
        bf2_strided(v2, v3);
        b.store(v0, 0);
        b.store(v2, 1);
        b.store(v1, 2);
        b.store(v3, 3);
    }
};
// butterflies.rs:323-472 (B5), :842-1089 (B11), :1168-1483 (B13), :1582-2059 (B17),
// :2061-2632 (B19), :2634-3425 (B23), :3761-4929 (B29), :4931-6241 (B31): all follow the scheme that
// tools/genbutterflies.py:53-114 prints; terms are accumulated left to right exactly as generated.
template <class T, int P> struct BPrime {
    static constexpr int H = (P + 1) / 2;  // halflen
    cx<T> tw[H];                             // tw[m] = twiddle(m, P), m = 1..H-1
    Direction direction;
    explicit BPrime(Direction d) : direction(d) {
        for (int m = 1; m < H; ++m) tw[m] = compute_twiddle<T>(m, P, d);
        tw[0] = cx<T>{1, 0};
    }
    void contiguous(LS<T> b) const {
        cx<T> xp[H], xn[H];
        for (int n = 1; n < H; ++n) {
            xp[n] = b.load(n) + b.load(P - n);
            xn[n] = b.load(n) - b.load(P - n);
        }
        cx<T> x0 = b.load(0);
        cx<T> sum = x0;
        for (int n = 1; n < H; ++n) sum = sum + xp[n];
        T re_a[H], re_b[H], im_a[H], im_b[H];
        for (int n = 1; n < H; ++n) {
            T a = x0.re;
            for (int m = 1; m < H; ++m) {
                int mn = (m * n) % P;
                if (2 * mn > P) mn = P - mn;
                a = a + tw[mn].re * xp[m].re;
            }
            re_a[n] = a;
            T bb = 0;
            for (int m = 1; m < H; ++m) {
                int mn = (m * n) % P;
                bool neg = false;
                if (2 * mn > P) {
                    mn = P - mn;
                    neg = true;
                }
                T term = (neg ? -tw[mn].im : tw[mn].im) * xn[m].im;
                bb = (m == 1) ? term : bb + term;
            }
            re_b[n] = bb;
        }
        for (int n = 1; n < H; ++n) {
            T a = x0.im;
            for (int m = 1; m < H; ++m) {
                int mn = (m * n) % P;
                if (2 * mn > P) mn = P - mn;
                a = a + tw[mn].re * xp[m].im;
            }
            im_a[n] = a;
            T bb = 0;
            for (int m = 1; m < H; ++m) {
                int mn = (m * n) % P;
                bool neg = false;
                if (2 * mn > P) {
                    mn = P - mn;
                    neg = true;
                }
                T term = (neg ? -tw[mn].im : tw[mn].im) * xn[m].re;
                bb = (m == 1) ? term : bb + term;
            }
            im_b[n] = bb;
        }
        b.store(sum, 0);
        for (int n = 1; n < P; ++n) {
            if (2 * n > P) {
                int nf = P - n;
                b.store(cx<T>{re_a[nf] + re_b[nf], im_a[nf] - im_b[nf]}, n);
            } else {
                b.store(cx<T>{re_a[n] - re_b[n], im_a[n] + im_b[n]}, n);
            }
        }
    }
};
// butterflies.rs:528-717 — Butterfly7 is hand-written (terms ordered by twiddle index, and the
// sign convention of x34im_b differs from the generated butterflies), so it is restated verbatim.
template <class T> struct B7 {
    cx<T> twiddle1, twiddle2, twiddle3;
    Direction direction;
    explicit B7(Direction d)
        : twiddle1(compute_twiddle<T>(1, 7, d)), twiddle2(compute_twiddle<T>(2, 7, d)),
          twiddle3(compute_twiddle<T>(3, 7, d)), direction(d) {}
    void contiguous(LS<T> b) const {
        cx<T> x16p = b.load(1) + b.load(6), x16n = b.load(1) - b.load(6);
        cx<T> x25p = b.load(2) + b.load(5), x25n = b.load(2) - b.load(5);
        cx<T> x34p = b.load(3) + b.load(4), x34n = b.load(3) - b.load(4);
        cx<T> x0 = b.load(0);
        cx<T> sum = x0 + x16p + x25p + x34p;
        T x16re_a = x0.re + twiddle1.re * x16p.re + twiddle2.re * x25p.re + twiddle3.re * x34p.re;
        T x16re_b = twiddle1.im * x16n.im + twiddle2.im * x25n.im + twiddle3.im * x34n.im;
        T x25re_a = x0.re + twiddle1.re * x34p.re + twiddle2.re * x16p.re + twiddle3.re * x25p.re;
        T x25re_b = -twiddle1.im * x34n.im + twiddle2.im * x16n.im - twiddle3.im * x25n.im;
        T x34re_a = x0.re + twiddle1.re * x25p.re + twiddle2.re * x34p.re + twiddle3.re * x16p.re;
        T x34re_b = -twiddle1.im * x25n.im + twiddle2.im * x34n.im + twiddle3.im * x16n.im;
        T x16im_a = x0.im + twiddle1.re * x16p.im + twiddle2.re * x25p.im + twiddle3.re * x34p.im;
        T x16im_b = twiddle1.im * x16n.re + twiddle2.im * x25n.re + twiddle3.im * x34n.re;
        T x25im_a = x0.im + twiddle1.re * x34p.im + twiddle2.re * x16p.im + twiddle3.re * x25p.im;
        T x25im_b = -twiddle1.im * x34n.re + twiddle2.im * x16n.re - twiddle3.im * x25n.re;
        T x34im_a = x0.im + twiddle1.re * x25p.im + twiddle2.re * x34p.im + twiddle3.re * x16p.im;
        T x34im_b = twiddle1.im * x25n.re - twiddle2.im * x34n.re - twiddle3.im * x16n.re;
        b.store(sum, 0);
        b.store(cx<T>{x16re_a - x16re_b, x16im_a + x16im_b}, 1);
        b.store(cx<T>{x25re_a - x25re_b, x25im_a + x25im_b}, 2);
        b.store(cx<T>{x34re_a - x34re_b, x34im_a - x34im_b}, 3);
        b.store(cx<T>{x34re_a + x34re_b, x34im_a + x34im_b}, 4);
        b.store(cx<T>{x25re_a + x25re_b, x25im_a - x25im_b}, 5);
        b.store(cx<T>{x16re_a + x16re_b, x16im_a - x16im_b}, 6);
    }
};
// butterflies.rs:474-526
template <class T> struct B6 {
    B3<T> butterfly3;
    explicit B6(Direction d) : butterfly3(d) {}
    void contiguous(LS<T> b) const {
        cx<T> sa[3] = {b.load(0), b.load(2), b.load(4)};
        cx<T> sb[3] = {b.load(3), b.load(5), b.load(1)};
        butterfly3.contiguous(ls_inplace(sa));
        butterfly3.contiguous(ls_inplace(sb));
        bf2_strided(sa[0], sb[0]);
        bf2_strided(sa[1], sb[1]);
        bf2_strided(sa[2], sb[2]);
        b.store(sa[0], 0);
        b.store(sb[1], 1);
        b.store(sa[2], 2);
        b.store(sb[0], 3);
        b.store(sa[1], 4);
        b.store(sb[2], 5);
    }
};
// butterflies.rs:719-778
template <class T> struct B8 {
    T root2;
    Direction direction;
    explicit B8(Direction d) : root2((T)std::sqrt(0.5)), direction(d) {}
    void contiguous(LS<T> b) const {
        B4<T> butterfly4(direction);
        cx<T> s0[4] = {b.load(0), b.load(2), b.load(4), b.load(6)};
        cx<T> s1[4] = {b.load(1), b.load(3), b.load(5), b.load(7)};
        butterfly4.contiguous(ls_inplace(s0));
        butterfly4.contiguous(ls_inplace(s1));
        s1[1] = (rotate_90(s1[1], direction) + s1[1]) * root2;
        s1[2] = rotate_90(s1[2], direction);
        s1[3] = (rotate_90(s1[3], direction) - s1[3]) * root2;
        for (int i = 0; i < 4; ++i) bf2_strided(s0[i], s1[i]);
        for (int i = 0; i < 4; ++i) b.store(s0[i], i);
        for (int i = 0; i < 4; ++i) b.store(s1[i], i + 4);
    }
};
// butterflies.rs:780-840
template <class T> struct B9 {
    B3<T> butterfly3;
    cx<T> twiddle1, twiddle2, twiddle4;
    explicit B9(Direction d)
        : butterfly3(d), twiddle1(compute_twiddle<T>(1, 9, d)), twiddle2(compute_twiddle<T>(2, 9, d)),
          twiddle4(compute_twiddle<T>(4, 9, d)) {}
    void contiguous(LS<T> b) const {
        cx<T> s0[3] = {b.load(0), b.load(3), b.load(6)};
        cx<T> s1[3] = {b.load(1), b.load(4), b.load(7)};
        cx<T> s2[3] = {b.load(2), b.load(5), b.load(8)};
        butterfly3.contiguous(ls_inplace(s0));
        butterfly3.contiguous(ls_inplace(s1));
        butterfly3.contiguous(ls_inplace(s2));
        s1[1] = s1[1] * twiddle1;
        s1[2] = s1[2] * twiddle2;
        s2[1] = s2[1] * twiddle2;
        s2[2] = s2[2] * twiddle4;
        butterfly3.strided(s0[0], s1[0], s2[0]);
        butterfly3.strided(s0[1], s1[1], s2[1]);
        butterfly3.strided(s0[2], s1[2], s2[2]);
        for (int i = 0; i < 3; ++i) b.store(s0[i], i);
        for (int i = 0; i < 3; ++i) b.store(s1[i], 3 + i);
        for (int i = 0; i < 3; ++i) b.store(s2[i], 6 + i);
    }
};
// butterflies.rs:1091-1166
template <class T> struct B12 {
    B3<T> butterfly3;
    B4<T> butterfly4;
    explicit B12(Direction d) : butterfly3(d), butterfly4(d) {}
    void contiguous(LS<T> b) const {
        cx<T> s0[4] = {b.load(0), b.load(3), b.load(6), b.load(9)};
        cx<T> s1[4] = {b.load(4), b.load(7), b.load(10), b.load(1)};
        cx<T> s2[4] = {b.load(8), b.load(11), b.load(2), b.load(5)};
        butterfly4.contiguous(ls_inplace(s0));
        butterfly4.contiguous(ls_inplace(s1));
        butterfly4.contiguous(ls_inplace(s2));
        for (int i = 0; i < 4; ++i) butterfly3.strided(s0[i], s1[i], s2[i]);
        b.store(s0[0], 0);
        b.store(s1[1], 1);
        b.store(s2[2], 2);
        b.store(s0[3], 3);
        b.store(s1[0], 4);
        b.store(s2[1], 5);
        b.store(s0[2], 6);
        b.store(s1[3], 7);
        b.store(s2[0], 8);
        b.store(s0[1], 9);
        b.store(s1[2], 10);
        b.store(s2[3], 11);
    }
};
// butterflies.rs:1485-1580
template <class T> struct B16 {
    B8<T> butterfly8;
    cx<T> twiddle1, twiddle2, twiddle3;
    Direction direction;
    explicit B16(Direction d)
        : butterfly8(d), twiddle1(compute_twiddle<T>(1, 16, d)), twiddle2(compute_twiddle<T>(2, 16, d)),
          twiddle3(compute_twiddle<T>(3, 16, d)), direction(d) {}
    void contiguous(LS<T> b) const {
        B4<T> butterfly4(direction);
        cx<T> ev[8], n1[4], n3[4];
        for (int i = 0; i < 8; ++i) ev[i] = b.load(2 * i);
        for (int i = 0; i < 4; ++i) n1[i] = b.load(1 + 4 * i);
        n3[0] = b.load(15);
        for (int i = 1; i < 4; ++i) n3[i] = b.load(4 * i - 1);
        butterfly8.contiguous(ls_inplace(ev));
        butterfly4.contiguous(ls_inplace(n1));
        butterfly4.contiguous(ls_inplace(n3));
        n1[1] = n1[1] * twiddle1;
        n3[1] = n3[1] * conj(twiddle1);
        n1[2] = n1[2] * twiddle2;
        n3[2] = n3[2] * conj(twiddle2);
        n1[3] = n1[3] * twiddle3;
        n3[3] = n3[3] * conj(twiddle3);
        for (int i = 0; i < 4; ++i) bf2_strided(n1[i], n3[i]);
        for (int i = 0; i < 4; ++i) n3[i] = rotate_90(n3[i], direction);
        for (int i = 0; i < 4; ++i) b.store(ev[i] + n1[i], i);
        for (int i = 0; i < 4; ++i) b.store(ev[4 + i] + n3[i], 4 + i);
        for (int i = 0; i < 4; ++i) b.store(ev[i] - n1[i], 8 + i);
        for (int i = 0; i < 4; ++i) b.store(ev[4 + i] - n3[i], 12 + i);
    }
};
// butterflies.rs:3427-3586
template <class T> struct B24 {
    B4<T> butterfly4;
    B6<T> butterfly6;
    cx<T> twiddle1, twiddle2, twiddle4, twiddle5, twiddle8, twiddle10;
    T root2;
    Direction direction;
    explicit B24(Direction d)
        : butterfly4(d), butterfly6(d), twiddle1(compute_twiddle<T>(1, 24, d)), twiddle2(compute_twiddle<T>(2, 24, d)),
          twiddle4(compute_twiddle<T>(4, 24, d)), twiddle5(compute_twiddle<T>(5, 24, d)),
          twiddle8(compute_twiddle<T>(8, 24, d)), twiddle10(compute_twiddle<T>(10, 24, d)), root2((T)std::sqrt(0.5)),
          direction(d) {}
    void contiguous(LS<T> b) const {
        cx<T> s0[6], s1[6], s2[6], s3[6];
        for (int i = 0; i < 6; ++i) {
            s0[i] = b.load(4 * i);
            s1[i] = b.load(4 * i + 1);
            s2[i] = b.load(4 * i + 2);
            s3[i] = b.load(4 * i + 3);
        }
        butterfly6.contiguous(ls_inplace(s0));
        butterfly6.contiguous(ls_inplace(s1));
        butterfly6.contiguous(ls_inplace(s2));
        butterfly6.contiguous(ls_inplace(s3));
        s1[1] = s1[1] * twiddle1;
        s1[2] = s1[2] * twiddle2;
        s1[3] = (rotate_90(s1[3], direction) + s1[3]) * root2;
        s1[4] = s1[4] * twiddle4;
        s1[5] = s1[5] * twiddle5;
        s2[1] = s2[1] * twiddle2;
        s2[2] = s2[2] * twiddle4;
        s2[3] = rotate_90(s2[3], direction);
        s2[4] = s2[4] * twiddle8;
        s2[5] = s2[5] * twiddle10;
        s3[1] = (rotate_90(s3[1], direction) + s3[1]) * root2;
        s3[2] = rotate_90(s3[2], direction);
        s3[3] = (rotate_90(s3[3], direction) - s3[3]) * root2;
        s3[4] = -s3[4];
        s3[5] = (rotate_90(s3[5], direction) + s3[5]) * (-root2);
        for (int i = 0; i < 6; ++i) butterfly4.strided(s0[i], s1[i], s2[i], s3[i]);
        for (int i = 0; i < 6; ++i) b.store(s0[i], i);
        for (int i = 0; i < 6; ++i) b.store(s1[i], 6 + i);
        for (int i = 0; i < 6; ++i) b.store(s2[i], 12 + i);
        for (int i = 0; i < 6; ++i) b.store(s3[i], 18 + i);
    }
};
// butterflies.rs:3588-3759
template <class T> struct B27 {
    B9<T> butterfly9;
    cx<T> tw[12];
    explicit B27(Direction d) : butterfly9(d) {
        const int idx[12] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};
        for (int i = 0; i < 12; ++i) tw[i] = compute_twiddle<T>(idx[i], 27, d);
    }
    void contiguous(LS<T> b) const {
        cx<T> s0[9], s1[9], s2[9];
        for (int i = 0; i < 9; ++i) {
            s0[i] = b.load(3 * i);
            s1[i] = b.load(3 * i + 1);
            s2[i] = b.load(3 * i + 2);
        }
        butterfly9.contiguous(ls_inplace(s0));
        butterfly9.contiguous(ls_inplace(s1));
        butterfly9.contiguous(ls_inplace(s2));
        for (int i = 1; i < 9; ++i) s1[i] = s1[i] * tw[i - 1];
        s2[1] = s2[1] * tw[1];
        s2[2] = s2[2] * tw[3];
        s2[3] = s2[3] * tw[5];
        s2[4] = s2[4] * tw[7];
        s2[5] = s2[5] * tw[8];
        s2[6] = s2[6] * tw[9];
        s2[7] = s2[7] * tw[10];
        s2[8] = s2[8] * tw[11];
        for (int i = 0; i < 9; ++i) butterfly9.butterfly3.strided(s0[i], s1[i], s2[i]);
        for (int i = 0; i < 9; ++i) b.store(s0[i], i);
        for (int i = 0; i < 9; ++i) b.store(s1[i], 9 + i);
        for (int i = 0; i < 9; ++i) b.store(s2[i], 18 + i);
    }
};
// butterflies.rs:6243-6393
template <class T> struct B32 {
    B16<T> butterfly16;
    B8<T> butterfly8;
    cx<T> tw[7];
    Direction direction;
    explicit B32(Direction d) : butterfly16(d), butterfly8(d), direction(d) {
        for (int i = 0; i < 7; ++i) tw[i] = compute_twiddle<T>(i + 1, 32, d);
    }
    void contiguous(LS<T> b) const {
        cx<T> ev[16], n1[8], n3[8];
        for (int i = 0; i < 16; ++i) ev[i] = b.load(2 * i);
        for (int i = 0; i < 8; ++i) n1[i] = b.load(1 + 4 * i);
        n3[0] = b.load(31);
        for (int i = 1; i < 8; ++i) n3[i] = b.load(4 * i - 1);
        butterfly16.contiguous(ls_inplace(ev));
        butterfly8.contiguous(ls_inplace(n1));
        butterfly8.contiguous(ls_inplace(n3));
        for (int i = 1; i < 8; ++i) {
            n1[i] = n1[i] * tw[i - 1];
            n3[i] = n3[i] * conj(tw[i - 1]);
        }
        for (int i = 0; i < 8; ++i) bf2_strided(n1[i], n3[i]);
        for (int i = 0; i < 8; ++i) n3[i] = rotate_90(n3[i], direction);
        for (int i = 0; i < 8; ++i) b.store(ev[i] + n1[i], i);
        for (int i = 0; i < 8; ++i) b.store(ev[8 + i] + n3[i], 8 + i);
        for (int i = 0; i < 8; ++i) b.store(ev[i] - n1[i], 16 + i);
        for (int i = 0; i < 8; ++i) b.store(ev[8 + i] - n3[i], 24 + i);
    }
};

// butterflies.rs:10-95 boilerplate_fft_butterfly!: Fft impl for a butterfly (all scratch lens 0)
template <class T> struct Butterfly : Fft<T> {
    typedef cx<T> C;
    size_t n;
    Direction direction;
    // only the member matching `n` is used
    B3<T> b3;
    B4<T> b4;
    BPrime<T, 5> b5;
    B6<T> b6;
    B7<T> b7;
    B8<T> b8;
    B9<T> b9;
    BPrime<T, 11> b11;
    B12<T> b12;
    BPrime<T, 13> b13;
    B16<T> b16;
    BPrime<T, 17> b17;
    BPrime<T, 19> b19;
    BPrime<T, 23> b23;
    B24<T> b24;
    B27<T> b27;
    BPrime<T, 29> b29;
    BPrime<T, 31> b31;
    B32<T> b32;
    static bool supported(size_t len) {
        switch (len) {
            case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 11: case 12: case 13:
            case 16: case 17: case 19: case 23: case 24: case 27: case 29: case 31: case 32:
                return true;
            default:
                return false;
        }
    }
    Butterfly(size_t len, Direction d)
        : n(len), direction(d), b3(d), b4(d), b5(d), b6(d), b7(d), b8(d), b9(d), b11(d), b12(d), b13(d), b16(d),
          b17(d), b19(d), b23(d), b24(d), b27(d), b29(d), b31(d), b32(d) {
        if (!supported(len)) throw FftPanic("no butterfly of that length");
    }
    size_t len() const override { return n; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return 0; }
    size_t get_outofplace_scratch_len() const override { return 0; }
    size_t get_immutable_scratch_len() const override { return 0; }
    const char* name() const override { return "Butterfly"; }
    void run(LS<T> b) const {
        switch (n) {
            case 1: b.store(b.load(0), 0); break;  // Butterfly1 (butterflies.rs:97-152): copy
            case 2: bf2_contiguous(b); break;
            case 3: b3.contiguous(b); break;
            case 4: b4.contiguous(b); break;
            case 5: b5.contiguous(b); break;
            case 6: b6.contiguous(b); break;
            case 7: b7.contiguous(b); break;
            case 8: b8.contiguous(b); break;
            case 9: b9.contiguous(b); break;
            case 11: b11.contiguous(b); break;
            case 12: b12.contiguous(b); break;
            case 13: b13.contiguous(b); break;
            case 16: b16.contiguous(b); break;
            case 17: b17.contiguous(b); break;
            case 19: b19.contiguous(b); break;
            case 23: b23.contiguous(b); break;
            case 24: b24.contiguous(b); break;
            case 27: b27.contiguous(b); break;
            case 29: b29.contiguous(b); break;
            case 31: b31.contiguous(b); break;
            case 32: b32.contiguous(b); break;
        }
    }
    void perform_fft_inplace(C* buffer, C*, size_t) const override { run(LS<T>{buffer, buffer}); }
    void perform_fft_out_of_place(C* in, C* out, C*, size_t) const override { run(LS<T>{in, out}); }
    void perform_fft_immut(const C* in, C* out, C*, size_t) const override { run(LS<T>{in, out}); }
};

// ------------------------------------------------------------------------------------------
// src/array_utils.rs — data movement
// ------------------------------------------------------------------------------------------
// array_utils.rs:9-18
template <class E> inline void transpose_small(size_t width, size_t height, const E* input, E* output) {
    for (size_t x = 0; x < width; ++x)
        for (size_t y = 0; y < height; ++y) output[y + x * height] = input[x + y * width];
}
// transpose crate 0.2: transpose(input, output, input_width, input_height): output[x*height + y] = input[y*width + x]
template <class E> inline void transpose(const E* input, E* output, size_t width, size_t height) {
    transpose_small(width, height, input, output);
}
// array_utils.rs:427-437
inline size_t reverse_bits(size_t value, size_t D, uint32_t rev_digits) {
    size_t result = 0;
    for (uint32_t i = 0; i < rev_digits; ++i) {
        result = (result * D) + (value % D);
        value = value / D;
    }
    return result;
}
// array_utils.rs:440-458
inline bool compute_logarithm(size_t value, size_t D, uint32_t* out) {
    if (value == 0 || D < 2) return false;
    uint32_t e = 0;
    size_t cur = value;
    while (cur % D == 0) {
        e += 1;
        cur /= D;
    }
    if (cur == 1) {
        *out = e;
        return true;
    }
    return false;
}
// array_utils.rs:372-422
template <class E> inline void bitreversed_transpose(size_t D, size_t height, const E* input, E* output, size_t n) {
    const size_t width = n / height;
    uint32_t rev_digits = 0;
    if (!compute_logarithm(width, D, &rev_digits)) throw FftPanic("bitreversed_transpose: width is not a power of D");
    const size_t strided_width = width / D;
    for (size_t x = 0; x < strided_width; ++x)
        for (size_t y = 0; y < height; ++y)
            for (size_t i = 0; i < D; ++i) {
                size_t fwd = D * x + i;
                size_t rev = reverse_bits(fwd, D, rev_digits);
                output[y + rev * height] = input[fwd + y * width];
            }
}
// array_utils.rs:460-463
struct TransposeFactor {
    uint8_t factor;  // the radix itself (2..7)
    uint8_t count;
};
// array_utils.rs:514-558
inline size_t reverse_remainders(size_t value, const std::vector<TransposeFactor>& factors) {
    size_t result = 0;
    for (auto& f : factors)
        for (uint8_t c = 0; c < f.count; ++c) {
            result = (result * f.factor) + (value % f.factor);
            value = value / f.factor;
        }
    return result;
}
// array_utils.rs:469-509
template <class E>
inline void factor_transpose(size_t D, size_t height, const E* input, E* output, size_t n,
                             const std::vector<TransposeFactor>& factors) {
    const size_t width = n / height;
    if (!(width % D == 0 && D > 1 && n % width == 0)) throw FftPanic("factor_transpose: bad arguments");
    const size_t strided_width = width / D;
    for (size_t x = 0; x < strided_width; ++x)
        for (size_t y = 0; y < height; ++y)
            for (size_t i = 0; i < D; ++i) {
                size_t fwd = D * x + i;
                size_t rev = reverse_remainders(fwd, factors);
                output[y + rev * height] = input[fwd + y * width];
            }
}

// ------------------------------------------------------------------------------------------
// src/algorithm/radixn.rs:338-490 — butterfly_2..7 column loops (also used by Radix4)
// ------------------------------------------------------------------------------------------
template <class T> struct CrossButterflies {
    Direction direction;
    B3<T> b3;
    B4<T> b4;
    BPrime<T, 5> b5;
    B6<T> b6;
    B7<T> b7;
    explicit CrossButterflies(Direction d) : direction(d), b3(d), b4(d), b5(d), b6(d), b7(d) {}
    void run(size_t radix, cx<T>* data, const cx<T>* twiddles, size_t num_columns) const {
        cx<T> scratch[7];
        for (size_t idx = 0; idx < num_columns; ++idx) {
            const size_t tw_idx = idx * (radix - 1);
            scratch[0] = data[idx];
            for (size_t r = 1; r < radix; ++r) scratch[r] = data[idx + r * num_columns] * twiddles[tw_idx + r - 1];
            LS<T> ls = ls_inplace(scratch);
            switch (radix) {
                case 2: bf2_contiguous(ls); break;
                case 3: b3.contiguous(ls); break;
                case 4: b4.contiguous(ls); break;
                case 5: b5.contiguous(ls); break;
                case 6: b6.contiguous(ls); break;
                case 7: b7.contiguous(ls); break;
                default: throw FftPanic("unsupported cross radix");
            }
            for (size_t r = 0; r < radix; ++r) data[idx + r * num_columns] = scratch[r];
        }
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/radix4.rs:27-203
// ------------------------------------------------------------------------------------------
template <class T> struct Radix4 : FftOop<T> {
    typedef cx<T> C;
    std::vector<C> twiddles;
    FftPtr<T> base_fft;
    size_t base_len, n;
    Direction direction;
    size_t inplace_scratch, oop_scratch, immut_scratch;
    CrossButterflies<T> cross;

    // radix4.rs:42-66
    static FftPtr<T> make(size_t len, Direction d) {
        if (len == 0 || (len & (len - 1)) != 0) {
            std::ostringstream m;
            m << "Radix4 algorithm requires a power-of-two input size. Got " << len;
            throw FftPanic(m.str());
        }
        uint32_t exponent = (uint32_t)__builtin_ctzll((unsigned long long)len);
        uint32_t base_exponent;
        switch (exponent) {
            case 0: base_exponent = 0; break;
            case 1: base_exponent = 1; break;
            case 2: base_exponent = 2; break;
            case 3: base_exponent = 3; break;
            default: base_exponent = (exponent % 2 == 1) ? 5 : 4;
        }
        FftPtr<T> base = std::make_shared<Butterfly<T>>((size_t)1 << base_exponent, d);
        return std::make_shared<Radix4<T>>((exponent - base_exponent) / 2, base);
    }
    // radix4.rs:69-119
    Radix4(uint32_t k, FftPtr<T> base)
        : base_fft(base), base_len(base->len()), n(base->len() * ((size_t)1 << (k * 2))),
          direction(base->fft_direction()), cross(base->fft_direction()) {
        const size_t ROW_COUNT = 4;
        size_t cross_fft_len = base_len;
        while (cross_fft_len < n) {
            size_t num_columns = cross_fft_len;
            cross_fft_len *= ROW_COUNT;
            for (size_t i = 0; i < num_columns; ++i)
                for (size_t kk = 1; kk < ROW_COUNT; ++kk)
                    twiddles.push_back(compute_twiddle<T>(i * kk, cross_fft_len, direction));
        }
        size_t base_inplace = base_fft->get_inplace_scratch_len();
        inplace_scratch = base_inplace > cross_fft_len ? cross_fft_len + base_inplace : cross_fft_len;
        oop_scratch = base_inplace > n ? base_inplace : 0;
        immut_scratch = base_inplace;
    }
    size_t len() const override { return n; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return inplace_scratch; }
    size_t get_outofplace_scratch_len() const override { return oop_scratch; }
    size_t get_immutable_scratch_len() const override { return immut_scratch; }
    const char* name() const override { return "Radix4"; }
    void cross_ffts(C* output) const {
        const size_t ROW_COUNT = 4;
        size_t cross_fft_len = base_len;
        const C* layer_twiddles = twiddles.data();
        while (cross_fft_len < n) {
            size_t num_columns = cross_fft_len;
            cross_fft_len *= ROW_COUNT;
            for (size_t off = 0; off + cross_fft_len <= n; off += cross_fft_len)
                cross.run(4, output + off, layer_twiddles, num_columns);
            layer_twiddles += num_columns * (ROW_COUNT - 1);
        }
    }
    // radix4.rs:131-165
    void perform_fft_immut(const C* input, C* output, C* scratch, size_t scratch_len) const override {
        if (n == base_len)
            std::memcpy(output, input, n * sizeof(C));
        else
            bitreversed_transpose<C>(4, base_len, input, output, n);
        base_fft->process_with_scratch(output, n, scratch, scratch_len);
        cross_ffts(output);
    }
    // radix4.rs:167-203
    void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const override {
        if (n == base_len)
            std::memcpy(output, input, n * sizeof(C));
        else
            bitreversed_transpose<C>(4, base_len, input, output, n);
        if (scratch_len > 0)
            base_fft->process_with_scratch(output, n, scratch, scratch_len);
        else
            base_fft->process_with_scratch(output, n, input, n);
        cross_ffts(output);
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/radixn.rs:35-333
// ------------------------------------------------------------------------------------------
template <class T> struct RadixN : FftOop<T> {
    typedef cx<T> C;
    std::vector<C> twiddles;
    FftPtr<T> base_fft;
    size_t base_len, n;
    std::vector<TransposeFactor> factors;  // reversed + run-length encoded (radixn.rs:85-104)
    std::vector<uint8_t> butterflies;      // in application order
    Direction direction;
    size_t inplace_scratch, oop_scratch, immut_scratch;
    CrossButterflies<T> cross;

    // radixn.rs:54-155
    RadixN(const std::vector<uint8_t>& radix_factors, FftPtr<T> base)
        : base_fft(base), base_len(base->len()), direction(base->fft_direction()), cross(base->fft_direction()) {
        size_t cross_fft_len = base_len;
        for (uint8_t f : radix_factors) {
            if (f < 2 || f > 7) throw FftPanic("RadixN factors must be in 2..=7");
            butterflies.push_back(f);
            cross_fft_len *= f;
        }
        n = cross_fft_len;
        for (auto it = radix_factors.rbegin(); it != radix_factors.rend(); ++it) {
            if (!factors.empty() && factors.back().factor == *it)
                factors.back().count += 1;
            else
                factors.push_back({*it, 1});
        }
        cross_fft_len = base_len;
        for (uint8_t f : radix_factors) {
            size_t cross_fft_columns = cross_fft_len;
            cross_fft_len *= f;
            for (size_t i = 0; i < cross_fft_columns; ++i)
                for (size_t k = 1; k < f; ++k) twiddles.push_back(compute_twiddle<T>(i * k, cross_fft_len, direction));
        }
        size_t base_inplace = base_fft->get_inplace_scratch_len();
        inplace_scratch = base_inplace > n ? n + base_inplace : n;
        oop_scratch = base_inplace > n ? base_inplace : 0;
        immut_scratch = base_inplace;
    }
    size_t len() const override { return n; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return inplace_scratch; }
    size_t get_outofplace_scratch_len() const override { return oop_scratch; }
    size_t get_immutable_scratch_len() const override { return immut_scratch; }
    const char* name() const override { return "RadixN"; }
    void reorder(const C* input, C* output) const {
        if (!factors.empty())
            factor_transpose<C>(factors.front().factor, base_len, input, output, n, factors);
        else
            std::memcpy(output, input, n * sizeof(C));
    }
    void cross_ffts(C* output) const {
        size_t cross_fft_len = base_len;
        const C* layer_twiddles = twiddles.data();
        for (uint8_t f : butterflies) {
            size_t cross_fft_columns = cross_fft_len;
            cross_fft_len *= f;
            for (size_t off = 0; off + cross_fft_len <= n; off += cross_fft_len)
                cross.run(f, output + off, layer_twiddles, cross_fft_columns);
            layer_twiddles += cross_fft_columns * (f - 1);
        }
    }
    // radixn.rs:167-248 (same structure as out-of-place, caller scratch goes to the base)
    void perform_fft_immut(const C* input, C* output, C* scratch, size_t scratch_len) const override {
        reorder(input, output);
        base_fft->process_with_scratch(output, n, scratch, scratch_len);
        cross_ffts(output);
    }
    // radixn.rs:250-333
    void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const override {
        reorder(input, output);
        if (scratch_len > 0)
            base_fft->process_with_scratch(output, n, scratch, scratch_len);
        else
            base_fft->process_with_scratch(output, n, input, n);
        cross_ffts(output);
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/mixed_radix.rs:35-230 (MixedRadix) and :266-405 (MixedRadixSmall)
// ------------------------------------------------------------------------------------------
template <class T> struct MixedRadix : Fft<T> {
    typedef cx<T> C;
    std::vector<C> twiddles;
    FftPtr<T> width_size_fft, height_size_fft;
    size_t width, height;
    size_t inplace_scratch, oop_scratch, immut_scratch;
    Direction direction;
    bool small;

    MixedRadix(FftPtr<T> width_fft, FftPtr<T> height_fft, bool small_variant)
        : width_size_fft(width_fft), height_size_fft(height_fft), width(width_fft->len()), height(height_fft->len()),
          direction(width_fft->fft_direction()), small(small_variant) {
        if (width_fft->fft_direction() != height_fft->fft_direction())
            throw FftPanic("width_fft and height_fft must have the same direction");
        const size_t n = width * height;
        if (small) {  // mixed_radix.rs:288-295
            if (width_fft->get_outofplace_scratch_len() != 0 || height_fft->get_outofplace_scratch_len() != 0)
                throw FftPanic("MixedRadixSmall should only be used with algorithms that require 0 out-of-place scratch");
            if (width_fft->get_inplace_scratch_len() > width || height_fft->get_inplace_scratch_len() > height)
                throw FftPanic("MixedRadixSmall should only be used with algorithms that require little inplace scratch");
        }
        twiddles.resize(n);
        for (size_t x = 0; x < width; ++x)  // mixed_radix.rs:66-71 / :299-304
            for (size_t y = 0; y < height; ++y) twiddles[x * height + y] = compute_twiddle<T>(x * y, n, direction);
        if (small) {  // mixed_radix.rs:399-405
            inplace_scratch = n;
            oop_scratch = 0;
            immut_scratch = n;
        } else {  // mixed_radix.rs:74-109
            size_t h_in = height_fft->get_inplace_scratch_len(), w_in = width_fft->get_inplace_scratch_len();
            size_t w_oop = width_fft->get_outofplace_scratch_len();
            size_t max_inner = std::max(h_in, w_in);
            oop_scratch = max_inner > n ? max_inner : 0;
            inplace_scratch = n + std::max(h_in > n ? h_in : (size_t)0, w_oop);
            immut_scratch = std::max(n + w_in, h_in);
        }
    }
    size_t len() const override { return width * height; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return inplace_scratch; }
    size_t get_outofplace_scratch_len() const override { return oop_scratch; }
    size_t get_immutable_scratch_len() const override { return immut_scratch; }
    const char* name() const override { return small ? "MixedRadixSmall" : "MixedRadix"; }
    void apply_twiddles(C* p) const {
        const size_t n = width * height;
        for (size_t i = 0; i < n; ++i) p[i] = p[i] * twiddles[i];
    }
    // mixed_radix.rs:128-158 / :319-341
    void perform_fft_inplace(C* buffer, C* scratch_all, size_t scratch_len) const override {
        const size_t n = width * height;
        C* scratch = scratch_all;
        C* inner_scratch = scratch_all + n;
        size_t inner_len = scratch_len - n;
        transpose(buffer, scratch, width, height);
        if (small) {
            height_size_fft->process_with_scratch(scratch, n, buffer, n);
        } else if (inner_len > n) {
            height_size_fft->process_with_scratch(scratch, n, inner_scratch, inner_len);
        } else {
            height_size_fft->process_with_scratch(scratch, n, buffer, n);
        }
        apply_twiddles(scratch);
        transpose(scratch, buffer, height, width);
        if (small)
            width_size_fft->process_outofplace_with_scratch(buffer, n, scratch, n, nullptr, 0);
        else
            width_size_fft->process_outofplace_with_scratch(buffer, n, scratch, n, inner_scratch, inner_len);
        transpose(scratch, buffer, width, height);
    }
    // mixed_radix.rs:160-189 / :343-366
    void perform_fft_immut(const C* input, C* output, C* scratch_raw, size_t scratch_len) const override {
        const size_t n = width * height;
        transpose(input, output, width, height);
        height_size_fft->process_with_scratch(output, n, scratch_raw, scratch_len);
        apply_twiddles(output);
        if (small) {
            transpose(output, scratch_raw, height, width);
            width_size_fft->process_with_scratch(scratch_raw, n, output, n);
            transpose(scratch_raw, output, width, height);
        } else {
            C* scratch = scratch_raw;
            C* inner_scratch = scratch_raw + n;
            transpose(output, scratch, height, width);
            width_size_fft->process_with_scratch(scratch, n, inner_scratch, scratch_len - n);
            transpose(scratch, output, width, height);
        }
    }
    // mixed_radix.rs:191-230 / :368-397
    void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const override {
        const size_t n = width * height;
        transpose(input, output, width, height);
        if (!small && scratch_len > n)
            height_size_fft->process_with_scratch(output, n, scratch, scratch_len);
        else
            height_size_fft->process_with_scratch(output, n, input, n);
        apply_twiddles(output);
        transpose(output, input, height, width);
        if (!small && scratch_len > n)
            width_size_fft->process_with_scratch(input, n, scratch, scratch_len);
        else
            width_size_fft->process_with_scratch(input, n, output, n);
        transpose(input, output, width, height);
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/good_thomas_algorithm.rs:344-517 (GoodThomasAlgorithmSmall) and :40-308 (large).
// Both variants realise the same index maps (the large one computes them incrementally,
// :144-222); the large variant is never emitted by any planner (plan.rs:140-144).
// ------------------------------------------------------------------------------------------
template <class T> struct GoodThomas : Fft<T> {
    typedef cx<T> C;
    size_t width, height;
    FftPtr<T> width_size_fft, height_size_fft;
    std::vector<size_t> input_map, output_map;
    Direction direction;
    bool small;
    size_t inplace_scratch, oop_scratch, immut_scratch;

    GoodThomas(FftPtr<T> width_fft, FftPtr<T> height_fft, bool small_variant)
        : width(width_fft->len()), height(height_fft->len()), width_size_fft(width_fft), height_size_fft(height_fft),
          direction(width_fft->fft_direction()), small(small_variant) {
        if (width_fft->fft_direction() != height_fft->fft_direction())
            throw FftPanic("n1_fft and height_fft must have the same direction");
        const size_t n = width * height;
        ExtGcd g = extended_gcd((int64_t)width, (int64_t)height);  // good_thomas_algorithm.rs:377
        if (g.gcd != 1) {
            std::ostringstream m;
            m << "Invalid input width and height to Good-Thomas Algorithm: (" << width << "," << height
              << "): Inputs must be coprime";
            throw FftPanic(m.str());
        }
        size_t width_inverse = (size_t)(g.x >= 0 ? g.x : g.x + (int64_t)height);   // :384-393
        size_t height_inverse = (size_t)(g.y >= 0 ? g.y : g.y + (int64_t)width);
        input_map.resize(n);
        output_map.resize(n);
        for (size_t i = 0; i < n; ++i) {  // :397-402
            size_t x = i % width, y = i / width;
            input_map[i] = (x * height + y * width) % n;
        }
        for (size_t i = 0; i < n; ++i) {
            size_t y = i % height, x = i / height;
            output_map[i] = (x * height * height_inverse + y * width * width_inverse) % n;
        }
        if (small) {  // :511-517
            inplace_scratch = n;
            oop_scratch = 0;
            immut_scratch = n;
        } else {  // :88-123
            size_t w_in = width_fft->get_inplace_scratch_len(), h_in = height_fft->get_inplace_scratch_len();
            size_t h_oop = height_fft->get_outofplace_scratch_len();
            size_t max_inner = std::max(h_in, w_in);
            oop_scratch = max_inner > n ? max_inner : 0;
            inplace_scratch = n + std::max(w_in > n ? w_in : (size_t)0, h_oop);
            immut_scratch = std::max(w_in, n + h_in);
        }
    }
    size_t len() const override { return width * height; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return inplace_scratch; }
    size_t get_outofplace_scratch_len() const override { return oop_scratch; }
    size_t get_immutable_scratch_len() const override { return immut_scratch; }
    const char* name() const override { return small ? "GoodThomasAlgorithmSmall" : "GoodThomasAlgorithm"; }
    // good_thomas_algorithm.rs:483-509 (:224-248 for the large variant)
    void perform_fft_inplace(C* buffer, C* scratch_all, size_t scratch_len) const override {
        const size_t n = width * height;
        C* scratch = scratch_all;
        C* inner = scratch_all + n;
        size_t inner_len = scratch_len - n;
        for (size_t i = 0; i < n; ++i) scratch[i] = buffer[input_map[i]];
        if (!small && inner_len > n)
            width_size_fft->process_with_scratch(scratch, n, inner, inner_len);
        else
            width_size_fft->process_with_scratch(scratch, n, buffer, n);
        transpose(scratch, buffer, width, height);
        height_size_fft->process_outofplace_with_scratch(buffer, n, scratch, n, small ? nullptr : inner,
                                                         small ? 0 : inner_len);
        for (size_t i = 0; i < n; ++i) buffer[output_map[i]] = scratch[i];
    }
    // good_thomas_algorithm.rs:419-449
    void perform_fft_immut(const C* input, C* output, C* scratch, size_t scratch_len) const override {
        const size_t n = width * height;
        for (size_t i = 0; i < n; ++i) output[i] = input[input_map[i]];
        if (small) {
            width_size_fft->process_with_scratch(output, n, scratch, scratch_len);
            transpose(output, scratch, width, height);
            height_size_fft->process_with_scratch(scratch, n, output, n);
            for (size_t i = 0; i < n; ++i) output[output_map[i]] = scratch[i];
        } else {  // :250-283: scratch = [n | inner]
            width_size_fft->process_with_scratch(output, n, scratch, scratch_len);
            C* s = scratch;
            C* inner = scratch + n;
            transpose(output, s, width, height);
            height_size_fft->process_with_scratch(s, n, inner, scratch_len - n);
            for (size_t i = 0; i < n; ++i) output[output_map[i]] = s[i];
        }
    }
    // good_thomas_algorithm.rs:451-481
    void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const override {
        const size_t n = width * height;
        for (size_t i = 0; i < n; ++i) output[i] = input[input_map[i]];
        if (!small && scratch_len > n)
            width_size_fft->process_with_scratch(output, n, scratch, scratch_len);
        else
            width_size_fft->process_with_scratch(output, n, input, n);
        transpose(output, input, width, height);
        if (!small && scratch_len > n)
            height_size_fft->process_with_scratch(input, n, scratch, scratch_len);
        else
            height_size_fft->process_with_scratch(input, n, output, n);
        for (size_t i = 0; i < n; ++i) output[output_map[i]] = input[i];
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/raders_algorithm.rs:41-283
// ------------------------------------------------------------------------------------------
template <class T> struct RadersAlgorithm : Fft<T> {
    typedef cx<T> C;
    FftPtr<T> inner_fft;
    std::vector<C> inner_fft_data;
    uint64_t primitive_root_, primitive_root_inverse, len_;
    size_t inplace_scratch, oop_scratch, immut_scratch;
    Direction direction;

    // raders_algorithm.rs:65-124
    explicit RadersAlgorithm(FftPtr<T> inner) : inner_fft(inner) {
        const size_t inner_fft_len = inner->len();
        const size_t len = inner_fft_len + 1;
        if (!is_prime_u64(len)) {
            std::ostringstream m;
            m << "For raders algorithm, inner_fft.len() + 1 must be prime. Expected prime number, got " << inner_fft_len
              << " + 1 = " << len;
            throw FftPanic(m.str());
        }
        direction = inner->fft_direction();
        len_ = len;
        if (!primitive_root((uint64_t)len, &primitive_root_)) throw FftPanic("no primitive root");
        ExtGcd g = extended_gcd((int64_t)primitive_root_, (int64_t)len);
        primitive_root_inverse = (uint64_t)(g.x >= 0 ? g.x : g.x + (int64_t)len);
        const T inner_fft_scale = (T)1 / (T)inner_fft_len;
        inner_fft_data.assign(inner_fft_len, C{0, 0});
        size_t twiddle_input = 1;
        for (auto& cell : inner_fft_data) {
            C tw = compute_twiddle<T>(twiddle_input, len, direction);
            cell = tw * inner_fft_scale;
            twiddle_input = (size_t)(((uint64_t)twiddle_input * primitive_root_inverse) % len_);
        }
        const size_t required_inner_scratch = inner->get_inplace_scratch_len();
        const size_t extra_inner_scratch = required_inner_scratch <= inner_fft_len ? 0 : required_inner_scratch;
        inplace_scratch = inner_fft_len + extra_inner_scratch;
        immut_scratch = inner_fft_len + required_inner_scratch;
        oop_scratch = extra_inner_scratch;
        std::vector<C> s(required_inner_scratch, C{0, 0});
        inner->process_with_scratch(inner_fft_data.data(), inner_fft_len, s.data(), s.size());
    }
    size_t len() const override { return (size_t)len_; }
    Direction fft_direction() const override { return direction; }
    size_t get_inplace_scratch_len() const override { return inplace_scratch; }
    size_t get_outofplace_scratch_len() const override { return oop_scratch; }
    size_t get_immutable_scratch_len() const override { return immut_scratch; }
    const char* name() const override { return "RadersAlgorithm"; }
    // raders_algorithm.rs:126-172
    void perform_fft_immut(const C* input_all, C* output_all, C* scratch_all, size_t scratch_len) const override {
        const size_t m = (size_t)len_ - 1;
        C* output_first = output_all;
        C* output = output_all + 1;
        const C input_first = input_all[0];
        const C* input = input_all + 1;
        C* scratch = scratch_all;
        C* extra = scratch_all + m;
        size_t extra_len = scratch_len - m;
        uint64_t input_index = 1;
        for (size_t i = 0; i < m; ++i) {
            input_index = (input_index * primitive_root_) % len_;
            scratch[i] = input[input_index - 1];
        }
        inner_fft->process_with_scratch(scratch, m, extra, extra_len);
        *output_first = input_first + scratch[0];
        for (size_t i = 0; i < m; ++i) scratch[i] = conj(scratch[i] * inner_fft_data[i]);
        scratch[0] = scratch[0] + conj(input_first);
        inner_fft->process_with_scratch(scratch, m, extra, extra_len);
        uint64_t output_index = 1;
        for (size_t i = 0; i < m; ++i) {
            output_index = (output_index * primitive_root_inverse) % len_;
            output[output_index - 1] = conj(scratch[i]);
        }
    }
    // raders_algorithm.rs:174-234
    void perform_fft_out_of_place(C* input_all, C* output_all, C* scratch, size_t scratch_len) const override {
        const size_t m = (size_t)len_ - 1;
        C* output_first = output_all;
        C* output = output_all + 1;
        C* input_first = input_all;
        C* input = input_all + 1;
        uint64_t input_index = 1;
        for (size_t i = 0; i < m; ++i) {
            input_index = (input_index * primitive_root_) % len_;
            output[i] = input[input_index - 1];
        }
        if (scratch_len > 0)
            inner_fft->process_with_scratch(output, m, scratch, scratch_len);
        else
            inner_fft->process_with_scratch(output, m, input, m);
        *output_first = *input_first + output[0];
        for (size_t i = 0; i < m; ++i) input[i] = conj(output[i] * inner_fft_data[i]);
        input[0] = input[0] + conj(*input_first);
        if (scratch_len > 0)
            inner_fft->process_with_scratch(input, m, scratch, scratch_len);
        else
            inner_fft->process_with_scratch(input, m, output, m);
        uint64_t output_index = 1;
        for (size_t i = 0; i < m; ++i) {
            output_index = (output_index * primitive_root_inverse) % len_;
            output[output_index - 1] = conj(input[i]);
        }
    }
    // raders_algorithm.rs:235-283
    void perform_fft_inplace(C* buffer_all, C* scratch_all, size_t scratch_len) const override {
        const size_t m = (size_t)len_ - 1;
        C* buffer_first = buffer_all;
        C* buffer = buffer_all + 1;
        const C buffer_first_val = *buffer_first;
        C* scratch = scratch_all;
        C* extra = scratch_all + m;
        size_t extra_len = scratch_len - m;
        uint64_t input_index = 1;
        for (size_t i = 0; i < m; ++i) {
            input_index = (input_index * primitive_root_) % len_;
            scratch[i] = buffer[input_index - 1];
        }
        C* inner_scratch = extra_len > 0 ? extra : buffer;
        size_t inner_scratch_len = extra_len > 0 ? extra_len : m;
        inner_fft->process_with_scratch(scratch, m, inner_scratch, inner_scratch_len);
        *buffer_first = *buffer_first + scratch[0];
        for (size_t i = 0; i < m; ++i) scratch[i] = conj(scratch[i] * inner_fft_data[i]);
        scratch[0] = scratch[0] + conj(buffer_first_val);
        inner_fft->process_with_scratch(scratch, m, inner_scratch, inner_scratch_len);
        uint64_t output_index = 1;
        for (size_t i = 0; i < m; ++i) {
            output_index = (output_index * primitive_root_inverse) % len_;
            buffer[output_index - 1] = conj(scratch[i]);
        }
    }
};

// ------------------------------------------------------------------------------------------
// src/algorithm/bluesteins_algorithm.rs:39-200
// ------------------------------------------------------------------------------------------
template <class T> struct BluesteinsAlgorithm : Fft<T> {
    typedef cx<T> C;
    FftPtr<T> inner_fft;
    std::vector<C> inner_fft_multiplier, twiddles;
    size_t n;
    Direction direction;

    // bluesteins_algorithm.rs:58-98
    BluesteinsAlgorithm(size_t len, FftPtr<T> inner) : inner_fft(inner), n(len) {
        const size_t inner_fft_len = inner->len();
        if (!(len * 2 - 1 <= inner_fft_len)) {
            std::ostringstream m;
            m << "Bluestein's algorithm requires inner_fft.len() >= self.len() * 2 - 1. Expected >= " << (len * 2 - 1)
              << ", got " << inner_fft_len;
            throw FftPanic(m.str());
        }
        const T inner_fft_scale = (T)1 / (T)inner_fft_len;
        direction = inner->fft_direction();
        inner_fft_multiplier.assign(inner_fft_len, C{0, 0});
        fill_bluesteins_twiddles<T>(inner_fft_multiplier.data(), len, opposite(direction));
        inner_fft_multiplier[0] = inner_fft_multiplier[0] * inner_fft_scale;
        for (size_t i = 1; i < len; ++i) {
            C tw = inner_fft_multiplier[i] * inner_fft_scale;
            inner_fft_multiplier[i] = tw;
            inner_fft_multiplier[inner_fft_len - i] = tw;
        }
        std::vector<C> s(inner->get_inplace_scratch_len(), C{0, 0});
        inner->process_with_scratch(inner_fft_multiplier.data(), inner_fft_len, s.data(), s.size());
        twiddles.assign(len, C{0, 0});
        fill_bluesteins_twiddles<T>(twiddles.data(), len, direction);
    }
    size_t len() const override { return n; }
    Direction fft_direction() const override { return direction; }
    size_t scratch_all() const { return inner_fft_multiplier.size() + inner_fft->get_inplace_scratch_len(); }
    size_t get_inplace_scratch_len() const override { return scratch_all(); }     // :191-200
    size_t get_outofplace_scratch_len() const override { return scratch_all(); }
    size_t get_immutable_scratch_len() const override { return scratch_all(); }
    const char* name() const override { return "BluesteinsAlgorithm"; }
    // bluesteins_algorithm.rs:139-180 (and :100-136 with output == input)
    void perform_fft_immut(const C* input, C* output, C* scratch, size_t scratch_len) const override {
        const size_t m = inner_fft_multiplier.size();
        C* inner_input = scratch;
        C* inner_scratch = scratch + m;
        const size_t inner_scratch_len = scratch_len - m;
        for (size_t i = 0; i < n; ++i) inner_input[i] = input[i] * twiddles[i];
        for (size_t i = n; i < m; ++i) inner_input[i] = C{0, 0};
        inner_fft->process_with_scratch(inner_input, m, inner_scratch, inner_scratch_len);
        for (size_t i = 0; i < m; ++i) inner_input[i] = conj(inner_input[i] * inner_fft_multiplier[i]);
        inner_fft->process_with_scratch(inner_input, m, inner_scratch, inner_scratch_len);
        for (size_t i = 0; i < n; ++i) output[i] = conj(inner_input[i]) * twiddles[i];
    }
    void perform_fft_inplace(C* buffer, C* scratch, size_t scratch_len) const override {
        perform_fft_immut(buffer, buffer, scratch, scratch_len);  // element i is read before it is written
    }
    void perform_fft_out_of_place(C* input, C* output, C* scratch, size_t scratch_len) const override {
        perform_fft_immut(input, output, scratch, scratch_len);
    }
};

// ------------------------------------------------------------------------------------------
// src/plan.rs — FftPlannerScalar (recipes :134-188, design :312-323 and :412-665, build :326-410)
// ------------------------------------------------------------------------------------------
struct Recipe {
    enum Kind { DftK, MixedRadixK, GoodThomasK, MixedRadixSmallK, GoodThomasSmallK, RadersK, BluesteinsK, RadixNK, Radix4K, ButterflyK };
    Kind kind;
    size_t len_;                           // Dft len / Bluestein len / butterfly len
    std::shared_ptr<Recipe> left, right;   // left/right or inner/base
    std::vector<uint8_t> factors;          // RadixN
    uint32_t k = 0;                        // Radix4
    size_t len() const {                   // plan.rs:191-230
        switch (kind) {
            case DftK: case ButterflyK: case BluesteinsK: return len_;
            case RadixNK: {
                size_t p = left->len();
                for (uint8_t f : factors) p *= f;
                return p;
            }
            case Radix4K: return left->len() * ((size_t)1 << (k * 2));
            case RadersK: return left->len() + 1;
            default: return left->len() * right->len();
        }
    }
    std::string str() const {
        std::ostringstream s;
        switch (kind) {
            case DftK: s << "Dft(" << len_ << ")"; break;
            case ButterflyK: s << "Butterfly" << len_; break;
            case MixedRadixK: s << "MixedRadix{" << left->str() << "," << right->str() << "}"; break;
            case GoodThomasK: s << "GoodThomasAlgorithm{" << left->str() << "," << right->str() << "}"; break;
            case MixedRadixSmallK: s << "MixedRadixSmall{" << left->str() << "," << right->str() << "}"; break;
            case GoodThomasSmallK: s << "GoodThomasAlgorithmSmall{" << left->str() << "," << right->str() << "}"; break;
            case RadersK: s << "RadersAlgorithm{" << left->str() << "}"; break;
            case BluesteinsK: s << "BluesteinsAlgorithm{" << len_ << "," << left->str() << "}"; break;
            case RadixNK: {
                s << "RadixN{[";
                for (size_t i = 0; i < factors.size(); ++i) s << (i ? "," : "") << (int)factors[i];
                s << "]," << left->str() << "}";
                break;
            }
            case Radix4K: s << "Radix4{" << k << "," << left->str() << "}"; break;
        }
        return s.str();
    }
};
typedef std::shared_ptr<Recipe> RecipePtr;

template <class T> struct FftPlannerScalar {
    static constexpr size_t MAX_RADIXN_FACTOR = 7;        // plan.rs:128
    static constexpr size_t MAX_RADER_PRIME_FACTOR = 23;  // plan.rs:129
    std::map<std::pair<size_t, int>, FftPtr<T>> algorithm_cache;  // fft_cache.rs:5-39
    std::map<size_t, RecipePtr> recipe_cache;

    FftPtr<T> plan_fft(size_t len, Direction d) {  // plan.rs:289-295
        RecipePtr r = design_fft_for_len(len);
        return build_fft(*r, d);
    }
    FftPtr<T> plan_fft_forward(size_t len) { return plan_fft(len, Direction::Forward); }
    FftPtr<T> plan_fft_inverse(size_t len) { return plan_fft(len, Direction::Inverse); }

    static RecipePtr mk(Recipe::Kind k, size_t len = 0, RecipePtr l = nullptr, RecipePtr r = nullptr) {
        auto p = std::make_shared<Recipe>();
        p->kind = k;
        p->len_ = len;
        p->left = l;
        p->right = r;
        return p;
    }
    // plan.rs:312-323
    RecipePtr design_fft_for_len(size_t len) {
        if (len < 2) return mk(Recipe::DftK, len);
        auto it = recipe_cache.find(len);
        if (it != recipe_cache.end()) return it->second;
        PrimeFactors factors = PrimeFactors::compute(len);
        RecipePtr r = design_fft_with_factors(len, factors);
        recipe_cache[len] = r;
        return r;
    }
    // plan.rs:326-335
    FftPtr<T> build_fft(const Recipe& recipe, Direction d) {
        size_t len = recipe.len();
        auto key = std::make_pair(len, (int)d);
        auto it = algorithm_cache.find(key);
        if (it != algorithm_cache.end()) return it->second;
        FftPtr<T> fft = build_new_fft(recipe, d);
        algorithm_cache[key] = fft;
        return fft;
    }
    // plan.rs:338-410
    FftPtr<T> build_new_fft(const Recipe& r, Direction d) {
        switch (r.kind) {
            case Recipe::DftK: return std::make_shared<Dft<T>>(r.len_, d);
            case Recipe::ButterflyK: return std::make_shared<Butterfly<T>>(r.len_, d);
            case Recipe::RadixNK: {
                FftPtr<T> base = build_fft(*r.left, d);
                return std::make_shared<RadixN<T>>(r.factors, base);
            }
            case Recipe::Radix4K: {
                FftPtr<T> base = build_fft(*r.left, d);
                return std::make_shared<Radix4<T>>(r.k, base);
            }
            case Recipe::MixedRadixK:
            case Recipe::MixedRadixSmallK: {
                FftPtr<T> l = build_fft(*r.left, d), rr = build_fft(*r.right, d);
                return std::make_shared<MixedRadix<T>>(l, rr, r.kind == Recipe::MixedRadixSmallK);
            }
            case Recipe::GoodThomasK:
            case Recipe::GoodThomasSmallK: {
                FftPtr<T> l = build_fft(*r.left, d), rr = build_fft(*r.right, d);
                return std::make_shared<GoodThomas<T>>(l, rr, r.kind == Recipe::GoodThomasSmallK);
            }
            case Recipe::RadersK: {
                FftPtr<T> inner = build_fft(*r.left, d);
                return std::make_shared<RadersAlgorithm<T>>(inner);
            }
            case Recipe::BluesteinsK: {
                FftPtr<T> inner = build_fft(*r.left, d);
                return std::make_shared<BluesteinsAlgorithm<T>>(r.len_, inner);
            }
        }
        throw FftPanic("unreachable recipe kind");
    }
    // plan.rs:412-425
    RecipePtr design_fft_with_factors(size_t len, const PrimeFactors& factors) {
        if (RecipePtr b = design_butterfly_algorithm(len)) return b;
        if (factors.is_prime()) return design_prime(len);
        if (RecipePtr bp = design_butterfly_product(len)) return bp;
        if (factors.has_factors_leq(MAX_RADIXN_FACTOR)) return design_radixn(factors);
        auto parts = factors.partition_factors();
        return design_mixed_radix(parts.first, parts.second);
    }
    // plan.rs:427-472
    RecipePtr design_butterfly_product(size_t len) {
        if (len > 992 || (len & (len - 1)) == 0) return nullptr;
        const size_t limit = (size_t)std::ceil(std::sqrt((double)len)) + 1;
        const size_t butterflies[] = {2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 17, 19, 23, 24, 27, 29, 31, 32};
        size_t min_sum = SIZE_MAX, fl = 0, fr = 0;
        bool found = false;
        for (size_t left : butterflies) {
            if (!(left < limit)) break;  // take_while
            size_t right = len / left;
            bool contains = false;
            for (size_t b : butterflies) contains = contains || (b == right);
            if (left * right == len && contains) {
                size_t sum = left + right;
                if (sum < min_sum) {
                    min_sum = sum;
                    fl = left;
                    fr = right;
                    found = true;
                }
            }
        }
        if (!found) return nullptr;
        RecipePtr l = design_fft_for_len(fl), r = design_fft_for_len(fr);
        return mk(gcd(fl, fr) == 1 ? Recipe::GoodThomasSmallK : Recipe::MixedRadixSmallK, 0, l, r);
    }
    // plan.rs:474-506
    RecipePtr design_mixed_radix(const PrimeFactors& lf, const PrimeFactors& rf) {
        size_t left_len = lf.get_product(), right_len = rf.get_product();
        RecipePtr l = design_fft_with_factors(left_len, lf), r = design_fft_with_factors(right_len, rf);
        if (left_len < 31 && right_len < 31)
            return mk(gcd(left_len, right_len) == 1 ? Recipe::GoodThomasSmallK : Recipe::MixedRadixSmallK, 0, l, r);
        return mk(Recipe::MixedRadixK, 0, l, r);
    }
    // plan.rs:508-607
    RecipePtr design_radixn(const PrimeFactors& factors) {
        uint32_t p2 = factors.power_two, p3 = factors.power_three, p5 = 0, p7 = 0;
        for (auto& f : factors.other_factors) {
            if (f.value == 5) p5 = f.count;
            if (f.value == 7) p7 = f.count;
        }
        size_t base_len;
        if (factors.has_factors_gt(MAX_RADIXN_FACTOR)) {
            base_len = factors.product_above(MAX_RADIXN_FACTOR);
        } else if (p7 == 0 && p5 == 0 && p3 < 2) {
            if (p3 == 0) {
                if (!(p2 > 5)) throw FftPanic("assertion failed: p2 > 5");
                base_len = (p2 % 2 == 1) ? 8 : 16;
            } else {
                if (!(p2 > 3)) throw FftPanic("assertion failed: p2 > 3");
                base_len = (p2 % 2 == 1) ? 24 : 12;
            }
        } else if (p2 > 0 && p3 > 0) {
            uint32_t excess_p2 = p2 > p3 ? p2 - p3 : 0;
            base_len = excess_p2 == 0 ? 6 : (excess_p2 == 1 ? 12 : 24);
        } else if (p3 > 2) {
            base_len = 27;
        } else if (p3 > 1) {
            base_len = 9;
        } else if (p7 > 0) {
            base_len = 7;
        } else {
            if (!(p5 > 0)) throw FftPanic("assertion failed: p5 > 0");
            base_len = 5;
        }
        RecipePtr base_fft = design_fft_for_len(base_len);
        size_t cross_len = factors.get_product() / base_len;
        uint32_t cross_bits = (uint32_t)__builtin_ctzll((unsigned long long)cross_len);
        if ((cross_len & (cross_len - 1)) == 0 && cross_bits % 2 == 0) {
            RecipePtr r = mk(Recipe::Radix4K, 0, base_fft);
            r->k = cross_bits / 2;
            return r;
        }
        std::vector<uint8_t> fs;
        while (cross_len % 7 == 0) { cross_len /= 7; fs.push_back(7); }
        while (cross_len % 6 == 0) { cross_len /= 6; fs.push_back(6); }
        while (cross_len % 5 == 0) { cross_len /= 5; fs.push_back(5); }
        while (cross_len % 3 == 0) { cross_len /= 3; fs.push_back(3); }
        if ((cross_len & (cross_len - 1)) != 0) throw FftPanic("assertion failed: cross_len.is_power_of_two()");
        cross_bits = (uint32_t)__builtin_ctzll((unsigned long long)cross_len);
        if (cross_bits % 2 == 1) fs.push_back(2);
        for (uint32_t i = 0; i < cross_bits / 2; ++i) fs.push_back(4);
        RecipePtr r = mk(Recipe::RadixNK, 0, base_fft);
        r->factors = fs;
        return r;
    }
    // plan.rs:610-634
    RecipePtr design_butterfly_algorithm(size_t len) {
        if (len >= 2 && Butterfly<T>::supported(len)) return mk(Recipe::ButterflyK, len);
        return nullptr;
    }
    // plan.rs:636-665
    RecipePtr design_prime(size_t len) {
        size_t inner_fft_len_rader = len - 1;
        PrimeFactors rf = PrimeFactors::compute(inner_fft_len_rader);
        bool too_large = false;
        for (auto& f : rf.other_factors) too_large = too_large || (f.value > MAX_RADER_PRIME_FACTOR);
        if (too_large) {
            size_t min_inner_len = 2 * len - 1;
            size_t inner_len_pow2 = 1;
            while (inner_len_pow2 < min_inner_len) inner_len_pow2 <<= 1;
            size_t inner_len_factor3 = inner_len_pow2 / 4 * 3;
            size_t inner_len = inner_len_factor3 >= min_inner_len ? inner_len_factor3 : inner_len_pow2;
            RecipePtr inner = design_fft_for_len(inner_len);
            return mk(Recipe::BluesteinsK, len, inner);
        }
        RecipePtr inner = design_fft_with_factors(inner_fft_len_rader, rf);
        return mk(Recipe::RadersK, 0, inner);
    }
};

}  // namespace rustfft_oracle
