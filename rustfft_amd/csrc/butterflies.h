// Register-resident forward DFT butterflies for the gfx950 kernels (and the host emulator).
//
// These replace the inner loops of the reference's src/algorithm/butterflies.rs (Butterfly2..32)
// and src/algorithm/radixn.rs:338-490 (butterfly_2..7): one thread holds the R inputs of a radix-R
// butterfly in VGPRs, everything is unrolled at compile time, constants are folded by the compiler.
// Only the FORWARD transform is implemented; the kernels obtain the inverse as
// conj(FFT(conj(x))) (sign flips on load/store), so no second set of constants exists.
//
// Small sizes follow the same decompositions the reference hard-codes:
//   3, 5, 7, 11, 13, ... : conjugate-symmetric prime DFT (butterflies.rs:230-248, 419-470, 615-716,
//                          tools/genbutterflies.py:53-114)
//   4 = 2x2 with a quarter turn (butterflies.rs:265-293), 8 = 4x2 with root-half shortcuts (:734-777),
//   16 / 32 / 6 / 9 / 10 / 12 / 14 / 15 ... : one mixed-radix step A x B in registers.
#pragma once
#include <type_traits>
#include <utility>

#include "cx.h"

namespace mi355 {

template <int I, int N, class F> MI_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// v *= exp(-2*pi*i*M/N) with the trivial rotations strength-reduced
template <int M0, int N, class T> MI_HD cx<T> mul_ctw(cx<T> v) {
    constexpr int M = ((M0 % N) + N) % N;
    if constexpr (M == 0) {
        return v;
    } else if constexpr (4 * M == N) {
        return mul_neg_i(v);
    } else if constexpr (2 * M == N) {
        return cx<T>{-v.re, -v.im};
    } else if constexpr (4 * M == 3 * N) {
        return mul_pos_i(v);
    } else if constexpr (8 * M == N) {  // (1 - i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210484903928;
        return cx<T>{(v.re + v.im) * h, (v.im - v.re) * h};
    } else if constexpr (8 * M == 3 * N) {  // (-1 - i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210484903928;
        return cx<T>{(v.im - v.re) * h, -(v.re + v.im) * h};
    } else if constexpr (8 * M == 5 * N) {  // (-1 + i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210484903928;
        return cx<T>{-(v.re + v.im) * h, (v.re - v.im) * h};
    } else if constexpr (8 * M == 7 * N) {  // (1 + i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210484903928;
        return cx<T>{(v.re - v.im) * h, (v.re + v.im) * h};
    } else {
        constexpr cx<T> w = ctw<T, M, N>();
        return v * w;
    }
}

constexpr bool is_prime_c(int n) {
    if (n < 2) return false;
    for (int d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}
// first factor of the in-register mixed-radix step
constexpr int first_factor(int r) {
    if (r % 4 == 0 && r > 4) return 4;
    if (r % 2 == 0 && r > 2) return 2;
    for (int d = 3; d * d <= r; d += 2)
        if (r % d == 0) return d;
    return r;
}

template <int R, class T> MI_HD void butterfly(cx<T>* v);

// conjugate-symmetric odd-prime DFT: (P-1)^2/2 real multiply-adds per output pair set
template <int P, class T> MI_HD void butterfly_prime(cx<T>* v) {
    constexpr int H = (P + 1) / 2;
    cx<T> xp[H], xn[H];
    static_for<1, H>([&](auto N_) {
        constexpr int n = N_;
        xp[n] = v[n] + v[P - n];
        xn[n] = v[n] - v[P - n];
    });
    const cx<T> x0 = v[0];
    cx<T> sum = x0;
    static_for<1, H>([&](auto N_) {
        constexpr int n = N_;
        sum = sum + xp[n];
    });
    v[0] = sum;
    static_for<1, H>([&](auto N_) {
        constexpr int n = N_;
        cx<T> a = x0, b = cx<T>{0, 0};
        static_for<1, H>([&](auto M_) {
            constexpr int m = M_;
            constexpr int mn = (m * n) % P;
            constexpr detail::cd u = detail::unit_root(mn, P);  // cos, sin of +2*pi*mn/P
            constexpr T c = (T)u.re, s = (T)u.im;
            a.re += c * xp[m].re;
            a.im += c * xp[m].im;
            b.re += s * xn[m].re;
            b.im += s * xn[m].im;
        });
        // X_n = A - i B ; X_{P-n} = A + i B
        v[n] = cx<T>{a.re + b.im, a.im - b.re};
        v[P - n] = cx<T>{a.re - b.im, a.im + b.re};
    });
}

template <int R, class T> MI_HD void butterfly(cx<T>* v) {
    if constexpr (R == 1) {
    } else if constexpr (R == 2) {
        cx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else if constexpr (R == 4) {
        cx<T> a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = mul_neg_i(v[1] - v[3]);
        v[0] = a + c;
        v[1] = b + d;
        v[2] = a - c;
        v[3] = b - d;
    } else if constexpr (is_prime_c(R)) {
        butterfly_prime<R>(v);
    } else {
        // one mixed-radix step R = A x B:  n = B*a + b,  k = ka + A*kb
        constexpr int A = first_factor(R), B = R / A;
        cx<T> t[R];
        static_for<0, B>([&](auto B_) {
            constexpr int b = B_;
            cx<T> u[A];
            static_for<0, A>([&](auto A_) {
                constexpr int a = A_;
                u[a] = v[B * a + b];
            });
            butterfly<A>(u);
            static_for<0, A>([&](auto K_) {
                constexpr int ka = K_;
                t[ka * B + b] = mul_ctw<b * ka, R>(u[ka]);
            });
        });
        static_for<0, A>([&](auto K_) {
            constexpr int ka = K_;
            cx<T> u[B];
            static_for<0, B>([&](auto B_) {
                constexpr int b = B_;
                u[b] = t[ka * B + b];
            });
            butterfly<B>(u);
            static_for<0, B>([&](auto KB_) {
                constexpr int kb = KB_;
                v[ka + A * kb] = u[kb];
            });
        });
    }
}

}  // namespace mi355
