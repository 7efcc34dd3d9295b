// Complex arithmetic + compile-time twiddle constants shared by the gfx950 kernels and the
// host-side kernel-body emulator (tests/emu).  Layout matches num_complex::Complex<T>
// (#[repr(C)] {re, im}), i.e. what RustFFT's Fft<T>::process() receives (src/lib.rs:195).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__ __forceinline__
#define MI_RESTRICT __restrict__
#else
#define MI_HD inline __attribute__((always_inline))
#define MI_RESTRICT __restrict__
#endif

// Scheduling fence between independent butterflies of one thread: keeps the compiler from interleaving them
// (which doubles the live temporaries and spills in the 32-values-per-thread tiles).  No-op on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MI_SCHED_FENCE() ((void)0)
#endif

// A value the code knows to be workgroup-uniform although the compiler cannot prove it (the result of a division expanded into
// vector instructions, a ticket read from LDS): v_readfirstlane moves it to an SGPR, and so everything derived from it.
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_UNIFORM(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define MI_UNIFORM(x) ((unsigned)(x))
#endif

namespace mi355 {

template <class T> struct cx {
    T re, im;
};

// Register barrier over eight complex values: an empty volatile asm that redefines them.  Volatile asm statements keep
// their order, so arithmetic that produces these values stays above the statement and arithmetic that consumes them
// stays below it -- a wall the IR-level vectoriser cannot pair instructions across (MI_SCHED_FENCE only binds the machine
// scheduler; without this the SLP vectoriser fuses neighbouring butterflies into packed operations, doubles the live
// twiddles and spills the 32-values-per-thread tiles).  No instruction is emitted.  No-op on the host.
template <class T> MI_HD void reg_wall8(cx<T>* x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x[0].re), "+v"(x[0].im), "+v"(x[1].re), "+v"(x[1].im), "+v"(x[2].re), "+v"(x[2].im), "+v"(x[3].re), "+v"(x[3].im),
                      "+v"(x[4].re), "+v"(x[4].im), "+v"(x[5].re), "+v"(x[5].im), "+v"(x[6].re), "+v"(x[6].im), "+v"(x[7].re), "+v"(x[7].im));
#else
    (void)x;
#endif
}
template <class T> MI_HD cx<T> operator+(cx<T> a, cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <class T> MI_HD cx<T> operator-(cx<T> a, cx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <class T> MI_HD cx<T> operator*(cx<T> a, cx<T> b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(MI355_PK_CMUL)
    // f32 on gfx950: two packed instructions, the re/im swap and the sign folded into the op_sel / neg_lo modifiers
    // (the compiler's own selection materialises swapped copies of loop-invariant factors in extra registers)
    if constexpr (sizeof(T) == 4) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f va = {a.re, a.im}, vb = {b.re, b.im}, t, r;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
        return {r.x, r.y};
    }
#endif
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class T> MI_HD cx<T> operator*(cx<T> a, T s) { return {a.re * s, a.im * s}; }
#if defined(__HIP_DEVICE_COMPILE__) && defined(MI355_PK)
// EXPERIMENT: Complex<float> arithmetic as <2 x float> vector operations (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
typedef float mi_v2f __attribute__((ext_vector_type(2)));
MI_HD cx<float> operator+(cx<float> a, cx<float> b) {
    mi_v2f r = mi_v2f{a.re, a.im} + mi_v2f{b.re, b.im};
    return {r.x, r.y};
}
MI_HD cx<float> operator-(cx<float> a, cx<float> b) {
    mi_v2f r = mi_v2f{a.re, a.im} - mi_v2f{b.re, b.im};
    return {r.x, r.y};
}
MI_HD cx<float> operator*(cx<float> a, float s) {
    mi_v2f r = mi_v2f{a.re, a.im} * mi_v2f{s, s};
    return {r.x, r.y};
}
#endif
template <class T> MI_HD cx<T> cconj(cx<T> a) { return {a.re, -a.im}; }
// multiply by -i (forward quarter turn; the reference's rotate_90, src/twiddles.rs:59-70)
template <class T> MI_HD cx<T> mul_neg_i(cx<T> a) { return {a.im, -a.re}; }
template <class T> MI_HD cx<T> mul_pos_i(cx<T> a) { return {-a.im, a.re}; }

// Non-temporal (streaming) global accesses.  Measured with tools/membench on MI355X: an in-place load-all /
// store-all block copy runs at 5.9 TB/s with nt loads + nt stores against 5.3 TB/s with plain accesses; for
// out-of-place streams only the nt LOAD helps (nt stores cost ~6 %).  Plain accesses on the host emulator.
template <class T> MI_HD cx<T> ld_nt(const cx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 t = __builtin_nontemporal_load((const vec2*)p);
    return cx<T>{t.x, t.y};
#else
    return *p;
#endif
}
template <class T> MI_HD void st_nt(cx<T>* p, cx<T> v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 t;
    t.x = v.re;
    t.y = v.im;
    __builtin_nontemporal_store(t, (vec2*)p);
#else
    *p = v;
#endif
}

// the same for a PAIR of adjacent values (16 bytes for Complex<float>: the two-columns-per-lane tiles); C2 = {cx<T> a, b}
template <class T, class C2> MI_HD C2 ld_nt2(const cx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T vec4 __attribute__((ext_vector_type(4)));
    vec4 t = __builtin_nontemporal_load((const vec4*)p);
    C2 q;
    q.a = cx<T>{t.x, t.y};
    q.b = cx<T>{t.z, t.w};
    return q;
#else
    return *(const C2*)p;
#endif
}
template <class T, class C2> MI_HD void st_nt2(cx<T>* p, const C2& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T vec4 __attribute__((ext_vector_type(4)));
    vec4 t;
    t.x = q.a.re;
    t.y = q.a.im;
    t.z = q.b.re;
    t.w = q.b.im;
    __builtin_nontemporal_store(t, (vec4*)p);
#else
    *(C2*)p = q;
#endif
}

// Agent-scope (device-coherent) accesses for data that another workgroup produces or consumes inside ONE launch (the fused
// two-pass kernel's ring): relaxed agent-scope atomics lower to global_load / global_store ... sc1 -- the load is served past
// this CU's L1 (which no other CU's store ever refreshes), the store is written through the XCD's L2 (whose dirty lines no
// other XCD can see) -- so neither a release fence (buffer_wbl2: the whole L2 writes back) nor an acquire fence (buffer_inv:
// the whole L1 is invalidated) is needed per tile.  Plain accesses on the host emulator.
template <class T> MI_HD cx<T> ld_agent(const cx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) {
        const unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cx<T> v;
        __builtin_memcpy(&v, &u, 8);
        return v;
    } else {
        const unsigned long long a = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cx<T> v;
        __builtin_memcpy(&v.re, &a, 8);
        __builtin_memcpy(&v.im, &b, 8);
        return v;
    }
#else
    return *p;
#endif
}
template <class T> MI_HD void st_agent(cx<T>* p, cx<T> v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) {
        unsigned long long u;
        __builtin_memcpy(&u, &v, 8);
        __hip_atomic_store((unsigned long long*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        // ONE 16-byte write-through store (the atomic builtins stop at 8 bytes: two of them per element cost Complex<f64> 16 - 26 %,
        // profiles/r4/ab_fused_f64_2p*.jsonl)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d t = {v.re, v.im};
        // (s_nop 1 inside the statement: the compiler pads nothing in an asm string, and its next instruction may otherwise overwrite the
        // data registers of a 16-byte store before the store has read them -- cdna_hip_programming.md section 5.7; found the hard way:
        // without it 2^17, 2^18 and 2^20 came out wrong on the device while every emulator test passed)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
    }
#else
    *p = v;
#endif
}

// ---- compile-time exp(-2*pi*i*m/n) -----------------------------------------------------------
// Exact octant reduction on the integers (m, n), then a Taylor series on [0, pi/4]; evaluated by
// the compiler, so radix-internal constants cost no table traffic.  Accuracy ~1e-16.
namespace detail {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double sin_small(double x) {
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i <= 14; ++i) {
        term *= -x2 / double((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double cos_small(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i <= 14; ++i) {
        term *= -x2 / double((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
struct cd {
    double re, im;
};
// cos/sin of 2*pi*m/n for 0 <= m < n
constexpr cd unit_root(long long m, long long n) {
    m %= n;
    if (m < 0) m += n;
    // quadrant q in 0..3 and remainder r: 4m = q*n + r, angle = q*pi/2 + (pi/2)*(r/n)
    long long q = (4 * m) / n, r = (4 * m) % n;
    double c = 0, s = 0;
    if (2 * r <= n) {  // phi <= pi/4
        double phi = (kPi / 2) * double(r) / double(n);
        c = cos_small(phi);
        s = sin_small(phi);
    } else {  // use the complement so the series argument stays <= pi/4
        double psi = (kPi / 2) * double(n - r) / double(n);
        c = sin_small(psi);
        s = cos_small(psi);
    }
    switch (q) {
        case 0: return {c, s};
        case 1: return {-s, c};
        case 2: return {-c, -s};
        default: return {s, -c};
    }
}
}  // namespace detail

// forward twiddle exp(-2*pi*i*m/n) as a compile-time constant of type cx<T>
template <class T, int M, int N> MI_HD constexpr cx<T> ctw() {
    constexpr detail::cd u = detail::unit_root(M, N);
    return cx<T>{(T)u.re, (T)(-u.im)};
}

}  // namespace mi355
