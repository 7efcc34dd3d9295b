// Run-time scheduled workgroup transform: the same Stockham law as engine.h, but the length, the radix list and
// the thread layout arrive as kernel parameters, so ONE compiled kernel serves every 13-smooth length that fits a
// workgroup (the GPU counterpart of the reference's RadixN, src/algorithm/radixn.rs:35-333, which also takes its
// factor list at run time).  Arithmetic per radix is still fully unrolled: the sub-pass loop dispatches through a
// switch to dyn_pass<R>, whose register indices are compile-time constants.
//
// Registers: EMAX complex values per thread; a radix-R sub-pass gives a thread up to EMAX / R butterflies.
#pragma once
#include "butterflies.h"

namespace mi355 {

constexpr int kDynMaxPass = 8;

struct DynSched {
    int n, np, tpf, f;             // length, sub-passes, threads per sequence, sequences per workgroup
    int radix[kDynMaxPass];
    int stride[kDynMaxPass];       // s_p = R_0 .. R_{p-1}
    int nb[kDynMaxPass];           // n / R_p
    int bpt[kDynMaxPass];          // butterflies per thread in sub-pass p
    int tw_off[kDynMaxPass];       // offset of sub-pass p's twiddles ([k-1][b mod s_p] layout, as Sched::tw_offset)
    unsigned rcp_stride[kDynMaxPass];  // ceil(2^32 / s_p): b / s_p == mulhi(b, rcp) for b, s_p < 2^16
    unsigned rcp_tpf;
    int pitch;                     // LDS elements between sequences
    int light;                     // radix set of the schedule: 1 = LIGHT (EMAX = 12 kernel), 2 = HEAVY (17 .. 31, EMAX = 32), 0 = full
};

MI_HD unsigned dyn_mulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
MI_HD int dyn_phys(int i) { return i + (i >> 5); }  // one padding slot per 32 elements

// one sub-pass for radix R: [gather | source] -> twiddle -> butterfly -> [scatter | destination]
template <int R, int EMAX, class T, class SRC, class DST>
MI_HD void dyn_pass_compute(const DynSched& s, int p, int f, int u, cx<T>* v, const cx<T>* MI_RESTRICT tw, cx<T>* ldsf, bool last,
                            DST& dst) {
    constexpr int MMAX = EMAX / R;
    const int nb = s.nb[p], st = s.stride[p], bpt = s.bpt[p];
    static_for<0, MMAX>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * s.tpf;
        if (m < bpt && b < nb) {
            const int q = (st > 1) ? (int)dyn_mulhi((unsigned)b, s.rcp_stride[p]) : b;
            const int r = (st > 1) ? b - q * st : 0;
            if (st > 1) {
                const cx<T>* t = tw + s.tw_off[p] + r;
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    v[m * R + k] = v[m * R + k] * t[(k - 1) * st];
                });
            }
            butterfly<R>(v + m * R);
            const int base = q * st * R + r;
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                if (last)
                    dst(f, base + k * st, v[m * R + k]);
                else
                    ldsf[dyn_phys(base + k * st)] = v[m * R + k];
            });
        }
    });
}
template <int R, int EMAX, class T, class SRC>
MI_HD void dyn_pass_load(const DynSched& s, int p, int f, int u, cx<T>* v, const cx<T>* ldsf, bool first, SRC& src) {
    constexpr int MMAX = EMAX / R;
    const int nb = s.nb[p], bpt = s.bpt[p];
    static_for<0, MMAX>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * s.tpf;
        if (m < bpt && b < nb) {
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                v[m * R + k] = first ? src(f, b + k * nb) : ldsf[dyn_phys(b + k * nb)];
            });
        }
    });
}

// LIGHT = 1 compiles only the radices {2,3,4,5,6,8,9,10,12}: a kernel without the 7/11/13/16-point butterflies
// needs about half the registers, i.e. twice the resident workgroups, for lengths of the form 2^a 3^b 5^c.
// LIGHT = 2 (HEAVY) adds the prime radices 17 .. 31 (the reference's Butterfly17 .. Butterfly31,
// src/algorithm/butterflies.rs:1582-6241) with 32 values per thread: lengths with such a factor beyond the compiled set.
#define MI_DYN_RADIX_SWITCH(LIGHT, RADIX, CALL)     \
    switch (RADIX) {                         \
        case 2: { constexpr int RR = 2; CALL; } break;   \
        case 3: { constexpr int RR = 3; CALL; } break;   \
        case 4: { constexpr int RR = 4; CALL; } break;   \
        case 5: { constexpr int RR = 5; CALL; } break;   \
        case 6: { constexpr int RR = 6; CALL; } break;   \
        case 7: if constexpr ((LIGHT) != 1) { constexpr int RR = 7; CALL; } break;   \
        case 8: { constexpr int RR = 8; CALL; } break;   \
        case 9: { constexpr int RR = 9; CALL; } break;   \
        case 10: { constexpr int RR = 10; CALL; } break; \
        case 11: if constexpr ((LIGHT) != 1) { constexpr int RR = 11; CALL; } break; \
        case 12: { constexpr int RR = 12; CALL; } break; \
        case 13: if constexpr ((LIGHT) != 1) { constexpr int RR = 13; CALL; } break; \
        case 16: if constexpr ((LIGHT) != 1) { constexpr int RR = 16; CALL; } break; \
        case 17: if constexpr ((LIGHT) == 2) { constexpr int RR = 17; CALL; } break; \
        case 19: if constexpr ((LIGHT) == 2) { constexpr int RR = 19; CALL; } break; \
        case 23: if constexpr ((LIGHT) == 2) { constexpr int RR = 23; CALL; } break; \
        case 29: if constexpr ((LIGHT) == 2) { constexpr int RR = 29; CALL; } break; \
        case 31: if constexpr ((LIGHT) == 2) { constexpr int RR = 31; CALL; } break; \
        default: break;                      \
    }

// X: executor (launch.h).  src(f, i), dst(f, i, value) as in engine.h.  SRC_IN_LDS as in wg_fft.
template <class T, int EMAX, int LIGHT, bool SRC_IN_LDS = false, class X, class SRC, class DST>
MI_HD void wg_fft_dyn(X& ex, const DynSched& s, void* lds_raw, const cx<T>* MI_RESTRICT tw, SRC src, DST dst) {
    cx<T>* lds = (cx<T>*)lds_raw;
    for (int p = 0; p < s.np; ++p) {
        const bool first = (p == 0), last = (p == s.np - 1);
        const int radix = s.radix[p];
        ex.for_threads([&](int tid, cx<T>* v) {
            const int f = s.tpf > 1 ? (int)dyn_mulhi((unsigned)tid, s.rcp_tpf) : tid, u = tid - f * s.tpf;
            if (f < s.f) {
                cx<T>* ldsf = lds + f * s.pitch;
                MI_DYN_RADIX_SWITCH(LIGHT, radix, (dyn_pass_load<RR, EMAX, T>(s, p, f, u, v, ldsf, first, src)));
            }
        });
        if (!first || SRC_IN_LDS) ex.barrier();  // every gather done before this sub-pass scatters into the same buffer
        ex.for_threads([&](int tid, cx<T>* v) {
            const int f = s.tpf > 1 ? (int)dyn_mulhi((unsigned)tid, s.rcp_tpf) : tid, u = tid - f * s.tpf;
            if (f < s.f) {
                cx<T>* ldsf = lds + f * s.pitch;
                MI_DYN_RADIX_SWITCH(LIGHT, radix, (dyn_pass_compute<RR, EMAX, T, SRC, DST>(s, p, f, u, v, tw, ldsf, last, dst)));
            }
        });
        if (!last) ex.barrier();
    }
}

}  // namespace mi355
