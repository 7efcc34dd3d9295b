// Workgroup-level Stockham FFT engine for gfx950 (wave64, LDS exchange), shared with the host
// emulator of the kernel bodies (tests/emu) through the `X` executor parameter.
//
// One workgroup transforms F independent length-N sequences.  TPF threads cooperate on each
// sequence; a thread keeps the inputs of its radix-R butterflies in VGPRs, and the only data
// movement between the NP sub-passes is one transposition through LDS.  Sub-pass p (radix R_p,
// stride s_p = R_0..R_{p-1}, nb_p = N/R_p butterflies), butterfly b:
//     in_k  = x[b + k*nb_p]                       * w_{s_p R_p}^{(b mod s_p) k}
//     out_k -> y[(b div s_p) s_p R_p + (b mod s_p) + k s_p]
// (autosort form: natural order in, natural order out, no digit reversal pass — this is what
// replaces the reference's bitreversed_transpose / factor_transpose + butterfly_N column loops,
// src/array_utils.rs:372-509 and src/algorithm/radixn.rs:338-490.)
//
// The first sub-pass reads straight from the caller's `src` functor (global memory, or LDS for the
// Rader/Bluestein bodies) and the last one writes straight to `dst`, so an NP-pass transform costs
// NP-1 LDS round trips.  Thread -> (sequence f, slot u) mappings:
//   MAP_EF  element-fastest: lanes walk along one sequence  (contiguous batched transforms)
//   MAP_FF  fft-fastest:     lanes walk across the F sequences (column tiles of the large-N passes)
// SPLIT = true exchanges the real and imaginary planes one after the other through a half-size
// LDS buffer (4 barriers instead of 2 per exchange) so that R*F = 16K-element tiles fit twice per CU.
#pragma once
#include "butterflies.h"

namespace mi355 {

enum Map { MAP_EF = 0, MAP_FF = 1, MAP_FFP = 2 };



// LIN: power-of-two schedules normally exchange through the XOR-swizzled layout; LIN = true selects the linear (padded)
// layout instead (SchedL below) -- the swizzle needs one live address register per element, which the two-transform
// bodies (Bluestein) cannot afford.  The choice is part of the type, so both can coexist in one library.
template <bool LIN, int N_, int TPF_, int... Rs> struct SchedImpl {
    static constexpr bool kLinearLds = LIN;
    static constexpr int N = N_, TPF = TPF_, NP = (int)sizeof...(Rs);
    static constexpr int R[sizeof...(Rs)] = {Rs...};
    static constexpr int radix(int p) { return R[p]; }
    static constexpr int stride(int p) {
        int s = 1;
        for (int q = 0; q < p; ++q) s *= R[q];
        return s;
    }
    static constexpr int nb(int p) { return N / R[p]; }
    static constexpr int bpt(int p) { return (nb(p) + TPF - 1) / TPF; }
    static constexpr int emax() {
        int e = 0;
        for (int p = 0; p < NP; ++p) e = (R[p] * bpt(p) > e) ? R[p] * bpt(p) : e;
        return e;
    }
    // sub-pass twiddle table: pass p >= 1 owns (R_p - 1) * s_p entries, laid out [k-1][b mod s_p]
    static constexpr int tw_offset(int p) {
        int o = 0;
        for (int q = 1; q < p; ++q) o += (R[q] - 1) * stride(q);
        return o;
    }
    static constexpr int tw_total() { return tw_offset(NP); }
    // ---- LDS layout of exchange x (written by sub-pass x, read by sub-pass x+1) ----------------------------
    // Power-of-two schedules use an XOR swizzle: sub-pass x scatters runs of s_x consecutive elements at stride
    // s_x R_x; XOR-ing the run index (scaled by s_x) into the bank bits puts the runs of one wave on distinct
    // banks, while any 32-aligned block of consecutive elements (what the gather side reads) only sees a
    // constant XOR, i.e. stays conflict-free (bank math: MI355X_MICROARCH.md §LDS).  Other schedules fall back to
    // one padding slot every R_0 elements.
    static constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
    static constexpr bool all_pow2() {
        for (int p = 0; p < NP; ++p)
            if (!is_pow2(R[p])) return false;
        return N >= 32;
    }
    static constexpr int ilog2(int v) {
        int l = 0;
        while ((1 << (l + 1)) <= v) ++l;
        return l;
    }
    static constexpr int paddiv() { return (!all_pow2() && NP > 1 && R[0] % 2 == 0) ? R[0] : 0; }
    static constexpr int phys(int i) {
        constexpr int d = paddiv();
        if constexpr (all_pow2() && (emax() > 16 || kLinearLds))
            return i + i / 32;
        else if constexpr (d != 0)
            return i + i / d;
        else
            return i;
    }
    // pitch between sequences: PITCH_MOD32 = 1 (odd) suits lanes that walk across >= 32 sequences; the column-tile
    // kernels with F < 32 ask for 32 / F so that the (32 / F) slots of one lane group land on the remaining banks
    template <int PITCH_MOD32> static constexpr int pitch_for() {
        int p = phys(N - 1) + 1;
        while ((p % 32) != (PITCH_MOD32 % 32)) ++p;
        return p;
    }
    static constexpr int pitch() { return pitch_for<1>(); }
    static constexpr bool valid() {
        int prod = 1;
        for (int p = 0; p < NP; ++p) prod *= R[p];
        return prod == N;
    }
};
template <int N_, int TPF_, int... Rs> struct Sched : SchedImpl<false, N_, TPF_, Rs...> {};
template <int N_, int TPF_, int... Rs> struct SchedL : SchedImpl<true, N_, TPF_, Rs...> {};

// The same schedule with its sub-passes in REVERSE order.  The last sub-pass of a transform (radix R) leaves butterfly b's
// outputs y[b + k N/R] in a thread's registers -- exactly the inputs x[b + k N/R] of a FIRST sub-pass of radix R with the
// same thread layout.  So a transform that follows another one element-wise (Bluestein's two inner transforms, with the
// spectrum multiply in between) can run the reversed schedule and pick its inputs up from the registers: no LDS staging
// of the intermediate spectrum, one exchange and one barrier fewer per row.
template <class Acc, int... Rs> struct rev_pack;
template <template <int, int, int...> class TT, int N_, int TPF_, int... Acc, int R0, int... Rs>
struct rev_pack<TT<N_, TPF_, Acc...>, R0, Rs...> : rev_pack<TT<N_, TPF_, R0, Acc...>, Rs...> {};
template <template <int, int, int...> class TT, int N_, int TPF_, int... Acc> struct rev_pack<TT<N_, TPF_, Acc...>> {
    using type = TT<N_, TPF_, Acc...>;
};
template <class S> struct reversed_sched;
template <int N_, int TPF_, int... Rs> struct reversed_sched<Sched<N_, TPF_, Rs...>> : rev_pack<Sched<N_, TPF_>, Rs...> {};
template <int N_, int TPF_, int... Rs> struct reversed_sched<SchedL<N_, TPF_, Rs...>> : rev_pack<SchedL<N_, TPF_>, Rs...> {};
// destination marker: the last sub-pass leaves its outputs in the register array (slot m R + k = output k of butterfly m)
struct KeepInRegs {};

// Two layouts for power-of-two schedules:
//  * swizzle (threads holding <= 16 values): XOR the run index of the scatter into the bank bits.  Sub-pass x scatters
//    runs of s_x consecutive elements at stride s_x R_x; the XOR puts the runs of one wave on distinct banks, and any
//    32-aligned block of consecutive elements (what the gather reads) only sees a constant XOR: conflict-free both ways
//    (measured: SQ_LDS_BANK_CONFLICT 43 % -> 0 of the LDS cycles).
//  * linear (the 32-values-per-thread tiles): one padding slot per 32 elements; 2-4-way conflicts remain on some
//    patterns, but every address is base + constant, which is what keeps those kernels from spilling.
template <class S> constexpr bool lds_swizzled() { return S::all_pow2() && S::emax() <= 16 && !S::kLinearLds; }
template <class S, int X> MI_HD int lds_phys(int i) {
    if constexpr (lds_swizzled<S>()) {
        constexpr int ST = S::stride(X), SR = S::stride(X) * S::R[X];
        if constexpr (ST < 32)
            return i ^ (((i >> S::ilog2(SR)) << S::ilog2(ST)) & 31);
        else
            return i;
    } else if constexpr (S::all_pow2()) {
        return i + (i >> 5);
    } else if constexpr (S::paddiv() != 0) {
        return i + i / S::paddiv();
    } else {
        return i;
    }
}
// compile-time part of phys(base + k*step) - phys(base) when it is separable (see above); -1 when it is not
template <class S> constexpr int lds_step_const(int k, int step, int span /* s_p R_p for scatters, 0 for gathers */) {
    if (lds_swizzled<S>()) return -1;
    if (!S::all_pow2()) {
        // one padding slot per D = R_0 elements (or none): every stride of the schedule past sub-pass 0 is a multiple
        // of R_0 and so is every gather step N / R_p, so these layouts are separable as well
        const int d = S::paddiv();
        if (d == 0) return k * step;
        if (step % d == 0) return k * (step + step / d);
        if (span != 0 && d % step == 0 && span % d == 0) return k * step + (k * step) / d;
        return -1;
    }
    if (step % 32 == 0) return k * (step + step / 32);
    if (span != 0 && 32 % step == 0 && span % 32 == 0) return k * step + (k * step) / 32;
    return -1;
}

//   MAP_FFP  fft-fastest, PAIRED: as MAP_FF, but the slots u and u + TPF/2 of one column sit in lanes l and l + 32 of
//            one wave, so the first exchange of a pair-fused schedule is a v_permlane32_swap instead of an LDS round trip
template <Map M, int F, int TPF> MI_HD void map_tid(int tid, int& f, int& u) {
    if constexpr (M == MAP_EF) {
        u = tid % TPF;
        f = tid / TPF;
    } else if constexpr (M == MAP_FFP) {
        static_assert(F <= 32 && 32 % F == 0 && TPF % 2 == 0, "paired map: a 32-lane half holds whole groups of F columns");
        const int lane = tid & 63, w = tid >> 6;
        f = lane % F;
        u = w * (32 / F) + (lane & 31) / F + (TPF / 2) * (lane >> 5);
    } else {
        f = tid % F;
        u = tid / F;
    }
}

// ---- one sub-pass worth of arithmetic on the registers ------------------------------------------
// Twiddles kept in the thread's register array (kernels that push many sequences through one workgroup load them
// once): slot TWREG + twreg_offset(P, m) + k - 1 holds the factor of input k of the thread's m-th butterfly in pass P.
template <class S> constexpr int twreg_offset(int P, int m) {
    int o = 0;
    for (int q = 1; q < P; ++q) o += S::bpt(q) * (S::R[q] - 1);
    return o + (P < S::NP ? m * (S::R[P] - 1) : 0);
}
template <class S> constexpr int twreg_count() { return twreg_offset<S>(S::NP, 0); }
template <class T, class S, int TWREG> MI_HD void preload_twiddles(cx<T>* v, int u, const cx<T>* MI_RESTRICT tw) {
    static_for<1, S::NP>([&](auto P_) {
        constexpr int P = P_;
        constexpr int R = S::R[P], NB = S::nb(P), ST = S::stride(P), BPT = S::bpt(P);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S::TPF;
            const cx<T>* t = tw + S::tw_offset(P) + (((m + 1) * S::TPF <= NB || b < NB) ? (b % ST) : 0);
            static_for<1, R>([&](auto K_) {
                constexpr int k = K_;
                v[TWREG + twreg_offset<S>(P, m) + k - 1] = t[(k - 1) * ST];
            });
        });
    });
}

// the same for one sub-pass: issued next to the gather that precedes the pass, i.e. BEFORE the barrier, so the table
// look-up's latency overlaps the barrier instead of following it (TWSTAGE kernels)
template <class T, class S, int P, int TWREG> MI_HD void preload_twiddles_pass(cx<T>* v, int u, const cx<T>* MI_RESTRICT tw) {
    constexpr int R = S::R[P], NB = S::nb(P), ST = S::stride(P), BPT = S::bpt(P);
    static_for<0, BPT>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * S::TPF;
        const cx<T>* t = tw + S::tw_offset(P) + (((m + 1) * S::TPF <= NB || b < NB) ? (b % ST) : 0);
        static_for<1, R>([&](auto K_) {
            constexpr int k = K_;
            v[TWREG + twreg_offset<S>(P, m) + k - 1] = t[(k - 1) * ST];
        });
    });
}

template <class T, class S, int P, int TWREG = -1> MI_HD void compute_pass(cx<T>* v, int u, const cx<T>* MI_RESTRICT tw) {
    constexpr int R = S::R[P], NB = S::nb(P), ST = S::stride(P), BPT = S::bpt(P);
    static_for<0, BPT>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * S::TPF;
        if ((m + 1) * S::TPF <= NB || b < NB) {
            if constexpr (ST > 1 && TWREG >= 0) {
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    v[m * R + k] = v[m * R + k] * v[TWREG + twreg_offset<S>(P, m) + k - 1];
                });
            } else if constexpr (ST > 1) {
                const cx<T>* t = tw + S::tw_offset(P) + (b % ST);
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    v[m * R + k] = v[m * R + k] * t[(k - 1) * ST];
                });
            }
            butterfly<R>(v + m * R);
        }
        if constexpr (BPT > 1 && R * BPT > 16) MI_SCHED_FENCE();
    });
}

// ---- LDS exchange halves ---------------------------------------------------------------------------
// PART: 0 = whole complex value (E = cx<T>), 1 = real plane, 2 = imaginary plane (E = T)
template <class T, class S, int P, int PART, class E> MI_HD void lds_scatter(const cx<T>* v, int u, E* ldsf) {
    constexpr int R = S::R[P], NB = S::nb(P), ST = S::stride(P), BPT = S::bpt(P);
    static_for<0, BPT>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * S::TPF;
        if ((m + 1) * S::TPF <= NB || b < NB) {
            const int base = (b / ST) * (ST * R) + (b % ST);
            const int pbase = lds_phys<S, P>(base);
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                constexpr int off = lds_step_const<S>(k, ST, ST * R);
                int o;
                if constexpr (off >= 0)
                    o = pbase + off;
                else
                    o = lds_phys<S, P>(base + k * ST);
                if constexpr (PART == 0)
                    ldsf[o] = v[m * R + k];
                else if constexpr (PART == 1)
                    ldsf[o] = v[m * R + k].re;
                else
                    ldsf[o] = v[m * R + k].im;
            });
        }
    });
}
template <class T, class S, int P, int PART, class E> MI_HD void lds_gather(cx<T>* v, int u, const E* ldsf) {
    constexpr int R = S::R[P], NB = S::nb(P), BPT = S::bpt(P);
    static_for<0, BPT>([&](auto M_) {
        constexpr int m = M_;
        const int b = u + m * S::TPF;
        if ((m + 1) * S::TPF <= NB || b < NB) {
            const int pbase = lds_phys<S, P - 1>(b);
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                constexpr int off = lds_step_const<S>(k, NB, 0);
                int o;
                if constexpr (off >= 0)
                    o = pbase + off;
                else
                    o = lds_phys<S, P - 1>(b + k * NB);
                if constexpr (PART == 0)
                    v[m * R + k] = ldsf[o];
                else if constexpr (PART == 1)
                    v[m * R + k].re = ldsf[o];
                else
                    v[m * R + k].im = ldsf[o];
            });
        }
    });
}

// ---- pair-fused first two sub-passes ------------------------------------------------------------------------------------
// Schedules that open with two radix-8 sub-passes and hold 32 values per thread (N = 32 TPF; the 512- and 1024-row column
// tiles): sub-pass 1's butterfly b' reads y[b' + k' N/8], i.e. output (b' mod 8) of the sub-pass-0 butterflies
// b'/8 + k' TPF/2 -- which the two threads u = g and u = g + TPF/2 hold between them (g = u mod TPF/2; thread h = u div
// TPF/2 has k' = 2m + h in its butterfly m).  With the paired lane map those two threads are lanes l and l + 32 of one
// wave: sixteen v_permlane32_swap per plane replace the whole first LDS exchange (32 ds_write + 32 ds_read per plane and
// two barriers; four barriers with the split exchange).  After the swap BOTH threads find input k' of their butterfly j
// (b' = 8 g + 4 h + j, j < 4) in register slot (k'/2) 8 + j + 4 (k' & 1), so the code is lane-uniform; the outputs go back to the
// same slots and from there to LDS in the ordinary exchange-1 layout, which sub-pass 2 gathers as usual.
template <class S> constexpr bool pair_fusable() {
    return S::NP >= 3 && S::R[0] == 8 && S::R[1] == 8 && S::bpt(0) == 4 && S::bpt(1) == 4 && S::N == 32 * S::TPF && S::all_pow2() && S::emax() > 16;
}
template <class T, class S> MI_HD void pair_subpass1(cx<T>* v, int u, const cx<T>* MI_RESTRICT tw) {
    constexpr int H = S::TPF / 2;
    const int h = u / H;
    const cx<T>* t = tw + S::tw_offset(1) + 4 * h;  // w_64^{(4h + j) k'} at [(k' - 1) 8 + 4h + j]
    static_for<0, 4>([&](auto J_) {
        constexpr int j = J_;
        cx<T> x[8];
        static_for<0, 8>([&](auto K_) {
            constexpr int k = K_;
            x[k] = v[(k / 2) * 8 + j + 4 * (k & 1)];
        });
        reg_wall8(x);  // butterfly j starts below this point ...
        static_for<1, 8>([&](auto K_) {
            constexpr int k = K_;
            x[k] = x[k] * t[(k - 1) * 8 + j];
        });
        butterfly<8>(x);
        reg_wall8(x);  // ... and is complete above this one: neighbouring butterflies cannot be fused into packed operations
        static_for<0, 8>([&](auto Q_) {
            constexpr int q = Q_;
            v[(q / 2) * 8 + j + 4 * (q & 1)] = x[q];
        });
        MI_SCHED_FENCE();
    });
}
// exchange 1 (written by sub-pass 1) from the pair-fused register layout: output q of butterfly b' = 8 g + 4 h + j lands at
// y[g 64 + (4 h + j) + 8 q]; linear layout (one padding slot per 32 elements): 66 g + 4 h + [j + 8 q + (q >= 4)]
template <class T, class S, int PART, class E> MI_HD void pair_scatter(const cx<T>* v, int u, E* ldsf) {
    constexpr int H = S::TPF / 2;
    const int g = u % H, h = u / H, pbase = 66 * g + 4 * h;
    static_for<0, 4>([&](auto J_) {
        constexpr int j = J_;
        static_for<0, 8>([&](auto Q_) {
            constexpr int q = Q_, slot = (q / 2) * 8 + j + 4 * (q & 1), off = j + 8 * q + (q >= 4 ? 1 : 0);
            if constexpr (PART == 0)
                ldsf[pbase + off] = v[slot];
            else if constexpr (PART == 1)
                ldsf[pbase + off] = v[slot].re;
            else
                ldsf[pbase + off] = v[slot].im;
        });
    });
}

// Destination adaptor: fn(f, i, value, integral_constant<slot>, v) also sees the compile-time register slot of the
// value and the thread's register array (for tables a kernel keeps next to the data registers).
template <class L> struct SlotDst {
    L fn;
};
template <class L> MI_HD SlotDst<L> slot_dst(L l) { return SlotDst<L>{l}; }
template <class D> struct is_slot_dst : std::false_type {};
template <class L> struct is_slot_dst<SlotDst<L>> : std::true_type {};
// The same two destinations with a PREFETCH hook: pre(f, u, v) runs in front of the last sub-pass's arithmetic, so table entries the
// destination (or the transform that takes the registers over) multiplies with are in flight while the last butterflies compute
// (round 5: the Bluestein bodies' spectrum multiplier and output chirp, which were fetched right when they were needed)
template <class P> struct KeepInRegsPre {
    P pre;
};
template <class L, class P> struct SlotDstPre {
    L fn;
    P pre;
};
template <class L, class P> struct is_slot_dst<SlotDstPre<L, P>> : std::true_type {};
template <class D> struct is_keep_in_regs : std::is_same<D, KeepInRegs> {};
template <class P> struct is_keep_in_regs<KeepInRegsPre<P>> : std::true_type {};
template <class D> struct has_pre : std::false_type {};
template <class P> struct has_pre<KeepInRegsPre<P>> : std::true_type {};
template <class L, class P> struct has_pre<SlotDstPre<L, P>> : std::true_type {};

template <class S, int P, Map MIN, Map MOUT> constexpr Map pass_map() {
    return P == 0 ? MIN : (P == S::NP - 1 ? MOUT : MAP_EF);
}

// register array length an executor must provide per thread
template <class S, bool SPLIT> constexpr int regs_needed() { return S::emax(); }
// LDS bytes one workgroup needs
template <class T, class S, int F, bool SPLIT, int PM = 1> constexpr size_t lds_bytes() {
    return (S::NP > 1) ? (size_t)F * S::template pitch_for<PM>() * (SPLIT ? sizeof(T) : sizeof(cx<T>)) : 0;
}

// ---- sub-pass twiddle tables staged in LDS (TWL) ----------------------------------------------------------------------
// TWL is a bit mask of sub-passes (bit p = sub-pass p >= 1; the set bits must be contiguous) whose twiddle tables the
// workgroup copies from global memory into LDS, behind its exchange buffer, while its first loads are in flight: the table
// loads are issued in front of the first row loads and written to LDS behind them (loads return in order, so the writes
// wait for the table entries only), and the barriers of the first exchange publish the copy before sub-pass 1 reads it.
// Why: a sub-pass fetches its factors right after a barrier.  As global loads they queue in the CU's vector-memory pipeline
// behind the co-resident workgroups' row bursts (128 KiB each for the column tiles), and their latency -- microseconds under
// load -- is exposed once per sub-pass; ds_read_b64 from LDS comes back in ~100 cycles whatever HBM is doing
// (measured on the 1024 x 16 column tiles of 2^20: first pass 5.41 -> 5.60 TB/s, later pass 5.10 -> 5.52, results bit-identical).
template <class S> constexpr int tw_pass_entries(int p) { return (p >= 1 && p < S::NP) ? (S::R[p] - 1) * S::stride(p) : 0; }
template <class S> constexpr int twl_first(int mask) {
    for (int p = 1; p < S::NP; ++p)
        if ((mask >> p) & 1) return p;
    return S::NP;
}
template <class S> constexpr int twl_total(int mask) {
    int o = 0;
    for (int p = 1; p < S::NP; ++p)
        if ((mask >> p) & 1) o += tw_pass_entries<S>(p);
    return o;
}
template <class S> constexpr bool twl_valid(int mask) {  // contiguous run of sub-passes >= 1
    if (mask == 0) return true;
    if (mask & 1) return false;
    int m = mask >> twl_first<S>(mask);
    return (m & (m + 1)) == 0 && (mask >> S::NP) == 0;
}
constexpr int twl_all(int np) { return ((1 << np) - 1) & ~1; }
constexpr size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }
// LDS bytes one workgroup needs, staged tables included
template <class T, class S, int F, bool SPLIT, int PM = 1, int TWL = 0> constexpr size_t lds_bytes_twl() {
    return TWL ? align16(lds_bytes<T, S, F, SPLIT, PM>()) + (size_t)twl_total<S>(TWL) * sizeof(cx<T>) : lds_bytes<T, S, F, SPLIT, PM>();
}

// ---- the workgroup transform -------------------------------------------------------------------------
// X: executor. X::for_threads(fn(tid, cx<T>* v)) runs fn for every thread of the workgroup with that
//    thread's private register array; X::barrier() is the workgroup barrier.
// src(f, i) -> cx<T>: input element i of sequence f;   dst(f, i, value): output element i.
// ABL (compile-time, tuning builds only): bit 2 skips the arithmetic, bit 3 skips the LDS exchange — ablation probes
// that keep the HBM access pattern; production instantiations use ABL = 0.
template <class T, class S, int F, Map MIN, Map MOUT, bool SPLIT, int PM, int ABL, int P, int TWREG = -1, bool TWSTAGE = false, int TWL = 0, int TWLOFF = 0, class X, class SRC, class DST>
MI_HD void wg_fft_stage(X& ex, void* lds_raw, const cx<T>* MI_RESTRICT tw, SRC& src, DST& dst) {
    constexpr int R = S::R[P], NB = S::nb(P), ST = S::stride(P), BPT = S::bpt(P);
    constexpr Map MP = pass_map<S, P, MIN, MOUT>();
    constexpr bool LAST = (P == S::NP - 1);
    // this sub-pass's factors: the global table, or its LDS copy (compute_pass adds S::tw_offset(P) to whatever it is given)
    const cx<T>* twp = tw;
    if constexpr (((TWL >> P) & 1) != 0)
        twp = (const cx<T>*)((char*)lds_raw + align16(lds_bytes<T, S, F, SPLIT, PM>()) + TWLOFF) - S::tw_offset(twl_first<S>(TWL));
    // arithmetic of pass P, then either the final store or the scatter half of the exchange
    ex.for_threads([&](int tid, cx<T>* v) {
        int f, u;
        map_tid<MP, F, S::TPF>(tid, f, u);
        if constexpr (LAST && has_pre<DST>::value) dst.pre(f, u, v);
        // (a sub-pass whose table is staged in LDS reads it there; the others take their factors from the registers when TWREG >= 0)
        if constexpr (!(ABL & 4)) compute_pass<T, S, P, (((TWL >> P) & 1) != 0 ? -1 : TWREG)>(v, u, twp);
        if constexpr (TWSTAGE && TWREG >= 0 && !LAST && ((TWL >> (P + 1)) & 1) == 0) {
            // the NEXT sub-pass's factors, fetched a whole exchange (scatter, barrier, gather, barrier) ahead of their use: the table look-up's
            // latency -- L2 under load: about a microsecond -- is off the row's critical path (round 5: the Bluestein bodies, whose tables are too
            // large for LDS or a rows loop; the registers are live from here to the next compute_pass only)
            int f2, u2;
            map_tid<pass_map<S, P + 1, MIN, MOUT>(), F, S::TPF>(tid, f2, u2);
            preload_twiddles_pass<T, S, P + 1, TWREG>(v, u2, tw);
        }
        if constexpr (LAST && is_keep_in_regs<DST>::value) {
        } else if constexpr (LAST) {
            static_for<0, BPT>([&](auto M_) {
                constexpr int m = M_;
                const int b = u + m * S::TPF;
                if ((m + 1) * S::TPF <= NB || b < NB) {
                    const int base = (b / ST) * (ST * R) + (b % ST);
                    static_for<0, R>([&](auto K_) {
                        constexpr int k = K_;
                        // destinations that keep per-output tables in the register array also get the
                        // compile-time slot of the value and the array itself
                        if constexpr (is_slot_dst<DST>::value)
                            dst.fn(f, base + k * ST, v[m * R + k], std::integral_constant<int, m * R + k>{}, v);
                        else
                            dst(f, base + k * ST, v[m * R + k]);
                    });
                }
            });
        } else if constexpr ((ABL & 8) != 0) {
        } else if constexpr (!SPLIT) {
            lds_scatter<T, S, P, 0>(v, u, (cx<T>*)lds_raw + f * S::template pitch_for<PM>());
        } else {
            lds_scatter<T, S, P, 1>(v, u, (T*)lds_raw + f * S::template pitch_for<PM>());
        }
    });
    if constexpr (!LAST && (ABL & 8) != 0) {
        wg_fft_stage<T, S, F, MIN, MOUT, SPLIT, PM, ABL, P + 1, TWREG, TWSTAGE, TWL, TWLOFF>(ex, lds_raw, tw, src, dst);
    } else if constexpr (!LAST) {
        constexpr Map MQ = pass_map<S, P + 1, MIN, MOUT>();
        ex.barrier();
        if constexpr (!SPLIT) {
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, P + 1, 0>(v, u, (const cx<T>*)lds_raw + f * S::template pitch_for<PM>());
            });
            ex.barrier();
        } else {
            // real plane first, then the imaginary plane, through the same half-size buffer.  The two planes
            // live in different registers, so the new real parts can land in v[].re while v[].im still holds the
            // old layout's imaginary parts: no side array is needed.
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, P + 1, 1>(v, u, (const T*)lds_raw + f * S::template pitch_for<PM>());
            });
            ex.barrier();
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MP, F, S::TPF>(tid, f, u);
                lds_scatter<T, S, P, 2>(v, u, (T*)lds_raw + f * S::template pitch_for<PM>());
            });
            ex.barrier();
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, P + 1, 2>(v, u, (const T*)lds_raw + f * S::template pitch_for<PM>());
            });
            ex.barrier();
        }
        wg_fft_stage<T, S, F, MIN, MOUT, SPLIT, PM, ABL, P + 1, TWREG, TWSTAGE, TWL, TWLOFF>(ex, lds_raw, tw, src, dst);
    }
}

// Source adaptor: element-wise functor -> the per-butterfly interface the first sub-pass uses.
// A source may instead implement bfly<R>() itself to share work across the R inputs of a butterfly
// (the inter-pass twiddle recurrence of the large-N passes does).
template <class L> struct ElemSrc {
    L load;
    template <int R, class T> MI_HD void bfly(int f, int b, int nb, cx<T>* v) const {
        static_for<0, R>([&](auto K_) {
            constexpr int k = K_;
            v[k] = load(f, b + k * nb);
        });
    }
};
template <class L> MI_HD ElemSrc<L> elem_src(L l) { return ElemSrc<L>{l}; }

// A source may go one step further and take over the whole first load of a thread -- every butterfly of sub-pass 0 at
// once -- by declaring `static constexpr bool kLoadsAll = true` and implementing load_all<S>(f, u, v): the column-tile
// passes share the inter-pass twiddle powers across a thread's butterflies that way.
template <class SRC, class = void> struct src_loads_all : std::false_type {};
template <class SRC> struct src_loads_all<SRC, std::enable_if_t<SRC::kLoadsAll>> : std::true_type {};

// SRC_IN_LDS: `src` reads the same LDS buffer the exchanges use (Rader/Bluestein second transform), so a
// barrier separates the loads from the first scatter.
// TWLOFF: byte offset of this transform's staged tables behind the exchange buffer (bodies that run two transforms through one
// buffer -- Bluestein -- give each its own region, so the second transform's staging cannot overwrite a table a slow wave of the
// first is still reading)
template <class T, class S, int F, Map MIN, Map MOUT, bool SPLIT, bool SRC_IN_LDS = false, int PM = 1, int ABL = 0, int TWREG = -1, bool TWSTAGE = false, int TWL = 0, int TWLOFF = 0, class X, class SRC, class DST>
MI_HD void wg_fft(X& ex, void* lds_raw, const cx<T>* MI_RESTRICT tw, SRC src, DST dst) {
    static_assert(S::valid(), "radices must multiply to N");
    static_assert(MIN != MAP_FFP || (pair_fusable<S>() && !SRC_IN_LDS && TWREG < 0 && sizeof(T) == 4), "the paired map is the pair-fused path");
    static_assert(TWL == 0 || (twl_valid<S>(TWL) && !SRC_IN_LDS && (TWREG < 0 || TWSTAGE) && MIN != MAP_FFP && S::NP >= 2), "staged tables: plain transforms whose first exchange publishes the copy");
    constexpr int R0 = S::R[0], NB0 = S::nb(0), BPT0 = S::bpt(0);
    constexpr int TWN = twl_total<S>(TWL), NT = F * S::TPF, TWPT = (TWN + NT - 1) / NT, TWSRC = S::tw_offset(twl_first<S>(TWL));
    cx<T>* twl = (cx<T>*)((char*)lds_raw + align16(lds_bytes<T, S, F, SPLIT, PM>()) + TWLOFF);
    // inputs of sub-pass 0 straight from the source
    ex.for_threads([&](int tid, cx<T>* v) {
        int f, u;
        map_tid<MIN, F, S::TPF>(tid, f, u);
        cx<T> twr[TWPT > 0 ? TWPT : 1];
        if constexpr (TWN > 0) {
            static_for<0, TWPT>([&](auto I_) {
                constexpr int i = I_;
                const int e = tid + i * NT;
                twr[i] = tw[TWSRC + (((i + 1) * NT <= TWN || e < TWN) ? e : 0)];
            });
        }
        if constexpr (src_loads_all<SRC>::value) {
            src.template load_all<S>(f, u, v);
        } else {
            static_for<0, BPT0>([&](auto M_) {
                constexpr int m = M_;
                const int b = u + m * S::TPF;
                if ((m + 1) * S::TPF <= NB0 || b < NB0) {
                    src.template bfly<R0>(f, b, NB0, v + m * R0);
                }
                if constexpr (BPT0 > 1 && R0 * BPT0 > 16) MI_SCHED_FENCE();
            });
        }
        if constexpr (TWN > 0) {
            static_for<0, TWPT>([&](auto I_) {
                constexpr int i = I_;
                const int e = tid + i * NT;
                if ((i + 1) * NT <= TWN || e < TWN) twl[e] = twr[i];
            });
        }
        if constexpr (TWSTAGE && TWREG >= 0 && S::NP > 1 && MIN != MAP_FFP && ((TWL >> 1) & 1) == 0) {  // sub-pass 1's factors ride behind the row loads
            int f1, u1;
            map_tid<pass_map<S, 1, MIN, MOUT>(), F, S::TPF>(tid, f1, u1);
            preload_twiddles_pass<T, S, 1, TWREG>(v, u1, tw);
        }
    });
    if constexpr (SRC_IN_LDS) ex.barrier();
    if constexpr (MIN == MAP_FFP) {
        constexpr Map MQ = pass_map<S, 2, MIN, MOUT>();
        constexpr int PITCH = S::template pitch_for<PM>();
        ex.for_threads([&](int tid, cx<T>* v) {
            int f, u;
            map_tid<MAP_FFP, F, S::TPF>(tid, f, u);
            compute_pass<T, S, 0>(v, u, tw);
        });
        ex.pair_swap();  // v[m 8 + k] of the upper half-wave <-> v[m 8 + k + 4] of the lower one, m, k < 4
        ex.for_threads([&](int tid, cx<T>* v) {
            int f, u;
            map_tid<MAP_FFP, F, S::TPF>(tid, f, u);
            pair_subpass1<T, S>(v, u, tw);
            if constexpr (!SPLIT)
                pair_scatter<T, S, 0>(v, u, (cx<T>*)lds_raw + f * PITCH);
            else
                pair_scatter<T, S, 1>(v, u, (T*)lds_raw + f * PITCH);
        });
        ex.barrier();
        if constexpr (!SPLIT) {
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, 2, 0>(v, u, (const cx<T>*)lds_raw + f * PITCH);
            });
            ex.barrier();
        } else {
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, 2, 1>(v, u, (const T*)lds_raw + f * PITCH);
            });
            ex.barrier();
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MAP_FFP, F, S::TPF>(tid, f, u);
                pair_scatter<T, S, 2>(v, u, (T*)lds_raw + f * PITCH);
            });
            ex.barrier();
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MQ, F, S::TPF>(tid, f, u);
                lds_gather<T, S, 2, 2>(v, u, (const T*)lds_raw + f * PITCH);
            });
            ex.barrier();
        }
        wg_fft_stage<T, S, F, MIN, MOUT, SPLIT, PM, ABL, 2, TWREG, TWSTAGE, TWL, TWLOFF>(ex, lds_raw, tw, src, dst);
    } else {
        wg_fft_stage<T, S, F, MIN, MOUT, SPLIT, PM, ABL, 0, TWREG, TWSTAGE, TWL, TWLOFF>(ex, lds_raw, tw, src, dst);
    }
}

}  // namespace mi355
