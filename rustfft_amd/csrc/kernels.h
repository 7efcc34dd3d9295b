// Kernel bodies (executor-generic) for the MI355X FFT passes.
//
//   k1_body : batched contiguous transforms, one workgroup = F sequences of length N <= LDS capacity.
//             Replaces Fft::process_with_scratch's chunk loop + the whole Radix4/RadixN pipeline
//             (src/fft_helper.rs:9-28, src/algorithm/radix4.rs:167-203, radixn.rs:250-333) for one chunk.
//   k2_body : one pass of the large-N decomposition N = R_1 R_2 .. R_P ("macro" Stockham pass, the GPU
//             form of the reference's six-step MixedRadix, src/algorithm/mixed_radix.rs:128-158):
//             a workgroup owns a tile of F adjacent macro-butterflies B = B0..B0+F-1 and does, for each,
//                 in_j  = X[B + j M] * w_{S R}^{(B mod S) j}     (M = N / R, S = R_1..R_{p-1})
//                 out_k -> Y[(B div S) S R + (B mod S) + k S]
//             reading F-element row segments (F*sizeof(C) contiguous bytes per row) and writing either a
//             fully contiguous F*R block (first pass, S = 1) or F-element segments (later passes).
//             The inter-pass twiddle comes from a two-level table w^e = lo[e & mask] * hi[e >> h], both
//             tables generated on the host the way src/twiddles.rs:6-23 does (f64 angle, round to T).
// The inverse transform is conj(FFT(conj(x))): `sgn_in` / `sgn_out` (+1 or -1) multiply the imaginary
// part on the way in / out, so one set of forward tables and butterflies serves both directions.
#pragma once
#include "engine.h"
#include "kernels_params.h"

namespace mi355 {

// ABL bits 7, 10, 11 (128 / 1024 / 2048; production options): sub-pass twiddle tables staged in LDS (engine.h TWL) -- all of
// them / sub-pass 1 only / the last sub-pass only
template <int ABL, int NP> constexpr int k1_twl() {
    return NP < 2 ? 0 : (ABL & 128) ? twl_all(NP) : (ABL & 1024) ? 2 : (ABL & 2048) ? (1 << (NP - 1)) : 0;
}
template <class T, class S, int F, bool SPLIT, int ABL = 0, class X>
MI_HD void k1_body(X& ex, const K1Params<T>& p, long long block, void* lds) {
    const long long fft0 = block * F;
    // workgroup-uniform base + 32-bit element offsets (F * N < 2^31): one address VGPR per access
    const cx<T>* in = p.in + fft0 * S::N;
    cx<T>* out = p.out + fft0 * S::N;
    const int rows = (int)((p.batch - fft0) < F ? (p.batch - fft0) : F);
    const T sgn = p.sgn;
    auto src = [=](int f, int i) -> cx<T> {
        if (f < rows) {
            cx<T> x;
            if constexpr ((ABL & 16) != 0)
                x = ld_nt(in + (unsigned)(f * S::N + i));
            else
                x = in[(unsigned)(f * S::N + i)];
            x.im *= sgn;
            return x;
        }
        return cx<T>{0, 0};
    };
    auto dst = [=](int f, int i, cx<T> x) {
        if (f < rows) {
            x.im *= sgn;
            if constexpr ((ABL & 32) != 0)
                st_nt(out + (unsigned)(f * S::N + i), x);
            else
                out[(unsigned)(f * S::N + i)] = x;
        }
    };
    // ABL bit 8 (256; tuning so far, round 5): the factors of the sub-passes whose tables are not staged in LDS are fetched one exchange ahead of
    // their use (engine.h TWSTAGE), as the Bluestein bodies do
    constexpr bool PFK = (ABL & 256) != 0;
    wg_fft<T, S, F, MAP_EF, MAP_EF, SPLIT, false, 1, (ABL & 15), (PFK ? S::emax() : -1), PFK, k1_twl<ABL, S::NP>()>(ex, lds, p.tw, elem_src(src), dst);
}
template <class S, bool SPLIT, int ABL> constexpr int k1_regs() { return regs_needed<S, SPLIT>() + ((ABL & 256) ? twreg_count<S>() : 0); }
template <class T, class S, int F, bool SPLIT, int ABL> constexpr size_t k1_lds_bytes() {
    return lds_bytes_twl<T, S, F, SPLIT, 1, k1_twl<ABL, S::NP>()>();
}

// ---- two-kernel Bluestein for lengths whose padded size M still fits ONE workgroup through the split exchange ------
// (4096 < n, 2n - 1 <= M <= 32768; bluesteins_algorithm.rs:100-136).  STAGE 1: rows of the caller (pitch n) times the
// chirp, zero-padded on the fly -> FFT_M -> conj(X * bf) into the workspace (pitch M).  STAGE 2: workspace -> FFT_M ->
// conj(X) * chirp, truncated to n, into the caller's rows.  Traffic 2 M + 2 n elements per transform.
template <class T, class S, int F, bool SPLIT, int STAGE, class X>
MI_HD void k1bs_body(X& ex, const BluesteinParams<T>& p, long long block, void* lds) {
    constexpr unsigned M = S::N;
    const long long fft0 = block * F;
    const unsigned n = (unsigned)p.n;
    const cx<T>* in = p.in + fft0 * (STAGE == 1 ? (long long)n : (long long)M);
    cx<T>* out = p.out + fft0 * (STAGE == 1 ? (long long)M : (long long)n);
    const cx<T>* MI_RESTRICT chirp = p.chirp;
    const cx<T>* MI_RESTRICT bf = p.bf;
    const int rows = (int)((p.batch - fft0) < F ? (p.batch - fft0) : F);
    const T sgn = p.sgn;
    auto src = [=](int f, int i) -> cx<T> {
        if constexpr (STAGE == 1) {
            if (f < rows && (unsigned)i < n) {
                cx<T> x = in[(unsigned)f * n + (unsigned)i];
                x.im *= sgn;
                return x * chirp[(unsigned)i];
            }
            return cx<T>{0, 0};
        } else {
            return f < rows ? in[(unsigned)f * M + (unsigned)i] : cx<T>{0, 0};
        }
    };
    auto dst = [=](int f, int j, cx<T> v) {
        if (f < rows) {
            if constexpr (STAGE == 1) {
                out[(unsigned)f * M + (unsigned)j] = cconj(v * bf[(unsigned)j]);
            } else if ((unsigned)j < n) {
                cx<T> y = cconj(v) * chirp[(unsigned)j];
                y.im *= sgn;
                out[(unsigned)f * n + (unsigned)j] = y;
            }
        }
    };
    wg_fft<T, S, F, MAP_EF, MAP_EF, SPLIT>(ex, lds, p.tw, elem_src(src), dst);
}

// LDS pitch residue for a tile of F columns: lanes walk across the columns first, so the (32 / F) row slots of one
// 32-lane group must fall on the banks the columns leave free
constexpr int k2_pitch_mod(int f) { return f < 32 ? 32 / f : 1; }

// Source of a large-N pass: F-element row segments, times the inter-pass twiddle w_Q^{c j} (Q = S R, c = B mod S,
// j = row).  A thread (column c, slot u) holds the rows j = u + m TPF + k NB (butterfly m, input k), so its factors are
//     w^{c j} = [w^{c u} (w^{c TPF})^m] * (w^{c NB})^k = B_m P^k :
// three two-level table look-ups per THREAD (B_0, the butterfly step Q = w^{c TPF} and P), the powers P^2, P^3, P^4
// shared by all butterflies of the thread, and per butterfly groups of four inputs: T, T P, T P^2, T P^3 with the group
// base T advancing by P^4.  2R + 1 complex multiplications per butterfly, a dependency chain R/4 + BPT long (rounding
// error ~ sqrt of that times eps), six table gathers per thread.  (History: 2 R divergent gathers per butterfly ran the
// later passes at 3.3 TB/s; base + step per butterfly with a depth-first power walk -- 16 gathers per thread -- at
// 4.4 - 4.8; divergent 8-byte gathers cost the L1 one cycle per lane, the row segments one per 16 lanes.)
// ABL bits 7, 10, 11 (128 / 1024 / 2048; production options): the sub-pass twiddle tables of the tile transform are staged in
// LDS (engine.h TWL) -- all of them / sub-pass 1 only / the last sub-pass only (for tiles whose LDS budget holds one table).
template <int ABL, int NP> constexpr int k2_twl() {
    return (ABL & 128) ? twl_all(NP) : (ABL & 1024) ? 2 : (ABL & 2048) ? (1 << (NP - 1)) : 0;
}
template <class T, bool FIRST, int ABL = 0, int F_ = 0, int RING = 0> struct K2Src {
    static constexpr bool kLoadsAll = true;
    const cx<T>* in;  // no restrict: the in-place last pass reads and writes the caller's buffer
    unsigned M, b0, bmod0;
    T sgn_in;
    const cx<T>* MI_RESTRICT tlo;
    const cx<T>* MI_RESTRICT thi;
    int hshift, lmask;
    MI_HD cx<T> lut(unsigned e) const { return tlo[e & (unsigned)lmask] * thi[e >> hshift]; }
    template <class S> MI_HD void load_all(int f, int u, cx<T>* v) const {
        constexpr int R = S::R[0], NB = S::nb(0), BPT = S::bpt(0), TPF = S::TPF;
        static_assert(NB % TPF == 0 && NB == BPT * TPF, "column-tile schedules fill every thread");
        // 32-bit element offsets from the (workgroup-uniform) transform base: one VGPR per address instead of two
        const unsigned col = b0 + (unsigned)f;
        if constexpr ((ABL & 4096) == 0) {  // (two columns per lane: k2_body has loaded the rows of both columns already)
            static_for<0, BPT>([&](auto M_) {
                constexpr int m = M_;
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    const unsigned row = (unsigned)(u + m * TPF + k * NB);
                    if constexpr ((RING & 2) != 0)
                        v[m * R + k] = ld_agent(in + (col + row * M));  // fused kernel: the ring another workgroup wrote in this launch
                    else if constexpr ((ABL & 16) != 0)
                        v[m * R + k] = ld_nt(in + (col + row * M));
                    else
                        v[m * R + k] = in[col + row * M];
                });
            });
        }
        if constexpr (FIRST || (ABL & 1)) {
            static_for<0, BPT * R>([&](auto I_) { v[decltype(I_)::value].im *= sgn_in; });
        } else {
            const unsigned c = bmod0 + (unsigned)f;
            const cx<T> s1 = lut(c * (unsigned)NB);
            cx<T> tb = lut(c * (unsigned)u), q = tb;
            if constexpr (BPT > 1) q = lut(c * (unsigned)TPF);
            const cx<T> s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2;
            static_for<0, BPT>([&](auto M_) {
                constexpr int m = M_;
                cx<T>* x = v + m * R;
                cx<T> t = tb;
                static_for<0, R / 4>([&](auto H_) {
                    constexpr int h = H_;
                    static_for<0, 4>([&](auto L_) { x[4 * h + decltype(L_)::value].im *= sgn_in; });
                    x[4 * h] = x[4 * h] * t;
                    x[4 * h + 1] = x[4 * h + 1] * (t * s1);
                    x[4 * h + 2] = x[4 * h + 2] * (t * s2);
                    x[4 * h + 3] = x[4 * h + 3] * (t * s3);
                    if constexpr (h + 1 < R / 4) t = t * s4;
                });
                if constexpr (m + 1 < BPT) tb = tb * q;
                if constexpr (BPT > 1 && R * BPT > 16) MI_SCHED_FENCE();
            });
        }
    }
};

// The tile of a power-of-two column-tile pass: transform g's rows at `in` / `out`, columns [tile F, (tile + 1) F).  k2_body maps a
// workgroup index to (g, tile) for the one-pass-per-launch kernels, k2f_body (below) for the fused two-pass kernel.
// RING (fused kernel): bit 0 = `out` is the ring: agent-scope stores; bit 1 = `in` is the ring: agent-scope loads (cx.h)
// `tile` places the columns the tile READS, `tile_out` the ones it WRITES (and its inter-pass factors): equal except for the units of a
// fused three-pass launch (kernels_params.h), whose ring holds a compact copy of a unit's columns.
template <class T, class S, int F, bool FIRST, bool SPLIT, int ABL = 0, int RING = 0, class X>
MI_HD void k2_tile(X& ex, const K2Params<T>& p, const cx<T>* in, cx<T>* out, unsigned tile, unsigned tile_out, void* lds) {
    constexpr int R = S::N;
    const unsigned b0 = tile * (unsigned)F, bo = tile_out * (unsigned)F;
    const unsigned M = (unsigned)p.m, s32 = (unsigned)p.s;
    const T sgn_out = p.sgn_out;
    // a tile never straddles a multiple of S (F | S whenever S > 1), so B div S is tile-uniform
    const unsigned bdiv = FIRST ? 0u : (bo >> p.s_shift);
    const unsigned bmod0 = FIRST ? 0u : (bo & (s32 - 1u));
    const unsigned obase = bdiv * s32 * (unsigned)R + bmod0;
    K2Src<T, FIRST, ABL, F, RING> src{in, M, b0, bmod0, p.sgn_in, p.tlo, p.thi, p.hshift, p.lmask};
    auto dst = [=](int f, int k, cx<T> x) {
        x.im *= sgn_out;
        cx<T>* o = FIRST ? out + ((bo + (unsigned)f) * (unsigned)R + (unsigned)k) : out + (obase + (unsigned)f + (unsigned)k * s32);
        if constexpr ((RING & 1) != 0)
            st_agent(o, x);
        else if constexpr ((ABL & 32) != 0)
            st_nt(o, x);
        else
            *o = x;
    };
    // first pass: lanes walk across the tile's columns on the way in and along each sequence on the way
    // out (the F*R output block is contiguous); later passes: across columns both ways
    if constexpr ((ABL & 4096) != 0) {
        // Two columns per lane (tuning; launch.h DevExecPair): virtual threads 2 t and 2 t + 1 are the columns f, f + 1 of the same
        // rows and run on one physical thread, whose register array holds the even column's values at v[i] and the odd column's at
        // v[NREG + i].  The even one fetches both columns' rows as 16-byte accesses in a phase of its own (no barrier: its partner
        // is the same thread; the emulator runs phases one after the other), the engine then applies the inter-pass factors
        // (K2Src skips its loads), and the later passes store both columns' outputs as 16 bytes the same way.
        constexpr int NREG = regs_needed<S, SPLIT>(), R0 = S::R[0], NB0 = S::nb(0), BPT0 = S::bpt(0), TPF = S::TPF;
        static_assert(F % 2 == 0 && sizeof(T) == 4, "pairs of adjacent columns; Complex<f32> only: a pair is one 16-byte access, the alignment the API guarantees");
        struct alignas(2 * sizeof(cx<T>)) cx2 {
            cx<T> a, b;
        };
        ex.for_threads([&](int tid, cx<T>* v) {
            int f, u;
            map_tid<MAP_FF, F, TPF>(tid, f, u);
            if ((f & 1) == 0) {
                const unsigned col = b0 + (unsigned)f;
                static_for<0, BPT0>([&](auto M_) {
                    constexpr int m = M_;
                    static_for<0, R0>([&](auto K_) {
                        constexpr int k = K_;
                        const unsigned row = (unsigned)(u + m * TPF + k * NB0);
                        cx2 q;
                        if constexpr ((ABL & 16) != 0)
                            q = ld_nt2<T, cx2>(in + (col + row * M));
                        else
                            q = *(const cx2*)(in + (col + row * M));
                        v[m * R0 + k] = q.a;
                        v[NREG + m * R0 + k] = q.b;
                    });
                });
            }
        });
        if constexpr (FIRST) {
            wg_fft<T, S, F, MAP_FF, MAP_EF, SPLIT, false, k2_pitch_mod(F), (ABL & 15), -1, false, k2_twl<ABL, S::NP>()>(ex, lds, p.tw, src, dst);
        } else {
            wg_fft<T, S, F, MAP_FF, MAP_FF, SPLIT, false, k2_pitch_mod(F), (ABL & 15), -1, false, k2_twl<ABL, S::NP>()>(ex, lds, p.tw, src, KeepInRegs{});
            constexpr int LP = S::NP - 1, RL = S::R[LP], NBL = S::nb(LP), STL = S::stride(LP), BPTL = S::bpt(LP);
            ex.for_threads([&](int tid, cx<T>* v) {
                int f, u;
                map_tid<MAP_FF, F, TPF>(tid, f, u);
                if ((f & 1) == 0) {
                    static_for<0, BPTL>([&](auto M_) {
                        constexpr int m = M_;
                        const int b = u + m * TPF;
                        const int base = (b / STL) * (STL * RL) + (b % STL);
                        static_for<0, RL>([&](auto K_) {
                            constexpr int k = K_;
                            cx2 q{v[m * RL + k], v[NREG + m * RL + k]};
                            q.a.im *= sgn_out;
                            q.b.im *= sgn_out;
                            if constexpr ((ABL & 32) != 0)
                                st_nt2<T, cx2>(out + (obase + (unsigned)f + (unsigned)(base + k * STL) * s32), q);
                            else
                                *(cx2*)(out + (obase + (unsigned)f + (unsigned)(base + k * STL) * s32)) = q;
                        });
                    });
                }
            });
        }
    } else {
        wg_fft<T, S, F, (ABL & 64) ? MAP_FFP : MAP_FF, FIRST ? MAP_EF : MAP_FF, SPLIT, false, k2_pitch_mod(F), (ABL & 15), -1, false, k2_twl<ABL, S::NP>()>(ex, lds, p.tw, src, dst);
    }
}
template <class T, class S, int F, bool FIRST, bool SPLIT, int ABL = 0, class X>
MI_HD void k2_body(X& ex, const K2Params<T>& p, long long block, void* lds) {
    // XCD-aware tile order.  Workgroup b is dispatched to XCD b % 8 (MI355X_MICROARCH.md, observed, used for speed
    // only).  Within every aligned group of 8 << xq consecutive workgroups (all tiles of one transform), XCD x takes the
    // tiles whose index has x in bits [xp, xp + 3): each XCD then streams runs of 2^xp ADJACENT tiles, and its requests
    // spread over the address bits that select L2 / memory channels instead of sharing them.  Measured with the tile
    // skeleton (tools/membench/skel.hip, 1024 x 16 tiles): identity order 5.1 / 4.9 TB/s (first / later pass shape),
    // x at address bits 9-11 5.6 / 5.5.  For tiles narrower than a 128-byte line this also keeps the tiles that share
    // a line on one XCD back to back (one L2 fetches the line once).
    if (p.xq > 0) {
        const int r = (int)(block & ((8LL << p.xq) - 1)), x = r & 7, i = r >> 3;
        const int t = ((i >> p.xp) << (p.xp + 3)) | (x << p.xp) | (i & ((1 << p.xp) - 1));
        block += t - r;
    }
    // M / F and S are powers of two for these plans: shifts, not 64-bit divisions (a 64-bit division is ~150 dependent
    // scalar instructions in front of the first load of the workgroup)
    const long long g = block >> p.tiles_shift;
    const unsigned tile = (unsigned)(block & ((1LL << p.tiles_shift) - 1));
    k2_tile<T, S, F, FIRST, SPLIT, ABL, 0>(ex, p, p.in + g * p.n, p.out + g * p.n, tile, tile, lds);
}
// ---- fused two-pass kernel: work-item decoding (shared by the gfx950 kernel and the host emulator) ------------------------
struct K2FItem {
    int pass;        // 0 / 1; -1: nothing to do (index past the end)
    long long g;     // transform
    unsigned tile;   // tile within the transform, XCD-aware order applied: where it reads ...
    unsigned tile_out;  // ... and where it writes (differs for the units of a three-pass plan, kernels_params.h)
    unsigned slot;   // ring slot: step % ns
    unsigned use;    // step / ns: how many times the slot has been used before
};
// Steps of t0 + t1 items.  Within a step the items go in groups of eight per pass -- items [16 q, 16 q + 8) are first-pass tiles,
// [16 q + 8, 16 q + 16) second-pass tiles (when t0 == t1 and both are multiples of 8; otherwise first-pass tiles, then
// second-pass tiles) -- so that, with workgroup b on XCD b % 8, tile slot i of either pass runs on XCD i % 8 exactly as in the
// one-pass kernels and the same XCD-aware tile permutation (K2Params::xp / xq) applies.
template <class T> MI_HD K2FItem k2f_decode(const K2FusedParams<T>& fp, long long w) {
    // 32-bit arithmetic (the grid is below 2^31); the quotients are workgroup-uniform
    const unsigned t0 = (unsigned)fp.tiles[0], t1 = (unsigned)fp.tiles[1], per = t0 + t1, w32 = (unsigned)w;
    const unsigned s = MI_UNIFORM(w32 / per), r = w32 - s * per;
    K2FItem it{};
    unsigned i;
    if (t0 == t1 && (t0 & 7u) == 0) {
        it.pass = (int)((r >> 3) & 1u);
        i = ((r >> 4) << 3) | (r & 7u);
    } else {
        it.pass = r < t0 ? 0 : 1;
        i = it.pass ? r - t0 : r;
    }
    const long long st = it.pass ? (long long)s - fp.lag : (long long)s;  // the step (transform x unit) this item belongs to
    if (st < 0 || st >= fp.batch) {
        it.pass = -1;
        return it;
    }
    const K2Params<T>& p = fp.pass[it.pass];
    if (p.xq > 0) {  // as k2_body: XCD id at tile-index bits [xp, xp + 3) within aligned groups of 8 << xq tiles
        const unsigned rr = i & ((8u << p.xq) - 1u), x = rr & 7u, j = rr >> 3;
        i += (((j >> p.xp) << (p.xp + 3)) | (x << p.xp) | (j & ((1u << p.xp) - 1u))) - rr;
    }
    const unsigned U = (unsigned)fp.units;
    unsigned u = 0;
    it.g = st;
    if (U > 1) {
        const unsigned g32 = MI_UNIFORM((unsigned)st / U);
        it.g = (long long)g32;
        u = (unsigned)st - g32 * U;
    }
    it.tile = it.pass ? i : u + i * U;
    it.tile_out = it.pass ? u * t1 + i : i;
    it.use = MI_UNIFORM((unsigned)st / (unsigned)fp.ns);
    it.slot = (unsigned)st - it.use * (unsigned)fp.ns;
    return it;
}
// LDS bytes of a column-tile workgroup: the exchange buffer + the staged twiddle tables
template <class T, class S, int F, bool SPLIT, int ABL> constexpr size_t k2_lds_bytes() {
    return lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), k2_twl<ABL, S::NP>()>() + ((ABL & 512) ? 4096 : 0);
}

// ---- large-N pass for lengths that are not powers of two ------------------------------------------------------------
// Same law as k2_body with the two alignment assumptions removed: the last tile of a transform may be ragged (M need
// not be a multiple of F: columns >= M are masked) and a tile may straddle a multiple of S (B div S and B mod S are
// taken per column).  This is the GPU form of the reference's generic MixedRadix (src/algorithm/mixed_radix.rs:128-158)
// for composite lengths; the power-of-two plans keep the specialised body above.
// FUSE (multi-kernel Bluestein, bluesteins_algorithm.rs:100-136): 1 = the FIRST pass reads the caller's rows (pitch n),
// multiplies by the chirp and zero-pads on the fly; 2 = the last pass stores conj(X * bf); 3 = the last pass stores
// conj(X) * chirp, truncated to n, into the caller's rows.  0 = plain pass.
// FUSE (multi-kernel Rader for primes p beyond one workgroup, raders_algorithm.rs:235-283, inner length N = p - 1): 4 = the FIRST
// pass of the first inner transform gathers x[g^(j+1) mod p] from the caller's rows (pitch p); 5 = its last pass stores
// conj(S[j] d[j]), folds conj(x[0]) into element 0 and writes X[0] = x[0] + S[0] to the caller's output row; 6 = the last pass
// of the second inner transform scatters conj(S[j]) to X[g^-(j+1) mod p] of the caller's rows.  The permutations are random
// within a row, i.e. within a 4(p - 1)-byte .. 8(p - 1)-byte span that the L2 holds: HBM still sees each row once.
template <class T, bool FIRST, int FUSE = 0> struct K2gSrc {
    const cx<T>* in;
    unsigned M, S, b0;
    T sgn_in;
    const cx<T>* MI_RESTRICT tlo;
    const cx<T>* MI_RESTRICT thi;
    int hshift, lmask;
    const cx<T>* MI_RESTRICT tab;
    unsigned n_valid;
    const int* MI_RESTRICT perm;
    MI_HD cx<T> lut(unsigned e) const { return tlo[e & (unsigned)lmask] * thi[e >> hshift]; }
    template <int R, int LOG, int K, int J0> MI_HD static void apply_tw(cx<T>* v, cx<T> w, const cx<T>* sp) {
        v[K] = v[K] * w;
        static_for<J0, LOG>([&](auto J_) {
            constexpr int j = J_;
            if constexpr (K + (1 << j) < R) apply_tw<R, LOG, K + (1 << j), j + 1>(v, w * sp[j], sp);
        });
    }
    template <int R, class TT> MI_HD void bfly(int f, int b, int nb, cx<TT>* v) const {
        const unsigned B = b0 + (unsigned)f;
        const unsigned col = B < M ? B : 0;  // masked columns read column 0 (never stored)
        static_for<0, R>([&](auto K_) {
            constexpr int k = K_;
            const unsigned idx = col + (unsigned)(b + k * nb) * M;
            if constexpr (FUSE == 1) {
                cx<T> x = cx<T>{0, 0};
                if (idx < n_valid) {
                    x = in[idx];
                    x.im *= sgn_in;
                    x = x * tab[idx];
                }
                v[k] = x;
            } else if constexpr (FUSE == 4) {
                cx<T> x = in[(unsigned)perm[idx]];
                x.im *= sgn_in;
                v[k] = x;
            } else {
                cx<T> x = in[idx];
                x.im *= sgn_in;
                v[k] = x;
            }
        });
        if constexpr (!FIRST) {
            const unsigned c = col % S;
            constexpr int LOG = (R > 16) ? 5 : (R > 8) ? 4 : (R > 4) ? 3 : (R > 2) ? 2 : (R > 1) ? 1 : 0;
            // the four table factors are fetched HERE, next to the row loads, and everything above stays above the fence: left to
            // itself the scheduler sinks part of the row loads between the twiddle products (three to four dependent memory round
            // trips per butterfly instead of one)
            const unsigned e0 = c * (unsigned)nb, e1 = c * (unsigned)b;
            const cx<T> a_lo = tlo[e0 & (unsigned)lmask], a_hi = thi[e0 >> hshift], b_lo = tlo[e1 & (unsigned)lmask], b_hi = thi[e1 >> hshift];
            MI_SCHED_FENCE();
            cx<T> sp[LOG > 0 ? LOG : 1];
            if constexpr (LOG > 0) {
                sp[0] = a_lo * a_hi;
                static_for<1, LOG>([&](auto I_) {
                    constexpr int i = I_;
                    sp[i] = sp[i - 1] * sp[i - 1];
                });
            }
            apply_tw<R, LOG, 0, 0>(v, b_lo * b_hi, sp);
        }
    }
};

template <class T, class S, int F, bool FIRST, int FUSE, bool SPLIT = false, int TWL = 0, class X>
MI_HD void k2g_body(X& ex, const K2Params<T>& p, long long block, void* lds) {
    static_assert(FUSE == 0 || (FUSE == 1 || FUSE == 4) == FIRST, "chirp-in / gather fuse into a first pass, the output stages into a last pass");
    constexpr int R = S::N;
    // XCD-aware tile order as in k2_body, over the GLOBAL workgroup index (tile counts are not multiples of 64 here): every
    // complete aligned group of 64 workgroups is permuted so that XCD x takes runs of 2^XP adjacent tiles (XCD id at address
    // bits 9 .. 11 of the row segment); the last, partial group keeps the identity order (xfull = number of complete groups).
    // A group may straddle two transforms: the map is a bijection on workgroup indices either way.  Tiles with row segments of
    // 512 bytes or more (XP = 0: the identity) and the fused Bluestein passes keep the plain order and the plain index code.
    constexpr int SEGB = F * (int)sizeof(cx<T>), XP = SEGB >= 512 ? 0 : SEGB >= 256 ? 1 : SEGB >= 128 ? 2 : 3;
    long long g;
    unsigned b0;
    if constexpr (FUSE == 4 || FUSE == 6) {
        // The gather / scatter passes of the multi-kernel Rader touch a caller's row (<= 8 (p - 1) bytes) at random 8- or 16-byte
        // granularity.  Workgroup b runs on XCD b % 8 and every XCD has its own L2: with the plain order the tiles of ONE row are
        // spread over all eight, so each L2 fetches the whole row for its gathers (8x the reads) and evicts partially written
        // lines of the scattered output (up to 16x the writes).  Here all tiles of transform g run on XCD g % 8: one L2 fetches the
        // row once and merges the scattered stores into full lines before they leave.  (Complete groups of eight transforms; the
        // last batch % 8 transforms keep the plain order.)
        const unsigned tpf = (unsigned)p.tiles_per_fft, full = ((unsigned)p.batch >> 3) << 3;
        const unsigned long long nfull = (unsigned long long)full * tpf;
        unsigned gi, tile;
        if ((unsigned long long)block < nfull) {
            const unsigned blk = (unsigned)block, x = blk & 7u, i = blk >> 3;
            gi = (i / tpf) * 8u + x;
            tile = i % tpf;
        } else {
            const unsigned r = (unsigned)((unsigned long long)block - nfull);
            gi = full + r / tpf;
            tile = r % tpf;
        }
        g = (long long)gi;
        b0 = tile * (unsigned)F;
    } else if constexpr (XP > 0 && FUSE == 0) {
        if (p.xq > 0 && (block >> 6) < (long long)p.xfull) {
            const int r = (int)(block & 63), x = r & 7, i = r >> 3;
            const int t = ((i >> XP) << (XP + 3)) | (x << XP) | (i & ((1 << XP) - 1));
            block += t - r;
        }
        // grid <= 2^31 - 1 (plan.cpp kMaxGrid): 32-bit division
        const unsigned tpf = (unsigned)p.tiles_per_fft;
        g = (long long)((unsigned)block / tpf);
        b0 = ((unsigned)block - (unsigned)g * tpf) * (unsigned)F;
    } else {
        g = block / p.tiles_per_fft;
        b0 = (unsigned)(block % p.tiles_per_fft) * (unsigned)F;
    }
    const cx<T>* in = p.in + g * ((FUSE == 1 || FUSE == 4) ? p.n_io : p.n);
    cx<T>* out = p.out + g * ((FUSE == 3 || FUSE == 6) ? p.n_io : p.n);
    const int* MI_RESTRICT perm = p.perm;
    const cx<T>* xin = FUSE == 5 ? p.xin + g * p.n_io : nullptr;
    cx<T>* xout = FUSE == 5 ? p.xout + g * p.n_io : nullptr;
    const T sgn_x = p.sgn_x;
    const unsigned M = (unsigned)p.m, Sg = (unsigned)p.s;
    const T sgn_out = p.sgn_out;
    const cx<T>* MI_RESTRICT tab = p.tab;
    const unsigned n_valid = p.n_valid;
    K2gSrc<T, FIRST, FUSE> src{in, M, Sg, b0, p.sgn_in, p.tlo, p.thi, p.hshift, p.lmask, tab, n_valid, perm};
    auto dst = [=](int f, int k, cx<T> x) {
        const unsigned B = b0 + (unsigned)f;
        if (B < M) {
            if constexpr (FIRST) {
                x.im *= sgn_out;
                out[B * (unsigned)R + (unsigned)k] = x;
            } else {
                const unsigned e = (B / Sg) * (Sg * (unsigned)R) + (B % Sg) + (unsigned)k * Sg;
                if constexpr (FUSE == 2) {
                    out[e] = cconj(x * tab[e]);
                } else if constexpr (FUSE == 3) {
                    if (e < n_valid) {
                        cx<T> y = cconj(x) * tab[e];
                        y.im *= sgn_out;
                        out[e] = y;
                    }
                } else if constexpr (FUSE == 5) {
                    cx<T> t = cconj(x * tab[e]);
                    if (e == 0) {  // raders_algorithm.rs:257-266: X[0] = x[0] + S[0]; S[0] d[0] picks up conj(x[0])
                        cx<T> x0 = xin[0];
                        x0.im *= sgn_x;
                        t = t + cconj(x0);
                        cx<T> X0 = x0 + x;
                        X0.im *= sgn_x;
                        xout[0] = X0;
                    }
                    out[e] = t;
                } else if constexpr (FUSE == 6) {
                    cx<T> y = cconj(x);
                    y.im *= sgn_out;
                    out[(unsigned)perm[e]] = y;
                } else {
                    x.im *= sgn_out;
                    out[e] = x;
                }
            }
        }
    };
    wg_fft<T, S, F, MAP_FF, FIRST ? MAP_EF : MAP_FF, SPLIT, false, k2_pitch_mod(F), 0, -1, false, (S::NP >= 2 ? TWL : 0)>(ex, lds, p.tw, src, dst);
}

// ---- large-N pass whose tile height is a PRIME P: Rader inside the tile ----------------------------------------------------
// The six-step MixedRadix of the reference (src/algorithm/mixed_radix.rs:53-158) for lengths with prime factors above 31
// (37 x 41, 101 x 103 ...: src/plan.rs:474-506 plans them as MixedRadix over Rader / Bluestein inner FFTs).  Same pass law as
// k2g_body -- a workgroup owns F adjacent columns, in_j = X[B + j M] w_{S P}^{(B mod S) j}, out_k -> Y[(B div S) S P + (B mod S) + k S]
// -- but the length-P column transform is Rader's algorithm (raders_algorithm.rs:235-283) run on the tile in LDS exactly as
// rader_body MODE 1 runs it on contiguous rows: the strided load scatters row j of every column to its convolution slot
// (perm = the inverse map j -> slot, x[0] to a spare slot), two inner transforms of length P - 1 with the spectrum multiply in
// between, the second one scattering to natural order, then the strided (first pass: contiguous) store.  HBM sees each element
// of the pass once, in F-element row segments, like every other column-tile pass.
template <class T, class S, int F, bool FIRST, class X>
MI_HD void k2r_body(X& ex, const K2Params<T>& p, long long block, void* lds) {
    constexpr int Mi = S::N, P = S::N + 1, PITCH = S::pitch(), NT = F * S::TPF;
    // x[0], then X[0]: a slot past both the exchange span and the natural-order outputs (see rader_body MODE 1)
    constexpr int XS = (S::phys(Mi - 1) + 1 > P) ? S::phys(Mi - 1) + 1 : P;
    static_assert(XS < PITCH && P <= PITCH, "row pitch must leave a spare slot");
    const unsigned tpf = (unsigned)p.tiles_per_fft;
    const long long g = (long long)((unsigned)block / tpf);  // grid <= 2^31 - 1 (plan.cpp kMaxGrid)
    const unsigned b0 = ((unsigned)block - (unsigned)g * tpf) * (unsigned)F;
    const cx<T>* in = p.in + g * p.n;
    cx<T>* out = p.out + g * p.n;
    const unsigned M = (unsigned)p.m, Sg = (unsigned)p.s;
    const T sgn_in = p.sgn_in, sgn_out = p.sgn_out;
    const cx<T>* MI_RESTRICT dtab = p.tab;
    const cx<T>* MI_RESTRICT tlo = p.tlo;
    const cx<T>* MI_RESTRICT thi = p.thi;
    const int hshift = p.hshift, lmask = p.lmask;
    const int* MI_RESTRICT perm_in = p.perm;
    const int* MI_RESTRICT perm_out = p.perm2;
    cx<T>* work = (cx<T>*)lds;
    // lanes walk across the tile's columns: thread t holds column t % F and the rows t / F, t / F + TPF, ...
    // The thread's rows are fetched in batches of eight -- every load of a batch (row element, its two twiddle-table factors, its
    // slot) is issued before the first one is consumed.  As a run-time loop (load, wait, LDS write, branch) the P / TPF rows of a
    // thread cost as many DEPENDENT memory round trips, which is what bound these passes at first (1.6 - 2.5 TB/s).
    ex.for_threads([&](int tid, cx<T>*) {
        const unsigned c = (unsigned)tid % (unsigned)F, B = b0 + c, col = B < M ? B : 0;  // masked columns read column 0 (never stored)
        const unsigned cs = FIRST ? 0u : col % Sg, r0 = (unsigned)tid / (unsigned)F;
        constexpr int NL = (P + S::TPF - 1) / S::TPF, CH = 8;
        static_for<0, (NL + CH - 1) / CH>([&](auto Q_) {
            constexpr int q = Q_, nq = (NL - q * CH < CH) ? NL - q * CH : CH;
            cx<T> xr[nq], wl[nq], wh[nq];
            unsigned slot[nq];
            static_for<0, nq>([&](auto I_) {
                constexpr int i = I_;
                const unsigned r = r0 + (unsigned)((q * CH + i) * S::TPF), rr = r < (unsigned)P ? r : 0u;
                xr[i] = in[col + rr * M];
                slot[i] = (unsigned)perm_in[rr];
                if constexpr (!FIRST) {
                    const unsigned e = cs * rr;
                    wl[i] = tlo[e & (unsigned)lmask];
                    wh[i] = thi[e >> hshift];
                }
            });
            static_for<0, nq>([&](auto I_) {
                constexpr int i = I_;
                const unsigned r = r0 + (unsigned)((q * CH + i) * S::TPF);
                if (r < (unsigned)P) {
                    cx<T> x = xr[i];
                    x.im *= sgn_in;
                    if constexpr (!FIRST) x = x * (wl[i] * wh[i]);
                    work[c * PITCH + (r == 0 ? (unsigned)XS : slot[i])] = x;
                }
            });
        });
    });
    ex.barrier();
    auto src1 = [=](int f, int j) -> cx<T> { return work[f * PITCH + j]; };
    auto dst1 = [=](int f, int j, cx<T> v) {
        cx<T> t = cconj(v * dtab[j]);
        if (j == 0) {
            const cx<T> x0 = work[f * PITCH + XS];
            t = t + cconj(x0);
            work[f * PITCH + XS] = x0 + v;  // X[0]
        }
        work[f * PITCH + j] = t;
    };
    wg_fft<T, S, F, MAP_EF, MAP_EF, false, true>(ex, lds, p.tw, elem_src(src1), dst1);
    ex.barrier();
    auto dst2 = [=](int f, int j, cx<T> v) {
        if (j == 0) work[f * PITCH] = work[f * PITCH + XS];  // slot 0 is no target of the g^-j scatter
        work[f * PITCH + perm_out[j]] = cconj(v);
    };
    wg_fft<T, S, F, MAP_EF, MAP_EF, false, true>(ex, lds, p.tw, elem_src(src1), dst2);
    ex.barrier();
    ex.for_threads([&](int tid, cx<T>*) {
        if constexpr (FIRST) {  // the tile's F x P outputs are one contiguous block: lanes walk along it
            for (int t = tid; t < F * P; t += NT) {
                const int f = t / P, i = t - f * P;
                if (b0 + (unsigned)f < M) {
                    cx<T> y = work[f * PITCH + i];
                    y.im *= sgn_out;
                    out[b0 * (unsigned)P + (unsigned)t] = y;
                }
            }
        } else {
            const unsigned c = (unsigned)tid % (unsigned)F, B = b0 + c;
            if (B < M) {
                const unsigned obase = (B / Sg) * (Sg * (unsigned)P) + (B % Sg);
                for (unsigned k = (unsigned)tid / (unsigned)F; k < (unsigned)P; k += (unsigned)S::TPF) {
                    cx<T> y = work[c * PITCH + k];
                    y.im *= sgn_out;
                    out[obase + k * Sg] = y;
                }
            }
        }
    });
}
template <class T, class S, int F> constexpr size_t k2r_lds_bytes() { return (size_t)F * S::pitch() * sizeof(cx<T>); }

// ---- Bluestein: any length n <= (M + 1) / 2 through two length-M workgroup transforms ---------------------
// The second transform runs the REVERSED schedule (engine.h reversed_sched): the first one leaves X[b + k M/R] in the
// registers of the thread that needs exactly those values as inputs of its first radix-R butterflies, so the spectrum
// multiply conj(X bf) happens in registers and the intermediate spectrum never goes through LDS (round 1 staged it in a
// natural-order LDS row: one more write + read of M elements and one more barrier per row).
// BFREG >= 0: the multiplier entries were fetched into v[BFREG + slot] in front of the first transform's last sub-pass (bluestein_body PF bit 2)
template <class T, class S2, int BFREG = -1> struct BluesteinRegSrc {
    static constexpr bool kLoadsAll = true;
    const cx<T>* MI_RESTRICT bf;
    template <class SS> MI_HD void load_all(int, int u, cx<T>* v) const {
        static_assert(std::is_same<SS, S2>::value, "source of the reversed schedule");
        constexpr int R = S2::R[0], NB = S2::nb(0), BPT = S2::bpt(0);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S2::TPF;
            if ((m + 1) * S2::TPF <= NB || b < NB) {
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    if constexpr (BFREG >= 0)
                        v[m * R + k] = cconj(v[m * R + k] * v[BFREG + m * R + k]);
                    else
                        v[m * R + k] = cconj(v[m * R + k] * bf[(unsigned)(b + k * NB)]);
                });
            }
        });
    }
};
// SPLIT (real / imaginary planes exchanged one after the other): with no staged spectrum, a padded length up to 32768 (f64:
// 16384) fits one workgroup's LDS, so lengths up to 16384 run in ONE kernel at 2 n elements of traffic (round 1: two
// whole-row kernels through an HBM workspace, 2 M + 2 n elements).
// TW1 (a per-inner-length measured option): sub-pass 1's twiddle table of BOTH transforms staged in LDS (engine.h TWL), each in
// its own region behind the exchange buffer; schedules with at least three sub-passes only (with two, sub-pass 1 is the last one
// and its table would be the big one)
template <class S> constexpr bool bluestein_tw1_ok() { return S::NP >= 3; }
// which sub-pass tables of both transforms are staged in LDS: TW1 = sub-pass 1; PF bit 1 (round 5, tuning): every sub-pass but the last (the last
// one's table is the big one: (R - 1) N / R entries)
template <class S, bool TW1, int PF> constexpr int bluestein_twl_mask() {
    if (!bluestein_tw1_ok<S>()) return 0;
    if ((PF & 2) != 0) return S::NP >= 4 ? (twl_all(S::NP) & ~(1 << (S::NP - 1))) : 2;
    return TW1 ? 2 : 0;
}
template <class T, class S, int F, bool SPLIT, bool TW1, int PF = 0> constexpr size_t bluestein_lds_bytes() {
    using S2 = typename reversed_sched<S>::type;
    const size_t ex = (size_t)F * (S::pitch() > S2::pitch() ? S::pitch() : S2::pitch()) * (SPLIT ? sizeof(T) : sizeof(cx<T>));
    constexpr int MASK = bluestein_twl_mask<S, TW1, PF>();
    if (MASK == 0) return ex;
    return align16(ex) + align16((size_t)twl_total<S>(MASK) * sizeof(cx<T>)) + (size_t)twl_total<S2>(MASK) * sizeof(cx<T>);
}
// PF (tuning so far, round 5) bit 0: the twiddle factors of the sub-passes whose tables are NOT staged in LDS are fetched one exchange ahead of
// their use into registers behind the data (engine.h TWSTAGE) instead of right after the barrier that precedes the sub-pass; bit 1: every
// sub-pass table but the last one's staged in LDS (bluestein_twl_mask).
// bit 2: the spectrum multiplier bf[] of a thread's hand-over values is fetched in FRONT of the first transform's last sub-pass (into registers
// behind the twiddle block) instead of right when the second transform starts; bit 3: the same for the output chirp of the second transform's
// last sub-pass.
template <class S, int PF> constexpr int bluestein_twregs() {
    using S2 = typename reversed_sched<S>::type;
    return (PF & 1) ? (twreg_count<S>() > twreg_count<S2>() ? twreg_count<S>() : twreg_count<S2>()) : 0;
}
template <class S, int PF> constexpr int bluestein_regs() { return S::emax() + bluestein_twregs<S, PF>() + ((PF & 12) ? S::emax() : 0); }
template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0, class X>
MI_HD void bluestein_body(X& ex, const BluesteinParams<T>& p, long long block, void* lds) {
    using S2 = typename reversed_sched<S>::type;
    static_assert(S2::R[0] == S::R[S::NP - 1] && S2::nb(0) == S::nb(S::NP - 1) && S2::bpt(0) == S::bpt(S::NP - 1), "register hand-over");
    const long long fft0 = block * F;
    // uniform row base + 32-bit offsets: one address register per access instead of a 64-bit pair
    const cx<T>* in = p.in + fft0 * p.n;
    cx<T>* out = p.out + fft0 * p.n;
    const cx<T>* MI_RESTRICT chirp = p.chirp;
    const int rows = (int)((p.batch - fft0) < F ? (p.batch - fft0) : F);
    const int n = p.n;
    const T sgn = p.sgn;
    auto src1 = [=](int f, int i) -> cx<T> {
        if (f < rows && i < n) {
            cx<T> x = in[(unsigned)(f * n + i)];
            x.im *= sgn;
            return x * chirp[(unsigned)i];
        }
        return cx<T>{0, 0};
    };
    // both transforms share one exchange buffer of max(pitch) entries per row; the staged tables sit behind it
    constexpr int MASK = bluestein_twl_mask<S, TW1, PF>();
    constexpr bool STG = MASK != 0;
    constexpr int TWR = (PF & 1) ? S::emax() : -1;
    constexpr size_t EXB = (size_t)F * (S::pitch() > S2::pitch() ? S::pitch() : S2::pitch()) * (SPLIT ? sizeof(T) : sizeof(cx<T>));
    constexpr int OFF1 = STG ? (int)(align16(EXB) - align16(lds_bytes<T, S, F, SPLIT>())) : 0;
    constexpr int OFF2 = STG ? (int)(align16(EXB) + align16((size_t)twl_total<S>(MASK) * sizeof(cx<T>)) - align16(lds_bytes<T, S2, F, SPLIT>())) : 0;
    // one spare register block behind data and twiddles: the multiplier entries, later the output chirp (the multiplier is dead by then)
    constexpr int XREG = S::emax() + bluestein_twregs<S, PF>();
    const cx<T>* MI_RESTRICT bfp = p.bf;
    auto pre1 = [=](int, int u, cx<T>* v) {  // geometry of the first transform's LAST sub-pass = the second one's first
        constexpr int LP = S::NP - 1, R = S::R[LP], NB = S::nb(LP), BPT = S::bpt(LP);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S::TPF;
            if ((m + 1) * S::TPF <= NB || b < NB) {
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    v[XREG + m * R + k] = bfp[(unsigned)(b + k * NB)];
                });
            }
        });
    };
    if constexpr ((PF & 4) != 0)
        wg_fft<T, S, F, MAP_EF, MAP_EF, SPLIT, false, 1, 0, TWR, (PF & 1) != 0, MASK, OFF1>(ex, lds, p.tw, elem_src(src1), KeepInRegsPre<decltype(pre1)>{pre1});
    else
        wg_fft<T, S, F, MAP_EF, MAP_EF, SPLIT, false, 1, 0, TWR, (PF & 1) != 0, MASK, OFF1>(ex, lds, p.tw, elem_src(src1), KeepInRegs{});
    // (the engine's barrier after the last gather of the first transform already orders it before the second one's scatters)
    auto dst2 = [=](int f, int j, cx<T> v) {
        if (f < rows && j < n) {
            cx<T> y = cconj(v) * chirp[(unsigned)j];
            y.im *= sgn;
            out[(unsigned)(f * n + j)] = y;
        }
    };
    auto dst2s = [=](int f, int j, cx<T> val, auto I_, cx<T>* v) {
        if (f < rows && j < n) {
            cx<T> y = cconj(val) * v[XREG + decltype(I_)::value];
            y.im *= sgn;
            out[(unsigned)(f * n + j)] = y;
        }
    };
    auto pre2 = [=](int, int u, cx<T>* v) {  // output chirp of the second transform's last sub-pass: element base + k stride of butterfly b
        constexpr int LP = S2::NP - 1, R = S2::R[LP], NB = S2::nb(LP), ST = S2::stride(LP), BPT = S2::bpt(LP);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S2::TPF;
            if ((m + 1) * S2::TPF <= NB || b < NB) {
                const int base = (b / ST) * (ST * R) + (b % ST);
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    const int j = base + k * ST;
                    v[XREG + m * R + k] = chirp[(unsigned)(j < n ? j : 0)];
                });
            }
        });
    };
    constexpr int BFR = (PF & 4) ? XREG : -1;
    if constexpr ((PF & 8) != 0)
        wg_fft<T, S2, F, MAP_EF, MAP_EF, SPLIT, false, 1, 0, TWR, (PF & 1) != 0, MASK, OFF2>(ex, lds, p.tw2, BluesteinRegSrc<T, S2, BFR>{p.bf},
                                                                                                 SlotDstPre<decltype(dst2s), decltype(pre2)>{dst2s, pre2});
    else
        wg_fft<T, S2, F, MAP_EF, MAP_EF, SPLIT, false, 1, 0, TWR, (PF & 1) != 0, MASK, OFF2>(ex, lds, p.tw2, BluesteinRegSrc<T, S2, BFR>{p.bf}, dst2);
}

// ---- run-time scheduled batched transform (13-smooth lengths) ------------------------------------------------
template <class T, int EMAX, int LIGHT, class X>
MI_HD void dyn_k1_body(X& ex, const DynK1Params<T>& p, long long block, void* lds) {
    const DynSched& s = p.s;
    const long long fft0 = block * s.f;
    const int n = s.n;
    const cx<T>* in = p.in + fft0 * n;
    cx<T>* out = p.out + fft0 * n;
    const int rows = (int)((p.batch - fft0) < s.f ? (p.batch - fft0) : s.f);
    const T sgn = p.sgn;
    auto src = [=](int f, int i) -> cx<T> {
        if (f < rows) {
            cx<T> x = in[(unsigned)(f * n + i)];
            x.im *= sgn;
            return x;
        }
        return cx<T>{0, 0};
    };
    auto dst = [=](int f, int i, cx<T> x) {
        if (f < rows) {
            x.im *= sgn;
            out[(unsigned)(f * n + i)] = x;
        }
    };
    wg_fft_dyn<T, EMAX, LIGHT>(ex, s, lds, p.tw, src, dst);
}

// ---- run-time scheduled Rader (any prime p with 13-smooth p - 1); same steps as rader_body ---------------------
template <class T, int EMAX, int LIGHT, class X>
MI_HD void dyn_rader_body(X& ex, const DynRaderParams<T>& p, long long block, void* lds) {
    const DynSched& s = p.s;
    const int M = s.n, P = s.n + 1, PITCH = s.pitch, F = s.f, NT = s.f * s.tpf;
    const long long fft0 = block * F;
    const cx<T>* in = p.in;
    cx<T>* out = p.out;
    const cx<T>* MI_RESTRICT dtab = p.d;
    const int* MI_RESTRICT perm_in = p.perm_in;
    const int* MI_RESTRICT perm_out = p.perm_out;
    const T sgn = p.sgn;
    cx<T>* work = (cx<T>*)lds;
    cx<T>* rows = work + F * PITCH;
    const long long rows_here = (p.batch - fft0) < F ? (p.batch - fft0) : F;
    const int valid = (int)(rows_here * P);
    ex.for_threads([&](int tid, cx<T>*) {
        for (int t = tid; t < F * P; t += NT) {
            cx<T> x = cx<T>{0, 0};
            if (t < valid) {
                x = in[fft0 * P + t];
                x.im *= sgn;
            }
            rows[t] = x;
        }
    });
    ex.barrier();
    auto src1 = [=](int f, int j) -> cx<T> { return rows[f * P + perm_in[j]]; };
    auto dst1 = [=](int f, int j, cx<T> v) {
        cx<T> t = cconj(v * dtab[j]);
        if (j == 0) {
            // rows[f*P] (= x[0]) is read by nobody else: the g^k gather never touches index 0
            const cx<T> x0 = rows[f * P];
            t = t + cconj(x0);
            rows[f * P] = x0 + v;  // X[0]
        }
        work[f * PITCH + j] = t;
    };
    wg_fft_dyn<T, EMAX, LIGHT>(ex, s, lds, p.tw, src1, dst1);
    ex.barrier();
    auto src2 = [=](int f, int i) -> cx<T> { return work[f * PITCH + i]; };
    auto dst2 = [=](int f, int j, cx<T> v) { rows[f * P + perm_out[j]] = cconj(v); };
    wg_fft_dyn<T, EMAX, LIGHT, true>(ex, s, lds, p.tw, src2, dst2);
    ex.barrier();
    ex.for_threads([&](int tid, cx<T>*) {
        for (int t = tid; t < valid; t += NT) {
            cx<T> y = rows[t];
            y.im *= sgn;
            out[fft0 * P + t] = y;
        }
    });
    (void)M;
}

// ---- element-wise stages of the multi-kernel Bluestein (one thread per output element, grid-stride) ---------
template <class T> MI_HD void pointwise_elem(const PointwiseParams<T>& p, long long idx) {
    if (p.stage == 0) {
        const long long r = idx / p.m, i = idx % p.m;
        cx<T> v{0, 0};
        if (i < p.n) {
            v = p.in[r * p.n + i];
            v.im *= p.sgn;
            v = v * p.tab[i];
        }
        p.out[idx] = v;
    } else if (p.stage == 1) {
        const long long j = idx % p.m;
        p.out[idx] = cconj(p.in[idx] * p.tab[j]);
    } else {
        const long long r = idx / p.n, i = idx % p.n;
        cx<T> v = cconj(p.in[r * p.m + i]) * p.tab[i];
        v.im *= p.sgn;
        p.out[idx] = v;
    }
}

// ---- Rader: prime length p = S::N + 1 ---------------------------------------------------------------------
// MODE 0 -- LDS: [F][PITCH] exchange/work buffer followed by [F][p] staging of the rows (the g^j permutations are
//   random within a row, so they are applied against LDS, never against HBM).
// MODE 1 -- one [F][PITCH] buffer: the coalesced load scatters x[t] straight to its convolution slot (perm_in holds the
//   INVERSE map t -> j with g^(j+1) = t), both transforms read their inputs linearly from LDS, and the second transform
//   scatters its outputs to their final positions in the same buffer.  Half the LDS, one LDS round trip fewer.
// MODE 5 -- MODE 1 with the REGISTER HAND-OVER the one-kernel Bluestein uses: the second transform runs the reversed schedule, so
//   the last sub-pass of the first leaves X[b + k NB] in exactly the registers the second's first radix-R butterflies read; the
//   d[] multiply (and the x[0] / X[0] step) happens in registers and the spectrum never goes through LDS: one LDS write + read of
//   p - 1 elements and one barrier fewer per row.  Rows sit at the larger pitch of the two schedules; x[0] / X[0] of row f live
//   in slot F PITCH + f, behind every row, because the two schedules' exchange spans differ.
// Body forms 2 / 3 / 4 are the rows loop (rader_rows_body); 6 is the rows loop with the same register hand-over.
// (MODE 9, round 5: MODE 3 with non-temporal row loads -- config 4's prime: +2.1 / +2.2 % at its full batch, profiles/r5/ab_c4_nt.jsonl; the same
// hint in every other rows loop measured nothing, ab_rader_nt_*.jsonl)
constexpr bool rader_rows_mode(int mode) { return (mode >= 2 && mode <= 4) || mode == 6 || mode == 9; }
template <class S> constexpr int rader5_pitch() {
    using S2 = typename reversed_sched<S>::type;
    return S::pitch() > S2::pitch() ? S::pitch() : S2::pitch();
}
template <class T, class S2> struct RaderRegSrc {
    static constexpr bool kLoadsAll = true;
    const cx<T>* MI_RESTRICT d;
    cx<T>* spare;  // x[0] on entry, X[0] on exit, one slot per row
    template <class SS> MI_HD void load_all(int f, int u, cx<T>* v) const {
        static_assert(std::is_same<SS, S2>::value, "source of the reversed schedule");
        constexpr int R = S2::R[0], NB = S2::nb(0), BPT = S2::bpt(0);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S2::TPF;
            if ((m + 1) * S2::TPF <= NB || b < NB) {
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    const cx<T> val = v[m * R + k];
                    cx<T> t = cconj(val * d[(unsigned)(b + k * NB)]);
                    if constexpr (m == 0 && k == 0) {
                        if (u == 0) {  // spectrum index 0 (raders_algorithm.rs:256-262)
                            const cx<T> x0 = spare[f];
                            t = t + cconj(x0);
                            spare[f] = x0 + val;  // X[0]
                        }
                    }
                    v[m * R + k] = t;
                });
            }
        });
    }
};
// MI355_RADER_PF (a probe build, tools/r5/build_rader_pf.sh): the sub-pass factors of the non-rows-loop bodies (MODE 1 / 5) fetched one exchange
// ahead of their sub-pass (engine.h TWSTAGE), as the Bluestein bodies do
#if defined(MI355_RADER_PF)
constexpr bool kRaderPF = true;
#else
constexpr bool kRaderPF = false;
#endif
template <class S> constexpr int rader_regs() {
    using S2 = typename reversed_sched<S>::type;
    return S::emax() + (kRaderPF ? (twreg_count<S>() > twreg_count<S2>() ? twreg_count<S>() : twreg_count<S2>()) : 0);
}
template <class T, class S, int F, int MODE, class X>
MI_HD void rader_body(X& ex, const RaderParams<T>& p, long long block, void* lds) {
    constexpr int RTW = kRaderPF ? S::emax() : -1;
    constexpr int M = S::N, P = S::N + 1, PITCH = (MODE == 5) ? rader5_pitch<S>() : S::pitch(), NT = F * S::TPF;
    const long long fft0 = block * F;
    const cx<T>* in = p.in;
    cx<T>* out = p.out;
    const cx<T>* MI_RESTRICT dtab = p.d;
    const int* MI_RESTRICT perm_in = p.perm_in;
    const int* MI_RESTRICT perm_out = p.perm_out;
    const long long batch = p.batch;
    const T sgn = p.sgn;
    cx<T>* work = (cx<T>*)lds;
    const long long rows_here = (batch - fft0) < F ? (batch - fft0) : F;
    const int valid = (int)(rows_here * P);
    if constexpr (MODE == 1 || MODE == 5) {
        // x[0], then X[0]: a slot past BOTH the exchange span (physical slots < phys(M-1) + 1) and the natural-order outputs
        // the last scatter writes (logical indices <= M = p - 1).  An unpadded layout has phys(M-1) + 1 == M, which IS the
        // target of the output with g^-(j+1) = p - 1: the slot must not be below p (round 2: a race that a rescheduling exposed).
        // MODE 5 keeps it behind all rows instead (slot F PITCH + f, addressed as row 0's slot F PITCH: f PITCH + XS' with
        // XS' = F PITCH + f - f PITCH is not a constant, so the accesses below go through xs_slot()).
        constexpr int XS = (MODE == 5) ? 0 : (S::phys(M - 1) + 1 > P) ? S::phys(M - 1) + 1 : P;
        static_assert(MODE == 5 || (XS < PITCH && P <= PITCH), "row pitch must leave a spare slot");
        static_assert(P <= PITCH, "a row holds its p natural-order outputs");
        auto xs_slot = [](int f) -> int { return (MODE == 5) ? F * PITCH + f : f * PITCH + XS; };
        // batches of eight elements per thread: all loads of a batch (element, slot) in flight before the first LDS write -- a
        // run-time loop costs one dependent memory round trip per element (F P / NT of them)
        ex.for_threads([&](int tid, cx<T>*) {
            constexpr int NL = (F * P + NT - 1) / NT, CH = 8;
            const cx<T>* rowsp = in + fft0 * P;
            static_for<0, (NL + CH - 1) / CH>([&](auto Q_) {
                constexpr int q = Q_, nq = (NL - q * CH < CH) ? NL - q * CH : CH;
                cx<T> xr[nq];
                int slot[nq];
                static_for<0, nq>([&](auto I_) {
                    constexpr int i = I_;
                    const int t = tid + (q * CH + i) * NT, tc = t < F * P ? t : 0, f = tc / P, e = tc - f * P;
                    xr[i] = rowsp[t < valid ? t : 0];
                    const int pj = perm_in[e];  // unconditional (entry 0 of the inverse map is a spare 0): no branch around the load
                    // (a select between two computed values, not a branch: a branch here waits for pj and serialises the batch)
                    if constexpr (MODE == 5) {
                        const int s0 = F * PITCH + f, s1 = f * PITCH + pj;
                        slot[i] = (e == 0) ? s0 : s1;
                    } else {
                        slot[i] = f * PITCH + (e == 0 ? XS : pj);
                    }
                });
                static_for<0, nq>([&](auto I_) {
                    constexpr int i = I_;
                    const int t = tid + (q * CH + i) * NT;
                    if (t < F * P) {
                        cx<T> x = xr[i];
                        x.im *= sgn;
                        const bool ok = t < valid;  // rows past the batch: zeros
                        work[slot[i]] = cx<T>{ok ? x.re : (T)0, ok ? x.im : (T)0};
                    }
                });
            });
        });
        ex.barrier();
        auto src1 = [=](int f, int j) -> cx<T> { return work[f * PITCH + j]; };
        auto dst2 = [=](int f, int j, cx<T> v) {
            if (j == 0) work[f * PITCH] = work[xs_slot(f)];  // slot 0 is no target of the g^-j scatter
            work[f * PITCH + perm_out[j]] = cconj(v);
        };
        if constexpr (MODE == 5) {
            using S2 = typename reversed_sched<S>::type;
            static_assert(S2::R[0] == S::R[S::NP - 1] && S2::nb(0) == S::nb(S::NP - 1) && S2::bpt(0) == S::bpt(S::NP - 1) && S2::TPF == S::TPF,
                          "register hand-over");
            wg_fft<T, S, F, MAP_EF, MAP_EF, false, true, 1, 0, RTW, kRaderPF>(ex, lds, p.tw, elem_src(src1), KeepInRegs{});
            // (the barrier after the last gather of the first transform orders it before the second one's scatters; the natural-
            // order outputs are written after the second transform's last gather + barrier, when no exchange data is live)
            wg_fft<T, S2, F, MAP_EF, MAP_EF, false, false, 1, 0, RTW, kRaderPF>(ex, lds, p.tw2, RaderRegSrc<T, S2>{dtab, work + F * PITCH}, dst2);
        } else {
            auto dst1 = [=](int f, int j, cx<T> v) {
                cx<T> t = cconj(v * dtab[j]);
                if (j == 0) {
                    const cx<T> x0 = work[f * PITCH + XS];
                    t = t + cconj(x0);
                    work[f * PITCH + XS] = x0 + v;  // X[0]
                }
                work[f * PITCH + j] = t;
            };
            wg_fft<T, S, F, MAP_EF, MAP_EF, false, true, 1, 0, RTW, kRaderPF>(ex, lds, p.tw, elem_src(src1), dst1);
            ex.barrier();
            wg_fft<T, S, F, MAP_EF, MAP_EF, false, true, 1, 0, RTW, kRaderPF>(ex, lds, p.tw, elem_src(src1), dst2);
        }
        ex.barrier();
        ex.for_threads([&](int tid, cx<T>*) {
            for (int t = tid; t < valid; t += NT) {
                const int f = t / P, i = t - f * P;
                cx<T> y = work[f * PITCH + i];
                y.im *= sgn;
                out[fft0 * P + t] = y;
            }
        });
    } else {
        cx<T>* rows = work + F * PITCH;
        // coalesced flat copy of this workgroup's rows into LDS
        ex.for_threads([&](int tid, cx<T>*) {
            for (int t = tid; t < F * P; t += NT) {
                cx<T> x = cx<T>{0, 0};
                if (t < valid) {
                    x = in[fft0 * P + t];
                    x.im *= sgn;
                }
                rows[t] = x;
            }
        });
        ex.barrier();
        auto src1 = [=](int f, int j) -> cx<T> { return rows[f * P + perm_in[j]]; };
        auto dst1 = [=](int f, int j, cx<T> v) {
            cx<T> t = cconj(v * dtab[j]);
            if (j == 0) {
                const cx<T> x0 = rows[f * P];
                t = t + cconj(x0);
                work[f * PITCH + M] = x0 + v;  // X[0]; slot M of the row is outside the transform's span (PITCH > M)
            }
            work[f * PITCH + j] = t;
        };
        wg_fft<T, S, F, MAP_EF, MAP_EF, false>(ex, lds, p.tw, elem_src(src1), dst1);
        ex.barrier();
        // X[0] must leave `work` before the second transform's exchanges reuse the buffer
        ex.for_threads([&](int tid, cx<T>*) {
            if (tid < F) rows[tid * P] = work[tid * PITCH + M];
        });
        auto src2 = [=](int f, int i) -> cx<T> { return work[f * PITCH + i]; };
        auto dst2 = [=](int f, int j, cx<T> v) { rows[f * P + perm_out[j]] = cconj(v); };
        wg_fft<T, S, F, MAP_EF, MAP_EF, false, true>(ex, lds, p.tw, elem_src(src2), dst2);
        ex.barrier();
        ex.for_threads([&](int tid, cx<T>*) {
            for (int t = tid; t < valid; t += NT) {
                cx<T> y = rows[t];
                y.im *= sgn;
                out[fft0 * P + t] = y;
            }
        });
    }
}

// ---- Rader, many rows per workgroup (MODE 2) ---------------------------------------------------------------
// The MODE 1 flow, but one workgroup pushes ROWS rows through the same LDS buffer one after another and keeps every
// per-thread table in registers across rows: the sub-pass twiddles, the d[] factors and g^-j targets of the thread's
// last-pass outputs, the scatter slots of the elements it loads, and the NEXT row's elements (prefetched while the
// current row is transformed).  Per row only x is read and X written; all index arithmetic is loop-invariant.
template <class T> MI_HD cx<T> idx_to_reg(int i) {
    cx<T> r{0, 0};
    if constexpr (sizeof(T) == 4)
        r.re = __builtin_bit_cast(T, i);
    else
        r.re = __builtin_bit_cast(T, (long long)i);
    return r;
}
template <class T> MI_HD int reg_to_idx(const cx<T>& r) {
    if constexpr (sizeof(T) == 4)
        return __builtin_bit_cast(int, r.re);
    else
        return (int)__builtin_bit_cast(long long, r.re);
}
// HO (MODE 6): the register hand-over of rader_body MODE 5 inside the rows loop -- the second transform runs the reversed schedule S2
// with ITS sub-pass factors in a second register block (TW1), the d[] factors already sit in the registers of the first
// transform's last-pass outputs (D0), and the spectrum never goes through LDS.  A tuning variant so far (256-VGPR budget).
template <class S, bool HO = false> struct RaderRows {
    using S2 = typename reversed_sched<S>::type;
    static constexpr int M = S::N, P = S::N + 1, NT = S::TPF, EM = S::emax();
    static constexpr int NL = (P + NT - 1) / NT;  // elements of a row each thread loads / stores
    static constexpr int TW0 = EM, TW1 = TW0 + twreg_count<S>(), D0 = TW1 + (HO ? twreg_count<S2>() : 0), PO0 = D0 + EM, PI0 = PO0 + EM,
                         XN0 = PI0 + NL, NREG = XN0 + NL;
    // first slot past the exchange span (of both schedules with HO) AND past the natural-order outputs (indices <= M): x[0], then
    // X[0]; XS + 1: dump slot
    static constexpr int SPAN = (HO && S2::phys(M - 1) > S::phys(M - 1)) ? S2::phys(M - 1) + 1 : S::phys(M - 1) + 1;
    static constexpr int XS = (SPAN > P) ? SPAN : P;
    // LDS slots of the one row buffer: the exchange span, the natural-order outputs and the two spare slots
    static constexpr int PITCH = (HO && S2::pitch() > S::pitch()) ? S2::pitch() : S::pitch();
    static constexpr int SLOTS = (PITCH > XS + 2) ? PITCH : XS + 2;
};
// source of the reversed schedule inside the rows loop: d[] from the thread's registers, x[0] / X[0] through the spare slot
template <class T, class S2, int D0> struct RaderRowsRegSrc {
    static constexpr bool kLoadsAll = true;
    cx<T>* spare;
    template <class SS> MI_HD void load_all(int, int u, cx<T>* v) const {
        static_assert(std::is_same<SS, S2>::value, "source of the reversed schedule");
        constexpr int R = S2::R[0], NB = S2::nb(0), BPT = S2::bpt(0);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = u + m * S2::TPF;
            if ((m + 1) * S2::TPF <= NB || b < NB) {
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    const cx<T> val = v[m * R + k];
                    cx<T> t = cconj(val * v[D0 + m * R + k]);
                    if constexpr (m == 0 && k == 0) {
                        if (u == 0) {
                            const cx<T> x0 = spare[0];
                            t = t + cconj(x0);
                            spare[0] = x0 + val;  // X[0]
                        }
                    }
                    v[m * R + k] = t;
                });
            }
        });
    }
};
template <class T, class S, int ROWS, bool PREFETCH, bool HO = false, bool NTL = false, class X>
MI_HD void rader_rows_body(X& ex, const RaderParams<T>& p, long long block, void* lds) {
    using L = RaderRows<S, HO>;
    using S2 = typename L::S2;
    constexpr int P = L::P, NT = L::NT, NL = L::NL, XS = L::XS;
    static_assert(XS + 1 < L::SLOTS && P <= L::SLOTS, "the row buffer holds two spare slots");
    const cx<T>* in = p.in;
    cx<T>* out = p.out;
    const T sgn = p.sgn;
    cx<T>* work = (cx<T>*)lds;
    const long long row0 = block * ROWS;
    const long long row_end = (row0 + ROWS < p.batch) ? row0 + ROWS : p.batch;
    ex.for_threads([&](int tid, cx<T>* v) {
        preload_twiddles<T, S, L::TW0>(v, tid, p.tw);
        if constexpr (HO) preload_twiddles<T, S2, L::TW1>(v, tid, p.tw2);
        constexpr int LP = S::NP - 1, R = S::R[LP], NB = S::nb(LP), ST = S::stride(LP), BPT = S::bpt(LP);
        static_for<0, BPT>([&](auto M_) {
            constexpr int m = M_;
            const int b = tid + m * S::TPF;
            const int bb = ((m + 1) * S::TPF <= NB || b < NB) ? b : 0;
            const int base = (bb / ST) * (ST * R) + (bb % ST);
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                v[L::D0 + m * R + k] = p.d[base + k * ST];  // factor of the first transform's last-pass output in slot m R + k
                if constexpr (!HO) v[L::PO0 + m * R + k] = idx_to_reg<T>(p.perm_out[base + k * ST]);
            });
        });
        if constexpr (HO) {  // the g^-j targets belong to the LAST pass of the second transform: the reversed schedule's
            constexpr int LP2 = S2::NP - 1, R2 = S2::R[LP2], NB2 = S2::nb(LP2), ST2 = S2::stride(LP2), BPT2 = S2::bpt(LP2);
            static_for<0, BPT2>([&](auto M_) {
                constexpr int m = M_;
                const int b = tid + m * S2::TPF;
                const int bb = ((m + 1) * S2::TPF <= NB2 || b < NB2) ? b : 0;
                const int base = (bb / ST2) * (ST2 * R2) + (bb % ST2);
                static_for<0, R2>([&](auto K_) {
                    constexpr int k = K_;
                    v[L::PO0 + m * R2 + k] = idx_to_reg<T>(p.perm_out[base + k * ST2]);
                });
            });
        }
        static_for<0, NL>([&](auto I_) {
            constexpr int i = I_;
            const int t = tid + i * NT;
            v[L::PI0 + i] = idx_to_reg<T>(t == 0 ? XS : (t < P ? p.perm_in[t] : XS + 1));
            if constexpr (PREFETCH) v[L::XN0 + i] = (t < P && row0 < row_end) ? in[row0 * P + t] : cx<T>{0, 0};
        });
    });
    auto src = [=](int, int j) -> cx<T> { return work[j]; };
    auto dst1 = [=](int, int j, cx<T> val, auto I_, cx<T>* v) {
        cx<T> t = cconj(val * v[L::D0 + decltype(I_)::value]);
        if (j == 0) {
            const cx<T> x0 = work[XS];
            t = t + cconj(x0);
            work[XS] = x0 + val;  // X[0]
        }
        work[j] = t;
    };
    auto dst2 = [=](int, int j, cx<T> val, auto I_, cx<T>* v) {
        if (j == 0) work[0] = work[XS];  // slot 0 is no target of the g^-j scatter
        work[reg_to_idx<T>(v[L::PO0 + decltype(I_)::value])] = cconj(val);
    };
    for (long long row = row0; row < row_end; ++row) {
        ex.relaunder();
        ex.for_threads([&](int tid, cx<T>* v) {
            static_for<0, NL>([&](auto I_) {
                constexpr int i = I_;
                cx<T> x;
                if constexpr (PREFETCH) {
                    x = v[L::XN0 + i];
                } else {
                    const int t = tid + i * NT;
#if defined(MI355_NT_PROBE)  // probe build (make tuning-min MINEXTRA=-DMI355_NT_PROBE): EVERY rows loop reads its rows with non-temporal loads
                    constexpr bool kNtRows = true;
#else
                    constexpr bool kNtRows = NTL;
#endif
                    if constexpr (kNtRows)
                        x = ((i + 1) * NT <= P || t < P) ? ld_nt(in + (row * P + t)) : cx<T>{0, 0};
                    else
                        x = ((i + 1) * NT <= P || t < P) ? in[row * P + t] : cx<T>{0, 0};
                }
                x.im *= sgn;
                work[reg_to_idx<T>(v[L::PI0 + i])] = x;
            });
            if (PREFETCH && row + 1 < row_end) {
                static_for<0, NL>([&](auto I_) {
                    constexpr int i = I_;
                    const int t = tid + i * NT;
                    if ((i + 1) * NT <= P || t < P) v[L::XN0 + i] = in[(row + 1) * P + t];
                });
            }
        });
        ex.barrier();
        if constexpr (HO) {
            static_assert(S2::R[0] == S::R[S::NP - 1] && S2::nb(0) == S::nb(S::NP - 1) && S2::bpt(0) == S::bpt(S::NP - 1) && S2::TPF == S::TPF,
                          "register hand-over");
            wg_fft<T, S, 1, MAP_EF, MAP_EF, false, true, 1, 0, L::TW0>(ex, lds, p.tw, elem_src(src), KeepInRegs{});
            wg_fft<T, S2, 1, MAP_EF, MAP_EF, false, false, 1, 0, L::TW1>(ex, lds, p.tw2, RaderRowsRegSrc<T, S2, L::D0>{work + XS}, slot_dst(dst2));
        } else {
            wg_fft<T, S, 1, MAP_EF, MAP_EF, false, true, 1, 0, L::TW0>(ex, lds, p.tw, elem_src(src), slot_dst(dst1));
            ex.barrier();
            wg_fft<T, S, 1, MAP_EF, MAP_EF, false, true, 1, 0, L::TW0>(ex, lds, p.tw, elem_src(src), slot_dst(dst2));
        }
        ex.barrier();
        ex.for_threads([&](int tid, cx<T>*) {
            static_for<0, NL>([&](auto I_) {
                constexpr int i = I_;
                const int t = tid + i * NT;
                if ((i + 1) * NT <= P || t < P) {
                    cx<T> y = work[t];
                    y.im *= sgn;
                    out[row * P + t] = y;
                }
            });
        });
        ex.barrier();
    }
}

}  // namespace mi355
