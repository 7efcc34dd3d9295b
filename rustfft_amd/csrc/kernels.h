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

template <class T, class S, int F, bool SPLIT, class X>
MI_HD void k1_body(X& ex, const K1Params<T>& p, long long block, void* lds) {
    const long long fft0 = block * F;
    const cx<T>* MI_RESTRICT in = p.in;
    cx<T>* MI_RESTRICT out = p.out;
    const long long batch = p.batch;
    const T sgn = p.sgn;
    auto src = [=](int f, int i) -> cx<T> {
        const long long g = fft0 + f;
        if (g < batch) {
            cx<T> x = in[g * S::N + i];
            x.im *= sgn;
            return x;
        }
        return cx<T>{0, 0};
    };
    auto dst = [=](int f, int i, cx<T> x) {
        const long long g = fft0 + f;
        if (g < batch) {
            x.im *= sgn;
            out[g * S::N + i] = x;
        }
    };
    wg_fft<T, S, F, MAP_EF, MAP_EF, SPLIT>(ex, lds, p.tw, src, dst);
}

template <class T, class S, int F, bool FIRST, bool SPLIT, class X>
MI_HD void k2_body(X& ex, const K2Params<T>& p, long long block, void* lds) {
    constexpr int R = S::N;
    const long long g = block / p.tiles_per_fft;
    const long long tile = block % p.tiles_per_fft;
    const long long b0 = tile * F;
    const cx<T>* MI_RESTRICT in = p.in + g * p.n;
    cx<T>* MI_RESTRICT out = p.out + g * p.n;
    const long long M = p.m, Sg = p.s;
    const T sgn_in = p.sgn_in, sgn_out = p.sgn_out;
    const cx<T>* MI_RESTRICT tlo = p.tlo;
    const cx<T>* MI_RESTRICT thi = p.thi;
    const int hshift = p.hshift, lmask = p.lmask;
    // a tile never straddles a multiple of S (F | S whenever S > 1), so B div S is tile-uniform
    const long long bdiv = FIRST ? 0 : (b0 / Sg);
    const long long bmod0 = FIRST ? 0 : (b0 % Sg);
    auto src = [=](int f, int j) -> cx<T> {
        cx<T> x = in[b0 + f + (long long)j * M];
        x.im *= sgn_in;
        if constexpr (!FIRST) {
            const unsigned e = (unsigned)(bmod0 + f) * (unsigned)j;
            const cx<T> w = tlo[e & lmask] * thi[e >> hshift];
            x = x * w;
        }
        return x;
    };
    auto dst = [=](int f, int k, cx<T> x) {
        x.im *= sgn_out;
        if constexpr (FIRST)
            out[(b0 + f) * R + k] = x;
        else
            out[bdiv * Sg * R + bmod0 + f + (long long)k * Sg] = x;
    };
    // first pass: lanes walk across the tile's columns on the way in and along each sequence on the way
    // out (the F*R output block is contiguous); later passes: across columns both ways
    wg_fft<T, S, F, MAP_FF, FIRST ? MAP_EF : MAP_FF, SPLIT>(ex, lds, p.tw, src, dst);
}

// ---- Bluestein: any length n <= (M + 1) / 2 through two length-M workgroup transforms ---------------------
template <class T, class S, int F, class X>
MI_HD void bluestein_body(X& ex, const BluesteinParams<T>& p, long long block, void* lds) {
    constexpr int M = S::N, PITCH = S::pitch();
    const long long fft0 = block * F;
    const cx<T>* MI_RESTRICT in = p.in;
    cx<T>* MI_RESTRICT out = p.out;
    const cx<T>* MI_RESTRICT chirp = p.chirp;
    const cx<T>* MI_RESTRICT bf = p.bf;
    const long long batch = p.batch;
    const int n = p.n;
    const T sgn = p.sgn;
    cx<T>* work = (cx<T>*)lds;  // natural-order spectrum; reuses the exchange buffer once the first transform is done
    auto src1 = [=](int f, int i) -> cx<T> {
        const long long g = fft0 + f;
        if (g < batch && i < n) {
            cx<T> x = in[g * n + i];
            x.im *= sgn;
            return x * chirp[i];
        }
        return cx<T>{0, 0};
    };
    auto dst1 = [=](int f, int j, cx<T> v) { work[f * PITCH + j] = cconj(v * bf[j]); };
    wg_fft<T, S, F, MAP_EF, MAP_EF, false>(ex, lds, p.tw, src1, dst1);
    ex.barrier();
    auto src2 = [=](int f, int i) -> cx<T> { return work[f * PITCH + i]; };
    auto dst2 = [=](int f, int j, cx<T> v) {
        const long long g = fft0 + f;
        if (g < batch && j < n) {
            cx<T> y = cconj(v) * chirp[j];
            y.im *= sgn;
            out[g * n + j] = y;
        }
    };
    wg_fft<T, S, F, MAP_EF, MAP_EF, false, true>(ex, lds, p.tw, src2, dst2);
    (void)M;
}

// ---- Rader: prime length p = S::N + 1 ---------------------------------------------------------------------
// LDS: [F][PITCH] exchange/work buffer followed by [F][p] staging of the rows (the g^j permutations are
// random within a row, so they are applied against LDS, never against HBM).
template <class T, class S, int F, class X>
MI_HD void rader_body(X& ex, const RaderParams<T>& p, long long block, void* lds) {
    constexpr int M = S::N, P = S::N + 1, PITCH = S::pitch(), NT = F * S::TPF;
    const long long fft0 = block * F;
    const cx<T>* MI_RESTRICT in = p.in;
    cx<T>* MI_RESTRICT out = p.out;
    const cx<T>* MI_RESTRICT dtab = p.d;
    const int* MI_RESTRICT perm_in = p.perm_in;
    const int* MI_RESTRICT perm_out = p.perm_out;
    const long long batch = p.batch;
    const T sgn = p.sgn;
    cx<T>* work = (cx<T>*)lds;
    cx<T>* rows = work + F * PITCH;
    const long long rows_here = (batch - fft0) < F ? (batch - fft0) : F;
    const int valid = (int)(rows_here * P);
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
    wg_fft<T, S, F, MAP_EF, MAP_EF, false>(ex, lds, p.tw, src1, dst1);
    ex.barrier();
    // X[0] must leave `work` before the second transform's exchanges reuse the buffer
    ex.for_threads([&](int tid, cx<T>*) {
        if (tid < F) rows[tid * P] = work[tid * PITCH + M];
    });
    auto src2 = [=](int f, int i) -> cx<T> { return work[f * PITCH + i]; };
    auto dst2 = [=](int f, int j, cx<T> v) { rows[f * P + perm_out[j]] = cconj(v); };
    wg_fft<T, S, F, MAP_EF, MAP_EF, false, true>(ex, lds, p.tw, src2, dst2);
    ex.barrier();
    ex.for_threads([&](int tid, cx<T>*) {
        for (int t = tid; t < valid; t += NT) {
            cx<T> y = rows[t];
            y.im *= sgn;
            out[fft0 * P + t] = y;
        }
    });
}

}  // namespace mi355
