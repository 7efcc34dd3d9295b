// Launch-parameter blocks of the kernels (plain data; shared by the kernels and the host planner).
#pragma once
#include "cx.h"
#include "dyn_engine.h"

namespace mi355 {

template <class T> struct K1Params {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;  // sub-pass twiddles (Sched::tw_total() entries)
    long long batch;  // number of sequences
    T sgn;            // +1 forward, -1 inverse (applied on load and on store)
};

template <class T> struct K2Params {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;   // sub-pass twiddles of the length-R workgroup transform
    const cx<T>* tlo;  // w_{S R}^e for e in [0, 2^h)
    const cx<T>* thi;  // w_{S R}^(e << h)
    int hshift;
    int lmask;
    long long n;              // full transform length N
    long long m;              // M = N / R
    long long s;              // S = product of earlier macro radices (1 for the first pass)
    long long batch;          // number of length-N transforms
    long long tiles_per_fft;  // M / F
    T sgn_in, sgn_out;
    int tiles_shift, s_shift;  // log2(tiles_per_fft), log2(s): the power-of-two passes (k2_body) index with shifts
    int xp, xq;  // XCD-aware tile order (k2_body): groups of 8 << xq workgroups, XCD id placed at tile-index bits [xp, xp+3); xq = 0: identity
    int dbg;   // measurement knobs (bit 0: skip the inter-pass twiddles); 0 in production
    int xfull;  // k2g_body: number of complete 64-workgroup groups of the grid (the permuted ones)
    // fused multi-kernel Bluestein (k2g_body FUSE != 0): the element-wise stages of bluesteins_algorithm.rs:100-136 ride
    // on the first load / last store of the two inner transforms
    const cx<T>* tab;   // FUSE 1, 3: chirp[n_valid]; FUSE 2: spectrum multiplier bf[N]
    long long n_io;     // FUSE 1: row pitch of `in`; FUSE 3: row pitch of `out` (the caller's length n, not N)
    unsigned n_valid;   // FUSE 1: elements >= n_valid of a row are zero padding; FUSE 3: only elements < n_valid are stored
    // fused multi-kernel Rader (k2g_body FUSE 4, 5, 6; raders_algorithm.rs:235-283 with inner length N = p - 1, caller rows of pitch
    // n_io = p): FUSE 4 loads x[perm[idx]] (perm = g^(j+1) mod p); FUSE 5 stores conj(X * tab) (tab = d), adds conj(x[0]) to
    // element 0 and writes X[0] = x[0] + S[0] into the caller's output row; FUSE 6 stores conj(X) at perm[e] (perm = g^-(j+1) mod p)
    const int* perm;
    const int* perm2;  // k2r_body (prime tile heights): g^-(j+1) mod P; `perm` holds the inverse map t -> j with g^(j+1) = t, `tab` d[P-1]
    const cx<T>* xin;  // the caller's input rows (x[0] of every row)
    cx<T>* xout;       // the caller's output rows (X[0] of every row)
    T sgn_x;           // -1 for the inverse plan (conj on the way in and out), +1 otherwise: every pass of the sequence sees it
};

// Fused two-pass kernel (kernels.h k2f_decode, launch.h k2f_kernel): ONE launch runs both column-tile passes of every
// transform of the batch; the intermediate lives in a ring of `ns` transform-sized slots that stays in the Infinity Cache.
// Work item w (= workgroup index, or a ticket): step s = w / (t0 + t1) holds the t0 first-pass tiles of transform s and the
// t1 second-pass tiles of transform s - lag.  Dependencies point to earlier steps only:
//   second-pass tile of transform g : all t0 first-pass tiles of g have been written    (written[g % ns] >= (g / ns + 1) t0)
//   first-pass tile of transform g  : all t1 second-pass tiles of g - ns have been read  (read[g % ns]    >= (g / ns) t1)
// ctrl (device, zeroed before every launch): [0] ticket counter, [1] error word (a bounded wait gave up), then per slot two
// counters on their own 128-byte lines: written at ctrl[32 + 64 slot], read at ctrl[64 + 64 slot].
template <class T> struct K2FusedParams {
    K2Params<T> pass[2];   // pass[0].in / pass[1].out: the caller's rows; pass[0].out == pass[1].in: the ring base
    unsigned* ctrl;
    int tiles[2];          // t0, t1: tiles per transform of the two passes
    int lag, ns;
    long long batch;       // STEPS of the launch: transforms x units
    int units;             // U >= 1: closed groups of tiles per transform (1 for a two-pass plan; three-pass plans: see below)
    long long slot_elems;  // elements of a ring slot (n / U)
    int mode;              // bit 0: dependency flags + fences (0 = timing probe only, results undefined); bit 1: work items by ticket
    int spin_limit;        // polls before a wait gives up and sets the error word
    unsigned* err;         // STICKY error word of the (plan, stream) slot: pinned host memory, never cleared by a launch (plan.cpp fused_check)
};
// Passes 0 and 1 of a THREE-pass plan N = R0 R1 R2 through the same kernel.  The tiles of the two passes close into U = R2 / F0 UNITS per
// transform: unit u is the R1 first-pass tiles with columns [u F0 + j R2, u F0 + j R2 + F0) (j < R1) and the F0 R0 / F1 second-pass tiles
// with B div R0 in [u F0, (u + 1) F0) -- the former write exactly what the latter read, F0 R0 R1 elements.  A step of the launch is a unit,
// a ring slot holds one unit in COMPACT form (first-pass tile j writes at column base j F0; the second pass reads it as an
// (R1) x (F0 R0) matrix: pass[1].m = F0 R0), and the tile index splits in two: where a tile reads (`tile`) and where it writes
// (`tile_out`) -- pass 0: u + j U / j, pass 1: i / u t1 + i.  U = 1 is the two-pass plan (both indices equal).
constexpr int k2f_ctrl_words(int ns) { return 32 + 64 * ns + 64; }
// workgroups of a fused launch: batch + lag steps of t0 + t1 items (the items of a step that have no transform do nothing)
constexpr long long k2f_grid(long long batch, int t0, int t1, int lag) { return (batch + lag) * (long long)(t0 + t1); }

// Bluestein (chirp-z) in one workgroup, src/algorithm/bluesteins_algorithm.rs:100-136:
//   a[i] = x[i] * chirp[i] (zero padded to M);  A = FFT_M(a);  A[j] = conj(A[j] * bf[j]);
//   A = FFT_M(A);  X[i] = conj(A[i]) * chirp[i]
template <class T> struct BluesteinParams {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;     // sub-pass twiddles of the length-M transform
    const cx<T>* tw2;    // the same for the REVERSED schedule (one-kernel Bluestein: second transform)
    const cx<T>* chirp;  // n entries, w[i] = twiddle(i^2 mod 2n, 2n), forward (src/twiddles.rs:25-57)
    const cx<T>* bf;     // M entries, FFT_M of the mirrored conjugate chirp / M (bluesteins_algorithm.rs:63-87)
    long long batch;
    int n;
    T sgn;
};

// Rader in one workgroup, src/algorithm/raders_algorithm.rs:235-283 (p prime, inner length p - 1):
//   s[j] = x[g^(j+1) mod p];  S = FFT(s);  X[0] = x[0] + S[0];  S[j] = conj(S[j] * d[j]);  S[0] += conj(x[0]);
//   S = FFT(S);  X[g^-(j+1) mod p] = conj(S[j])
template <class T> struct RaderParams {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;       // sub-pass twiddles of the length-(p-1) transform
    const cx<T>* d;        // p-1 entries: FFT of twiddle(g^-j mod p, p) / (p-1)  (raders_algorithm.rs:87-113)
    const int* perm_in;    // g^(j+1) mod p
    const int* perm_out;   // g^-(j+1) mod p
    long long batch;
    int p;
    T sgn;
    const cx<T>* tw2;      // MODE 5: sub-pass twiddles of the REVERSED schedule (the second transform)
};

// Run-time scheduled variants (dyn_engine.h): any 13-smooth length that fits one workgroup, and Rader for any
// prime p whose p - 1 is 13-smooth.
template <class T> struct DynK1Params {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;
    long long batch;
    T sgn;
    DynSched s;
};
template <class T> struct DynRaderParams {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;
    const cx<T>* d;
    const int* perm_in;
    const int* perm_out;
    long long batch;
    T sgn;
    DynSched s;  // schedule of the inner length p - 1
};

// Element-wise stages of the multi-kernel Bluestein used for lengths that do not fit one workgroup
// (same algebra as BluesteinParams; the two length-M transforms are ordinary power-of-two plans):
//   stage 0: w[r][i] = (i < n) ? x[r][i] * chirp[i] : 0          (zero-pad to M)
//   stage 1: w[r][j] = conj(w[r][j] * bf[j])
//   stage 2: y[r][i] = conj(w[r][i]) * chirp[i],  i < n
template <class T> struct PointwiseParams {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tab;   // chirp (stages 0, 2) or bf (stage 1)
    long long rows;
    long long n;        // outer length
    long long m;        // inner (padded) length
    int stage;
    T sgn;              // conj-in (stage 0) / conj-out (stage 2) for the inverse direction
};

}  // namespace mi355
