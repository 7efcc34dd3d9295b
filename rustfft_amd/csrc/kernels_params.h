// Launch-parameter blocks of the kernels (plain data; shared by the kernels and the host planner).
#pragma once
#include "cx.h"

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
};

}  // namespace mi355
