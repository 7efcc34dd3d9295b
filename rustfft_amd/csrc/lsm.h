// LDS stage machine ("lsm"): ONE compiled kernel per precision and block size that executes a run-time PROGRAM of in-place
// butterfly stages on rows that live in LDS.  It is the GPU form of what the reference's planner builds for lengths with
// large prime factors -- MixedRadix over Rader inner FFTs, Rader over MixedRadix inner FFTs (src/plan.rs:412-425, 474-506,
// 636-665; src/algorithm/mixed_radix.rs:128-158; src/algorithm/raders_algorithm.rs:65-283) -- with the factor list, the tree
// shape and every permutation decided at PLAN time on the host (lsm_plan.h), the way RadixN takes its factor list at run time
// (src/algorithm/radixn.rs:54-155).
//
// Execution model.  A workgroup holds F rows of the transform in LDS (row f at slots f RP ..).  The program is
//     LOAD   : coalesced global read, every element scattered to the LDS slot the host computed for it (one table composes
//              every input permutation of the tree: digit reversals, Rader's g^j gathers, the six-step transposes);
//     stages : each one runs independent radix-R butterflies IN PLACE (read R slots, compute, write the same R slots), so ONE
//              barrier separates two stages (the Stockham engine of engine.h needs two per exchange).  A butterfly (work item)
//              reads the slots  base + k * astep;  its base and its twiddle row come from a per-stage DESCRIPTOR table the host
//              wrote -- one 32-bit word per work item of the workgroup, fetched a stage ahead -- so the kernel does no index
//              arithmetic beyond one add per element, whatever the tree looks like (a sub-pass of a strided FFT batched over the
//              other axes of the tree: decimation in time = twiddle before the butterfly; the transposed flow graph, decimation
//              in frequency = twiddle after it);
//     FIX    : Rader's x[0] / X[0] step for every Rader instance (raders_algorithm.rs:256-262, restated without conjugates);
//     BFLY2  : the LAST stage of Rader's first inner transform and the FIRST stage of its second one in one stage -- they work on the same
//              R slots (a decimation-in-frequency first stage is the transpose of the decimation-in-time last stage), so the spectrum
//              multiply and the x[0] step happen in registers between two butterflies and a Rader costs 2 P - 1 stages instead of 2 P + 1
//              (the register hand-over of the compiled Rader / Bluestein bodies, kernels.h rader_body MODE 5);
//     STORE  : coalesced global write, every element gathered from the slot the host computed.
// Optional per-stage pre-multiplier (a second descriptor word per item indexes its table): Rader's spectrum multiply and the
// six-step twiddles ride on the first stage that follows them.  Stage tables are copied to LDS at kernel start (the big six-step
// table may stay in global memory); stage headers are read through scalar loads (the stage index is workgroup-uniform).
// Bank conflicts are the host's business: it orders the work items of a stage and picks odd pitches.
#pragma once
#include "butterflies.h"

namespace mi355 {

constexpr int kLsmMaxStages = 48;
constexpr int kLsmItems = 4;  // work items per thread and stage, at most
constexpr int kLsmEmax = 16;  // complex values a thread holds in a stage of radix 2 .. 16
constexpr int kLsmIoMax = 64; // elements of the workgroup's rows a thread moves in LOAD / STORE, at most (batches of eight)
enum LsmOp { LSM_BFLY = 1, LSM_FIX = 2, LSM_BFLY2 = 3, LSM_X0MUL = 4 };
enum LsmFlag {
    LSM_TW_PRE = 1,      // multiply input k by w^(row k) before the butterfly (decimation in time)
    LSM_TW_POST = 2,     // multiply output k by w^(row k) after it (decimation in frequency)
    LSM_PRE_MUL = 4,     // multiply every input by a table entry first
    LSM_PRE_GLOBAL = 8,  // ... whose table lives in global memory (LsmParams::gtab) instead of LDS
    LSM_PRE2 = 16        // BFLY2: a second table multiplies the INPUTS (a six-step twiddle in front of Rader's middle stage): index = the spectrum
                         // multiplier's + p2_delta, in LDS or (LSM_PRE_GLOBAL) in global memory; the spectrum multiplier itself is always in LDS
};

struct LsmStage {
    int op, radix, flags;
    int total;      // work items of the WORKGROUP (all F rows)
    int astep;      // slot of input k = base + k astep              (FIX: base = S[0], base + astep = x[0])
    int tw_kstep;   // factor of input k >= 1: ltab[row + (k - 1) tw_kstep]
    int p_kstep;    // pre-multiplier of input k: tab[pidx + k p_kstep]     (FIX: of x[0]: tab[pidx])
    int desc_off;   // first descriptor word of the stage: base | row << 16
    int pdesc_off;  // first pre-multiplier index of the stage (stages with LSM_PRE_MUL)
    float cfix;     // FIX / BFLY2: S[0] += cfix * x[0]   (= -(q - 1): 1 / D[0], lsm_plan.h)
    int fix_off;    // BFLY2: slot of x[0] relative to the base of the item that holds S[0] (bit 31 of its pre-multiplier word marks that item)
    int p2_delta;   // BFLY2 with LSM_PRE2: index of the input table = index of the spectrum multiplier + p2_delta
    int round;      // work items the workgroup takes per ROUND: NT x the items a thread holds at this radix; a stage with more runs several rounds
};

template <class T> struct LsmParams {
    const cx<T>* in;
    cx<T>* out;
    const LsmStage* stages;
    const unsigned* desc;          // work-item descriptors of every stage
    const cx<T>* ltab;             // stage tables, copied to LDS behind the rows
    const cx<T>* gtab;             // stage tables read from global memory
    const unsigned short* ldperm;  // [f n]: LDS slot of input element i of the workgroup
    const unsigned short* stperm;  // [f n]: LDS slot of output element k of the workgroup
    long long batch;
    int nstages, ltab_n;
    int n, f;                      // row length, rows per workgroup
    int tab_off;                   // first table element in LDS (elements)
    int nt, lds_bytes;             // block size and dynamic LDS bytes of the launch
    T sgn;
};

// one radix-R stage for the work items of one thread: item = tid + j NT, descriptor words in dw[] (pre-multiplier indices in dp[])
// Register budget (the kernel should hold four waves per SIMD, i.e. 128 VGPRs): the data (2 EMAX), the twiddles of every item (fetched
// with the data: < 2 EMAX) and one word per item.
template <int R, int NT, class T>
MI_HD void lsm_bfly(const LsmStage& st, const LsmParams<T>& p, int tid, cx<T>* lds, const cx<T>* ltab, const unsigned* dw, const unsigned* dp, int left) {
    constexpr int IMAX = (kLsmEmax / R) < kLsmItems ? (kLsmEmax / R) : kLsmItems;
    static_assert(IMAX >= 1, "radix exceeds the register budget");
    cx<T> v[IMAX * R];  // the data of the thread's items: live within the stage only (nothing but the descriptor words crosses a barrier in registers)
    const int total = left, flags = st.flags;  // (items of this round and beyond: the guards below stop at the thread's IMAX)
    const bool has_tw = (flags & (LSM_TW_PRE | LSM_TW_POST)) != 0;
    // Factors: radices up to 8 fetch them WITH the data (one LDS round trip per stage); the larger ones -- one item per thread, 2 R data
    // registers and a butterfly that needs as many temporaries -- fetch them right where they are used, so that they are not live across
    // the butterfly (radix 16: 168 -> under 128 VGPRs; the extra LDS round trip hides behind the other waves of the SIMD)
    constexpr bool AHEAD = R <= 8;
    cx<T> tw[IMAX * R];  // pre-multipliers, then twiddles (slot j R + k)
    auto load_pre = [&](int j0, int pi) {
        if (flags & LSM_PRE_GLOBAL) {
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                tw[j0 + k] = p.gtab[(unsigned)(pi + k * st.p_kstep)];
            });
        } else {
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                tw[j0 + k] = ltab[pi + k * st.p_kstep];
            });
        }
    };
    auto load_tw = [&](int j0, int tr) {
        static_for<1, R>([&](auto K_) {
            constexpr int k = K_;
            tw[j0 + k] = ltab[tr + (k - 1) * st.tw_kstep];
        });
    };
    // phase 1: issue every LDS read of the thread (data, then the pre-multipliers or the twiddles)
    static_for<0, IMAX>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < total && tid + j * NT < total) {
            const int base = (int)(dw[j] & 0xffffu), tr = (int)(dw[j] >> 16);
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                v[j * R + k] = lds[base + k * st.astep];
            });
            if (flags & LSM_PRE_MUL) {
                load_pre(j * R, (int)dp[j]);  // (the six-step table may be global memory: always requested with the data)
            } else if (AHEAD && has_tw) {
                load_tw(j * R, tr);
            }
        }
    });
    // phase 2: arithmetic, write back in place
    static_for<0, IMAX>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < total && tid + j * NT < total) {
            const int base = (int)(dw[j] & 0xffffu), tr = (int)(dw[j] >> 16);
            cx<T>* x = v + j * R;
            if (flags & LSM_PRE_MUL) {
                static_for<0, R>([&](auto K_) {
                    constexpr int k = K_;
                    x[k] = x[k] * tw[j * R + k];
                });
                if (AHEAD && has_tw) load_tw(j * R, tr);
            }
            if (flags & LSM_TW_PRE) {
                if (!AHEAD) load_tw(j * R, tr);
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    x[k] = x[k] * tw[j * R + k];
                });
            }
            butterfly<R>(x);
            if (flags & LSM_TW_POST) {
                if (!AHEAD) load_tw(j * R, tr);
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    x[k] = x[k] * tw[j * R + k];
                });
            }
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                lds[base + k * st.astep] = x[k];
            });
        }
        if constexpr (IMAX > 1) MI_SCHED_FENCE();
    });
}

// LSM_BFLY2: [twiddle] butterfly, x[0] step, spectrum multiply, butterfly [the same twiddle] on the same R slots (see the header).  The two
// twiddle multiplies use ONE set of factors: the decimation-in-time last stage and the decimation-in-frequency first stage of a length share
// stride and radix, hence the table.
template <int R, int NT, class T>
MI_HD void lsm_bfly2(const LsmStage& st, const LsmParams<T>& p, int tid, cx<T>* lds, const cx<T>* ltab, const unsigned* dw, const unsigned* dp, int left) {
    constexpr int IMAX = (kLsmEmax / R) < kLsmItems ? (kLsmEmax / R) : kLsmItems;
    const int total = left, flags = st.flags;
    const bool has_tw = (flags & (LSM_TW_PRE | LSM_TW_POST)) != 0;
    constexpr bool AHEAD = R <= 8;
    cx<T> v[IMAX * R], tw[IMAX * R];
    auto load_tw = [&](int j0, int tr) {
        static_for<1, R>([&](auto K_) {
            constexpr int k = K_;
            tw[j0 + k] = ltab[tr + (k - 1) * st.tw_kstep];
        });
    };
    static_for<0, IMAX>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < total && tid + j * NT < total) {
            const int base = (int)(dw[j] & 0xffffu), tr = (int)(dw[j] >> 16);
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                v[j * R + k] = lds[base + k * st.astep];
            });
            if (AHEAD && has_tw) load_tw(j * R, tr);
        }
    });
    static_for<0, IMAX>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < total && tid + j * NT < total) {
            const int base = (int)(dw[j] & 0xffffu), tr = (int)(dw[j] >> 16), pi = (int)(dp[j] & 0x7fffffffu);
            cx<T>* x = v + j * R;
            if (flags & LSM_PRE2) {
                if (flags & LSM_PRE_GLOBAL) {
                    static_for<0, R>([&](auto K_) {
                        constexpr int k = K_;
                        x[k] = x[k] * p.gtab[(unsigned)(pi + st.p2_delta + k * st.p_kstep)];
                    });
                } else {
                    static_for<0, R>([&](auto K_) {
                        constexpr int k = K_;
                        x[k] = x[k] * ltab[pi + st.p2_delta + k * st.p_kstep];
                    });
                }
            }
            if (has_tw) {
                if (!AHEAD) load_tw(j * R, tr);
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    x[k] = x[k] * tw[j * R + k];
                });
            }
            butterfly<R>(x);
            if (dp[j] >> 31) {  // this item holds S[0]: X[0] = x[0] + S[0]; S[0] += x[0] / D[0]   (lsm_fix)
                const cx<T> x0 = lds[base + st.fix_off];
                lds[base + st.fix_off] = x0 + x[0];
                x[0] = x[0] + x0 * (T)st.cfix;
            }
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                x[k] = x[k] * ltab[pi + k * st.p_kstep];
            });
            butterfly<R>(x);
            if (has_tw) {
                if (!AHEAD) load_tw(j * R, tr);
                static_for<1, R>([&](auto K_) {
                    constexpr int k = K_;
                    x[k] = x[k] * tw[j * R + k];
                });
            }
            static_for<0, R>([&](auto K_) {
                constexpr int k = K_;
                lds[base + k * st.astep] = x[k];
            });
        }
        if constexpr (IMAX > 1) MI_SCHED_FENCE();
    });
}
// LSM_X0MUL: a Rader node that FOLLOWS a six-step twiddle takes the factor of its x[0] here (the fused stage has one table index per item)
template <int NT, class T> MI_HD void lsm_x0mul(const LsmStage& st, const LsmParams<T>& p, int tid, cx<T>* lds, const cx<T>* ltab, const unsigned* dw, const unsigned* dp, int left) {
    static_for<0, kLsmItems>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < left && tid + j * NT < left) {
            const int px = (int)(dw[j] & 0xffffu);
            lds[px] = lds[px] * ((st.flags & LSM_PRE_GLOBAL) ? p.gtab[dp[j]] : ltab[dp[j]]);
        }
    });
}

// Rader's x[0] / X[0] step, one work item per Rader instance: with S[0] = sum of the gathered inputs (the first inner
// transform's bin 0) and D[0] = -1 / (q - 1) exactly (the sum of all non-trivial q-th roots of unity is -1),
//     X[0] = x[0] + S[0];      S[0] += x[0] / D[0]
// so that the spectrum multiply that follows turns bin 0 into S[0] D[0] + x[0], which the second inner transform spreads
// as "+ x[0]" over every output (raders_algorithm.rs:256-266 does the same with conj(x[0]) between its two conjugations).
template <int NT, class T> MI_HD void lsm_fix(const LsmStage& st, const LsmParams<T>& p, int tid, cx<T>* lds, const cx<T>* ltab, const unsigned* dw, const unsigned* dp, int left) {
    static_for<0, kLsmItems>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < left && tid + j * NT < left) {
            const int ps = (int)(dw[j] & 0xffffu), px = ps + st.astep;
            cx<T> s0 = lds[ps], x0 = lds[px];
            if (st.flags & LSM_PRE_MUL) x0 = x0 * ((st.flags & LSM_PRE_GLOBAL) ? p.gtab[dp[j]] : ltab[dp[j]]);
            lds[px] = x0 + s0;
            lds[ps] = s0 + x0 * (T)st.cfix;
        }
    });
}

#define MI_LSM_RADIX_SWITCH(RADIX, CALL)                     \
    switch (RADIX) {                                         \
        case 2: { constexpr int RR = 2; CALL; } break;       \
        case 3: { constexpr int RR = 3; CALL; } break;       \
        case 4: { constexpr int RR = 4; CALL; } break;       \
        case 5: { constexpr int RR = 5; CALL; } break;       \
        case 6: { constexpr int RR = 6; CALL; } break;       \
        case 7: { constexpr int RR = 7; CALL; } break;       \
        case 8: { constexpr int RR = 8; CALL; } break;       \
        case 9: { constexpr int RR = 9; CALL; } break;       \
        case 10: { constexpr int RR = 10; CALL; } break;     \
        case 11: { constexpr int RR = 11; CALL; } break;     \
        case 12: { constexpr int RR = 12; CALL; } break;     \
        case 13: { constexpr int RR = 13; CALL; } break;     \
        case 14: { constexpr int RR = 14; CALL; } break;     \
        case 15: { constexpr int RR = 15; CALL; } break;     \
        case 16: { constexpr int RR = 16; CALL; } break;     \
        default: break;                                      \
    }

// descriptor words of a stage for this thread: item j of the thread is item tid + j NT of the workgroup.  Only the words the stage has
// are requested (the guards are workgroup-uniform): a fetch is a global load plus its address arithmetic, and most stages have one or two
// items per thread and no pre-multiplier
template <int NT> MI_HD void lsm_fetch_desc(const LsmStage& st, const unsigned* MI_RESTRICT desc, int tid, unsigned* dw, unsigned* dp, int r0 = 0) {
    static_for<0, kLsmItems>([&](auto J_) {
        constexpr int j = J_;
        if (j * NT < st.round && r0 + j * NT < st.total) {
            const int item = r0 + tid + j * NT, ic = item < st.total ? item : 0;
            dw[j] = desc[(unsigned)(st.desc_off + ic)];
            if (st.flags & LSM_PRE_MUL) dp[j] = desc[(unsigned)(st.pdesc_off + ic)];
        }
    });
}

// X: executor with 4 kLsmItems words per thread (words(tid): the descriptor words of the current and of the next
// stage), NT threads; lsm_launch.h has the gfx950 and the host forms
template <class T, int NT, class X> MI_HD void lsm_body(X& ex, const LsmParams<T>& p, long long block, void* lds_raw) {
    constexpr int EMAX = kLsmEmax, CH = 8;
    cx<T>* lds = (cx<T>*)lds_raw;
    const cx<T>* ltab = lds + p.tab_off;
    const int F = p.f, N = p.n, total = F * N;
    const long long row0 = block * F;
    const long long left = p.batch - row0;
    const int valid = (int)((left < F ? left : F) * N);
    const cx<T>* in = p.in + row0 * N;
    cx<T>* out = p.out + row0 * N;
    const T sgn = p.sgn;
    // stage headers through SCALAR loads: the constant address space tells the compiler that nothing in this kernel writes them (as plain
    // global memory every field was a vector load + s_waitcnt + v_readfirstlane in front of the stage's first branch: a dependent L2 round
    // trip per field and stage)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const LsmStage __attribute__((address_space(4))) * StagePtr;
    StagePtr stages = (StagePtr)(unsigned long long)p.stages;
#else
    const LsmStage* stages = p.stages;
#endif
    // tables -> LDS; rows -> LDS, every element to the slot the program wants it in.  Batches of eight elements per thread: all
    // loads of a batch (element, slot) are in flight before the first LDS write.  The first stage's descriptor words ride along.
    ex.for_threads([&](int tid, cx<T>*) {
        unsigned* cur = ex.words(tid);
        const LsmStage st0 = stages[0];
        lsm_fetch_desc<NT>(st0, p.desc, tid, cur, cur + kLsmItems);
        static_for<0, kLsmIoMax / CH>([&](auto Q_) {
            constexpr int q = Q_;
            if ((q + 1) * CH * NT <= valid) {  // (workgroup-uniform) a full batch of a full workgroup: no per-element checks
                cx<T> xr[CH];
                int slot[CH];
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    const unsigned e = (unsigned)(tid + (q * CH + i) * NT);
                    xr[i] = in[e];
                    slot[i] = (int)p.ldperm[e];
                });
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    lds[slot[i]] = cx<T>{xr[i].re, xr[i].im * sgn};
                });
            } else if (q * CH * NT < total) {
                cx<T> xr[CH];
                int slot[CH];
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    const int e = tid + (q * CH + i) * NT;
                    xr[i] = in[(unsigned)(e < valid ? e : 0)];
                    slot[i] = (int)p.ldperm[e < total ? e : 0];
                });
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    const int e = tid + (q * CH + i) * NT;
                    if (e < total) {
                        const bool ok = e < valid;  // rows past the batch: zeros
                        lds[slot[i]] = cx<T>{ok ? xr[i].re : (T)0, ok ? xr[i].im * sgn : (T)0};
                    }
                });
            }
        });
        for (int i = tid; i < p.ltab_n; i += NT) lds[p.tab_off + i] = p.ltab[i];
    });
    for (int s = 0; s < p.nstages; ++s) {
        ex.barrier();
        const LsmStage st = stages[s];
        const LsmStage stn = stages[s + 1 < p.nstages ? s + 1 : s];
        ex.for_threads([&](int tid, cx<T>*) {
            unsigned* cur = ex.words(tid);
            unsigned* nxt = cur + 2 * kLsmItems;
            // the NEXT stage's words are requested now and consumed behind the next barrier: their latency hides behind this stage
            if (s + 1 < p.nstages) lsm_fetch_desc<NT>(stn, p.desc, tid, nxt, nxt + kLsmItems);
            // a stage with more work items than the workgroup holds at once runs in ROUNDS (in-place items are independent: no barrier in
            // between); the first round's words were fetched a stage ahead, the later ones are fetched when their round starts
            for (int r0 = 0; r0 < st.total; r0 += st.round) {
                if (r0 > 0) lsm_fetch_desc<NT>(st, p.desc, tid, cur, cur + kLsmItems, r0);
                const int left = (st.total - r0) < st.round ? (st.total - r0) : st.round;
                if (st.op == LSM_FIX) {
                    lsm_fix<NT, T>(st, p, tid, lds, ltab, cur, cur + kLsmItems, left);
                } else if (st.op == LSM_X0MUL) {
                    lsm_x0mul<NT, T>(st, p, tid, lds, ltab, cur, cur + kLsmItems, left);
                } else if (st.op == LSM_BFLY2) {
                    MI_LSM_RADIX_SWITCH(st.radix, (lsm_bfly2<RR, NT, T>(st, p, tid, lds, ltab, cur, cur + kLsmItems, left)));
                } else {
                    MI_LSM_RADIX_SWITCH(st.radix, (lsm_bfly<RR, NT, T>(st, p, tid, lds, ltab, cur, cur + kLsmItems, left)));
                }
            }
            static_for<0, 2 * kLsmItems>([&](auto I_) { cur[decltype(I_)::value] = nxt[decltype(I_)::value]; });
        });
    }
    ex.barrier();
    ex.for_threads([&](int tid, cx<T>*) {
        static_for<0, kLsmIoMax / CH>([&](auto Q_) {
            constexpr int q = Q_;
            if ((q + 1) * CH * NT <= valid) {
                cx<T> y[CH];
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    y[i] = lds[(int)p.stperm[(unsigned)(tid + (q * CH + i) * NT)]];
                });
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    out[(unsigned)(tid + (q * CH + i) * NT)] = cx<T>{y[i].re, y[i].im * sgn};
                });
            } else if (q * CH * NT < valid) {
                cx<T> y[CH];
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    const int e = tid + (q * CH + i) * NT;
                    y[i] = lds[(int)p.stperm[e < valid ? e : 0]];
                });
                static_for<0, CH>([&](auto I_) {
                    constexpr int i = I_;
                    const int e = tid + (q * CH + i) * NT;
                    if (e < valid) out[(unsigned)e] = cx<T>{y[i].re, y[i].im * sgn};
                });
            }
        });
    });
}

}  // namespace mi355
