// The compiled kernel instantiations.  Sched<N, TPF, radices...>: TPF threads per sequence; a thread
// holds at most 16 (32 for the 1024-row tile) complex values in VGPRs.
// K1 lists: batched contiguous transforms (one launch per Fft::process call).
// K2 lists: column-tile passes of the large-N decomposition, tile width F chosen so that
//           F * sizeof(Complex<T>) >= 128 contiguous bytes per row and two workgroups fit a CU's LDS.
#pragma once

#define MI_K1_LIST(T, PREC)                           \
    MI_K1(T, PREC, 256, false, 2, 1, 2);              \
    MI_K1(T, PREC, 256, false, 4, 1, 4);              \
    MI_K1(T, PREC, 128, false, 8, 1, 8);              \
    MI_K1(T, PREC, 64, false, 16, 4, 4, 4);           \
    MI_K1(T, PREC, 64, false, 32, 4, 8, 4);           \
    MI_K1(T, PREC, 32, false, 64, 8, 8, 8);           \
    MI_K1(T, PREC, 32, false, 128, 8, 16, 8)
// (256 ... 4096 are registered per precision in kernels_k1_*.hip: their tilings are measured choices, tools/ab.py)


// Bluestein bodies, one per inner power-of-two length M (serves every n with 2n - 1 <= M).  Workgroups of one wave (64
// threads) where the schedule allows: a barrier costs a single-wave workgroup nothing, and these bodies are bound by
// their barrier-separated phases, not by HBM (measured +4 .. +34 % over 256-thread workgroups for M <= 1024).
#define MI_BS_LIST(T, PREC)                        \
    MI_BS(T, PREC, 64, 8, 2, 4, 2);                \
    MI_BS(T, PREC, 64, 16, 4, 4, 4);               \
    MI_BS(T, PREC, 64, 32, 4, 8, 4);               \
    MI_BS(T, PREC, 8, 64, 8, 8, 8);                \
    MI_BS(T, PREC, 8, 128, 8, 16, 8);              \
    MI_BS(T, PREC, 2, 256, 32, 8, 8, 4);           \
    MI_BS(T, PREC, 1, 4096, 512, 8, 8, 8, 8)

// Bluestein bodies over the 3 * 2^k inner lengths (the reference's second family, src/plan.rs:649-657): the planner takes
// the smallest compiled M >= 2n - 1, so the worst-case padding drops from 2x to 1.33x
// (sub-pass order: measured per inner length with tools/bs_ladder.py, profiles/r2/bluestein_schedules_ab.txt -- the smallest
// radix first and the radix-16 sub-passes next to the register hand-over run 3 - 25 % faster than largest-first)
#define MI_BS_LIST3_F32(T, PREC)                  \
    MI_BS(T, PREC, 256, 12, 1, 12);  \
    MI_BS(T, PREC, 128, 24, 2, 12, 2);  \
    MI_BS(T, PREC, 16, 48, 4, 12, 4);  \
    MI_BS(T, PREC, 8, 96, 8, 16, 6);  \
    MI_BS(T, PREC, 16, 192, 16, 16, 12);  \
    MI_BS(T, PREC, 1, 384, 64, 6, 8, 8);  \
    MI_BS(T, PREC, 1, 768, 64, 12, 8, 8);  \
    MI_BS(T, PREC, 1, 1536, 256, 6, 16, 16);  \
    MI_BS(T, PREC, 1, 3072, 256, 12, 16, 16);  \
    MI_BS(T, PREC, 1, 6144, 768, 8, 8, 8, 12)
// (6144, round 5: 768 threads x 8 values -- 12 in the last sub-pass -- instead of 512 x 12 on 6 x 8 x 8 x 16: +5 % with the tables staged and the last
// sub-pass's factors fetched ahead, profiles/r5/ab_bs_threads_3067.jsonl; the old schedule: tuning 6)
#define MI_BS_LIST3_F64(T, PREC)                  \
    MI_BS(T, PREC, 256, 12, 1, 12);  \
    MI_BS(T, PREC, 128, 24, 2, 12, 2);  \
    MI_BS(T, PREC, 16, 48, 4, 12, 4);  \
    MI_BS(T, PREC, 8, 96, 8, 16, 6);  \
    MI_BS(T, PREC, 16, 192, 16, 16, 12);  \
    MI_BS(T, PREC, 1, 384, 64, 6, 8, 8);  \
    MI_BS(T, PREC, 1, 768, 64, 12, 8, 8);  \
    MI_BS(T, PREC, 1, 1536, 256, 6, 16, 16);  \
    MI_BS(T, PREC, 1, 3072, 256, 12, 16, 16);  \
    MI_BS(T, PREC, 1, 6144, 512, 16, 16, 24)
