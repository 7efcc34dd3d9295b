// K2 (column-tile passes of the large-N decomposition) instantiations, Complex<float>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2_f32(std::vector<KernelEntry>& reg) {
    MI_K2(float, 32, 64, false, 64, 8, 8, 8);
    // (sub-pass twiddle tables staged in LDS -- suffix "t", engine.h TWL -- wherever an interleaved A/B measured a gain:
    // profiles/r3/ab_twl_round{1,2}.jsonl; the pre-staging kernels stay as tuning variant 40)
    MI_K2_FIRST(float, 32, 64, false, 128, 8, 16, 8);
    MI_K2X_LATER(float, 32, 64, false, 128, "t", 128, 8, 16, 8);  // last pass of the three-pass plans: 4.78 - 4.90 -> 4.97 - 5.31 TB/s
    MI_K2(float, 32, 32, false, 256, 16, 16, 16);
    // 512-row tile: 32 columns (256-byte row segments), 32 values per thread, split exchange (measured +12 % at 2^18 over the
    // 16-column full-complex tile, kept as variant 9)
    // (interleaved A/B on one box, round 2: as a FIRST pass the 16-column full-complex tile moves 5.45 TB/s against 5.18, as a
    // later pass 4.79 against 5.12 -- each kind gets its better tiling)
    MI_K2X_LATER(float, 32, 32, true, 128, "t", 512, 16, 8, 8, 8);   // staged tables: 5.12 -> 5.69 TB/s (2^18)
    MI_K2X_FIRST(float, 32, 16, false, 128, "t", 512, 32, 16, 8, 4);  // 5.19 -> 5.84
    // 1024-row tile: 16 columns (128-byte row segments), 32 values per thread, real/imaginary planes exchanged
    // one after the other so two workgroups fit a CU's LDS; the twiddled sub-passes are radix 8 (fewer live twiddles)
    // round 3: the sub-pass twiddle tables (1016 entries) staged in LDS: first pass 5.41 -> 5.61 TB/s, later pass 5.10 -> 5.53
    // (bit-identical results; the other radix orders measured within 1 % of this one with staged tables)
    MI_K2X(float, 32, 16, true, 128, "t", 1024, 32, 8, 8, 16);
    // 2048-row tile (2^21 and 2^22 in two passes instead of three): 16 columns = 128-byte row segments through the split
    // exchange (135 KB of LDS, one 1024-thread workgroup per CU).  Measured at 2^22: 14.0 TFLOP/s against 13.0 for the
    // 8-column tile (64-byte segments, two workgroups per CU, tiles paired per XCD), kept as variant 2.
    // Round 2 (XCD-aware tile order in place): as a FIRST pass the 8-column tile (two 512-thread workgroups per CU, contiguous
    // 128 KiB writes) moves 4.94 - 5.02 TB/s against 4.41 - 4.47 for the 16-column one; as a later pass (64-byte strided
    // writes) 3.85 against 4.25.
    MI_K2_FIRST(float, 32, 8, true, 2048, 64, 8, 16, 16);
    // round 4: the LATER pass as two columns per lane (launch.h DevExecPair: 1024 physical threads run 2048 virtual ones, adjacent
    // columns of the same rows; row loads / stores of 16 bytes): 4.08 -> 3.92 ms per 8 GiB launch at 2^22 (+4 %,
    // profiles/r4/ab_fused_1024thr_2p22.jsonl variant 54; as a FIRST pass the form loses 9 %: ab_2048_persist.jsonl)
    MI_K2X_LATER(float, 32, 16, true, 4096, "p2", 2048, 128, 8, 16, 16);
#if defined(MI355_TUNING)
    MI_K2_LATER(float, 32, 16, true, 2048, 64, 8, 16, 16);  // tuning 41: one column per lane (the round-3 tile)
    reg.back().variant = 41;
#endif
    MI_K2V(40, float, 32, 64, false, 128, 8, 16, 8);       // tuning 40: the kernels before the tables were staged in LDS
    MI_K2V(40, float, 32, 16, true, 1024, 32, 8, 8, 16);
#if defined(MI355_TUNING)
    MI_K2_LATER(float, 32, 32, true, 512, 16, 8, 8, 8);
    reg.back().variant = 40;
    MI_K2_FIRST(float, 32, 16, false, 512, 32, 16, 8, 4);
    reg.back().variant = 40;
#endif
    MI_K2V(1, float, 32, 8, false, 1024, 64, 16, 16, 4);   // tuning: 64-byte row segments paired per XCD, full-complex exchange
    MI_K2V(3, float, 32, 8, true, 1024, 32, 8, 8, 16);     // tuning: 8-column tiles (paired per XCD), 256 threads, four workgroups per CU
    // tuning: pair-fused first two sub-passes (v_permlane32_swap instead of the first LDS exchange)
#if defined(MI355_TUNING)
    MI_K2P_FIRST(float, 32, 16, true, 1024, 32, 8, 8, 16);
    reg.back().variant = 12;
    MI_K2P_LATER(float, 32, 16, true, 1024, 32, 8, 8, 16);
    reg.back().variant = 12;
    MI_K2P_LATER(float, 32, 32, true, 512, 16, 8, 8, 8);
    reg.back().variant = 12;
#endif
    // tuning 50 - 53: two columns per lane (DevExecPair, ABL bit 4096) with the staged tables (128): the tile as 16 columns x 64
    // virtual slots of 16 values, 512 physical threads x 2 x 16 values -- the same registers, waves and LDS as the shipped tile,
    // row loads / stores of 16 bytes per lane; 52 / 53: the 2048-row tile the same way (1024 physical threads)
    MI_K2ABL(50, 4224, float, 32, 16, true, 1024, 64, 8, 8, 16);
    MI_K2ABL(51, 4224, float, 32, 16, true, 1024, 64, 16, 8, 8);
    MI_K2ABL(52, 4096, float, 32, 16, true, 2048, 128, 8, 16, 16);
    MI_K2ABL(53, 4096, float, 32, 16, true, 2048, 128, 16, 16, 8);
    // tuning 60: PERSISTENT workgroups (ABL bit 8192) for the one-workgroup-per-CU 2048-row later tile and, for comparison, the 1024-row tiles
    MI_K2ABL(60, 8192, float, 32, 16, true, 2048, 64, 8, 16, 16);
    MI_K2ABL(60, 8320, float, 32, 16, true, 1024, 32, 8, 8, 16);
#if defined(MI355_TUNING)
    // round 5, tuning 70 / 71 / 72: the 2048-row tiles of 2^22 (config 5's per-GPU kernels) with non-temporal loads + stores / loads / stores
    reg.push_back(make_k2<float, Sched<2048, 64, 8, 16, 16>, 8, true, true, 48>(32, "k2first<2048, 64, 8, 16, 16>xF8nt"));
    reg.back().variant = 70;
    reg.push_back(make_k2<float, Sched<2048, 128, 8, 16, 16>, 16, false, true, 4096 + 48>(32, "k2later<2048, 128, 8, 16, 16>xF16p2nt"));
    reg.back().variant = 70;
    reg.push_back(make_k2<float, Sched<2048, 64, 8, 16, 16>, 8, true, true, 16>(32, "k2first<2048, 64, 8, 16, 16>xF8ntl"));
    reg.back().variant = 71;
    reg.push_back(make_k2<float, Sched<2048, 128, 8, 16, 16>, 16, false, true, 4096 + 16>(32, "k2later<2048, 128, 8, 16, 16>xF16p2ntl"));
    reg.back().variant = 71;
    reg.push_back(make_k2<float, Sched<2048, 64, 8, 16, 16>, 8, true, true, 32>(32, "k2first<2048, 64, 8, 16, 16>xF8nts"));
    reg.back().variant = 72;
    reg.push_back(make_k2<float, Sched<2048, 128, 8, 16, 16>, 16, false, true, 4096 + 32>(32, "k2later<2048, 128, 8, 16, 16>xF16p2nts"));
    reg.back().variant = 72;
#endif
    MI_K2V(10, float, 32, 16, true, 1024, 32, 32, 32);     // tuning: two radix-32 sub-passes, one exchange (the later pass spills)
    MI_K2V(11, float, 32, 16, true, 1024, 32, 4, 16, 16);  // tuning: radix-4 first sub-pass (eight butterflies per thread)
    // ablation probes of the default 1024-row tile (wrong results by design; MI355FFT_VARIANT=5..8, tuning only)
    MI_K2ABL(5, 13, float, 32, 16, true, 1024, 32, 8, 8, 16);  // loads + stores only
    MI_K2ABL(6, 9, float, 32, 16, true, 1024, 32, 8, 8, 16);   // arithmetic, no exchange, no inter-pass twiddles
    MI_K2ABL(7, 4, float, 32, 16, true, 1024, 32, 8, 8, 16);   // exchange + twiddles, no butterflies
    MI_K2ABL(8, 1, float, 32, 16, true, 1024, 32, 8, 8, 16);   // everything but the inter-pass twiddles
    // (non-temporal loads / stores, ABL bits 16 / 32: measured -0.5 % / -13 % on this tile, not instantiated)
    MI_K2ABL(20, 128, float, 32, 16, true, 1024, 32, 8, 8, 16);  // sub-pass twiddles staged in LDS (kernels.h K2Src)
    MI_K2ABL(20, 128, float, 32, 64, false, 64, 8, 8, 8);
    MI_K2ABL(20, 128, float, 32, 64, false, 128, 8, 16, 8);
    MI_K2ABL(20, 128, float, 32, 32, false, 256, 16, 16, 16);
#if defined(MI355_TUNING)
    MI_K2X_LATER(float, 32, 32, true, 128, "t", 512, 16, 8, 8, 8);
    reg.back().variant = 20;
    MI_K2X_FIRST(float, 32, 16, false, 128, "t", 512, 32, 16, 8, 4);
    reg.back().variant = 20;
    MI_K2X_FIRST(float, 32, 32, true, 128, "t", 512, 16, 8, 8, 8);
    reg.back().variant = 32;
    MI_K2X_LATER(float, 32, 32, true, 128, "t", 512, 16, 8, 8, 8);
    reg.back().variant = 32;
    MI_K2X_LATER(float, 32, 16, true, 128, "t", 2048, 64, 8, 16, 16);
    reg.back().variant = 20;
    MI_K2X_FIRST(float, 32, 8, true, 1024, "t1", 2048, 64, 8, 16, 16);
    reg.back().variant = 20;
#endif
    MI_K2ABL(28, 128, float, 32, 16, true, 1024, 32, 16, 16, 4);
    MI_K2ABL(29, 128, float, 32, 16, true, 1024, 32, 4, 8, 32);
    MI_K2ABL(30, 128, float, 32, 16, true, 1024, 32, 4, 4, 4, 16);
#if defined(MI355_TUNING)
    MI_K2X_LATER(float, 32, 32, true, 128, "t", 512, 16, 4, 8, 16);
    reg.back().variant = 31;
    MI_K2X_FIRST(float, 32, 16, false, 128, "t", 512, 32, 4, 8, 16);
    reg.back().variant = 31;
#endif
    MI_K2ABL(21, 132, float, 32, 16, true, 1024, 32, 8, 8, 16);  // probe: staging, no arithmetic
    MI_K2ABL(22, 256, float, 32, 16, true, 1024, 32, 8, 8, 16);  // probe: LDS allocation of the staged variant, twiddles from global
    MI_K2ABL(23, 640, float, 32, 16, true, 1024, 32, 8, 8, 16);  // probe: staged + 4 KiB more LDS
    MI_K2ABL(24, 128, float, 32, 16, true, 1024, 32, 8, 16, 8);  // staged, other radix orders
    MI_K2ABL(25, 128, float, 32, 16, true, 1024, 32, 16, 8, 8);
    MI_K2ABL(26, 128, float, 32, 16, true, 1024, 32, 4, 16, 16);
}
}  // namespace mi355
