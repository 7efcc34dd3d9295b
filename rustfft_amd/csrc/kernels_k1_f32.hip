// K1 (batched contiguous) instantiations, Complex<float>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k1_f32(std::vector<KernelEntry>& reg) {
    MI_K1_LIST(float, 32);
    // 256 ... 2048: one kernel per row block; the other tilings below (tuning entries) measured within 2 % of these
    MI_K1(float, 32, 16, false, 256, 16, 16, 16);
    MI_K1(float, 32, 8, false, 512, 32, 16, 8, 4);
    // round 3: sub-pass 1's twiddle table staged in LDS ("t1", engine.h TWL; interleaved A/B profiles/r3/ab_twl_round1.jsonl):
    // 1024: 5.22 -> 5.56 TB/s, 2048: 5.15 -> 5.42, 4096: 5.14 -> 5.37, 8192: 4.96 -> 5.38, 16384: 4.83 -> 5.09, 32768: 4.37 -> 4.68
    // (staging every table costs the 4096-point kernel its occupancy: 4.77; no gain at 256 / 512); the old kernels: tuning variant 40
    MI_K1X(float, 32, 4, false, 1024, "t1", 1024, 64, 16, 16, 4);
    MI_K1X(float, 32, 2, false, 1024, "t1", 2048, 128, 16, 16, 8);
    MI_K1V(40, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1V(40, float, 32, 2, false, 2048, 128, 16, 16, 8);
    // tuning: other tilings of the small whole-row kernels (tools/ab.py --log2n 8 .. 11 min:MI355FFT_VARIANT=v)
    MI_K1V(5, float, 32, 8, false, 256, 32, 8, 8, 4);
    MI_K1V(6, float, 32, 32, false, 256, 16, 16, 16);
    MI_K1V(5, float, 32, 4, false, 512, 64, 8, 8, 8);
    MI_K1V(6, float, 32, 8, false, 512, 64, 8, 8, 8);
    MI_K1V(7, float, 32, 16, false, 512, 32, 16, 8, 4);
    MI_K1V(5, float, 32, 2, false, 1024, 128, 8, 8, 16);
    MI_K1V(6, float, 32, 8, false, 1024, 64, 16, 16, 4);
    MI_K1V(7, float, 32, 4, false, 1024, 128, 16, 8, 8);
    MI_K1V(5, float, 32, 1, false, 2048, 256, 8, 16, 16);
    MI_K1V(6, float, 32, 4, false, 2048, 128, 16, 16, 8);
    MI_K1V(7, float, 32, 1, false, 2048, 128, 16, 16, 8);
    MI_K1V(40, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1V(40, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1V(40, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1V(40, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    MI_K1X(float, 32, 1, false, 1024, "t1", 4096, 256, 16, 16, 16);  // interleaved A/B: 5.08 TB/s against 4.82 for 8 x 8 x 8 x 8 on 512 threads
    // 2^13 .. 2^15 in ONE kernel: the real and imaginary planes go through LDS one after the other (split exchange), so a
    // whole 32768-point row fits 132 KB.  Measured on MI355X: 18.0 / 20.8 / 20.3 TFLOP/s (4.6 / 4.8 / 4.3 TB/s) against
    // 10.6 / 12.2 / 12.7 for two column-tile passes.  Variants: the 16-values-per-thread schedules (4.4 - 4.5 TB/s at 8192).
    // round 5: 8192 and 16384 read their rows with NON-TEMPORAL loads ("n"; ABL bit 16): +2.8 ... +2.9 % and +3.7 ... +4.0 % in two interleaved
    // runs (5.39 -> 5.54 and 5.13 -> 5.33 TB/s), bit-identical results; 1024: +0.3 ... +1.9 %, 2048 / 4096 / 32768: -1 ... -6 %; non-temporal
    // STORES lose everywhere (up to -31 % at 16384): profiles/r5/ab_k1_nt_2p*.jsonl, ab_k1_ntload_confirm_2p*.jsonl
    MI_K1X(float, 32, 1, true, 1040, "t1n", 8192, 256, 8, 32, 32);
    MI_K1X(float, 32, 1, true, 1040, "t1n", 16384, 512, 16, 32, 32);
    MI_K1X(float, 32, 1, true, 1024, "t1", 32768, 1024, 32, 32, 32);
    MI_K1V(3, float, 32, 1, false, 8192, 512, 16, 8, 8, 8);
    // tuning: split exchange for the LDS-bound 2^10 .. 2^12 kernels (half the LDS per workgroup: six instead of four per CU)
    MI_K1V(20, float, 32, 4, true, 1024, 64, 16, 16, 4);
    MI_K1V(20, float, 32, 2, true, 2048, 128, 16, 16, 8);
    MI_K1V(20, float, 32, 1, true, 4096, 256, 16, 16, 16);
    MI_K1V(21, float, 32, 2, false, 1024, 64, 16, 16, 4);
    MI_K1V(21, float, 32, 1, false, 2048, 128, 16, 16, 8);
    // two-kernel Bluestein for 4096 < n <= 16384 (padded lengths 3 * 2^12, 2^14, 3 * 2^13, 2^15)
    // one-kernel Bluestein for 4096 < n <= 8192: split exchange, spectrum handed over in registers (measured 626 / 1020 GB/s at
    // n = 4099 / 7919 against 470 / 690 for the two-kernel form; the 1024-thread bodies for M = 24576, 32768 spill under the
    // 128-VGPR cap and lose to it: 458 against 642 at n = 10007, so 8192 < n <= 16384 keeps two kernels)
    // (the production bodies for 12288 / 16384 and the two-kernel pair: kernels_bs_f32.hip, compiled without the SLP vectoriser)
    MI_BSSV(1, float, 32, 1, 12288, 512, 32, 24, 16);  // tuning: the schedules these replaced; (2): without the split exchange
    MI_BSSV(1, float, 32, 1, 16384, 512, 16, 32, 32);
    MI_BSV(2, float, 32, 1, 16384, 512, 16, 32, 32);
    MI_K1V(4, float, 32, 1, true, 8192, 512, 16, 8, 8, 8);
    // tuning: lighter sub-passes / more threads for the whole-row kernels (tools/ab.py --log2n k min:MI355FFT_VARIANT=v)
    MI_K1V(5, float, 32, 1, false, 4096, 512, 8, 8, 8, 8);
    MI_K1V(6, float, 32, 2, false, 4096, 256, 16, 16, 16);
    // (profiles/r2/ab_k1_pow2_f32.jsonl: none of these beats the shipped whole-row schedules in f32)
    MI_K1V(5, float, 32, 1, true, 8192, 512, 8, 8, 8, 16);
    MI_K1V(6, float, 32, 1, false, 8192, 512, 16, 16, 32);
    MI_K1V(7, float, 32, 1, false, 8192, 512, 8, 8, 8, 16);
    MI_K1V(8, float, 32, 1, true, 8192, 1024, 8, 8, 8, 16);
    MI_K1V(5, float, 32, 1, true, 16384, 1024, 8, 8, 16, 16);
    MI_K1V(6, float, 32, 1, true, 16384, 1024, 16, 16, 8, 8);
    MI_K1V(7, float, 32, 1, true, 16384, 512, 8, 8, 16, 16);
    MI_K1V(5, float, 32, 1, true, 32768, 1024, 8, 16, 16, 16);
    MI_K1V(6, float, 32, 1, true, 32768, 1024, 16, 16, 16, 8);
    // ablation probes of the 1024-point kernel (MI355FFT_VARIANT=5..7, wrong results by design): measured 5.36 TB/s for the
    // load/store skeleton against 5.1 - 5.2 TB/s for the full kernel.  Tuning history (no gain, removed): F = 2 / 8 rows per
    // workgroup, radix-8 schedules, 128-thread 4096 kernel, non-temporal loads/stores (tools/membench shows +11 % for an
    // in-place copy, the real kernels lose 1 - 3 %).
    // round 5: non-temporal accesses re-measured on the kernels with staged tables (tools/membench/skel4: +3 .. +8 % on this shape):
    // 50 = loads and stores, 51 = loads only, 52 = stores only (ABL bits 16 / 32 on top of the shipped "t1" = 1024)
    MI_K1ABL(50, 1072, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(51, 1040, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(52, 1056, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(50, 1072, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(51, 1040, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(52, 1056, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(50, 1072, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(51, 1040, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(52, 1056, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(50, 1072, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1ABL(51, 1040, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1ABL(52, 1056, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1ABL(50, 1072, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(51, 1040, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(52, 1056, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(50, 1072, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    MI_K1ABL(51, 1040, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    MI_K1ABL(52, 1056, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    // round 5, tuning 53: the shipped kernels + sub-pass factors fetched one exchange ahead (ABL bit 256)
    MI_K1ABL(53, 1280, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(53, 1280, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(53, 1280, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(53, 1296, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1ABL(53, 1296, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(53, 1280, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    // sub-pass twiddle tables staged in LDS: all (30) / sub-pass 1 only (31) / last sub-pass only (32)
    MI_K1ABL(30, 128, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(31, 1024, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(32, 2048, float, 32, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(30, 128, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(31, 1024, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(32, 2048, float, 32, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(30, 128, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(31, 1024, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(32, 2048, float, 32, 1, false, 4096, 256, 16, 16, 16);
    MI_K1ABL(31, 1024, float, 32, 1, true, 8192, 256, 8, 32, 32);
    MI_K1ABL(31, 1024, float, 32, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(31, 1024, float, 32, 1, true, 32768, 1024, 32, 32, 32);
    MI_K1ABL(30, 128, float, 32, 16, false, 256, 16, 16, 16);
    MI_K1ABL(30, 128, float, 32, 8, false, 512, 32, 16, 8, 4);
    MI_K1ABL(5, 12, float, 32, 4, false, 1024, 64, 16, 16, 4);  // loads + stores only
    MI_K1ABL(6, 8, float, 32, 4, false, 1024, 64, 16, 16, 4);   // arithmetic without the exchanges
    MI_K1ABL(7, 4, float, 32, 4, false, 1024, 64, 16, 16, 4);   // exchanges without the arithmetic
}
}  // namespace mi355
