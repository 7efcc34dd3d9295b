// One-kernel and two-kernel Bluestein bodies, Complex<float>, over the 2^k and 3 * 2^k inner lengths (the 5 * 2^k / 7 * 2^k ones:
// kernels_bs57_f32.hip).  This unit and kernels_bs57_f32.hip are compiled with -fno-slp-vectorize (Makefile NOSLP): the SLP vectoriser
// pairs the re / im parts of DIFFERENT values into v_pk_*_f32 operations and pays for the pairing in register moves (a quarter of the
// VALU instructions of these bodies); without it every inner length listed here runs 1 ... 9.5 % faster (416 primes <= 4096: median
// +4.9 %; the two-kernel pair at n = 10007: +25 %; profiles/r4/ab_noslp_*.jsonl).  2048 and 1792 lose and stay in kernels_np2_f32.hip.
#define MI355_PK_CMUL 1
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_bs_f32(std::vector<KernelEntry>& reg) {
    MI_BS_LIST(float, 32);
    MI_BS(float, 32, 4, 512, 64, 8, 8, 8);
    MI_BS(float, 32, 1, 1024, 128, 8, 8, 16);  // 3.98 ns per row against 4.25 for 16 x 16 x 4 on one wave
    // 8192: four lighter sub-passes (16 values per thread, 94 VGPRs, five waves per SIMD).  Rounds 2 - 3 measured 16 x 16 x 32 (132 VGPRs)
    // ahead, 1.22 against 0.97 TB/s -- with the SLP vectoriser; without it (this unit) the light schedule runs n = 4093 in 2.66 ms against
    // 3.78: +42 % (profiles/r4/ab_bs8192_variants.jsonl; 16 x 8 x 8 x 8: 2.72).  The old schedule: tuning 6.
    MI_BS(float, 32, 1, 8192, 512, 8, 8, 8, 16);
    MI_BSV(6, float, 32, 1, 8192, 512, 16, 16, 32);
    MI_BS_LIST3_F32(float, 32);
    MI_BSV(5, float, 32, 1, 8192, 512, 16, 8, 8, 8);
    // one-kernel Bluestein for 4096 < n <= 8192 through the split exchange, two-kernel Bluestein for 8192 < n <= 16384 (see kernels_k1_f32.hip
    // for the measurements behind the split)
    MI_BSS(float, 32, 1, 12288, 768, 12, 8, 8, 16);   // four lighter sub-passes: 100.5 ns per row against 115.4 for 32 x 24 x 16 on 512 threads
    MI_BSS(float, 32, 1, 16384, 1024, 8, 8, 16, 16);  // 111.8 against 125.0 for 16 x 32 x 32 on 512 threads
    MI_BS2(float, 32, 1, true, 24576, 1024, 32, 32, 24);
    MI_BS2(float, 32, 1, true, 32768, 1024, 32, 32, 32);
    // round 5, tuning 70 / 71 / 72: the shipped body of each inner length (bs_tw1 / bs_pf) + the spectrum multiplier fetched in front of the first
    // transform's last sub-pass (70), + the output chirp in front of the second one's (71), the chirp alone (72)
    MI_BSPV(70, 20, float, 32, 1, 1024, 128, 8, 8, 16);
    MI_BSPV(71, 28, float, 32, 1, 1024, 128, 8, 8, 16);
    MI_BSPV(72, 24, float, 32, 1, 1024, 128, 8, 8, 16);
    MI_BSPV(70, 20, float, 32, 1, 1536, 256, 6, 16, 16);
    MI_BSPV(71, 28, float, 32, 1, 1536, 256, 6, 16, 16);
    MI_BSPV(72, 24, float, 32, 1, 1536, 256, 6, 16, 16);
    MI_BSPV(70, 21, float, 32, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(71, 29, float, 32, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(72, 25, float, 32, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(70, 20, float, 32, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(71, 28, float, 32, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(72, 24, float, 32, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(70, 7, float, 32, 1, 6144, 512, 6, 8, 8, 16);
    MI_BSPV(71, 15, float, 32, 1, 6144, 512, 6, 8, 8, 16);
    MI_BSPV(72, 11, float, 32, 1, 6144, 512, 6, 8, 8, 16);
    MI_BSPV(70, 7, float, 32, 1, 8192, 512, 8, 8, 8, 16);
    MI_BSPV(71, 15, float, 32, 1, 8192, 512, 8, 8, 8, 16);
    MI_BSPV(72, 11, float, 32, 1, 8192, 512, 8, 8, 8, 16);
    // round 5, tuning 80: the shipped body of each inner length with TWO rows per physical thread (launch.h DevExecRows2)
    MI_BSR2V(80, 17, float, 32, 3072, 256, 12, 16, 16);
    MI_BSR2V(80, 16, float, 32, 4096, 512, 8, 8, 8, 8);
    MI_BSR2V(80, 7, float, 32, 6144, 512, 6, 8, 8, 16);
    MI_BSR2V(80, 3, float, 32, 8192, 512, 8, 8, 8, 16);
    MI_BSR2V(80, 16, float, 32, 1024, 128, 8, 8, 16);
    MI_BSR2V(80, 16, float, 32, 1536, 256, 6, 16, 16);
    // round 5, tuning 81 / 82: MORE threads per row, 8 values per thread (the two-rows-per-thread form, tuning 80, measured -17 ... -45 %: these bodies want more waves, not fewer)
    MI_BSPV(81, 3, float, 32, 1, 8192, 1024, 8, 8, 8, 16);
    MI_BSPV(82, 1, float, 32, 1, 8192, 1024, 8, 8, 8, 16);
    MI_BSPV(6, 7, float, 32, 1, 6144, 512, 6, 8, 8, 16);  // the schedule shipped until round 5 (with its staging / prefetch choice)
    MI_BSPV(82, 1, float, 32, 1, 6144, 768, 8, 8, 8, 12);
    MI_BSPV(81, 17, float, 32, 1, 3072, 384, 8, 8, 8, 6);
    MI_BSPV(82, 1, float, 32, 1, 3072, 384, 8, 8, 8, 6);
    MI_BSPV(83, 1, float, 32, 1, 8192, 1024, 8, 8, 8, 8, 2);  // five sub-passes, 8 values per thread throughout
    MI_BSPV(83, 1, float, 32, 1, 6144, 768, 8, 8, 8, 6, 2);
}
}  // namespace mi355
