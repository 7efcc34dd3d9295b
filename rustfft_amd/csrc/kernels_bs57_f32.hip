// One-kernel Bluestein bodies over the 5 * 2^k and 7 * 2^k inner lengths, Complex<float>: with the 2^k and 3 * 2^k ones
// (kernels_np2_f32.hip, kernel_lists.h) the planner's "smallest compiled M >= 2n - 1" pads by less than 1.25x instead of
// 1.33x (the reference pads to 2^k or 3 * 2^k, src/plan.rs:649-657; every pass of the kernel runs over M, not n).
#define MI355_PK_CMUL 1
#include "launch.h"
namespace mi355 {
void register_bs57_f32(std::vector<KernelEntry>& reg) {
    MI_BS(float, 32, 1, 320, 64, 5, 8, 8);
    MI_BS(float, 32, 1, 448, 64, 7, 8, 8);
    MI_BS(float, 32, 1, 640, 64, 10, 8, 8);
    MI_BS(float, 32, 1, 896, 64, 14, 8, 8);
    MI_BS(float, 32, 1, 1280, 128, 10, 8, 16);
    // (1792: kernels_np2_f32.hip -- it is the one 7 * 2^k body that loses without the SLP vectoriser, which this unit is compiled without)
    MI_BS(float, 32, 1, 2560, 256, 10, 16, 16);
    MI_BS(float, 32, 1, 3584, 256, 14, 16, 16);
    MI_BS(float, 32, 1, 5120, 512, 10, 8, 8, 8);  // four lighter sub-passes: 25.6 ns per row against 33.0 for 16 x 16 x 20
    MI_BS(float, 32, 1, 7168, 512, 16, 16, 28);
    MI_BSS(float, 32, 1, 10240, 640, 10, 8, 8, 16);  // 88.5 against 101.5 for 32 x 20 x 16 on 512 threads
    MI_BSS(float, 32, 1, 14336, 512, 32, 28, 16);
    // Measured and NOT compiled: the 9 * 2^k and 15 * 2^k inner lengths (576 ... 4608, 960 ... 7680; worst-case padding 1.25x -> 1.17x): every
    // prime that would move, old ladder against new in one process (profiles/r4/ab_ladder915_f32_rep*.jsonl): 576 -7 %, 960 -17 %, 1152 -8 %,
    // 2304 -5 %, 3840 -15 %, 4608 -2 %, 1920 +4 %, 7680 +32 % -- and 7680 only beat the old 8192 body, which the lighter 8192 schedule
    // (kernels_bs_f32.hip) now beats by more: the radix-9 / 15 sub-passes cost more than the padding they save.
    MI_BSV(4, float, 32, 1, 7168, 512, 14, 8, 8, 8);  // tuning 4 / 5: four lighter sub-passes instead of 16 x 16 x 28
    MI_BSV(5, float, 32, 1, 7168, 512, 8, 8, 8, 14);
    MI_BSV(1, float, 32, 1, 640, 80, 8, 8, 10);  // tuning: the largest-first order
    MI_BSV(1, float, 32, 1, 1280, 128, 16, 10, 8);  // tuning: the largest-first order
    MI_BSV(1, float, 32, 1, 2560, 256, 16, 16, 10);  // tuning: the largest-first order
    MI_BSV(1, float, 32, 1, 3584, 256, 16, 16, 14);  // tuning: the largest-first order
    MI_BSV(1, float, 32, 1, 896, 112, 8, 8, 14);  // tuning: the largest-first order
    MI_BSV(1, float, 32, 1, 5120, 512, 16, 16, 20);  // tuning: the three-sub-pass schedules
    MI_BSSV(1, float, 32, 1, 10240, 512, 32, 20, 16);
    MI_BSSV(4, float, 32, 1, 14336, 1024, 14, 16, 8, 8);  // tuning 4 / 5: 14336 in four lighter sub-passes instead of 32 x 28 x 16
    MI_BSSV(5, float, 32, 1, 14336, 896, 16, 14, 8, 8);
    // round 5, tuning 70 / 71 / 72: the shipped body of each inner length (bs_tw1 / bs_pf) + the spectrum multiplier fetched in front of the first
    // transform's last sub-pass (70), + the output chirp in front of the second one's (71), the chirp alone (72)
    MI_BSPV(70, 20, float, 32, 1, 1280, 128, 10, 8, 16);
    MI_BSPV(71, 28, float, 32, 1, 1280, 128, 10, 8, 16);
    MI_BSPV(72, 24, float, 32, 1, 1280, 128, 10, 8, 16);
    MI_BSPV(70, 20, float, 32, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(71, 28, float, 32, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(72, 24, float, 32, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(70, 5, float, 32, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(71, 13, float, 32, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(72, 9, float, 32, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(70, 20, float, 32, 1, 5120, 512, 10, 8, 8, 8);
    MI_BSPV(71, 28, float, 32, 1, 5120, 512, 10, 8, 8, 8);
    MI_BSPV(72, 24, float, 32, 1, 5120, 512, 10, 8, 8, 8);
    MI_BSPV(70, 20, float, 32, 1, 7168, 512, 16, 16, 28);
    MI_BSPV(71, 28, float, 32, 1, 7168, 512, 16, 16, 28);
    MI_BSPV(72, 24, float, 32, 1, 7168, 512, 16, 16, 28);
    // round 5, tuning 80: the shipped body of each inner length with TWO rows per physical thread (launch.h DevExecRows2)
    MI_BSR2V(80, 28, float, 32, 2560, 256, 10, 16, 16);
    MI_BSR2V(80, 1, float, 32, 3584, 256, 14, 16, 16);
    MI_BSR2V(80, 16, float, 32, 5120, 512, 10, 8, 8, 8);
    MI_BSR2V(80, 16, float, 32, 7168, 512, 16, 16, 28);
    MI_BSR2V(80, 16, float, 32, 1280, 128, 10, 8, 16);
    // round 5, tuning 81 / 82: more threads per row, 8 values per thread
    MI_BSPV(81, 16, float, 32, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(82, 1, float, 32, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(81, 16, float, 32, 1, 7168, 896, 8, 8, 8, 14);
    MI_BSPV(82, 1, float, 32, 1, 7168, 896, 8, 8, 8, 14);
    MI_BSPV(81, 16, float, 32, 1, 3584, 448, 8, 8, 8, 7);
    MI_BSPV(82, 1, float, 32, 1, 3584, 448, 8, 8, 8, 7);
    MI_BSPV(81, 16, float, 32, 1, 2560, 320, 8, 8, 8, 5);
    MI_BSPV(82, 1, float, 32, 1, 2560, 320, 8, 8, 8, 5);
}
}  // namespace mi355
