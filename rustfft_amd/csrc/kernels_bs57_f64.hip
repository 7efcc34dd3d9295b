// One-kernel Bluestein bodies over the 5 * 2^k and 7 * 2^k inner lengths, Complex<double> (see kernels_bs57_f32.hip).
#include "launch.h"
namespace mi355 {
void register_bs57_f64(std::vector<KernelEntry>& reg) {
    MI_BS(double, 64, 1, 320, 64, 5, 8, 8);
    MI_BS(double, 64, 1, 448, 64, 7, 8, 8);
    MI_BS(double, 64, 1, 640, 80, 10, 8, 8);
    MI_BS(double, 64, 1, 896, 112, 8, 8, 14);
    MI_BS(double, 64, 1, 1280, 128, 5, 16, 16);
    MI_BS(double, 64, 1, 1792, 128, 16, 16, 7);
    MI_BS(double, 64, 1, 2560, 256, 10, 16, 16);
    MI_BS(double, 64, 1, 3584, 256, 14, 16, 16);
    MI_BS(double, 64, 1, 5120, 640, 8, 8, 8, 10);  // 50.5 ns per row against 58.9 for 16 x 16 x 20
    MI_BS(double, 64, 1, 7168, 512, 16, 16, 28);
    MI_BSS(double, 64, 1, 10240, 640, 10, 8, 8, 16);  // 120.0 against 131.8 for 16 x 16 x 10 x 4
    MI_BSS(double, 64, 1, 14336, 512, 16, 16, 14, 4);  // 896 threads cap a thread at 128 VGPRs and spill
    MI_BSV(1, double, 64, 1, 640, 80, 8, 8, 10);  // tuning: the largest-first order
    MI_BSV(1, double, 64, 1, 1280, 128, 16, 10, 8);  // tuning: the largest-first order
    MI_BSV(1, double, 64, 1, 2560, 256, 16, 16, 10);  // tuning: the largest-first order
    MI_BSV(1, double, 64, 1, 3584, 256, 16, 16, 14);  // tuning: the largest-first order
    MI_BSV(1, double, 64, 1, 5120, 512, 16, 16, 20);  // tuning: the schedules these replaced
    MI_BSSV(1, double, 64, 1, 10240, 640, 16, 16, 10, 4);
    // round 5, tuning 60 .. 63 (Complex<f64>): sub-pass factors fetched one exchange ahead (60), every table but the last staged in LDS (61), both (62),
    // sub-pass 1 staged + the others fetched ahead (63) -- kernels.h bluestein_body PF
    MI_BSPV(60, 1, double, 64, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(61, 2, double, 64, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(62, 3, double, 64, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(63, 17, double, 64, 1, 2560, 256, 10, 16, 16);
    MI_BSPV(60, 1, double, 64, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(61, 2, double, 64, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(62, 3, double, 64, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(63, 17, double, 64, 1, 3584, 256, 14, 16, 16);
    MI_BSPV(60, 1, double, 64, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(61, 2, double, 64, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(62, 3, double, 64, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(63, 17, double, 64, 1, 5120, 640, 8, 8, 8, 10);
    MI_BSPV(60, 1, double, 64, 1, 7168, 512, 16, 16, 28);
    MI_BSPV(61, 2, double, 64, 1, 7168, 512, 16, 16, 28);
    MI_BSPV(62, 3, double, 64, 1, 7168, 512, 16, 16, 28);
    MI_BSPV(63, 17, double, 64, 1, 7168, 512, 16, 16, 28);
}
}  // namespace mi355
