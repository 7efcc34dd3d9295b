// K1 (batched contiguous) instantiations, Complex<double>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k1_f64(std::vector<KernelEntry>& reg) {
    MI_K1_LIST(double, 64);
    // 256 ... 2048: one kernel per row block; the other tilings below (tuning entries) measured within 2 % of these
    MI_K1(double, 64, 4, false, 256, 32, 8, 8, 4);  // interleaved A/B (profiles/r2/ab_k1_small_f64.jsonl): 5.47 TB/s against 5.18 for 16 x 16 on 16 threads x 16 rows
    MI_K1(double, 64, 8, false, 512, 32, 16, 8, 4);
    // round 5: non-temporal row LOADS ("n", ABL bit 16) where two interleaved runs agree: 1024 +1.9 / +2.8 %, 8192 +7.8 / +8.1 % (5.27 -> 5.69 TB/s);
    // 2048 +-1 %, 4096 -3.7 %, 16384 -4 ... -7 % keep plain loads (profiles/r5/ab_k1_f64_ntload_2p*.jsonl)
    MI_K1X(double, 64, 4, false, 16, "n", 1024, 64, 16, 16, 4);
    MI_K1(double, 64, 2, false, 2048, 128, 16, 16, 8);
    // tuning: other tilings of the small whole-row kernels (tools/ab.py --log2n 8 .. 11 min:MI355FFT_VARIANT=v)
    MI_K1V(5, double, 64, 16, false, 256, 16, 16, 16);
    MI_K1V(6, double, 64, 16, false, 256, 16, 16, 16);
    MI_K1V(5, double, 64, 2, false, 512, 64, 8, 8, 8);
    MI_K1V(6, double, 64, 4, false, 512, 64, 8, 8, 8);
    MI_K1V(7, double, 64, 8, false, 512, 32, 16, 8, 4);
    MI_K1V(5, double, 64, 1, false, 1024, 128, 8, 8, 16);
    MI_K1V(6, double, 64, 4, false, 1024, 64, 16, 16, 4);
    MI_K1V(7, double, 64, 2, false, 1024, 128, 16, 8, 8);
    MI_K1V(5, double, 64, 1, false, 2048, 256, 8, 16, 16);
    MI_K1V(6, double, 64, 2, false, 2048, 128, 16, 16, 8);
    MI_K1V(7, double, 64, 1, false, 2048, 128, 16, 16, 8);
    MI_K1(double, 64, 1, false, 4096, 512, 8, 8, 8, 8);  // interleaved A/B (profiles/r2/ab_k1_pow2_f64.jsonl): 5.25 TB/s against 5.00 for 16 x 16 x 16 on 256 threads
    // 2^13, 2^14 in one kernel (split exchange): 10.0 / 10.1 TFLOP/s (5.05 / 4.7 TB/s) against 5.8 / 6.1 for two passes
    MI_K1X(double, 64, 1, true, 16, "n", 8192, 512, 8, 8, 8, 16);  // 5.44 TB/s against 5.22 for 16 x 8 x 8 x 8
    MI_K1(double, 64, 1, true, 16384, 512, 16, 32, 32);
    // round 5, tuning 51: non-temporal loads (ABL bit 16) on the whole-row kernels
    MI_K1ABL(51, 16, double, 64, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(51, 16, double, 64, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(51, 16, double, 64, 1, false, 4096, 512, 8, 8, 8, 8);
    MI_K1ABL(51, 16, double, 64, 1, true, 8192, 512, 8, 8, 8, 16);
    MI_K1ABL(51, 16, double, 64, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(31, 1024, double, 64, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(31, 1024, double, 64, 2, false, 2048, 128, 16, 16, 8);
    MI_K1ABL(31, 1024, double, 64, 1, false, 4096, 512, 8, 8, 8, 8);
    MI_K1ABL(31, 1024, double, 64, 1, true, 8192, 512, 8, 8, 8, 16);
    MI_K1ABL(31, 1024, double, 64, 1, true, 16384, 512, 16, 32, 32);
    MI_K1ABL(30, 128, double, 64, 4, false, 1024, 64, 16, 16, 4);
    MI_K1ABL(30, 128, double, 64, 8, false, 512, 32, 16, 8, 4);
    MI_K1V(3, double, 64, 1, true, 16384, 1024, 16, 16, 8, 8);
    MI_K1V(5, double, 64, 1, false, 4096, 256, 16, 16, 16);  // tuning: the schedules the two above replaced
    MI_K1V(5, double, 64, 1, true, 8192, 512, 16, 8, 8, 8);
    MI_K1V(6, double, 64, 1, true, 8192, 1024, 8, 8, 8, 16);
    MI_K1V(5, double, 64, 1, true, 16384, 1024, 8, 8, 16, 16);
    // one-kernel Bluestein for 4096 < n <= 8192: split exchange, spectrum handed over in registers (960 GB/s at n = 4099 against
    // 800 for the two-kernel form it replaces)
    MI_BSS(double, 64, 1, 12288, 768, 16, 16, 16, 3);
    MI_BSS(double, 64, 1, 16384, 1024, 16, 16, 16, 4);
}
}  // namespace mi355
