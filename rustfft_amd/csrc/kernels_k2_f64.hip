// K2 (column-tile passes of the large-N decomposition) instantiations, Complex<double>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2_f64(std::vector<KernelEntry>& reg) {
    MI_K2(double, 64, 32, false, 64, 8, 8, 8);
    MI_K2(double, 64, 32, false, 128, 8, 16, 8);
    MI_K2(double, 64, 16, false, 256, 16, 16, 16);
    MI_K2(double, 64, 8, false, 512, 32, 16, 8, 4);
    // (16-column split tile for 512 rows: measured 5 % slower in f64, not instantiated)
    // 1024-row tile: 8 columns (128-byte segments) on 512 threads x 16 values, split exchange, two workgroups per CU.  Round 1 ran it
    // on 256 threads x 32 values (168 - 180 VGPRs, 8 waves per CU): interleaved A/B at 2^20 f64, first / later pass 3.95 / 4.09 TB/s
    // against 5.20 / 4.67 (radices 16 16 4) and 5.10 / 4.83 (8 8 16) with 16 values per thread -- each kind takes its better schedule.
    // round 3: sub-pass 1's twiddle table staged in LDS ("t1"; the last sub-pass's 16 KB table does not fit beside two tiles):
    // first pass 5.18 -> 5.29 TB/s, later pass 4.60 - 4.80 -> 4.97 - 5.13 (profiles/r3/ab_twl_round2.jsonl); no gain on the smaller f64 tiles
    MI_K2X_FIRST(double, 64, 8, true, 1024, "t1", 1024, 64, 16, 16, 4);
    MI_K2X_LATER(double, 64, 8, true, 1024, "t1", 1024, 64, 8, 8, 16);
#if defined(MI355_TUNING)
    MI_K2_FIRST(double, 64, 8, true, 1024, 64, 16, 16, 4);
    reg.back().variant = 40;
    MI_K2_LATER(double, 64, 8, true, 1024, 64, 8, 8, 16);
    reg.back().variant = 40;
#endif
    MI_K2V(39, double, 64, 8, true, 1024, 32, 16, 16, 4);
    MI_K2V(42, double, 64, 8, true, 512, 32, 16, 8, 4);
    // 2048-row tile (2^21, 2^22 in two passes instead of three): 8 columns on 1024 threads x 16 values, one workgroup per CU
    MI_K2(double, 64, 8, true, 2048, 128, 16, 16, 8);
    // sub-pass twiddle tables staged in LDS (tuning variants 20: all that fit; 21 / 22: alternatives)
    MI_K2ABL(20, 128, double, 64, 32, false, 64, 8, 8, 8);
    MI_K2ABL(20, 128, double, 64, 32, false, 128, 8, 16, 8);
    MI_K2ABL(20, 128, double, 64, 16, false, 256, 16, 16, 16);
    MI_K2ABL(20, 128, double, 64, 8, false, 512, 32, 16, 8, 4);
#if defined(MI355_TUNING)
    MI_K2X_FIRST(double, 64, 8, true, 1024, "t1", 1024, 64, 16, 16, 4);
    reg.back().variant = 20;
    MI_K2X_LATER(double, 64, 8, true, 1024, "t1", 1024, 64, 8, 8, 16);
    reg.back().variant = 20;
    MI_K2X_FIRST(double, 64, 8, true, 2048, "tl", 1024, 64, 16, 16, 4);
    reg.back().variant = 21;
    MI_K2X_LATER(double, 64, 8, true, 1024, "t1", 1024, 64, 8, 8, 16);
    reg.back().variant = 21;
#endif
    MI_K2ABL(20, 1024, double, 64, 8, true, 2048, 128, 16, 16, 8);
}
}  // namespace mi355
