// Fused two-pass kernels (launch.h k2f_kernel), Complex<double>: the two-pass power-of-two plans whose two tiles run on the same
// number of threads (2^15 ... 2^21; the later tiles of 2^19 and 2^21 in a wider shape than the plan's).
// The 256 x 256 pair also fuses the first two passes of the three-pass plans 2^23 and 2^24 (+21 % / +8 %, ab_fused3_f64_2p2*.jsonl).  First macro
// argument: 1 = the planner's default: with ONE 16-byte write-through store per element (cx.h st_agent) the fused launch gains
// 20 % / 14 % / 12 % / 26 % at 2^16 / 2^17 / 2^18 / 2^20, results bit-identical to the two-launch plan
// (profiles/r4/ab_fused_f64_16B_2p*.jsonl; with two 8-byte stores per element it LOST 16 - 26 %: ab_fused_f64_2p*.jsonl).
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2f_f64(std::vector<KernelEntry>& reg) {
    using S256 = Sched<256, 16, 16, 16>;
    using S512 = Sched<512, 32, 16, 8, 4>;
    using S1024F = Sched<1024, 64, 16, 16, 4>;
    using S1024L = Sched<1024, 64, 8, 8, 16>;
    using S128 = Sched<128, 8, 16, 8>;
    MI_K2F(1, double, 64, "k2first<256, 16, 16, 16>xF16", 16, false, 0, S256, "k2later<128, 8, 16, 8>xF32", 32, false, 0, S128);                      // 2^15: 5.92 -> 5.00 ms (+18 %, ab_fused_f64_2p15.jsonl)
    // 2^19: the plan's 512-row later tile has 8 columns on 256 threads; here it runs as 16 columns on 512 threads through the split
    // exchange (one block size per launch)
    MI_K2F(1, double, 64, "k2first<1024, 64, 16, 16, 4>xF8t1", 8, true, 1024, S1024F, "k2later<512, 32, 16, 8, 4>xF8", 16, true, 0, S512);              // 2^19: 6.31 -> 5.05 ms (+25 %, ab_fused_f64_2p19.jsonl)
    MI_K2F(1, double, 64, "k2first<256, 16, 16, 16>xF16", 16, false, 0, S256, "k2later<256, 16, 16, 16>xF16", 16, false, 0, S256);                    // 2^16
    MI_K2F(1, double, 64, "k2first<512, 32, 16, 8, 4>xF8", 8, false, 0, S512, "k2later<256, 16, 16, 16>xF16", 16, false, 0, S256);                     // 2^17
    MI_K2F(1, double, 64, "k2first<512, 32, 16, 8, 4>xF8", 8, false, 0, S512, "k2later<512, 32, 16, 8, 4>xF8", 8, false, 0, S512);                      // 2^18
    MI_K2F(1, double, 64, "k2first<1024, 64, 16, 16, 4>xF8t1", 8, true, 1024, S1024F, "k2later<1024, 64, 8, 8, 16>xF8t1", 8, true, 1024, S1024L);     // 2^20
    // 2^21: the 1024-row later tile as 16 columns on 1024 threads (the plan's has 8 on 512; the 2048-row first tile needs 1024):
    // 6.77 -> 6.46 ms (+4.8 %, profiles/r4/ab_fused_f64_2p21.jsonl).  Measured and not compiled: 2^21 as 1024 x 2048 with a 4-column later
    // tile on 512 threads +2 % (ab_fused_rev_f64_2p21.jsonl); 2^22, both 2048-row tiles on 1024 threads, one workgroup per CU: -11 %
    // (ab_fused_f64_2p22.jsonl)
    using S2048 = Sched<2048, 128, 16, 16, 8>;
    MI_K2F(1, double, 64, "k2first<2048, 128, 16, 16, 8>xF8", 8, true, 0, S2048, "k2later<1024, 64, 8, 8, 16>xF8t1", 16, true, 1024, S1024L);        // 2^21
}
}  // namespace mi355
