// Fused two-pass kernels (launch.h k2f_kernel), Complex<double>: the two-pass power-of-two plans whose two tiles run on the same
// number of threads (2^16, 2^17, 2^18, 2^20; 2^19 and 2^21 pair a 512-thread tile with a 256- / 1024-thread one).  First macro
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
    MI_K2F(1, double, 64, "k2first<256, 16, 16, 16>xF16", 16, false, 0, S256, "k2later<256, 16, 16, 16>xF16", 16, false, 0, S256);                    // 2^16
    MI_K2F(1, double, 64, "k2first<512, 32, 16, 8, 4>xF8", 8, false, 0, S512, "k2later<256, 16, 16, 16>xF16", 16, false, 0, S256);                     // 2^17
    MI_K2F(1, double, 64, "k2first<512, 32, 16, 8, 4>xF8", 8, false, 0, S512, "k2later<512, 32, 16, 8, 4>xF8", 8, false, 0, S512);                      // 2^18
    MI_K2F(1, double, 64, "k2first<1024, 64, 16, 16, 4>xF8t1", 8, true, 1024, S1024F, "k2later<1024, 64, 8, 8, 16>xF8t1", 8, true, 1024, S1024L);     // 2^20
}
}  // namespace mi355
