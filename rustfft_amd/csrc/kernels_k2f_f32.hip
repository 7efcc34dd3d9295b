// Fused two-pass kernels (launch.h k2f_kernel): both column-tile passes of a two-pass power-of-two plan in ONE launch, the
// intermediate in a cache-resident ring.  One entry per two-pass plan 2^16 .. 2^22, Complex<float> (the 256 x 256 pair also fuses the first two passes of the three-pass plans 2^23 and 2^24
// over units of a transform: kernels_params.h); each names the two
// one-pass kernels (kernels_k2_f32.hip) whose tiles it runs.  First macro argument: 1 = the planner's default for that length
// (interleaved A/B against the two-launch plan, three plan instances per arm, the final lag / ring rule,
// profiles/r4/ab_fused_final_2p*.jsonl: 2^16 +10.9 %, 2^19 +13.6 %, 2^20 +18.3 % (12.66 -> 10.71 ms per forward + inverse pair); steady-state
// forward-only launches at 4 GiB, profiles/r4/fused_warmup_2p*.jsonl: 2^16 +8 %, 2^17 +11 %, 2^19 +14 %, 2^20 +11 %, 2^18 (512 x 512) +-0 -- that split stays off, 2^18 runs as 256 x 1024 (below);
// 2^21 -4 %, 2^22 -13 % -- the 2048-row tiles spill in the fused kernel and 2^22's second pass has to run on 8-column tiles).
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2f_f32(std::vector<KernelEntry>& reg) {
    using S256 = Sched<256, 16, 16, 16>;
    using S512F = Sched<512, 32, 16, 8, 4>;
    using S512L = Sched<512, 16, 8, 8, 8>;
    using S1024 = Sched<1024, 32, 8, 8, 16>;
    using S2048 = Sched<2048, 64, 8, 16, 16>;
    MI_K2F(1, float, 32, "k2first<256, 16, 16, 16>xF32", 32, false, 0, S256, "k2later<256, 16, 16, 16>xF32", 32, false, 0, S256);              // 2^16
    // 2^17 in the REVERSED pass order (256-row tile first): two launches 12.32 -> 11.93 ms per pair, fused 10.89 (+13 %); the
    // standard order fused: 12.36 (profiles/r4/ab_fused_rev_2p17.jsonl).  An AUTO entry that exists for the reversed order only makes
    // the planner reverse the passes (plan.cpp choose_macro_radices).
    MI_K2F(1, float, 32, "k2first<256, 16, 16, 16>xF32", 32, false, 0, S256, "k2later<512, 16, 8, 8, 8>xF32t", 32, true, 128, S512L);        // 2^17
    MI_K2F(0, float, 32, "k2first<512, 32, 16, 8, 4>xF16t", 16, false, 128, S512F, "k2later<512, 16, 8, 8, 8>xF32t", 32, true, 128, S512L);   // 2^18
    // 2^18 as 256 x 1024: forward-only, 4 GiB, three plan instances per arm (profiles/r4/ab_fused_2p18_splits.jsonl): 512 x 512 two launches
    // 5.92 ms, fused 5.81; 1024 x 256 6.18 / 5.69; 256 x 1024 6.33 / 5.40 (+9.6 % over the balanced two-launch plan).  The planner takes
    // the split that has a default fused kernel (plan.cpp choose_macro_radices).
    MI_K2F(1, float, 32, "k2first<256, 16, 16, 16>xF32", 32, false, 0, S256, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024);
    MI_K2F(1, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<512, 16, 8, 8, 8>xF32t", 32, true, 128, S512L);   // 2^19
    MI_K2F(1, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024); // 2^20
    MI_K2F(0, float, 32, "k2first<2048, 64, 8, 16, 16>xF8", 8, true, 0, S2048, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024);     // 2^21
    // 2^21 as 1024 x 2048, the later pass on 8-column tiles of 512 threads (the plan's later tile has 16 columns on 1024 threads; one launch
    // has one block size): 7.00 ms (2048 x 1024, two launches) -> 6.49 (+7.8 %, profiles/r4/ab_fused_rev_2p21.jsonl); the planner takes
    // this split because it is the one with a default fused kernel.  (The standard order below spills in its 2048-row first tile: -4 %.)
    MI_K2F(1, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<2048, 128, 8, 16, 16>xF16p2", 8, true, 0, S2048);
#if defined(MI355_TUNING)
    // other splits of 2^16 ... 2^20 (MI355FFT_R0 selects them): candidates
    using S128 = Sched<128, 8, 16, 8>;
    MI_K2F(0, float, 32, "k2first<128, 8, 16, 8>xF64", 64, false, 0, S128, "k2later<512, 16, 8, 8, 8>xF32t", 32, true, 128, S512L);            // 2^16 = 128 x 512
    MI_K2F(0, float, 32, "k2first<512, 32, 16, 8, 4>xF16t", 16, false, 128, S512F, "k2later<128, 8, 16, 8>xF64t", 64, false, 128, S128);      // 2^16 = 512 x 128
    MI_K2F(0, float, 32, "k2first<128, 8, 16, 8>xF64", 64, false, 0, S128, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024);         // 2^17 = 128 x 1024
    MI_K2F(0, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<128, 8, 16, 8>xF64t", 64, false, 128, S128);      // 2^17 = 1024 x 128
    MI_K2F(0, float, 32, "k2first<128, 8, 16, 8>xF64", 64, false, 0, S128, "k2later<2048, 128, 8, 16, 16>xF16p2", 8, true, 0, S2048);         // 2^18 = 128 x 2048
    MI_K2F(0, float, 32, "k2first<256, 16, 16, 16>xF32", 32, false, 0, S256, "k2later<2048, 128, 8, 16, 16>xF16p2", 8, true, 0, S2048);       // 2^19 = 256 x 2048
    MI_K2F(0, float, 32, "k2first<512, 32, 16, 8, 4>xF16t", 16, false, 128, S512F, "k2later<2048, 128, 8, 16, 16>xF16p2", 8, true, 0, S2048);  // 2^20 = 512 x 2048
#endif
    MI_K2FR(0, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024);
    MI_K2FR(3, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024);
    // 2^22: the one-pass plan runs its second pass on 16-column tiles of 1024 threads; a fused launch has ONE block size, so
    // the second pass runs here on 8-column tiles of 512 threads (two workgroups per CU, like the first pass)
    MI_K2F(0, float, 32, "k2first<2048, 64, 8, 16, 16>xF8", 8, true, 0, S2048, "k2later<2048, 128, 8, 16, 16>xF16p2", 8, true, 0, S2048);         // 2^22
    // tuning 23: the 256 x 256 pair (2^16; the fused front of 2^23 / 2^24) on 16-column tiles of 256 threads (the shape the Complex<f64> pair has):
    // 2^16 5.65 -> 6.50 ms, 2^23 8.88 -> 9.47, 2^24 9.33 -> 10.02 (profiles/r4/ab_fused_f16tiles_2p*.jsonl): slower
    MI_K2FV(23, float, 32, "k2first<256, 16, 16, 16>xF32", 16, false, 0, S256, "k2later<256, 16, 16, 16>xF32", 16, false, 0, S256);
    // tuning 21: 2^20 with the SECOND pass as two columns per lane (16-byte ring loads and output stores)
    using S1024P = Sched<1024, 64, 8, 8, 16>;
    MI_K2FV(21, float, 32, "k2first<1024, 32, 8, 8, 16>xF16t", 16, true, 128, S1024, "k2later<1024, 32, 8, 8, 16>xF16t", 16, true, 4224, S1024P);
}
}  // namespace mi355
