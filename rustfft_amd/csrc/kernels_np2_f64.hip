// Non-power-of-two instantiations, Complex<double>: native mixed radix for BASELINE config 3 (N = 1200),
// Rader for config 4 (N = 1009, inner 1008 = 16 x 9 x 7) and the Bluestein bodies that cover every other length.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_np2_f64(std::vector<KernelEntry>& reg) {
    // round 5: rows read AND written with non-temporal accesses ("n", ABL bits 16 + 32): 5.48 - 5.62 -> 5.93 - 5.97 TB/s in three interleaved
    // runs (+6 ... +7 %; loads only: +2 ... +3 %), bit-identical (profiles/r5/ab_c3_nt*.jsonl) -- the one whole-row kernel where the
    // stores gain too (the Complex<f32> power-of-two rows lose up to 31 % with them)
    MI_K1X(double, 64, 2, false, 48, "n", 1200, 120, 10, 10, 12);
    // tuning: other orders / tilings of 1200 (tools/ab.py --n 1200 --dtype f64 min:MI355FFT_VARIANT=k)
    MI_K1ABL(51, 16, double, 64, 2, false, 1200, 120, 10, 10, 12);  // round 5: non-temporal loads (51) / loads and stores (50)
    MI_K1ABL(50, 48, double, 64, 2, false, 1200, 120, 10, 10, 12);
    MI_K1V(1, double, 64, 2, false, 1200, 120, 12, 10, 10);
    MI_K1V(2, double, 64, 2, false, 1200, 120, 10, 12, 10);
    MI_K1V(3, double, 64, 4, false, 1200, 120, 10, 10, 12);
    MI_K1V(4, double, 64, 1, false, 1200, 120, 10, 10, 12);
    MI_K1V(5, double, 64, 2, false, 1200, 150, 8, 10, 15);
    MI_K1V(6, double, 64, 1, false, 1200, 240, 5, 15, 16);
    MI_K1V(7, double, 64, 3, false, 1200, 80, 15, 16, 5);
    MI_K1V(8, double, 64, 2, false, 1200, 150, 8, 15, 10);
    MI_K1V(9, double, 64, 2, false, 1200, 100, 12, 10, 10);
    // f64: scatter on load, one row per workgroup.  (Rounds 1 - 2: the rows loop without the next-row prefetch, 2.6 TB/s against
    // 2.25 for this body; with the row loads batched -- round 3 -- this body runs 3.23 TB/s against 2.72 in a one-process A/B,
    // profiles/r3/rader_mode1_back_f64.jsonl.)
    MI_RADER(double, 64, 1, 1, 1008, 144, 16, 9, 7);
    MI_RADERV(1, double, 64, 4, 0, 1008, 144, 16, 9, 7);
    MI_RADERV(2, double, 64, 8, 3, 1008, 144, 16, 9, 7);
    MI_RADERV(3, double, 64, 8, 2, 1008, 144, 16, 9, 7);
    // tuning / emulator: side-by-side bodies with the register hand-over (rader_body MODE 5) for primes of every schedule shape
    MI_RADERV(5, double, 64, 2, 5, 1008, 126, 14, 9, 8);
    MI_RADERV(5, double, 64, 32, 5, 96, 8, 16, 6);
    MI_RADERV(5, double, 64, 8, 5, 270, 30, 10, 9, 3);
    MI_RADERV(5, double, 64, 1, 5, 4056, 312, 13, 13, 8, 3);
    MI_RADERV(5, double, 64, 16, 5, 192, 16, 16, 12);
    // tuning / emulator: the rows loop with the register hand-over (rader_rows_body HO, MODE 6)
    MI_RADERV(6, double, 64, 8, 6, 1008, 126, 14, 9, 8);
    MI_RADERV(6, double, 64, 8, 6, 540, 108, 12, 9, 5);
    MI_RADERV(6, double, 64, 8, 6, 4050, 450, 10, 9, 9, 5);
    MI_RADERV(6, double, 64, 8, 6, 192, 64, 8, 8, 3);
    MI_BS_LIST(double, 64);
    MI_BS(double, 64, 2, 512, 64, 8, 8, 8);
    MI_BS(double, 64, 2, 1024, 128, 8, 8, 16);  // 6.55 ns per row against 8.04 for 16 x 16 x 4
    MI_BS(double, 64, 2, 2048, 128, 16, 16, 8);
    MI_BS(double, 64, 1, 8192, 512, 16, 8, 8, 8);  // the radix-32 last pass spills in f64
    MI_BS_LIST3_F64(double, 64);
    // tuning: the orders the defaults were measured against (see kernels_np2_f32.hip)
    MI_BSV(1, double, 64, 1, 512, 64, 8, 8, 8);
    MI_BSV(1, double, 64, 4, 1024, 64, 16, 16, 4);
    MI_BSV(1, double, 64, 2, 1536, 128, 16, 16, 6);
    MI_BSV(1, double, 64, 1, 3072, 256, 16, 16, 12);
    MI_BSV(1, double, 64, 1, 4096, 256, 16, 16, 16);
    MI_BSV(1, double, 64, 1, 768, 96, 8, 8, 12);
    reg.push_back(make_pointwise<double>(64));
    reg.push_back(make_dyn_k1<double>(64));
    reg.push_back(make_dyn_rader<double>(64));
    // round 5, tuning 60 .. 63 (Complex<f64>): sub-pass factors fetched one exchange ahead (60), every table but the last staged in LDS (61), both (62),
    // sub-pass 1 staged + the others fetched ahead (63) -- kernels.h bluestein_body PF
    MI_BSPV(60, 1, double, 64, 2, 2048, 128, 16, 16, 8);
    MI_BSPV(61, 2, double, 64, 2, 2048, 128, 16, 16, 8);
    MI_BSPV(62, 3, double, 64, 2, 2048, 128, 16, 16, 8);
    MI_BSPV(63, 17, double, 64, 2, 2048, 128, 16, 16, 8);
    MI_BSPV(60, 1, double, 64, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(61, 2, double, 64, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(62, 3, double, 64, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(63, 17, double, 64, 1, 3072, 256, 12, 16, 16);
    MI_BSPV(60, 1, double, 64, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(61, 2, double, 64, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(62, 3, double, 64, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(63, 17, double, 64, 1, 4096, 512, 8, 8, 8, 8);
    MI_BSPV(60, 1, double, 64, 1, 6144, 512, 16, 16, 24);
    MI_BSPV(61, 2, double, 64, 1, 6144, 512, 16, 16, 24);
    MI_BSPV(62, 3, double, 64, 1, 6144, 512, 16, 16, 24);
    MI_BSPV(63, 17, double, 64, 1, 6144, 512, 16, 16, 24);
    MI_BSPV(60, 1, double, 64, 1, 8192, 512, 16, 8, 8, 8);
    MI_BSPV(61, 2, double, 64, 1, 8192, 512, 16, 8, 8, 8);
    MI_BSPV(62, 3, double, 64, 1, 8192, 512, 16, 8, 8, 8);
    MI_BSPV(63, 17, double, 64, 1, 8192, 512, 16, 8, 8, 8);
}
}  // namespace mi355
