// Non-power-of-two instantiations, Complex<float>: native mixed radix for BASELINE config 3 (N = 1200),
// Rader for config 4 (N = 1009, inner 1008 = 16 x 9 x 7) and the Bluestein bodies that cover every other length.
// The Rader / Bluestein bodies are VALU-issue bound, not HBM bound: this unit opts into cx.h's two-instruction packed
// complex multiply (measured on MI355X: Rader 1009 +16 %; the HBM-bound pow2 tiles lose 3-6 % with it and keep the
// compiler's own selection).
#define MI355_PK_CMUL 1
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_np2_f32(std::vector<KernelEntry>& reg) {
    MI_K1(float, 32, 4, false, 1200, 120, 10, 10, 12);  // four rows per workgroup: 4.90 TB/s against 4.64 with two (f64 is fastest with two)
    // tuning: other orders / tilings of 1200 (tools/ab.py --n 1200 min:MI355FFT_VARIANT=k)
    MI_K1V(1, float, 32, 2, false, 1200, 120, 12, 10, 10);
    MI_K1V(2, float, 32, 2, false, 1200, 120, 10, 12, 10);
    MI_K1V(3, float, 32, 4, false, 1200, 120, 10, 10, 12);
    MI_K1V(4, float, 32, 1, false, 1200, 120, 10, 10, 12);
    MI_K1V(5, float, 32, 2, false, 1200, 150, 8, 10, 15);
    MI_K1V(6, float, 32, 1, false, 1200, 240, 5, 15, 16);
    MI_K1V(7, float, 32, 3, false, 1200, 80, 15, 16, 5);
    MI_K1V(8, float, 32, 2, false, 1200, 150, 8, 15, 10);
    MI_K1V(9, float, 32, 2, false, 1200, 100, 12, 10, 10);
    // Rader 1009: eight rows per workgroup, one after another, every per-thread table in registers (kernels.h
    // rader_rows_body).  Measured on MI355X (2 GiB of rows): 3.0 TB/s, against 2.3 (one row per workgroup, staged rows,
    // variant 1), 2.1 (variant 2: scatter on load, 128 threads) and 2.4 (variant 4: no prefetch of the next row).
    // (round 2, interleaved A/B of seven schedules of 1008: 14 x 9 x 8 on 126 threads = two full waves: 3.36 TB/s against 3.22)
    // (the PRODUCTION body of 1009 is generated now -- kernels_rader_f32_ns*.hip, tools/gen_rader_kernels.py FORCE: the rows loop without
    // the next-row prefetch at four waves per SIMD, compiled without the SLP vectoriser; the round-1 .. 3 body is tuning variant 72)
    MI_RADERV(72, float, 32, 8, 2, 1008, 126, 14, 9, 8);
    MI_RADERV(73, float, 32, 8, 9, 1008, 126, 14, 9, 8);  // mode 9 = mode 3 + non-temporal row loads (round 5: +1.3 % at config 4's batch, -2 % at 1 GiB; not shipped)
#if defined(MI355_MINIMAL) && !defined(MI355_MINIMAL_RADER)
    MI_RADER(float, 32, 8, 3, 1008, 126, 14, 9, 8);  // `make tuning-min` carries no generated Rader unit: a default body, so that MI355FFT_VARIANT finds the prime
#endif
    MI_RADERV(37, float, 32, 8, 2, 1008, 144, 16, 9, 7);
    MI_RADERV(1, float, 32, 1, 0, 1008, 144, 16, 9, 7);
    MI_RADERV(2, float, 32, 1, 1, 1008, 128, 16, 9, 7);
    MI_RADERV(3, float, 32, 32, 2, 1008, 144, 16, 9, 7);
    MI_RADERV(4, float, 32, 8, 3, 1008, 144, 16, 9, 7);
    MI_RADERV(64, float, 32, 8, 3, 1008, 126, 14, 9, 8);  // the production schedule without the next-row prefetch: four waves per SIMD
    MI_RADERV(65, float, 32, 16, 3, 1008, 126, 14, 9, 8);
    MI_RADERV(66, float, 32, 8, 3, 1008, 126, 8, 9, 14);
    MI_RADERV(67, float, 32, 8, 3, 1008, 112, 16, 9, 7);
    MI_RADERV(68, float, 32, 8, 3, 1008, 126, 16, 9, 7);
    MI_RADERV(69, float, 32, 8, 3, 1008, 84, 12, 12, 7);
    MI_RADERV(70, float, 32, 4, 3, 1008, 126, 14, 9, 8);
    MI_RADERV(71, float, 32, 32, 3, 1008, 126, 14, 9, 8);
    // tuning: other schedules of the inner length 1008 = 2^4 3^2 7 in the rows loop
    MI_RADERV(30, float, 32, 8, 2, 1008, 126, 8, 9, 14);
    MI_RADERV(31, float, 32, 8, 2, 1008, 126, 14, 9, 8);
    MI_RADERV(32, float, 32, 8, 2, 1008, 84, 12, 12, 7);
    MI_RADERV(33, float, 32, 8, 2, 1008, 112, 16, 9, 7);
    MI_RADERV(34, float, 32, 8, 2, 1008, 126, 16, 9, 7);
    MI_RADERV(35, float, 32, 16, 2, 1008, 144, 16, 9, 7);
    MI_RADERV(36, float, 32, 4, 2, 1008, 144, 16, 9, 7);
    // tuning / emulator: side-by-side bodies with the register hand-over (rader_body MODE 5) for primes of every schedule shape
    MI_RADERV(5, float, 32, 2, 5, 1008, 126, 14, 9, 8);
    MI_RADERV(5, float, 32, 32, 5, 96, 8, 16, 6);
    MI_RADERV(5, float, 32, 8, 5, 270, 30, 10, 9, 3);
    MI_RADERV(5, float, 32, 1, 5, 4056, 312, 13, 13, 8, 3);
    MI_RADERV(5, float, 32, 16, 5, 192, 16, 16, 12);
    // tuning / emulator: the rows loop with the register hand-over (rader_rows_body HO, MODE 6)
    MI_RADERV(6, float, 32, 8, 6, 1008, 126, 14, 9, 8);
    MI_RADERV(61, float, 32, 8, 6, 1008, 144, 12, 7, 12);   // palindromic schedule: both transforms run the same radix order
    MI_RADERV(62, float, 32, 16, 6, 1008, 126, 14, 9, 8);  // sixteen rows per workgroup
    MI_RADERV(63, float, 32, 8, 6, 1008, 126, 8, 9, 14);
    MI_RADERV(6, float, 32, 8, 6, 540, 108, 12, 9, 5);
    MI_RADERV(6, float, 32, 8, 6, 4050, 450, 10, 9, 9, 5);
    MI_RADERV(6, float, 32, 8, 6, 192, 64, 8, 8, 3);
    // The other Complex<float> Bluestein bodies live in kernels_bs_f32.hip / kernels_bs57_f32.hip, which are compiled WITHOUT the SLP
    // vectoriser (+1 ... +9.5 % per inner length, profiles/r4/ab_noslp_primes_f32.jsonl); these two inner lengths lose without it
    // (2048: -4.8 %, 1792: -2.6 %) and stay in this unit.
    MI_BS(float, 32, 1, 2048, 128, 8, 16, 16);  // 8.84 against 9.57 for 16 x 16 x 8
    MI_BS(float, 32, 1, 1792, 128, 16, 16, 7);
    // tuning: the orders / thread counts the defaults were measured against, and the large inner lengths without the split exchange
    MI_BSV(1, float, 32, 2, 512, 64, 8, 8, 8);
    MI_BSV(1, float, 32, 1, 1024, 64, 16, 16, 4);
    MI_BSV(1, float, 32, 1, 2048, 128, 16, 16, 8);
    MI_BSV(1, float, 32, 1, 1536, 128, 16, 16, 6);
    MI_BSV(1, float, 32, 1, 3072, 256, 16, 16, 12);
    MI_BSV(1, float, 32, 1, 4096, 256, 16, 16, 16);
    MI_BSV(1, float, 32, 1, 768, 96, 8, 8, 12);
    MI_BSV(1, float, 32, 1, 6144, 512, 16, 16, 24);
    MI_BSV(3, float, 32, 1, 8192, 512, 32, 16, 16);
    reg.push_back(make_pointwise<float>(32));
    reg.push_back(make_dyn_k1<float>(32));
    reg.push_back(make_dyn_rader<float>(32));
    // round 5, tuning 70 / 71 / 72: the shipped body of each inner length (bs_tw1 / bs_pf) + the spectrum multiplier fetched in front of the first
    // transform's last sub-pass (70), + the output chirp in front of the second one's (71), the chirp alone (72)
    MI_BSPV(70, 5, float, 32, 1, 2048, 128, 8, 16, 16);
    MI_BSPV(71, 13, float, 32, 1, 2048, 128, 8, 16, 16);
    MI_BSPV(72, 9, float, 32, 1, 2048, 128, 8, 16, 16);
    // round 5, tuning 80: the shipped body of each inner length with TWO rows per physical thread (launch.h DevExecRows2)
    MI_BSR2V(80, 1, float, 32, 2048, 128, 8, 16, 16);
}
}  // namespace mi355
