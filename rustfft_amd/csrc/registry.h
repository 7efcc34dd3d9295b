// Kernel registry: one KernelEntry per compiled (schedule, tile, precision) instantiation.  The host
// planner (plan.cpp) picks entries by length; `launch` hides whether the body runs as a gfx950 kernel
// (HIP build) or under the host emulator of the kernel bodies (tests/emu build).
#pragma once
#include <cstddef>
#include <vector>

namespace mi355 {

enum KernelKind { KIND_K1 = 1, KIND_K2_FIRST = 2, KIND_K2_LATER = 3, KIND_RADER = 4, KIND_BLUESTEIN = 5, KIND_POINTWISE = 6, KIND_DYN_K1 = 7, KIND_DYN_RADER = 8, KIND_K2G_FIRST = 9, KIND_K2G_LATER = 10,
                  // fused multi-kernel Bluestein passes (k2g_body FUSE = 1, 2, 3)
                  KIND_K2G_FIRST_CHIRP = 11, KIND_K2G_LAST_MUL = 12, KIND_K2G_LAST_CHIRP = 13,
                  // two-kernel Bluestein over one-workgroup transforms (k1bs_body STAGE 1 / 2)
                  KIND_BS2_FIRST = 14, KIND_BS2_SECOND = 15,
                  // multi-kernel Rader for primes beyond one workgroup (k2g_body FUSE = 4, 5, 6): the g^j gather rides on the first
                  // load of the first inner transform, the spectrum multiply + x[0] / X[0] step on its last store, the g^-j
                  // scatter on the last store of the second one
                  KIND_K2G_FIRST_GATHER = 16, KIND_K2G_LAST_RMUL = 17, KIND_K2G_LAST_SCATTER = 18,
                  // column-tile passes whose tile height is a PRIME: Rader inside the tile (k2r_body); n = the prime
                  KIND_K2R_FIRST = 19, KIND_K2R_LATER = 20,
                  // both column-tile passes of a two-pass power-of-two plan in ONE launch (launch.h k2f_kernel); n = N, part[] = the names of
                  // the two one-pass kernels it fuses
                  KIND_K2_FUSED = 21 };

struct KernelEntry {
    int kind;
    int prec;  // 32 / 64
    int n;     // transform length handled by one workgroup sequence (N for K1, R for K2)
    int f;     // sequences per workgroup (K1) / tile width in columns (K2)
    int tpf, np, radix[8];
    int threads;
    size_t lds_bytes;
    int tw_total;  // entries in the sub-pass twiddle table
    bool split;
    int aux;      // kind-specific (Rader: the prime p)
    int variant;  // 0 = default; other values are alternative tilings selectable with MI355FFT_VARIANT (tuning)
    const char* name;
    void (*launch)(const void* params, long long grid, void* stream);
    int (*prepare)();  // one-time setup (dynamic-LDS attribute); returns 0 on success
    const char* part[2];  // KIND_K2_FUSED: names of the first-pass and second-pass kernels this entry fuses
    int f2;               // KIND_K2_FUSED: tile width of the second pass (f = the first pass's)
    int (*blocks_per_cu)();  // KIND_K2_FUSED: workgroups of this kernel one compute unit holds (the runtime's occupancy calculation over the
                             // kernel's registers, LDS and block size); nullptr elsewhere
};

std::vector<KernelEntry>& registry();
void ensure_registry();  // populate once (thread-safe)

// each kernels_*.{hip,cpp} translation unit contributes through one of these
void register_k1_f32(std::vector<KernelEntry>&);
void register_k1_f64(std::vector<KernelEntry>&);
void register_k2_f32(std::vector<KernelEntry>&);
void register_k2_f64(std::vector<KernelEntry>&);
void register_k2f_f32(std::vector<KernelEntry>&);  // fused two-pass kernels (kernels_k2f_f32.hip)
void register_k2f_f64(std::vector<KernelEntry>&);
void register_np2_f32(std::vector<KernelEntry>&);  // non-power-of-two: mixed radix, Rader, Bluestein
void register_np2_f64(std::vector<KernelEntry>&);
void register_bs57_f32(std::vector<KernelEntry>&);  // Bluestein bodies over 5 * 2^k and 7 * 2^k
void register_bs_f32(std::vector<KernelEntry>&);    // Complex<float> Bluestein bodies over 2^k and 3 * 2^k (compiled without the SLP vectoriser)
void register_bs57_f64(std::vector<KernelEntry>&);
// generated: large-N pass kernels for 7-smooth tile heights (tools/gen_k2g_kernels.py)
void register_k2g_f32_0(std::vector<KernelEntry>&);
void register_k2g_f32_1(std::vector<KernelEntry>&);
void register_k2g_f32_2(std::vector<KernelEntry>&);
void register_k2g_f32_3(std::vector<KernelEntry>&);
void register_k2g_f32_4(std::vector<KernelEntry>&);
void register_k2g_f32_5(std::vector<KernelEntry>&);
void register_k2g_f32_6(std::vector<KernelEntry>&);
void register_k2g_f32_7(std::vector<KernelEntry>&);
void register_k2g_f32_ns0(std::vector<KernelEntry>&);  // the Complex<float> tile heights compiled without the SLP vectoriser (tools/k2g_noslp_choice.json)
void register_k2g_f32_ns1(std::vector<KernelEntry>&);
void register_k2g_f32_ns2(std::vector<KernelEntry>&);
void register_k2g_f32_ns3(std::vector<KernelEntry>&);
void register_k2g_f64_0(std::vector<KernelEntry>&);
void register_k2g_f64_1(std::vector<KernelEntry>&);
void register_k2g_f64_2(std::vector<KernelEntry>&);
void register_k2g_f64_3(std::vector<KernelEntry>&);
void register_k2g_f64_4(std::vector<KernelEntry>&);
void register_k2g_f64_5(std::vector<KernelEntry>&);
void register_k2g_f64_6(std::vector<KernelEntry>&);
void register_k2g_f64_7(std::vector<KernelEntry>&);
void register_k2gr_f32_0(std::vector<KernelEntry>&);  // generated: the Rader-fused forms of the same tiles
void register_k2gr_f32_1(std::vector<KernelEntry>&);
void register_k2gr_f32_2(std::vector<KernelEntry>&);
void register_k2gr_f32_3(std::vector<KernelEntry>&);
void register_k2gr_f64_0(std::vector<KernelEntry>&);
void register_k2gr_f64_1(std::vector<KernelEntry>&);
void register_k2gr_f64_2(std::vector<KernelEntry>&);
void register_k2gr_f64_3(std::vector<KernelEntry>&);
void register_k2r_f32(std::vector<KernelEntry>&);  // generated: prime tile heights (tools/gen_k2g_kernels.py)
void register_k2r_f64(std::vector<KernelEntry>&);
// generated: compiled schedules for the 13-smooth lengths in (16, 4096] (tools/gen_smooth_kernels.py)
void register_smooth_f32_0(std::vector<KernelEntry>&);
void register_smooth_f32_1(std::vector<KernelEntry>&);
void register_smooth_f32_2(std::vector<KernelEntry>&);
void register_smooth_f32_3(std::vector<KernelEntry>&);
void register_smooth_f32_4(std::vector<KernelEntry>&);
void register_smooth_f32_5(std::vector<KernelEntry>&);
void register_smooth_f32_6(std::vector<KernelEntry>&);
void register_smooth_f32_7(std::vector<KernelEntry>&);
void register_smooth_f64_0(std::vector<KernelEntry>&);
void register_smooth_f64_1(std::vector<KernelEntry>&);
void register_smooth_f64_2(std::vector<KernelEntry>&);
void register_smooth_f64_3(std::vector<KernelEntry>&);
void register_smooth_f64_4(std::vector<KernelEntry>&);
void register_smooth_f64_5(std::vector<KernelEntry>&);
void register_smooth_f64_6(std::vector<KernelEntry>&);
void register_smooth_f64_7(std::vector<KernelEntry>&);

// generated: single-kernel schedules for the 7-smooth lengths in (4096, 16384] (tools/gen_smooth_kernels.py main_big)
void register_smooth2_f32_0(std::vector<KernelEntry>&);
void register_smooth2_f32_1(std::vector<KernelEntry>&);
void register_smooth2_f32_2(std::vector<KernelEntry>&);
void register_smooth2_f32_3(std::vector<KernelEntry>&);
void register_smooth2_f64_0(std::vector<KernelEntry>&);
void register_smooth2_f64_1(std::vector<KernelEntry>&);
void register_smooth2_f64_2(std::vector<KernelEntry>&);
void register_smooth2_f64_3(std::vector<KernelEntry>&);

// generated: single-kernel schedules for the 13-smooth lengths in (4096, 16384] (Complex<f32>: 32768] with a factor 11 / 13 (tools/gen_smooth_kernels.py main_big13)
void register_smooth4_f32_0(std::vector<KernelEntry>&);
void register_smooth4_f32_1(std::vector<KernelEntry>&);
void register_smooth4_f32_ns0(std::vector<KernelEntry>&);
void register_smooth4_f32_ns1(std::vector<KernelEntry>&);
void register_smooth4_f32_ns2(std::vector<KernelEntry>&);
void register_smooth4_f32_ns3(std::vector<KernelEntry>&);
void register_smooth4_f32_ns4(std::vector<KernelEntry>&);
void register_smooth4_f32_ns5(std::vector<KernelEntry>&);
void register_smooth4_f32_ns6(std::vector<KernelEntry>&);
void register_smooth4_f32_ns7(std::vector<KernelEntry>&);
void register_smooth4_f32_ns8(std::vector<KernelEntry>&);
void register_smooth4_f32_ns9(std::vector<KernelEntry>&);
void register_smooth4_f64_0(std::vector<KernelEntry>&);
void register_smooth4_f64_1(std::vector<KernelEntry>&);
void register_smooth4_f64_2(std::vector<KernelEntry>&);
void register_smooth4_f64_3(std::vector<KernelEntry>&);
void register_smooth4_f64_4(std::vector<KernelEntry>&);
void register_smooth4_f64_5(std::vector<KernelEntry>&);
void register_smooth4_f64_6(std::vector<KernelEntry>&);
void register_smooth4_f64_7(std::vector<KernelEntry>&);
void register_smooth5_f32_ns0(std::vector<KernelEntry>&);  // round 5: lengths with a prime factor 17 .. 31 above the smooth3 limits, up to 8192 (tools/gen_smooth_kernels.py main_primes_big)
void register_smooth5_f32_ns1(std::vector<KernelEntry>&);
void register_smooth5_f32_ns2(std::vector<KernelEntry>&);
void register_smooth5_f32_ns3(std::vector<KernelEntry>&);
void register_smooth5_f32_ns4(std::vector<KernelEntry>&);
void register_smooth5_f32_ns5(std::vector<KernelEntry>&);
void register_smooth5_f32_ns6(std::vector<KernelEntry>&);
void register_smooth5_f32_ns7(std::vector<KernelEntry>&);
void register_smooth5_f64_0(std::vector<KernelEntry>&);
void register_smooth5_f64_1(std::vector<KernelEntry>&);
void register_smooth5_f64_2(std::vector<KernelEntry>&);
void register_smooth5_f64_3(std::vector<KernelEntry>&);
void register_smooth5_f64_4(std::vector<KernelEntry>&);
void register_smooth5_f64_5(std::vector<KernelEntry>&);
void register_smooth5_f64_6(std::vector<KernelEntry>&);
void register_smooth5_f64_7(std::vector<KernelEntry>&);

// generated: compiled schedules for the lengths <= 2048 (f64: 1024) with a prime factor 17 .. 31 (tools/gen_smooth_kernels.py main_primes)
void register_smooth3_f32_0(std::vector<KernelEntry>&);
void register_smooth3_f32_1(std::vector<KernelEntry>&);
void register_smooth3_f32_2(std::vector<KernelEntry>&);
void register_smooth3_f32_3(std::vector<KernelEntry>&);
void register_smooth3_f32_4(std::vector<KernelEntry>&);
void register_smooth3_f32_5(std::vector<KernelEntry>&);
void register_smooth3_f32_6(std::vector<KernelEntry>&);
void register_smooth3_f32_7(std::vector<KernelEntry>&);
void register_smooth3_f32_8(std::vector<KernelEntry>&);
void register_smooth3_f32_9(std::vector<KernelEntry>&);
void register_smooth3_f32_10(std::vector<KernelEntry>&);
void register_smooth3_f32_11(std::vector<KernelEntry>&);
void register_smooth3_f32_12(std::vector<KernelEntry>&);
void register_smooth3_f32_13(std::vector<KernelEntry>&);
void register_smooth3_f64_0(std::vector<KernelEntry>&);
void register_smooth3_f64_1(std::vector<KernelEntry>&);
void register_smooth3_f64_2(std::vector<KernelEntry>&);
void register_smooth3_f64_3(std::vector<KernelEntry>&);
void register_smooth3_f64_4(std::vector<KernelEntry>&);
void register_smooth3_f64_5(std::vector<KernelEntry>&);
void register_smooth3_f64_6(std::vector<KernelEntry>&);
void register_smooth3_f64_7(std::vector<KernelEntry>&);
// generated: the Complex<float> schedules that measured faster without the SLP vectoriser (tools/smooth_noslp_choice.json; Makefile NOSLP)
void register_smooth_f32_ns0(std::vector<KernelEntry>&);
void register_smooth2_f32_ns0(std::vector<KernelEntry>&);
void register_smooth2_f32_ns1(std::vector<KernelEntry>&);
void register_smooth2_f32_ns2(std::vector<KernelEntry>&);
void register_smooth3_f32_ns0(std::vector<KernelEntry>&);
void register_smooth3_f32_ns1(std::vector<KernelEntry>&);
void register_smooth3_f32_ns2(std::vector<KernelEntry>&);
void register_smooth3_f32_ns3(std::vector<KernelEntry>&);
void register_smooth3_f32_ns4(std::vector<KernelEntry>&);
void register_smooth3_f32_ns5(std::vector<KernelEntry>&);
void register_smooth3_f32_ns6(std::vector<KernelEntry>&);
// generated: compiled Rader bodies for the primes <= 4096 with 13-smooth p - 1 (tools/gen_rader_kernels.py)
void register_rader_f32_0(std::vector<KernelEntry>&);
void register_rader_f32_1(std::vector<KernelEntry>&);
void register_rader_f32_2(std::vector<KernelEntry>&);
void register_rader_f32_3(std::vector<KernelEntry>&);
void register_rader_f32_ns0(std::vector<KernelEntry>&);  // the Complex<float> bodies compiled without the SLP vectoriser (tools/gen_rader_kernels.py NOSLP_F32)
void register_rader_f32_ns1(std::vector<KernelEntry>&);
void register_rader_f64_0(std::vector<KernelEntry>&);
void register_rader_f64_1(std::vector<KernelEntry>&);
void register_rader_f64_2(std::vector<KernelEntry>&);
void register_rader_f64_3(std::vector<KernelEntry>&);

template <class S> inline void fill_sched(KernelEntry& e) {
    e.tpf = S::TPF;
    e.np = S::NP;
    for (int i = 0; i < 8; ++i) e.radix[i] = i < S::NP ? S::R[i] : 0;
    e.tw_total = S::tw_total();
}

}  // namespace mi355
