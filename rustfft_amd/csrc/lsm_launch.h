// Launch thunks of the LDS stage machine (lsm.h): one registry entry per precision; the block size (256 / 512 / 1024 threads) and the
// LDS bytes are properties of the PROGRAM and travel in the parameter block.
#pragma once
#include "launch.h"
#include "lsm.h"

namespace mi355 {

constexpr int KIND_LSM = 22;  // registry.h KernelKind continues here

#if !defined(MI355_EMU)
template <class T, int NT> struct LsmDevExec {
    unsigned w[4 * kLsmItems];
    template <class Fn> __device__ __forceinline__ void for_threads(Fn&& fn) { fn((int)threadIdx.x, (cx<T>*)nullptr); }
    __device__ __forceinline__ unsigned* words(int) { return w; }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
};
// four waves per SIMD (128 VGPRs) is what hides the LDS round trips and the barriers of a program whose stages are short
// Waves per SIMD the register allocator is asked for: FOUR for Complex<f32> (the unit is compiled without the SLP vectoriser -- Makefile NOSLP:
// 115 VGPRs, nothing spilled; with it 142 VGPRs left alone, 19 spilled at 128) and TWO for Complex<f64> (183 VGPRs, nothing spilled).  The stages
// are short and separated by barriers, so resident waves are what hides their LDS round trips -- but a SPILL costs more than a wave buys:
// measured with one build against the other, Complex<f32> 128 against 142 VGPRs: 512-thread programs x1.35 .. x1.66, 64- / 128-thread x1.14 ..
// x1.19 (profiles/r6/lsm_w4_ab_f32.jsonl); Complex<f64> at three waves (168 VGPRs, 46 spilled) against two: the two-wave build x1.07 .. x1.40 at
// 17 of 17 lengths (profiles/r6/lsm_w2_ab_f64.jsonl).
#if !defined(MI355_LSM_WAVES)
#define MI355_LSM_WAVES(T) (sizeof(T) == 4 ? 4 : 2)
#endif
template <class T, int NT> __global__ __launch_bounds__(NT, MI355_LSM_WAVES(T)) void lsm_kernel(LsmParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LsmDevExec<T, NT> ex;
    lsm_body<T, NT>(ex, p, (long long)blockIdx.x, smem);
}
template <class T> const void* lsm_fn(int nt) {
    if (nt == 64) return (const void*)lsm_kernel<T, 64>;
    if (nt == 128) return (const void*)lsm_kernel<T, 128>;
    if (nt == 256) return (const void*)lsm_kernel<T, 256>;
    if (nt == 512) return (const void*)lsm_kernel<T, 512>;
    if (nt == 1024) return (const void*)lsm_kernel<T, 1024>;
    return nullptr;
}
template <class T> KernelEntry make_lsm(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_LSM;
    e.prec = prec;
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        const LsmParams<T>* p = (const LsmParams<T>*)params;
        void* args[] = {const_cast<void*>(params)};
        if (const void* fn = lsm_fn<T>(p->nt)) (void)hipLaunchKernel(fn, dim3((unsigned)grid), dim3((unsigned)p->nt), args, (size_t)p->lds_bytes, (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        int rc = 0;
        for (int nt : {64, 128, 256, 512, 1024})
            if (const void* fn = lsm_fn<T>(nt)) {
                const int r = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                rc = rc ? rc : r;
            }
        return rc;
    };
    return e;
}
#else
template <class T> struct LsmHostExec {
    int nt;
    bool reverse;
    std::vector<cx<T>> regs;
    std::vector<unsigned> wd;
    explicit LsmHostExec(int n) : nt(n), reverse(emu_reverse_order()), regs((size_t)n * kLsmEmax, cx<T>{0, 0}), wd((size_t)n * 4 * kLsmItems, 0u) {}
    template <class Fn> void for_threads(Fn&& fn) {
        if (reverse)
            for (int t = nt - 1; t >= 0; --t) fn(t, regs.data() + (size_t)t * kLsmEmax);
        else
            for (int t = 0; t < nt; ++t) fn(t, regs.data() + (size_t)t * kLsmEmax);
    }
    unsigned* words(int tid) { return wd.data() + (size_t)tid * 4 * kLsmItems; }
    void barrier() {}
};
template <class T> KernelEntry make_lsm(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_LSM;
    e.prec = prec;
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        const LsmParams<T>* p = (const LsmParams<T>*)params;
        std::vector<char> lds((size_t)p->lds_bytes + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            LsmHostExec<T> ex(p->nt);
            if (p->nt == 64)
                lsm_body<T, 64>(ex, *p, b, lds.data());
            else if (p->nt == 128)
                lsm_body<T, 128>(ex, *p, b, lds.data());
            else if (p->nt == 256)
                lsm_body<T, 256>(ex, *p, b, lds.data());
            else if (p->nt == 512)
                lsm_body<T, 512>(ex, *p, b, lds.data());
            else
                lsm_body<T, 1024>(ex, *p, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
#endif

void register_lsm_f32(std::vector<KernelEntry>&);
void register_lsm_f64(std::vector<KernelEntry>&);

}  // namespace mi355
