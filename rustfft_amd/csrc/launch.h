// Executors + launch thunks.  HIP build: DevExec + __global__ wrappers.  Emulator build
// (MI355_EMU, tests/emu only): HostExec runs every thread of a workgroup phase by phase on the CPU,
// which checks all index arithmetic of the kernel bodies without a GPU.
#pragma once
#include <cstdlib>
#include <string>
#include <vector>

#include "kernels.h"
#include "registry.h"

namespace mi355 {

#if !defined(MI355_EMU)
// -------------------------------------------------------------------------------- gfx950
template <class T, int NREG> struct DevExec {
    cx<T> v[NREG];
    template <class Fn> __device__ __forceinline__ void for_threads(Fn&& fn) { fn((int)threadIdx.x, v); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    __device__ __forceinline__ void relaunder() {}
    // engine.h pair-fused schedules: slot m 8 + k of lanes 32..63 <-> slot m 8 + k + 4 of lanes 0..31 (m, k < 4), both planes.
    // v_permlane32_swap_b32 vdst, src swaps vdst[32..63] with src[0..31]; the s_nop covers the VALU-write -> permlane-read
    // hazard the compiler does not pad inside an asm statement (cdna_hip_programming.md T21).
    __device__ __forceinline__ void pair_swap() {
        if constexpr (sizeof(T) == 4 && NREG >= 32) {
            static_for<0, 4>([&](auto M_) {
                constexpr int m = M_;
                asm volatile(
                    "s_nop 1\n\t"
                    "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
                    "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\tv_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
                    : "+v"(v[m * 8 + 0].re), "+v"(v[m * 8 + 0].im), "+v"(v[m * 8 + 1].re), "+v"(v[m * 8 + 1].im), "+v"(v[m * 8 + 2].re),
                      "+v"(v[m * 8 + 2].im), "+v"(v[m * 8 + 3].re), "+v"(v[m * 8 + 3].im), "+v"(v[m * 8 + 4].re), "+v"(v[m * 8 + 4].im),
                      "+v"(v[m * 8 + 5].re), "+v"(v[m * 8 + 5].im), "+v"(v[m * 8 + 6].re), "+v"(v[m * 8 + 6].im), "+v"(v[m * 8 + 7].re),
                      "+v"(v[m * 8 + 7].im));
            });
        }
    }
};
// Two VIRTUAL threads per physical thread (tuning: ABL bit 4096 of the column-tile kernels): the engine sees a workgroup of
// 2 x blockDim threads; physical thread t runs virtual threads 2 t and 2 t + 1 one after the other in every phase, each on its own
// half of the register array.  With the column-fastest map these are two ADJACENT columns of the tile (same rows), so the row
// loads / stores of the pair are 16 contiguous bytes -- the "two columns per lane" form of the tile skeleton
// (tools/membench/skel.hip v4: + 1.4 .. 6 % on the later-pass shape) without a second engine.
template <class T, int NREG> struct DevExecPair {
    cx<T> v[2 * NREG];
    template <class Fn> __device__ __forceinline__ void for_threads(Fn&& fn) {
        fn(2 * (int)threadIdx.x, v);
        fn(2 * (int)threadIdx.x + 1, v + NREG);
    }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    __device__ __forceinline__ void relaunder() {}
    __device__ __forceinline__ void pair_swap() {}
};
// Executor for bodies that loop over many sequences: relaunder() makes the thread index opaque to the optimiser from
// there on, so the (cheap) index arithmetic is redone per sequence instead of being hoisted out of the loop into
// hundreds of live registers.
template <class T, int NREG> struct DevExecLoop {
    cx<T> v[NREG];
    int tid = (int)threadIdx.x;
    template <class Fn> __device__ __forceinline__ void for_threads(Fn&& fn) { fn(tid, v); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    __device__ __forceinline__ void relaunder() { asm volatile("" : "+v"(tid)); }
    __device__ __forceinline__ void pair_swap() {}
};

template <class T, class S, int F, bool SPLIT, int ABL = 0>
__global__ __launch_bounds__(F* S::TPF) void k1_kernel(K1Params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, k1_regs<S, SPLIT, ABL>()> ex;
    k1_body<T, S, F, SPLIT, ABL>(ex, p, (long long)blockIdx.x, smem);
}
// two workgroups per CU is what keeps HBM busy while the other workgroup computes: ask the register
// allocator for (2 * threads / 256) waves per SIMD
// ABL bit 6 (64): pair-fused first two sub-passes (engine.h; a production option, not an ablation)
// ABL bit 12 (4096, tuning): two virtual threads per physical thread (DevExecPair)
template <class S, int F, int ABL> constexpr int k2_threads() { return (ABL & 4096) ? F * S::TPF / 2 : F * S::TPF; }
template <class T, class S, int F, bool FIRST, bool SPLIT, int ABL = 0>
__global__ __launch_bounds__((k2_threads<S, F, ABL>()), (k2_threads<S, F, ABL>() >= 512 ? 4 : 2)) void k2_kernel(K2Params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr ((ABL & 4096) != 0) {
        static_assert(F % 2 == 0 && !(ABL & 64), "pairs of adjacent columns");
        DevExecPair<T, regs_needed<S, SPLIT>()> ex;
        k2_body<T, S, F, FIRST, SPLIT, ABL>(ex, p, (long long)blockIdx.x, smem);
    } else if constexpr ((ABL & 8192) != 0) {
        // tuning: PERSISTENT workgroups (grid = what the chip holds; each walks the tile list with that stride).  The stores of
        // tile i and the loads of tile i + 1 are in flight together: the barrier between two tiles is a bare s_barrier (it
        // orders the LDS reuse; the last LDS reads of a tile are complete before its last sub-pass computes), NOT a
        // __syncthreads(), whose s_waitcnt vmcnt(0) would drain the stores first.
        DevExec<T, regs_needed<S, SPLIT>()> ex;
        const long long total = p.batch * p.tiles_per_fft;
        for (long long b = (long long)blockIdx.x; b < total; b += (long long)gridDim.x) {
            k2_body<T, S, F, FIRST, SPLIT, ABL>(ex, p, b, smem);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        DevExec<T, regs_needed<S, SPLIT>()> ex;
        k2_body<T, S, F, FIRST, SPLIT, ABL>(ex, p, (long long)blockIdx.x, smem);
    }
}

template <class T, class S, int F, bool SPLIT, int STAGE>
__global__ __launch_bounds__(F* S::TPF) void k1bs_kernel(BluesteinParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, regs_needed<S, SPLIT>()> ex;
    k1bs_body<T, S, F, SPLIT, STAGE>(ex, p, (long long)blockIdx.x, smem);
}
template <class T, class S, int F, bool SPLIT, int STAGE> KernelEntry make_k1bs(int prec, const char* name) {
    KernelEntry e{};
    e.kind = STAGE == 1 ? KIND_BS2_FIRST : KIND_BS2_SECOND;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = lds_bytes<T, S, F, SPLIT>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)k1bs_kernel<T, S, F, SPLIT, STAGE>, dim3((unsigned)grid), dim3(F * S::TPF), args,
                              lds_bytes<T, S, F, SPLIT>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k1bs_kernel<T, S, F, SPLIT, STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes<T, S, F, SPLIT>());
    };
    return e;
}
template <class T, class S, int F, bool SPLIT, int ABL = 0> KernelEntry make_k1(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_K1;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = k1_lds_bytes<T, S, F, SPLIT, ABL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)k1_kernel<T, S, F, SPLIT, ABL>, dim3((unsigned)grid), dim3(F * S::TPF), args,
                              k1_lds_bytes<T, S, F, SPLIT, ABL>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k1_kernel<T, S, F, SPLIT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)k1_lds_bytes<T, S, F, SPLIT, ABL>());
    };
    return e;
}
template <class T, class S, int F, bool FIRST, bool SPLIT, int ABL = 0> KernelEntry make_k2(int prec, const char* name) {
    KernelEntry e{};
    e.kind = FIRST ? KIND_K2_FIRST : KIND_K2_LATER;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = k2_threads<S, F, ABL>();
    e.lds_bytes = k2_lds_bytes<T, S, F, SPLIT, ABL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        if ((ABL & 8192) != 0) {  // persistent: as many workgroups as the chip holds at 128 VGPRs
            const long long resident = 256LL * (k2_threads<S, F, ABL>() >= 1024 ? 1 : 1024 / k2_threads<S, F, ABL>());
            if (grid > resident) grid = resident;
        }
        (void)hipLaunchKernel((const void*)k2_kernel<T, S, F, FIRST, SPLIT, ABL>, dim3((unsigned)grid), dim3(k2_threads<S, F, ABL>()), args,
                              k2_lds_bytes<T, S, F, SPLIT, ABL>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k2_kernel<T, S, F, FIRST, SPLIT, ABL>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2_lds_bytes<T, S, F, SPLIT, ABL>());
    };
    return e;
}
// ---- fused two-pass kernel (K2FusedParams, kernels.h k2f_decode) --------------------------------------------------------------
// Cross-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): a producer
// tile's waves drain their stores (s_waitcnt vmcnt(0)), the workgroup barrier collects them, lane 0 issues the agent-scope
// release (buffer_wbl2 sc1: the XCD's L2 writes its dirty lines back), waits for it (the explicit s_waitcnt: the compiler may
// drop its own) and bumps the slot's counter with a relaxed agent-scope atomic; a consumer's lane 0 polls the counter with
// relaxed agent-scope loads (L1-bypassing) and s_sleep, then issues ONE agent-scope acquire (invalidates this CU's L1) before
// the barrier that releases its waves to plain loads.  Waits are bounded: a wait that gives up sets the STICKY error word of
// the (plan, stream) slot (K2FusedParams::err: pinned host memory, cleared by no launch) and the tile proceeds -- the launch always
// terminates, and the give-up cannot be missed: the next device call / synchronize / destroy on that plan and stream fails with it and the
// host-slice path re-runs the affected rows as two launches (plan.cpp fused_check, capi.cpp process_host_impl).
// Deadlock freedom: a work item only waits for items of EARLIER steps, i.e. lower indices.  With tickets (mode bit 1) every
// lower index has been claimed by a workgroup that is already running when an item starts to wait.  Without tickets the same
// holds per XCD because a dispatcher walks its share of the grid in order (identical resource needs, nothing to reorder), and
// then for the chip: the XCD with the lowest dispatch frontier cannot hold a waiter whose dependency lies beyond a frontier.
// returns true when the wait gave up
template <class T> __device__ __forceinline__ bool k2f_wait(unsigned* ctr, unsigned target, const K2FusedParams<T>& fp) {
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > fp.spin_limit) {
            // the word lives in pinned host memory and no launch clears it: a plain system-scope store (no PCIe atomic needed), visible to
            // the host at the latest when this launch completes
            __hip_atomic_store(fp.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return true;
        }
    }
    return false;
}
// A tile whose wait gave up still runs (the launch must terminate and the counters must advance), but everything it stores is POISONED: the
// sign it multiplies imaginary parts with on the way out becomes NaN.  A first-pass tile that overwrites a ring slot too early therefore
// writes NaN, not plausible numbers, and every second-pass tile that reads a poisoned or half-written column produces NaN in all of that
// column's outputs: the rows a give-up touched come back as NaN whatever the caller does next (src/lib.rs:184: never silently wrong).
template <class T> __device__ __forceinline__ K2Params<T> k2f_poisoned(const K2Params<T>& p, bool gave_up) {
    K2Params<T> q = p;
    if (gave_up) q.sgn_out = __builtin_nanf("");
    return q;
}
// RINGV: how the ring is accessed -- bit 0: the first pass stores it with agent-scope (write-through) stores, so no release fence;
// bit 1: the second pass loads it with agent-scope (L1-bypassing) loads, so no acquire fence (cx.h st_agent / ld_agent)
template <class T, class S0, int F0, bool SPLIT0, int ABL0, class S1, int F1, bool SPLIT1, int ABL1, int RINGV>
__global__ __launch_bounds__((k2_threads<S0, F0, ABL0>()), (k2_threads<S0, F0, ABL0>() >= 512 ? 4 : 2)) void k2f_kernel(K2FusedParams<T> fp) {
    static_assert(k2_threads<S0, F0, ABL0>() == k2_threads<S1, F1, ABL1>(), "one block size for both passes");
    static_assert(!(ABL0 & 4096), "the first pass runs on a plain executor (its ring stores are per element)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_ticket;
    __shared__ unsigned s_gave_up;
    if (threadIdx.x == 0) s_gave_up = 0u;  // (published by the barrier every path below passes before the flag is read)
    long long w = (long long)blockIdx.x;
    if (fp.mode & 2) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(fp.ctrl, 1u);
        __syncthreads();
        w = (long long)__builtin_amdgcn_readfirstlane((int)s_ticket);  // workgroup-uniform: everything derived from it stays in SGPRs
    }
    const K2FItem it = k2f_decode(fp, w);
    if (it.pass < 0) return;
    unsigned* written = fp.ctrl + 32 + 64 * it.slot;
    unsigned* rd = fp.ctrl + 64 + 64 * it.slot;
    const long long n = fp.pass[0].n;
    cx<T>* ring = fp.pass[0].out + (long long)it.slot * fp.slot_elems;
    const bool sync = (fp.mode & 1) != 0;
    const bool release = sync && !(RINGV & 1) && !(fp.mode & 4), acquire = sync && !(RINGV & 2) && !(fp.mode & 8);  // mode bits 2, 3: probes
    if (it.pass == 0) {
        bool gave_up = false;
        if (sync && it.use > 0) {
            if (threadIdx.x == 0 && k2f_wait(rd, it.use * (unsigned)fp.tiles[1], fp)) s_gave_up = 1u;
            __syncthreads();
            gave_up = __builtin_amdgcn_readfirstlane((int)s_gave_up) != 0;
        }
        DevExec<T, regs_needed<S0, SPLIT0>()> ex;
        k2_tile<T, S0, F0, true, SPLIT0, ABL0, (RINGV & 1)>(ex, k2f_poisoned(fp.pass[0], gave_up), fp.pass[0].in + it.g * n, ring, it.tile, it.tile_out, smem);
        if (sync) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(written, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        bool gave_up = false;
        if (sync) {
            if (threadIdx.x == 0) {
                if (k2f_wait(written, (it.use + 1u) * (unsigned)fp.tiles[0], fp)) s_gave_up = 1u;
                if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            gave_up = __builtin_amdgcn_readfirstlane((int)s_gave_up) != 0;
        }
        // (ABL bit 4096: the second pass as two columns per lane -- 16-byte ring loads and 16-byte stores, launch.h DevExecPair)
        typename std::conditional<(ABL1 & 4096) != 0, DevExecPair<T, regs_needed<S1, SPLIT1>()>, DevExec<T, regs_needed<S1, SPLIT1>()>>::type ex;
        k2_tile<T, S1, F1, false, SPLIT1, ABL1, (RINGV & 2)>(ex, k2f_poisoned(fp.pass[1], gave_up), (const cx<T>*)ring, fp.pass[1].out + it.g * n, it.tile, it.tile_out, smem);
        if (sync) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(rd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <class T, class S0, int F0, bool SPLIT0, int ABL0, class S1, int F1, bool SPLIT1, int ABL1> constexpr size_t k2f_lds_bytes() {
    return k2_lds_bytes<T, S0, F0, SPLIT0, ABL0>() > k2_lds_bytes<T, S1, F1, SPLIT1, ABL1>() ? k2_lds_bytes<T, S0, F0, SPLIT0, ABL0>() : k2_lds_bytes<T, S1, F1, SPLIT1, ABL1>();
}
template <class T, class S0, int F0, bool SPLIT0, int ABL0, class S1, int F1, bool SPLIT1, int ABL1, int RINGV = 1, int VARIANT = 0>
KernelEntry make_k2f(int prec, const char* name, const char* part0, const char* part1) {
    KernelEntry e{};
    e.kind = KIND_K2_FUSED;
    e.prec = prec;
    e.n = S0::N * S1::N;
    e.f = F0;
    e.f2 = F1;
    e.threads = k2_threads<S0, F0, ABL0>();
    e.lds_bytes = k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>();
    e.name = name;
    e.part[0] = part0;
    e.part[1] = part1;
    e.variant = VARIANT;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)k2f_kernel<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1, RINGV>, dim3((unsigned)grid), dim3(k2_threads<S0, F0, ABL0>()), args,
                              k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k2f_kernel<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1, RINGV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>());
    };
    e.blocks_per_cu = []() -> int {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k2f_kernel<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1, RINGV>, k2_threads<S0, F0, ABL0>(),
                                                         k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>()) != hipSuccess)
            return 0;
        return n;
    };
    return e;
}
template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0>
__global__ __launch_bounds__(F* S::TPF) void bluestein_kernel(BluesteinParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, bluestein_regs<S, PF>()> ex;
    bluestein_body<T, S, F, SPLIT, TW1, PF>(ex, p, (long long)blockIdx.x, smem);
}
// Two ROWS per physical thread (round 5, tuning): the engine sees a workgroup of 2 x blockDim threads transforming F = 2 rows (MAP_EF: virtual
// thread t + blockDim is slot t of row 1); physical thread t runs both, one after the other in every phase, each on its own half of the register
// array.  A barrier then serves two rows -- half the barriers per row, two independent dependency chains per thread -- at half the waves per row.
template <class T, int NREG> struct DevExecRows2 {
    cx<T> v[2 * NREG];
    template <class Fn> __device__ __forceinline__ void for_threads(Fn&& fn) {
        fn((int)threadIdx.x, v);
        fn((int)threadIdx.x + (int)blockDim.x, v + NREG);
    }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    __device__ __forceinline__ void relaunder() {}
    __device__ __forceinline__ void pair_swap() {}
};
template <class T, class S, bool SPLIT, bool TW1, int PF>
__global__ __launch_bounds__(S::TPF) void bluestein_rows2_kernel(BluesteinParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExecRows2<T, bluestein_regs<S, PF>()> ex;
    bluestein_body<T, S, 2, SPLIT, TW1, PF>(ex, p, (long long)blockIdx.x, smem);
}
template <class T, class S, bool SPLIT, bool TW1, int PF> KernelEntry make_bluestein_rows2(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_BLUESTEIN;
    e.prec = prec;
    e.n = S::N;
    e.f = 2;
    fill_sched<S>(e);
    e.threads = S::TPF;
    e.lds_bytes = bluestein_lds_bytes<T, S, 2, SPLIT, TW1, PF>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)bluestein_rows2_kernel<T, S, SPLIT, TW1, PF>, dim3((unsigned)grid), dim3(S::TPF), args,
                              bluestein_lds_bytes<T, S, 2, SPLIT, TW1, PF>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)bluestein_rows2_kernel<T, S, SPLIT, TW1, PF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bluestein_lds_bytes<T, S, 2, SPLIT, TW1, PF>());
    };
    return e;
}
template <class T, class S, int F, int MODE>
// MODE 2: rows loop with next-row prefetch (>= 3 waves per SIMD in f32); MODE 3: without the prefetch (MODE 9: + non-temporal row loads); MODE 4: as MODE 2 for the
// larger primes whose per-thread tables need up to 256 VGPRs (two waves per SIMD)
__global__ __launch_bounds__((rader_rows_mode(MODE) ? 1 : F) * S::TPF, (MODE == 2 ? (sizeof(T) == 4 ? 3 : 2) : (MODE == 3 || MODE == 9) ? (sizeof(T) == 4 ? 4 : 2) : (MODE == 4 || MODE == 6) ? 2 : 1)) void rader_kernel(RaderParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (rader_rows_mode(MODE)) {  // F = rows pushed through one workgroup one after another
        DevExecLoop<T, RaderRows<S, MODE == 6>::NREG> ex;
        rader_rows_body<T, S, F, MODE == 2 || MODE == 4 || MODE == 6, MODE == 6, MODE == 9>(ex, p, (long long)blockIdx.x, smem);
    } else {
        DevExec<T, rader_regs<S>()> ex;
        rader_body<T, S, F, MODE>(ex, p, (long long)blockIdx.x, smem);
    }
}
template <class T> __global__ __launch_bounds__(256) void pointwise_kernel(PointwiseParams<T> p, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) pointwise_elem<T>(p, i);
}
template <class T> KernelEntry make_pointwise(int prec) {
    KernelEntry e{};
    e.kind = KIND_POINTWISE;
    e.prec = prec;
    e.threads = 256;
    e.name = "bluestein_pointwise";
    e.launch = [](const void* params, long long total, void* stream) {
        long long blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        void* args[] = {const_cast<void*>(params), &total};
        (void)hipLaunchKernel((const void*)pointwise_kernel<T>, dim3((unsigned)blocks), dim3(256), args, 0, (hipStream_t)stream);
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0> constexpr size_t bluestein_lds() {
    return bluestein_lds_bytes<T, S, F, SPLIT, TW1, PF>();  // exchange buffer of both schedules (+ the staged tables)
}
template <class T, class S, int F, int MODE> constexpr size_t rader_lds() {
    if (rader_rows_mode(MODE)) return (size_t)RaderRows<S, MODE == 6>::SLOTS * sizeof(cx<T>);  // one row at a time
    if (MODE == 5) return (size_t)F * (rader5_pitch<S>() + 1) * sizeof(cx<T>);       // rows at the larger pitch of the two schedules + F spare slots
    return (size_t)F * (S::pitch() + (MODE >= 1 ? 0 : S::N + 1)) * sizeof(cx<T>);
}

template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0> KernelEntry make_bluestein(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_BLUESTEIN;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = bluestein_lds<T, S, F, SPLIT, TW1, PF>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)bluestein_kernel<T, S, F, SPLIT, TW1, PF>, dim3((unsigned)grid), dim3(F * S::TPF), args,
                              bluestein_lds<T, S, F, SPLIT, TW1, PF>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)bluestein_kernel<T, S, F, SPLIT, TW1, PF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bluestein_lds<T, S, F, SPLIT, TW1, PF>());
    };
    return e;
}
template <class T, class S, int F, int MODE> KernelEntry make_rader(int prec, const char* name) {
    KernelEntry e{};
    e.split = (MODE >= 1);  // Rader: perm_in is the inverse map (kernels.h rader_body MODE 1, rader_rows_body)
    e.kind = KIND_RADER;
    e.prec = prec;
    e.n = S::N;
    e.aux = S::N + 1;
    e.f = F;
    fill_sched<S>(e);
    e.threads = (rader_rows_mode(MODE) ? 1 : F) * S::TPF;
    e.lds_bytes = rader_lds<T, S, F, MODE>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)rader_kernel<T, S, F, MODE>, dim3((unsigned)grid), dim3((rader_rows_mode(MODE) ? 1 : F) * S::TPF), args,
                              rader_lds<T, S, F, MODE>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)rader_kernel<T, S, F, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)rader_lds<T, S, F, MODE>());
    };
    return e;
}
// Waves per SIMD the compiler must keep (caps the VGPR count).  Split tiles: as for the power-of-two ones.  f64 non-split tiles:
// four (<= 128 VGPRs; measured +3.4 % in the median over 100 general plans, profiles/r2/k2g_xcd_ab.txt) except the two fused-
// Bluestein last passes of the 256-row tile (150 / 166 VGPRs).  f32: no request (the compiler then schedules the light kernels
// for 5 - 8 waves; a request of four costs those 16 - 19 %) except three radix-15-first later passes that sit at 130 - 132 VGPRs
// without it and lose a fifth of their rate to the occupancy step (measured on k2glater<75>: 4.5 -> 3.4 TB/s at 130 VGPRs).
template <class T, class S, bool FIRST, int FUSE, bool SPLIT, int THREADS> constexpr int k2g_min_waves() {
    if (SPLIT) return THREADS >= 512 ? 4 : 2;
    if (sizeof(T) == 8) return (FUSE >= 2 && S::N >= 256) ? 1 : 4;
    return (!FIRST && FUSE == 0 && S::R[0] == 15 && (S::N == 300 || S::N == 375 || S::N == 450)) ? 4 : 1;
}
template <class T, class S, int F, bool FIRST, int FUSE, bool SPLIT, int TWL>
__global__ __launch_bounds__(F* S::TPF, (k2g_min_waves<T, S, FIRST, FUSE, SPLIT, F * S::TPF>())) void k2g_kernel(K2Params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, regs_needed<S, SPLIT>()> ex;
    k2g_body<T, S, F, FIRST, FUSE, SPLIT, TWL>(ex, p, (long long)blockIdx.x, smem);
}
// TWL: sub-pass twiddle tables staged in LDS (engine.h); -1 = all of them
template <class S> constexpr int k2g_twl(int twl) { return S::NP < 2 ? 0 : twl < 0 ? twl_all(S::NP) : twl; }
template <class T, class S, int F, bool FIRST, int FUSE = 0, bool SPLIT = false, int TWL_ = 0> KernelEntry make_k2g(int prec, const char* name) {
    constexpr int TWL = k2g_twl<S>(TWL_);
    KernelEntry e{};
    e.kind = FUSE == 1 ? KIND_K2G_FIRST_CHIRP : FUSE == 2 ? KIND_K2G_LAST_MUL : FUSE == 3 ? KIND_K2G_LAST_CHIRP : FUSE == 4 ? KIND_K2G_FIRST_GATHER : FUSE == 5 ? KIND_K2G_LAST_RMUL : FUSE == 6 ? KIND_K2G_LAST_SCATTER : FIRST ? KIND_K2G_FIRST : KIND_K2G_LATER;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), TWL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)k2g_kernel<T, S, F, FIRST, FUSE, SPLIT, TWL>, dim3((unsigned)grid), dim3(F * S::TPF), args,
                              lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), TWL>(), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k2g_kernel<T, S, F, FIRST, FUSE, SPLIT, TWL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), TWL>());
    };
    return e;
}
template <class T, class S, int F, bool FIRST>
__global__ __launch_bounds__(F* S::TPF) void k2r_kernel(K2Params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, regs_needed<S, false>()> ex;
    k2r_body<T, S, F, FIRST>(ex, p, (long long)blockIdx.x, smem);
}
template <class T, class S, int F, bool FIRST> KernelEntry make_k2r(int prec, const char* name) {
    KernelEntry e{};
    e.kind = FIRST ? KIND_K2R_FIRST : KIND_K2R_LATER;
    e.prec = prec;
    e.n = S::N + 1;  // the tile height: the prime
    e.aux = S::N + 1;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = k2r_lds_bytes<T, S, F>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void* stream) {
        void* args[] = {const_cast<void*>(params)};
        (void)hipLaunchKernel((const void*)k2r_kernel<T, S, F, FIRST>, dim3((unsigned)grid), dim3(F * S::TPF), args, k2r_lds_bytes<T, S, F>(),
                              (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)k2r_kernel<T, S, F, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2r_lds_bytes<T, S, F>());
    };
    return e;
}
constexpr int kDynEmax = 16, kDynEmaxLight = 12, kDynEmaxHeavy = 32;
constexpr int dyn_emax(int set) { return set == 1 ? kDynEmaxLight : set == 2 ? kDynEmaxHeavy : kDynEmax; }
// the HEAVY set (prime radices 17 .. 31, 32 values per thread) runs workgroups of at most 256 threads: 256 VGPRs each
template <class T, int LIGHT> __global__ __launch_bounds__(LIGHT == 2 ? 256 : 512) void dyn_k1_kernel(DynK1Params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, dyn_emax(LIGHT)> ex;
    dyn_k1_body<T, dyn_emax(LIGHT), LIGHT>(ex, p, (long long)blockIdx.x, smem);
}
template <class T, int LIGHT> __global__ __launch_bounds__(LIGHT == 2 ? 256 : 512) void dyn_rader_kernel(DynRaderParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<T, dyn_emax(LIGHT)> ex;
    dyn_rader_body<T, dyn_emax(LIGHT), LIGHT>(ex, p, (long long)blockIdx.x, smem);
}
// the run-time scheduled kernels take their block size and LDS bytes from the schedule in the parameter block
template <class T> KernelEntry make_dyn_k1(int prec) {
    KernelEntry e{};
    e.kind = KIND_DYN_K1;
    e.prec = prec;
    e.name = "dyn_k1";
    e.launch = [](const void* params, long long grid, void* stream) {
        const DynK1Params<T>* p = (const DynK1Params<T>*)params;
        void* args[] = {const_cast<void*>(params)};
        // only the HEAVY radix set is planned as a plain transform (plan.cpp); the other sets serve the run-time scheduled Rader
        (void)hipLaunchKernel((const void*)dyn_k1_kernel<T, 2>, dim3((unsigned)grid), dim3(p->s.f * p->s.tpf), args,
                              (size_t)p->s.f * p->s.pitch * sizeof(cx<T>), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        return (int)hipFuncSetAttribute((const void*)dyn_k1_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    };
    return e;
}
template <class T> KernelEntry make_dyn_rader(int prec) {
    KernelEntry e{};
    e.kind = KIND_DYN_RADER;
    e.prec = prec;
    e.name = "dyn_rader";
    e.launch = [](const void* params, long long grid, void* stream) {
        const DynRaderParams<T>* p = (const DynRaderParams<T>*)params;
        void* args[] = {const_cast<void*>(params)};
        const void* fn = p->s.light == 1 ? (const void*)dyn_rader_kernel<T, 1> : p->s.light == 2 ? (const void*)dyn_rader_kernel<T, 2> : (const void*)dyn_rader_kernel<T, 0>;
        (void)hipLaunchKernel(fn, dim3((unsigned)grid), dim3(p->s.f * p->s.tpf), args,
                              (size_t)p->s.f * (p->s.pitch + p->s.n + 1) * sizeof(cx<T>), (hipStream_t)stream);
    };
    e.prepare = []() -> int {
        int a = (int)hipFuncSetAttribute((const void*)dyn_rader_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int b = (int)hipFuncSetAttribute((const void*)dyn_rader_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int c = (int)hipFuncSetAttribute((const void*)dyn_rader_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return a ? a : b ? b : c;
    };
    return e;
}
#else
// -------------------------------------------------------------------------------- host emulator
// MI355_EMU_ORDER=reverse runs the threads of every phase from the last to the first: a write-write or read-write race
// between threads of one phase (which the GPU resolves by timing) then shows up as a changed result -- the emulator's
// cheap race detector (tests/test_kernel_bodies_emu.py::test_thread_order_independence).
inline bool emu_reverse_order() {
    const char* e = getenv("MI355_EMU_ORDER");
    return e && e[0] == 'r';
}
template <class T, int NREG> struct HostExec {
    int nt;
    bool reverse;
    std::vector<cx<T>> regs;
    explicit HostExec(int n) : nt(n), reverse(emu_reverse_order()), regs((size_t)n * NREG, cx<T>{0, 0}) {}
    template <class Fn> void for_threads(Fn&& fn) {
        if (reverse)
            for (int t = nt - 1; t >= 0; --t) fn(t, regs.data() + (size_t)t * NREG);
        else
            for (int t = 0; t < nt; ++t) fn(t, regs.data() + (size_t)t * NREG);
    }
    void barrier() {}
    void relaunder() {}
    // lanes l < 32 and l + 32 of each 64-thread wave exchange register slots (DevExec::pair_swap)
    void pair_swap() {
        if constexpr (NREG >= 32)
            for (int t = 0; t < nt; ++t)
                if ((t & 63) < 32 && t + 32 < nt)
                    for (int m = 0; m < 4; ++m)
                        for (int k = 0; k < 4; ++k) std::swap(regs[(size_t)(t + 32) * NREG + m * 8 + k], regs[(size_t)t * NREG + m * 8 + k + 4]);
    }
};
template <class T, class S, int F, bool SPLIT, int STAGE> KernelEntry make_k1bs(int prec, const char* name) {
    KernelEntry e{};
    e.kind = STAGE == 1 ? KIND_BS2_FIRST : KIND_BS2_SECOND;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = lds_bytes<T, S, F, SPLIT>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(lds_bytes<T, S, F, SPLIT>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, regs_needed<S, SPLIT>()> ex(F * S::TPF);
            k1bs_body<T, S, F, SPLIT, STAGE>(ex, *(const BluesteinParams<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, int F, bool SPLIT, int ABL = 0> KernelEntry make_k1(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_K1;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = k1_lds_bytes<T, S, F, SPLIT, ABL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(k1_lds_bytes<T, S, F, SPLIT, ABL>() + 64, (char)0x5a);  // poisoned LDS
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, k1_regs<S, SPLIT, ABL>()> ex(F * S::TPF);
            k1_body<T, S, F, SPLIT, ABL>(ex, *(const K1Params<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, int F, bool FIRST, bool SPLIT, int ABL = 0> KernelEntry make_k2(int prec, const char* name) {
    KernelEntry e{};
    e.kind = FIRST ? KIND_K2_FIRST : KIND_K2_LATER;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = k2_lds_bytes<T, S, F, SPLIT, ABL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(k2_lds_bytes<T, S, F, SPLIT, ABL>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, regs_needed<S, SPLIT>()> ex(F * S::TPF);
            k2_body<T, S, F, FIRST, SPLIT, ABL>(ex, *(const K2Params<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
// Fused two-pass kernel on the host: the work items run one after another in index order; every item CHECKS the dependency
// counters the device kernel would wait on (an unsatisfied one means the item order is wrong: error word bit 1) and updates them.
template <class T, class S0, int F0, bool SPLIT0, int ABL0, class S1, int F1, bool SPLIT1, int ABL1> constexpr size_t k2f_lds_bytes() {
    return k2_lds_bytes<T, S0, F0, SPLIT0, ABL0>() > k2_lds_bytes<T, S1, F1, SPLIT1, ABL1>() ? k2_lds_bytes<T, S0, F0, SPLIT0, ABL0>() : k2_lds_bytes<T, S1, F1, SPLIT1, ABL1>();
}
template <class T, class S0, int F0, bool SPLIT0, int ABL0, class S1, int F1, bool SPLIT1, int ABL1, int RINGV = 1, int VARIANT = 0>
KernelEntry make_k2f(int prec, const char* name, const char* part0, const char* part1) {
    KernelEntry e{};
    e.kind = KIND_K2_FUSED;
    e.variant = VARIANT;
    e.prec = prec;
    e.n = S0::N * S1::N;
    e.f = F0;
    e.f2 = F1;
    e.threads = F0 * S0::TPF;
    e.lds_bytes = k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>();
    e.name = name;
    e.part[0] = part0;
    e.part[1] = part1;
    e.launch = [](const void* params, long long grid, void*) {
        const K2FusedParams<T>& fp = *(const K2FusedParams<T>*)params;
        std::vector<char> lds(k2f_lds_bytes<T, S0, F0, SPLIT0, ABL0, S1, F1, SPLIT1, ABL1>() + 64, (char)0x5a);
        const long long n = fp.pass[0].n;
        for (long long w = 0; w < grid; ++w) {
            const K2FItem it = k2f_decode(fp, w);
            if (it.pass < 0) continue;
            unsigned* written = fp.ctrl + 32 + 64 * it.slot;
            unsigned* rd = fp.ctrl + 64 + 64 * it.slot;
            cx<T>* ring = fp.pass[0].out + (long long)it.slot * fp.slot_elems;
            if (it.pass == 0) {
                if (it.use > 0 && *rd < it.use * (unsigned)fp.tiles[1]) *fp.err |= 2u;
                HostExec<T, regs_needed<S0, SPLIT0>()> ex(F0 * S0::TPF);
                k2_tile<T, S0, F0, true, SPLIT0, ABL0>(ex, fp.pass[0], fp.pass[0].in + it.g * n, ring, it.tile, it.tile_out, lds.data());
                *written += 1;
            } else {
                if (*written < (it.use + 1u) * (unsigned)fp.tiles[0]) *fp.err |= 2u;
                HostExec<T, regs_needed<S1, SPLIT1>()> ex(F1 * S1::TPF);
                k2_tile<T, S1, F1, false, SPLIT1, ABL1>(ex, fp.pass[1], (const cx<T>*)ring, fp.pass[1].out + it.g * n, it.tile, it.tile_out, lds.data());
                *rd += 1;
            }
        }
    };
    e.prepare = []() -> int { return 0; };
    e.blocks_per_cu = []() -> int { return F0 * S0::TPF >= 1024 ? 1 : 2; };
    return e;
}
template <class T> KernelEntry make_pointwise(int prec) {
    KernelEntry e{};
    e.kind = KIND_POINTWISE;
    e.prec = prec;
    e.threads = 256;
    e.name = "bluestein_pointwise";
    e.launch = [](const void* params, long long total, void*) {
        for (long long i = 0; i < total; ++i) pointwise_elem<T>(*(const PointwiseParams<T>*)params, i);
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0> constexpr size_t bluestein_lds() {
    return bluestein_lds_bytes<T, S, F, SPLIT, TW1, PF>();  // exchange buffer of both schedules (+ the staged tables)
}
template <class T, class S, int F, int MODE> constexpr size_t rader_lds() {
    if (rader_rows_mode(MODE)) return (size_t)RaderRows<S, MODE == 6>::SLOTS * sizeof(cx<T>);  // one row at a time
    if (MODE == 5) return (size_t)F * (rader5_pitch<S>() + 1) * sizeof(cx<T>);       // rows at the larger pitch of the two schedules + F spare slots
    return (size_t)F * (S::pitch() + (MODE >= 1 ? 0 : S::N + 1)) * sizeof(cx<T>);
}
template <class T, class S, int F, bool SPLIT = false, bool TW1 = false, int PF = 0> KernelEntry make_bluestein(int prec, const char* name) {
    KernelEntry e{};
    e.kind = KIND_BLUESTEIN;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = bluestein_lds<T, S, F, SPLIT, TW1, PF>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(bluestein_lds<T, S, F, SPLIT, TW1, PF>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, bluestein_regs<S, PF>()> ex(F * S::TPF);
            bluestein_body<T, S, F, SPLIT, TW1, PF>(ex, *(const BluesteinParams<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, bool SPLIT, bool TW1, int PF> KernelEntry make_bluestein_rows2(int prec, const char* name) {
    KernelEntry e = make_bluestein<T, S, 2, SPLIT, TW1, PF>(prec, name);  // the emulator runs every (virtual) thread anyway
    e.threads = S::TPF;
    return e;
}
template <class T, class S, int F, int MODE> KernelEntry make_rader(int prec, const char* name) {
    KernelEntry e{};
    e.split = (MODE >= 1);  // Rader: perm_in is the inverse map (kernels.h rader_body MODE 1, rader_rows_body)
    e.kind = KIND_RADER;
    e.prec = prec;
    e.n = S::N;
    e.aux = S::N + 1;
    e.f = F;
    fill_sched<S>(e);
    e.threads = (rader_rows_mode(MODE) ? 1 : F) * S::TPF;
    e.lds_bytes = rader_lds<T, S, F, MODE>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(rader_lds<T, S, F, MODE>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            if constexpr (rader_rows_mode(MODE)) {
                HostExec<T, RaderRows<S, MODE == 6>::NREG> ex(S::TPF);
                rader_rows_body<T, S, F, MODE == 2 || MODE == 4 || MODE == 6, MODE == 6, MODE == 9>(ex, *(const RaderParams<T>*)params, b, lds.data());
            } else {
                HostExec<T, rader_regs<S>()> ex(F * S::TPF);
                rader_body<T, S, F, MODE>(ex, *(const RaderParams<T>*)params, b, lds.data());
            }
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class S> constexpr int k2g_twl(int twl) { return S::NP < 2 ? 0 : twl < 0 ? twl_all(S::NP) : twl; }
template <class T, class S, int F, bool FIRST, int FUSE = 0, bool SPLIT = false, int TWL_ = 0> KernelEntry make_k2g(int prec, const char* name) {
    constexpr int TWL = k2g_twl<S>(TWL_);
    KernelEntry e{};
    e.kind = FUSE == 1 ? KIND_K2G_FIRST_CHIRP : FUSE == 2 ? KIND_K2G_LAST_MUL : FUSE == 3 ? KIND_K2G_LAST_CHIRP : FUSE == 4 ? KIND_K2G_FIRST_GATHER : FUSE == 5 ? KIND_K2G_LAST_RMUL : FUSE == 6 ? KIND_K2G_LAST_SCATTER : FIRST ? KIND_K2G_FIRST : KIND_K2G_LATER;
    e.prec = prec;
    e.n = S::N;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), TWL>();
    e.split = SPLIT;
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(lds_bytes_twl<T, S, F, SPLIT, k2_pitch_mod(F), TWL>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, regs_needed<S, SPLIT>()> ex(F * S::TPF);
            k2g_body<T, S, F, FIRST, FUSE, SPLIT, TWL>(ex, *(const K2Params<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T, class S, int F, bool FIRST> KernelEntry make_k2r(int prec, const char* name) {
    KernelEntry e{};
    e.kind = FIRST ? KIND_K2R_FIRST : KIND_K2R_LATER;
    e.prec = prec;
    e.n = S::N + 1;
    e.aux = S::N + 1;
    e.f = F;
    fill_sched<S>(e);
    e.threads = F * S::TPF;
    e.lds_bytes = k2r_lds_bytes<T, S, F>();
    e.name = name;
    e.launch = [](const void* params, long long grid, void*) {
        std::vector<char> lds(k2r_lds_bytes<T, S, F>() + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, regs_needed<S, false>()> ex(F * S::TPF);
            k2r_body<T, S, F, FIRST>(ex, *(const K2Params<T>*)params, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
constexpr int kDynEmax = 16, kDynEmaxLight = 12, kDynEmaxHeavy = 32;
template <class T> KernelEntry make_dyn_k1(int prec) {
    KernelEntry e{};
    e.kind = KIND_DYN_K1;
    e.prec = prec;
    e.name = "dyn_k1";
    e.launch = [](const void* params, long long grid, void*) {
        const DynK1Params<T>* p = (const DynK1Params<T>*)params;
        std::vector<char> lds((size_t)p->s.f * p->s.pitch * sizeof(cx<T>) + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, kDynEmaxHeavy> ex(p->s.f * p->s.tpf);
            dyn_k1_body<T, kDynEmaxHeavy, 2>(ex, *p, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
template <class T> KernelEntry make_dyn_rader(int prec) {
    KernelEntry e{};
    e.kind = KIND_DYN_RADER;
    e.prec = prec;
    e.name = "dyn_rader";
    e.launch = [](const void* params, long long grid, void*) {
        const DynRaderParams<T>* p = (const DynRaderParams<T>*)params;
        std::vector<char> lds((size_t)p->s.f * (p->s.pitch + p->s.n + 1) * sizeof(cx<T>) + 64, (char)0x5a);
        for (long long b = 0; b < grid; ++b) {
            HostExec<T, kDynEmaxHeavy> ex(p->s.f * p->s.tpf);
            if (p->s.light == 1)
                dyn_rader_body<T, kDynEmaxLight, 1>(ex, *p, b, lds.data());
            else if (p->s.light == 2)
                dyn_rader_body<T, kDynEmaxHeavy, 2>(ex, *p, b, lds.data());
            else
                dyn_rader_body<T, kDynEmax, 0>(ex, *p, b, lds.data());
        }
    };
    e.prepare = []() -> int { return 0; };
    return e;
}
#endif

// Variant (V) and ablation (ABL) instantiations are tuning aids: compiled only with -DMI355_TUNING, so the shipped
// library contains no kernel that is wrong by design and no alternative tiling an environment variable could select.
#if defined(MI355_TUNING)
#define MI_K1ABL(V, ABL, T, PREC, F, SPLIT, ...)                                                               \
    reg.push_back(make_k1<T, Sched<__VA_ARGS__>, F, SPLIT, ABL>(PREC, "k1<" #__VA_ARGS__ ">xF" #F "abl" #ABL)); \
    reg.back().variant = V
#define MI_K1V(V, T, PREC, F, SPLIT, ...)                                                       \
    reg.push_back(make_k1<T, Sched<__VA_ARGS__>, F, SPLIT>(PREC, "k1<" #__VA_ARGS__ ">xF" #F "v" #V)); \
    reg.back().variant = V
#else
#define MI_K1ABL(V, ABL, T, PREC, F, SPLIT, ...) (void)0
#define MI_K1V(V, T, PREC, F, SPLIT, ...) (void)0
#endif
#define MI_K1(T, PREC, F, SPLIT, ...) reg.push_back(make_k1<T, Sched<__VA_ARGS__>, F, SPLIT>(PREC, "k1<" #__VA_ARGS__ ">xF" #F))
#if defined(MI355_TUNING)
#define MI_K2V(V, T, PREC, F, SPLIT, ...)                                                                   \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F "v" #V));  \
    reg.back().variant = V;                                                                                  \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F "v" #V)); \
    reg.back().variant = V
#define MI_K2ABL(V, ABL, T, PREC, F, SPLIT, ...)                                                              \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT, ABL>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F "abl" #ABL));  \
    reg.back().variant = V;                                                                                  \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT, ABL>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F "abl" #ABL)); \
    reg.back().variant = V
#else
#define MI_K2V(V, T, PREC, F, SPLIT, ...) (void)0
#define MI_K2ABL(V, ABL, T, PREC, F, SPLIT, ...) (void)0
#endif
// one pass kind only: the first pass (contiguous writes) and the later passes (strided writes) of one tile height may
// prefer different tilings
#define MI_K2_FIRST(T, PREC, F, SPLIT, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F))
#define MI_K2_LATER(T, PREC, F, SPLIT, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F))
// pair-fused variants (engine.h pair_fusable schedules, Complex<float>)
#define MI_K2P_FIRST(T, PREC, F, SPLIT, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT, 64>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F "p"))
#define MI_K2P_LATER(T, PREC, F, SPLIT, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT, 64>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F "p"))
#define MI_K2(T, PREC, F, SPLIT, ...)                                                                  \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F)); \
    reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F))

// both passes of a two-pass plan in one launch; NAME0 / NAME1 are the registry names of the one-pass kernels it fuses
// (the ring is stored with agent-scope write-through stores and read with plain loads behind one acquire per tile: RINGV = 1,
// the measured choice -- profiles/r4/ab_proto_2p20.jsonl: a release fence per tile costs 8 ms of 11, write-through stores 0.4)
// AUTO = 1: the planner uses the fused launch for this pair by default (a per-size measured choice, profiles/r4/ab_fused_*.jsonl);
// 0: compiled, used on request only (mi355fft_plan_set_fused)
#define MI_K2F(AUTO, T, PREC, NAME0, F0, SP0, ABL0, SCHED0, NAME1, F1, SP1, ABL1, SCHED1)                                                          \
    reg.push_back(make_k2f<T, SCHED0, F0, SP0, ABL0, SCHED1, F1, SP1, ABL1, 1, 0>(PREC, "k2fused[" NAME0 " | " NAME1 "]", NAME0, NAME1)); \
    reg.back().aux = AUTO
// tuning: other ring-access forms (RINGV 0: plain accesses + release / acquire fences, 2: agent-scope loads, 3: both agent-scope),
// registered as variant 10 + RINGV (MI355FFT_FUSE_RING)
#if defined(MI355_TUNING)
// tuning: any other fused pairing, registered as variant V (MI355FFT_FUSE_RING = 100 + V - 10 selects it)
#define MI_K2FV(V, T, PREC, NAME0, F0, SP0, ABL0, SCHED0, NAME1, F1, SP1, ABL1, SCHED1) \
    reg.push_back(make_k2f<T, SCHED0, F0, SP0, ABL0, SCHED1, F1, SP1, ABL1, 1, V>(PREC, "k2fused[" NAME0 " | " NAME1 "]v" #V, NAME0, NAME1))
#define MI_K2FR(RINGV, T, PREC, NAME0, F0, SP0, ABL0, SCHED0, NAME1, F1, SP1, ABL1, SCHED1) \
    reg.push_back(make_k2f<T, SCHED0, F0, SP0, ABL0, SCHED1, F1, SP1, ABL1, RINGV, 10 + RINGV>(PREC, "k2fused[" NAME0 " | " NAME1 "]r" #RINGV, NAME0, NAME1))
#else
#define MI_K2FR(RINGV, T, PREC, NAME0, F0, SP0, ABL0, SCHED0, NAME1, F1, SP1, ABL1, SCHED1) (void)0
#define MI_K2FV(V, T, PREC, NAME0, F0, SP0, ABL0, SCHED0, NAME1, F1, SP1, ABL1, SCHED1) (void)0
#endif
// production instantiations with options (ABL: 64 pair-fused, 128 / 1024 / 2048 sub-pass twiddle tables staged in LDS: all /
// sub-pass 1 / the last sub-pass); SUF is appended to the kernel name ("t", "t1", "tl")
#define MI_K1X(T, PREC, F, SPLIT, ABL, SUF, ...) reg.push_back(make_k1<T, Sched<__VA_ARGS__>, F, SPLIT, ABL>(PREC, "k1<" #__VA_ARGS__ ">xF" #F SUF))
#define MI_K2X_FIRST(T, PREC, F, SPLIT, ABL, SUF, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, true, SPLIT, ABL>(PREC, "k2first<" #__VA_ARGS__ ">xF" #F SUF))
#define MI_K2X_LATER(T, PREC, F, SPLIT, ABL, SUF, ...) reg.push_back(make_k2<T, Sched<__VA_ARGS__>, F, false, SPLIT, ABL>(PREC, "k2later<" #__VA_ARGS__ ">xF" #F SUF))
#define MI_K2X(T, PREC, F, SPLIT, ABL, SUF, ...)         \
    MI_K2X_FIRST(T, PREC, F, SPLIT, ABL, SUF, __VA_ARGS__); \
    MI_K2X_LATER(T, PREC, F, SPLIT, ABL, SUF, __VA_ARGS__)
// general column tiles: MI355_K2G_TWL (a build-wide switch of the A/B builds) / MI_K2GT: every sub-pass table staged in LDS
#if defined(MI355_K2G_TWL)
#define MI_K2G(T, PREC, F, ...) MI_K2GT(T, PREC, F, __VA_ARGS__)
#else
#define MI_K2G(T, PREC, F, ...)                                                                        \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, true>(PREC, "k2gfirst<" #__VA_ARGS__ ">xF" #F));  \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false>(PREC, "k2glater<" #__VA_ARGS__ ">xF" #F))
#endif
#define MI_K2GT(T, PREC, F, ...)                                                                                      \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, true, 0, false, -1>(PREC, "k2gfirst<" #__VA_ARGS__ ">xF" #F "t"));  \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 0, false, -1>(PREC, "k2glater<" #__VA_ARGS__ ">xF" #F "t"))
// tall general tiles (split exchange, 32 values per thread)
#define MI_K2GS(T, PREC, F, ...)                                                                                      \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, true, 0, true>(PREC, "k2gfirst<" #__VA_ARGS__ ">xF" #F "s"));  \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 0, true>(PREC, "k2glater<" #__VA_ARGS__ ">xF" #F "s"))
// the two kernels of the two-kernel Bluestein for one padded length
#define MI_BS2(T, PREC, F, SPLIT, ...)                                                                          \
    reg.push_back(make_k1bs<T, Sched<__VA_ARGS__>, F, SPLIT, 1>(PREC, "bluestein2_first<" #__VA_ARGS__ ">xF" #F)); \
    reg.push_back(make_k1bs<T, Sched<__VA_ARGS__>, F, SPLIT, 2>(PREC, "bluestein2_second<" #__VA_ARGS__ ">xF" #F))
// the three fused passes of the multi-kernel Bluestein for one tile height
#define MI_K2GF(T, PREC, F, ...)                                                                              \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, true, 1>(PREC, "k2gfirst_chirp<" #__VA_ARGS__ ">xF" #F)); \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 2>(PREC, "k2glast_mul<" #__VA_ARGS__ ">xF" #F));   \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 3>(PREC, "k2glast_chirp<" #__VA_ARGS__ ">xF" #F))
// the three fused passes of the multi-kernel Rader for one tile height
#define MI_K2GR(T, PREC, F, ...)                                                                                \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, true, 4>(PREC, "k2gfirst_gather<" #__VA_ARGS__ ">xF" #F));  \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 5>(PREC, "k2glast_rmul<" #__VA_ARGS__ ">xF" #F));    \
    reg.push_back(make_k2g<T, Sched<__VA_ARGS__>, F, false, 6>(PREC, "k2glast_scatter<" #__VA_ARGS__ ">xF" #F))
// column-tile passes of a prime tile height (Rader inside the tile); the Sched is the inner length P - 1
#define MI_K2R(T, PREC, F, ...)                                                                  \
    reg.push_back(make_k2r<T, Sched<__VA_ARGS__>, F, true>(PREC, "k2rfirst<" #__VA_ARGS__ ">xF" #F)); \
    reg.push_back(make_k2r<T, Sched<__VA_ARGS__>, F, false>(PREC, "k2rlater<" #__VA_ARGS__ ">xF" #F))
// Bluestein bodies take the linear exchange layout (SchedL): 164 -> 124 VGPRs for the power-of-two inner lengths
// Sub-pass 1's twiddle table of both transforms staged in LDS (kernels.h bluestein_body TW1, name suffix "t1"): a per-inner-length
// measured choice (round 3, shipped build vs a build with the tables staged in every body, 151 / 149 lengths over all inner
// lengths, one process: profiles/r3/ab_bluestein_tw1_{f32,f64}.jsonl) -- the inner lengths whose median gain is >= 4 %:
// f32 +6 ... 25 % (256 ... 1536: +7 ... 25 %), f64 +5 ... 8 %; the others are within +-3 % or lose (16384: -15 %, f64 3072: -14 %).
template <class T, class S> constexpr bool bs_tw1() {
#if defined(MI355_BS_TW1)  // A/B builds: every body
    return bluestein_tw1_ok<S>();
#else
    constexpr int M = S::N;
    if (sizeof(T) == 4)
        return bluestein_tw1_ok<S>() && (M == 256 || M == 320 || M == 384 || M == 512 || M == 640 || M == 1024 || M == 1280 || M == 1536 || M == 2560 || M == 3072 ||
                                         M == 4096 || M == 5120 || M == 7168 || M == 12288);  // (3072: together with bs_pf, round 5)
    return bluestein_tw1_ok<S>() && (M == 896 || M == 1024 || M == 1536 || M == 2560 || M == 4096);
#endif
}
// Round 5: where the sub-pass factors that are NOT staged in LDS come from (kernels.h bluestein_body PF) -- a per-inner-length measured choice of
// the Complex<f32> bodies (profiles/r5/ab_bs_stage_*.jsonl, one prime per inner length, all variants interleaved in one process):
//   1 = fetched one exchange AHEAD of their sub-pass instead of right behind its barrier: 2048 +1.7 %, 3584 +3.7 %, 3072 (with "t1") +6.2 %;
//   3 = every table but the last staged in LDS + the last sub-pass's factors fetched ahead: 6144 +10.9 %, 8192 +7.2 %;
//   0 = as before (4096, 5120, 7168: the staged sub-pass-1 table alone measured best; the prefetch alone loses 8 - 18 % there);
//   + 4 = the spectrum multiplier fetched in front of the first transform's last sub-pass, + 8 = the output chirp in front of the second one's
//   (profiles/r5/ab_bs_pre_*.jsonl): 2560 + 12: +3.0 %; within +-2 % or slower everywhere else (5120 / 7168: -23 ... -36 %).
// Kernel names carry the value as "p<PF>".
// Complex<f64> (profiles/r5/ab_bs_f64_*.jsonl): 1 at 2048 +5.5 %, 2560 (with "t1") +2.3 %, 3072 +4.3 %, 3584 +4.8 %, 5120 +8.0 %, 7168 +1.7 %, 8192 +8.8 %;
// 4096 and 6144 within +-1 %; staging more tables (2 / 3) loses up to 37 % there (an f64 row fills the LDS: fewer workgroups per CU).
template <class T, class S> constexpr int bs_pf() {
    constexpr int M = S::N;
    if (!bluestein_tw1_ok<S>()) return 0;
    if (sizeof(T) == 8) return (M == 2048 || M == 2560 || M == 3072 || M == 3584 || M == 5120 || M == 7168 || M == 8192) ? 1 : 0;
    return (M == 2048 || M == 3072 || M == 3584) ? 1 : (M == 8192 || M == 6144) ? 3 : M == 2560 ? 12 : 0;
}
inline const char* bs_name(const char* base, bool tw1, int pf) {
    std::string* s = new std::string(base);  // (registered once per kernel at start-up, lives as long as the registry)
    if (tw1 && !(pf & 2)) *s += "t1";
    if (pf) *s += "p" + std::to_string(pf);
    return s->c_str();
}
#define MI_BS(T, PREC, F, ...)                                                                                                            \
    reg.push_back(make_bluestein<T, SchedL<__VA_ARGS__>, F, false, bs_tw1<T, SchedL<__VA_ARGS__>>(), bs_pf<T, SchedL<__VA_ARGS__>>()>( \
        PREC, bs_name("bluestein<" #__VA_ARGS__ ">xF" #F, bs_tw1<T, SchedL<__VA_ARGS__>>(), bs_pf<T, SchedL<__VA_ARGS__>>())))
// one-kernel Bluestein through the split exchange (padded lengths above 8192: one workgroup per row)
#define MI_BSS(T, PREC, F, ...)                                                                                \
    reg.push_back(make_bluestein<T, SchedL<__VA_ARGS__>, F, true, bs_tw1<T, SchedL<__VA_ARGS__>>()>(            \
        PREC, bs_tw1<T, SchedL<__VA_ARGS__>>() ? "bluestein<" #__VA_ARGS__ ">xF" #F "st1" : "bluestein<" #__VA_ARGS__ ">xF" #F "s"))
#if defined(MI355_TUNING)
#define MI_BSV(V, T, PREC, F, ...)                                                                    \
    reg.push_back(make_bluestein<T, SchedL<__VA_ARGS__>, F>(PREC, "bluestein<" #__VA_ARGS__ ">xF" #F "v" #V)); \
    reg.back().variant = V
#define MI_BSPV(V, PF, T, PREC, F, ...)                                                                                          \
    reg.push_back(make_bluestein<T, SchedL<__VA_ARGS__>, F, false, ((PF) & 16) != 0, ((PF) & 15)>(PREC, "bluestein<" #__VA_ARGS__ ">xF" #F "pf" #PF "v" #V)); \
    reg.back().variant = V
#define MI_BSR2V(V, PF, T, PREC, ...)                                                                                            \
    reg.push_back(make_bluestein_rows2<T, SchedL<__VA_ARGS__>, false, ((PF) & 16) != 0, ((PF) & 15)>(PREC, "bluestein<" #__VA_ARGS__ ">xR2pf" #PF "v" #V)); \
    reg.back().variant = V
#define MI_BSSV(V, T, PREC, F, ...)                                                                          \
    reg.push_back(make_bluestein<T, SchedL<__VA_ARGS__>, F, true>(PREC, "bluestein<" #__VA_ARGS__ ">xF" #F "sv" #V)); \
    reg.back().variant = V
#else
#define MI_BSV(V, T, PREC, F, ...) (void)0
#define MI_BSSV(V, T, PREC, F, ...) (void)0
#define MI_BSPV(V, PF, T, PREC, F, ...) (void)0
#define MI_BSR2V(V, PF, T, PREC, ...) (void)0
#endif
#define MI_RADER(T, PREC, F, MODE, ...) reg.push_back(make_rader<T, Sched<__VA_ARGS__>, F, MODE>(PREC, "rader<" #__VA_ARGS__ ">xF" #F "m" #MODE))
#if defined(MI355_TUNING)
#define MI_RADERV(V, T, PREC, F, MODE, ...)                                                                   \
    reg.push_back(make_rader<T, Sched<__VA_ARGS__>, F, MODE>(PREC, "rader<" #__VA_ARGS__ ">xF" #F "m" #MODE "v" #V)); \
    reg.back().variant = V
#else
#define MI_RADERV(V, T, PREC, F, MODE, ...) (void)0
#endif

}  // namespace mi355
