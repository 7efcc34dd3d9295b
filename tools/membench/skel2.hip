// Round-3 skeletons of the 1024-row x 16-column tile pass (2^20 = 1024 x 1024, Complex<float>): does a SOFTWARE-PIPELINED
// persistent workgroup (next tile's loads issued before the current tile's arithmetic, one workgroup per CU) move the tile
// faster than the shipped structure (one workgroup per tile, two per CU, overlap by chance)?  Unlike tools/membench/skel.hip
// the tiles here carry a stand-in for the transform's work between load and store: `lds_rounds` full-tile LDS exchanges
// (write, barrier, transposed read, barrier) and `fma_iters` dependent FMA sweeps over the thread's values, so the
// memory / arithmetic overlap is part of what is measured.  GB/s = read + write bytes.
//   shape "first": strided read (128-byte row segments at an 8 KiB pitch), contiguous 128 KiB write
//   shape "later": strided read, strided write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

// XCD-aware order of the shipped kernels (kernels.h k2_body, xp = 2, xq = 3 for 64 tiles per transform)
__device__ __forceinline__ void tile_of(long long b, int order, long long& g, int& j) {
    g = b >> 6;
    int r = (int)(b & 63);
    if (order) {
        const int x = r & 7, i = r >> 3;
        r = ((i >> 2) << 5) | (x << 2) | (i & 3);
    }
    j = r;
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the stand-in for the transform: LDS exchanges + dependent FMAs
template <int NT, int E> __device__ __forceinline__ void fake_work(v2 (&v)[E], v2* lds, int lds_rounds, int fma_iters, float a, float b, bool split = false) {
    const int tid = threadIdx.x;
    for (int r = 0; r < lds_rounds; ++r) {
        for (int it = 0; it < fma_iters; ++it) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                v[k].x = __builtin_fmaf(v[k].x, a, b);
                v[k].y = __builtin_fmaf(v[k].y, a, b);
            }
        }
        if (lds && !split) {
#pragma unroll
            for (int k = 0; k < E; ++k) lds[(tid + k * NT) + ((tid + k * NT) >> 5)] = v[k];
            lds_barrier();
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = (tid * E + k) & (NT * E - 1);
                v[k] = lds[i + (i >> 5)];
            }
            lds_barrier();
        } else if (lds) {  // real plane, then imaginary plane through a half-size buffer (the shipped split exchange)
            float* p = (float*)lds;
#pragma unroll
            for (int k = 0; k < E; ++k) p[(tid + k * NT) + ((tid + k * NT) >> 5)] = v[k].x;
            lds_barrier();
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = (tid * E + k) & (NT * E - 1);
                v[k].x = p[i + (i >> 5)];
            }
            lds_barrier();
#pragma unroll
            for (int k = 0; k < E; ++k) p[(tid + k * NT) + ((tid + k * NT) >> 5)] = v[k].y;
            lds_barrier();
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = (tid * E + k) & (NT * E - 1);
                v[k].y = p[i + (i >> 5)];
            }
            lds_barrier();
        }
    }
}

// ---- float2 lanes: 16 lanes walk across the tile's columns ------------------------------------------------------
template <int NT, int E> __device__ __forceinline__ void load_v2(v2 (&v)[E], const v2* in, long long b, int order) {
    constexpr int RS = NT / 16;
    const int f = threadIdx.x & 15, u = threadIdx.x >> 4;
    long long g;
    int j;
    tile_of(b, order, g, j);
    const v2* src = in + (g << 20) + j * 16 + f + (size_t)u * 1024;
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = src[(size_t)k * RS * 1024];
}
template <int NT, int E> __device__ __forceinline__ void store_v2(const v2 (&v)[E], v2* out, long long b, int order, int later) {
    constexpr int RS = NT / 16;
    const int f = threadIdx.x & 15, u = threadIdx.x >> 4;
    long long g;
    int j;
    tile_of(b, order, g, j);
    if (later) {
        v2* dst = out + (g << 20) + j * 16 + f + (size_t)u * 1024;
#pragma unroll
        for (int k = 0; k < E; ++k) dst[(size_t)k * RS * 1024] = v[k];
    } else {
        v2* dst = out + (g << 20) + (size_t)j * 16384 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < E; ++k) dst[k * NT] = v[k];
    }
}
// ---- float4 lanes: 8 lanes per 128-byte segment, a lane moves two adjacent columns -------------------------------
template <int NT, int E> __device__ __forceinline__ void load_v4(v2 (&v)[E], const v2* in, long long b, int order) {
    constexpr int RS = NT / 8;
    const int f = threadIdx.x & 7, u = threadIdx.x >> 3;
    long long g;
    int j;
    tile_of(b, order, g, j);
    const v4* src = (const v4*)(in + (g << 20) + j * 16 + 2 * f + (size_t)u * 1024);
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
        v4 t = src[(size_t)k * RS * 512];
        v[2 * k] = v2{t.x, t.y};
        v[2 * k + 1] = v2{t.z, t.w};
    }
}
template <int NT, int E> __device__ __forceinline__ void store_v4(const v2 (&v)[E], v2* out, long long b, int order, int later) {
    constexpr int RS = NT / 8;
    const int f = threadIdx.x & 7, u = threadIdx.x >> 3;
    long long g;
    int j;
    tile_of(b, order, g, j);
    if (later) {
        v4* dst = (v4*)(out + (g << 20) + j * 16 + 2 * f + (size_t)u * 1024);
#pragma unroll
        for (int k = 0; k < E / 2; ++k) dst[(size_t)k * RS * 512] = v4{v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y};
    } else {
        v4* dst = (v4*)(out + (g << 20) + (size_t)j * 16384) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < E / 2; ++k) dst[k * NT] = v4{v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y};
    }
}

// one workgroup per tile (the shipped structure)
template <int NT, int WPS, int VW> __global__ __launch_bounds__(NT, WPS) void tile_plain(const v2* __restrict__ in, v2* __restrict__ out, int later, int order, int lds_rounds, int fma_iters, int use_lds, float a, float b) {
    constexpr int E = 16384 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2 v[E];
    if constexpr (VW == 2)
        load_v2<NT, E>(v, in, blockIdx.x, order);
    else
        load_v4<NT, E>(v, in, blockIdx.x, order);
    fake_work<NT, E>(v, use_lds ? (v2*)smem : nullptr, lds_rounds, fma_iters, a, b, true);
    if constexpr (VW == 2)
        store_v2<NT, E>(v, out, blockIdx.x, order, later);
    else
        store_v4<NT, E>(v, out, blockIdx.x, order, later);
}

// persistent, software-pipelined: tile t + G is in flight while tile t is worked on and stored
template <int NT, int WPS, int VW> __global__ __launch_bounds__(NT, WPS) void tile_pipe(const v2* __restrict__ in, v2* __restrict__ out, int later, int order, int lds_rounds, int fma_iters, int use_lds, float a, float b, long long ntiles) {
    constexpr int E = 16384 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2* lds = use_lds ? (v2*)smem : nullptr;
    v2 va[E], vb[E];
    const long long G = gridDim.x;
    long long t = blockIdx.x;
    auto ld = [&](v2 (&v)[E], long long tt) {
        if constexpr (VW == 2)
            load_v2<NT, E>(v, in, tt, order);
        else
            load_v4<NT, E>(v, in, tt, order);
    };
    auto st = [&](const v2 (&v)[E], long long tt) {
        if constexpr (VW == 2)
            store_v2<NT, E>(v, out, tt, order, later);
        else
            store_v4<NT, E>(v, out, tt, order, later);
    };
    // the prefetch is UNCONDITIONAL (the tail re-reads the last tile): a conditional load makes the compiler's wait-count
    // bookkeeping assume the shorter queue on every path, i.e. wait for the prefetch itself before touching the current tile
    const long long last = ntiles - 1;
    if (t >= ntiles) return;
    ld(va, t);
    while (true) {
        ld(vb, t + G < ntiles ? t + G : last);
        fake_work<NT, E>(va, lds, lds_rounds, fma_iters, a, b);
        st(va, t);
        t += G;
        if (t >= ntiles) break;
        // one loop body + a register copy (the wait for the prefetched tile sits here, behind the stores just issued: vmcnt(E)).
        // Unrolling by two with swapped roles instead makes the loop header a join of two different queue states, and the
        // compiler then waits for the stores it has just issued.
#pragma unroll
        for (int k = 0; k < E; ++k) va[k] = vb[k];
    }
}

template <class K> float time_it(K&& launch, int reps = 5) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)4 << 30;  // 512 transforms of 2^20 complex<f32>
    void *a, *b;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 0, bytes);
    (void)hipMemset(b, 0, bytes);
    const long long ntiles = 512 * 64;
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    auto rw = [&](const char* name, float ms) {
        printf("%-92s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
    };
    char nm[200];
#define OPTIN(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    OPTIN((tile_plain<512, 4, 2>));
    OPTIN((tile_plain<512, 4, 4>));
    OPTIN((tile_pipe<512, 2, 2>));
    OPTIN((tile_pipe<512, 2, 4>));
    OPTIN((tile_pipe<1024, 4, 2>));
    OPTIN((tile_pipe<1024, 4, 4>));
    OPTIN((tile_pipe<256, 1, 2>));
    const float fa = 1.0f, fb = 0.0f;
    // work levels: (lds_rounds, fma_iters per round).  The shipped later pass issues ~1600 VALU instructions per wave of 64
    // threads x 32 values = 50 per value; a sweep of 2 FMAs per value and iteration makes (2, 12) ~ 48 per value.
    const int levels[][2] = {{0, 0}, {2, 0}, {2, 6}, {2, 12}, {2, 18}};
    for (int later = 0; later < 2; ++later) {
        const char* sh = later ? "later" : "first";
        for (auto& lv : levels) {
            const int R = lv[0], C = lv[1], use_lds = R > 0;
            const int Rr = R > 0 ? R : 1, Cc = R > 0 ? C : 0;
            if (quick && !(R == 2 && C == 12) && R != 0) continue;
            snprintf(nm, sizeof nm, "%s plain 512thr x32 v2, 2 WG/CU            work(lds %d, fma %d)", sh, R, C);
            rw(nm, time_it([&] { tile_plain<512, 4, 2><<<ntiles, 512, 66 * 1024>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb); }));
            snprintf(nm, sizeof nm, "%s plain 512thr x32 v4, 2 WG/CU            work(lds %d, fma %d)", sh, R, C);
            rw(nm, time_it([&] { tile_plain<512, 4, 4><<<ntiles, 512, 66 * 1024>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb); }));
            for (int gm : {1, 2}) {
                if (gm == 2 && use_lds) continue;  // two 512-thread pipelines per CU: registers allow one
                snprintf(nm, sizeof nm, "%s pipe  512thr x32 v2, grid %d/CU          work(lds %d, fma %d)", sh, gm, R, C);
                rw(nm, time_it([&] { tile_pipe<512, 2, 2><<<256 * gm, 512, use_lds ? 136 * 1024 : 0>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb, ntiles); }));
            }
            snprintf(nm, sizeof nm, "%s pipe  512thr x32 v4, grid 1/CU          work(lds %d, fma %d)", sh, R, C);
            rw(nm, time_it([&] { tile_pipe<512, 2, 4><<<256, 512, use_lds ? 136 * 1024 : 0>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb, ntiles); }));
            snprintf(nm, sizeof nm, "%s pipe 1024thr x16 v2, grid 1/CU          work(lds %d, fma %d)", sh, R, C);
            rw(nm, time_it([&] { tile_pipe<1024, 4, 2><<<256, 1024, use_lds ? 136 * 1024 : 0>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb, ntiles); }));
            snprintf(nm, sizeof nm, "%s pipe 1024thr x16 v4, grid 1/CU          work(lds %d, fma %d)", sh, R, C);
            rw(nm, time_it([&] { tile_pipe<1024, 4, 4><<<256, 1024, use_lds ? 136 * 1024 : 0>>>((v2*)a, (v2*)b, later, 1, Rr, Cc, use_lds, fa, fb, ntiles); }));
        }
    }
    return 0;
}
