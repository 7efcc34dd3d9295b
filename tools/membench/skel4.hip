// Round-5 skeleton of the whole-row kernels (K1: one workgroup = one contiguous row of 4096 Complex<float> = 32 KiB, 256 threads, 16 values
// per thread, contiguous read + contiguous write; 2^10 .. 2^14 run 0.64 - 0.69 of 8 TB/s, VERDICT r4 item 6 asks for 0.70): where is the
// ceiling of this shape, and do the levers that did not fit the column tiles move it?
//   plain     : global_load_dwordx2 straight into the butterfly layout (the shipped structure), 8-byte stores
//   v4        : 16-byte loads and stores (row-contiguous register layout)
//   nt        : plain with non-temporal loads / stores
//   dma       : the row arrives by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging VGPRs), is gathered from LDS
//   dma-pipe  : persistent workgroups, two row buffers: the NEXT row's DMA is in flight while the current row is worked on and stored
// work = `lds_rounds` LDS exchanges + dependent FMA sweeps (stand-in for the transform).  GB/s = read + write bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096, NT = 256, E = 16;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void fake_work(v2 (&v)[E], v2* lds, int lds_rounds, int fma_iters, float a, float b) {
    const int tid = threadIdx.x;
    for (int r = 0; r < lds_rounds; ++r) {
        for (int it = 0; it < fma_iters; ++it) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                v[k].x = __builtin_fmaf(v[k].x, a, b);
                v[k].y = __builtin_fmaf(v[k].y, a, b);
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) lds[(tid + k * NT) + ((tid + k * NT) >> 5)] = v[k];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int i = (tid * E + k) & (N - 1);
            v[k] = lds[i + (i >> 5)];
        }
        lds_barrier();
    }
}

template <int MODE>  // 0 plain, 1 v4, 2 nt
__global__ __launch_bounds__(NT) void row_plain(const v2* __restrict__ in, v2* __restrict__ out, int lds_rounds, int fma_iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t base = (size_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    v2 v[E];
    if constexpr (MODE == 1) {
        const v4* s = (const v4*)(in + base);
#pragma unroll
        for (int k = 0; k < E / 2; ++k) {
            v4 t = s[tid + k * NT];
            v[2 * k] = v2{t.x, t.y};
            v[2 * k + 1] = v2{t.z, t.w};
        }
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = MODE == 2 ? __builtin_nontemporal_load(in + base + tid + k * NT) : in[base + tid + k * NT];
    }
    fake_work(v, (v2*)smem, lds_rounds, fma_iters, a, b);
    if constexpr (MODE == 1) {
        v4* d = (v4*)(out + base);
#pragma unroll
        for (int k = 0; k < E / 2; ++k) d[tid + k * NT] = v4{v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y};
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if constexpr (MODE == 2)
                __builtin_nontemporal_store(v[k], out + base + tid + k * NT);
            else
                out[base + tid + k * NT] = v[k];
        }
    }
}

// one row by LDS-DMA into `buf` (32 KiB, lane-linear): wave w's instruction i moves bytes [(i 4 + w) 1024, + 1024) of the row
__device__ __forceinline__ void dma_row(const v2* row, char* buf) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* g = (const char*)row + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int piece = i * 4 + w;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
    }
}

template <int V4ST> __global__ __launch_bounds__(NT) void row_dma(const v2* __restrict__ in, v2* __restrict__ out, int lds_rounds, int fma_iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t base = (size_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    dma_row(in + base, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    v2 v[E];
    const v2* l = (const v2*)smem;
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = l[tid + k * NT];
    lds_barrier();  // the row buffer becomes the exchange buffer
    fake_work(v, (v2*)(smem + 0), lds_rounds, fma_iters, a, b);
    if constexpr (V4ST) {
        // the fake exchange leaves thread t with elements t E .. t E + E - 1: 16-byte stores of neighbours
        v4* d = (v4*)(out + base) + tid * (E / 2);
#pragma unroll
        for (int k = 0; k < E / 2; ++k) d[k] = v4{v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y};
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) out[base + tid + k * NT] = v[k];
    }
}

// persistent: rows blockIdx.x, + gridDim.x, ...; buffers A / B alternate; exchange buffer separate (3 x 33 KiB = 2 workgroups per CU at most ... 1 with 160 KiB? no: 99 KiB -> 1 per CU; so the exchange aliases the row buffer just consumed)
__global__ __launch_bounds__(NT) void row_dma_pipe(const v2* __restrict__ in, v2* __restrict__ out, int lds_rounds, int fma_iters, float a, float b, int rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, G = gridDim.x;
    constexpr int BUF = 34 * 1024;  // a row + the exchange layout's padding
    int r = blockIdx.x;
    if (r >= rows) return;
    dma_row(in + (size_t)r * N, smem);
    int cur = 0;
    bool first = true;
    while (true) {
        const int rn = r + G < rows ? r + G : r;  // unconditional prefetch (the tail re-reads its own row)
        dma_row(in + (size_t)rn * N, smem + (cur ^ 1) * BUF);
        // outstanding, in issue order: [DMA(cur) 8] [stores of the previous row 16 (first iteration: none)] [DMA(next) 8]: the current row
        // has landed when at most 24 operations are outstanding
        if (first)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // no stores yet: [DMA(cur) 8] [DMA(next) 8]
        else
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        v2 v[E];
        const v2* l = (const v2*)(smem + cur * BUF);
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = l[tid + k * NT];
        lds_barrier();
        fake_work(v, (v2*)(smem + cur * BUF), lds_rounds, fma_iters, a, b);
#pragma unroll
        for (int k = 0; k < E; ++k) out[(size_t)r * N + tid + k * NT] = v[k];
        if (r + G >= rows) break;
        r += G;
        cur ^= 1;
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

int main() {
    const size_t bytes = (size_t)4 << 30;
    void *a, *b;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 0, bytes);
    (void)hipMemset(b, 0, bytes);
    const int rows = (int)(bytes / (N * 8));
    auto rw = [&](const char* name, float ms) {
        printf("%-84s %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0);
        fflush(stdout);
    };
    char nm[200];
#define OPTIN(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    OPTIN(row_plain<0>);
    OPTIN(row_plain<1>);
    OPTIN(row_plain<2>);
    OPTIN(row_dma<0>);
    OPTIN(row_dma<1>);
    OPTIN(row_dma_pipe);
    const float fa = 1.0f, fb = 0.0f;
    const int levels[][2] = {{1, 0}, {2, 6}, {2, 12}};
    for (int inplace = 0; inplace < 2; ++inplace) {
        void* o = inplace ? a : b;
        for (auto& lv : levels) {
            const int R = lv[0], C = lv[1];
            const char* ip = inplace ? "in place " : "a -> b   ";
            snprintf(nm, sizeof nm, "%s plain dwordx2                               work(lds %d, fma %d)", ip, R, C);
            rw(nm, time_it([&] { row_plain<0><<<rows, NT, 34 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb); }));
            snprintf(nm, sizeof nm, "%s dwordx4 loads and stores                    work(lds %d, fma %d)", ip, R, C);
            rw(nm, time_it([&] { row_plain<1><<<rows, NT, 34 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb); }));
            snprintf(nm, sizeof nm, "%s non-temporal dwordx2                        work(lds %d, fma %d)", ip, R, C);
            rw(nm, time_it([&] { row_plain<2><<<rows, NT, 34 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb); }));
            snprintf(nm, sizeof nm, "%s LDS-DMA loads, dwordx2 stores               work(lds %d, fma %d)", ip, R, C);
            rw(nm, time_it([&] { row_dma<0><<<rows, NT, 34 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb); }));
            snprintf(nm, sizeof nm, "%s LDS-DMA loads, dwordx4 stores               work(lds %d, fma %d)", ip, R, C);
            rw(nm, time_it([&] { row_dma<1><<<rows, NT, 34 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb); }));
            for (int per_cu : {2, 1}) {
                snprintf(nm, sizeof nm, "%s LDS-DMA persistent double buffer, %d WG/CU    work(lds %d, fma %d)", ip, per_cu, R, C);
                rw(nm, time_it([&] { row_dma_pipe<<<256 * per_cu, NT, 68 * 1024>>>((v2*)a, (v2*)o, R, C, fa, fb, rows); }));
            }
        }
    }
    return 0;
}
