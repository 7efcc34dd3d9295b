// Round-5 skeleton of the 2048-row x 16-column tile (2^22 = 2048 x 2048, Complex<float>; config 5's per-GPU kernel): ONE workgroup of 1024
// threads per CU holds the 256 KiB tile in registers, so nothing overlaps its memory phases with its exchanges.  Question: does the
// SAME workgroup, run as two column HALVES (2048 x 8 each: the first / second 64 bytes of every 128-byte row segment) one after the
// other -- both halves' loads issued up front, half A worked on while half B is still landing, half A's stores in flight while half B
// is worked on -- move the tile faster, and does it matter that the two 64-byte halves of a line are then written microseconds apart?
//   plain   : 32 values per thread, split exchange (the shipped structure)
//   halves/T: loads A, loads B, work A, work B, stores A + B together
//   halves/E: loads A, loads B, work A, stores A, work B, stores B     (early stores: 64-byte halves of a line written apart)
// work = `lds_rounds` LDS exchanges (A half-tile fits LDS unsplit) + dependent FMA sweeps.  GB/s = read + write bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int LOGN = 22, ROWS = 2048, PITCH = 2048;  // later-pass shape: element (row r, column c) of transform g at g 2^22 + r 2048 + c

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XCD-aware order over the 128 tiles of a transform (16 columns = 128 bytes each: XCD bits are tile-index bits 2..4)
__device__ __forceinline__ void tile_of(long long b, int order, long long& g, int& j) {
    g = b >> 7;
    int r = (int)(b & 127);
    if (order) {
        const int x = r & 7, i = r >> 3;  // x: the XCD this workgroup runs on (b % 8)
        r = ((i >> 2) << 5) | (x << 2) | (i & 3);
    }
    j = r;
}

template <int NT, int E, bool SPLIT> __device__ __forceinline__ void fake_work(v2 (&v)[E], v2* lds, int lds_rounds, int fma_iters, float a, float b) {
    const int tid = threadIdx.x;
    for (int r = 0; r < lds_rounds; ++r) {
        for (int it = 0; it < fma_iters; ++it) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                v[k].x = __builtin_fmaf(v[k].x, a, b);
                v[k].y = __builtin_fmaf(v[k].y, a, b);
            }
        }
        if (!lds) continue;
        if constexpr (!SPLIT) {
#pragma unroll
            for (int k = 0; k < E; ++k) lds[(tid + k * NT) + ((tid + k * NT) >> 5)] = v[k];
            lds_barrier();
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = (tid * E + k) & (NT * E - 1);
                v[k] = lds[i + (i >> 5)];
            }
            lds_barrier();
        } else {
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

// plain: 16 lanes across the 16 columns, thread (f, u) holds rows u + 64 k, k < 32
__global__ __launch_bounds__(1024, 4) void tile_plain(const v2* __restrict__ in, v2* __restrict__ out, int order, int lds_rounds, int fma_iters, int use_lds, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long g;
    int j;
    tile_of(blockIdx.x, order, g, j);
    const int f = threadIdx.x & 15, u = threadIdx.x >> 4;
    const size_t base = ((size_t)g << LOGN) + (size_t)u * PITCH + j * 16 + f;
    v2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = in[base + (size_t)k * 64 * PITCH];
    fake_work<1024, 32, true>(v, use_lds ? (v2*)smem : nullptr, lds_rounds, fma_iters, a, b);
#pragma unroll
    for (int k = 0; k < 32; ++k) out[base + (size_t)k * 64 * PITCH] = v[k];
}

// halves: 8 lanes across the 8 columns of a half, thread (f, u) holds rows u + 128 k, k < 16, of half A and of half B
template <int EARLY> __global__ __launch_bounds__(1024, 4) void tile_halves(const v2* __restrict__ in, v2* __restrict__ out, int order, int lds_rounds, int fma_iters, int use_lds, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long g;
    int j;
    tile_of(blockIdx.x, order, g, j);
    const int f = threadIdx.x & 7, u = threadIdx.x >> 3;
    const size_t base = ((size_t)g << LOGN) + (size_t)u * PITCH + j * 16 + f;
    v2 va[16], vb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) va[k] = in[base + (size_t)k * 128 * PITCH];
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = in[base + 8 + (size_t)k * 128 * PITCH];
    v2* lds = use_lds ? (v2*)smem : nullptr;
    fake_work<1024, 16, false>(va, lds, lds_rounds, fma_iters, a, b);
    if (EARLY) {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[base + (size_t)k * 128 * PITCH] = va[k];
    }
    fake_work<1024, 16, false>(vb, lds, lds_rounds, fma_iters, a, b);
    if (!EARLY) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            out[base + (size_t)k * 128 * PITCH] = va[k];
            out[base + 8 + (size_t)k * 128 * PITCH] = vb[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[base + 8 + (size_t)k * 128 * PITCH] = vb[k];
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
    const size_t bytes = (size_t)4 << 30;  // 128 transforms of 2^22 complex<f32>
    void *a, *b;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 0, bytes);
    (void)hipMemset(b, 0, bytes);
    const int ntiles = 128 * 128;
    auto rw = [&](const char* name, float ms) {
        printf("%-84s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
    };
    char nm[200];
#define OPTIN(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    OPTIN(tile_plain);
    OPTIN(tile_halves<0>);
    OPTIN(tile_halves<1>);
    const float fa = 1.0f, fb = 0.0f;
    const int levels[][2] = {{0, 0}, {2, 0}, {2, 6}, {2, 12}, {2, 18}, {3, 12}};
    for (int order = 1; order >= 0; --order)
        for (auto& lv : levels) {
            const int R = lv[0], C = lv[1], use_lds = R > 0;
            const int Rr = R > 0 ? R : 1, Cc = R > 0 ? C : 0;
            if (order == 0 && !(R == 2 && C == 12)) continue;
            snprintf(nm, sizeof nm, "order %d  plain 1024thr x32, split exchange            work(lds %d, fma %d)", order, R, C);
            rw(nm, time_it([&] { tile_plain<<<ntiles, 1024, 136 * 1024>>>((v2*)a, (v2*)b, order, Rr, Cc, use_lds, fa, fb); }));
            snprintf(nm, sizeof nm, "order %d  halves, stores together                      work(lds %d, fma %d)", order, R, C);
            rw(nm, time_it([&] { tile_halves<0><<<ntiles, 1024, 136 * 1024>>>((v2*)a, (v2*)b, order, Rr, Cc, use_lds, fa, fb); }));
            snprintf(nm, sizeof nm, "order %d  halves, half A stored before half B's work   work(lds %d, fma %d)", order, R, C);
            rw(nm, time_it([&] { tile_halves<1><<<ntiles, 1024, 136 * 1024>>>((v2*)a, (v2*)b, order, Rr, Cc, use_lds, fa, fb); }));
        }
    return 0;
}
