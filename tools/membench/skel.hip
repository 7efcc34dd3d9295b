// Data-movement skeletons of the 2^20 = 1024 x 1024 two-pass plan (no arithmetic): which launch structure moves a
// 1024-row x 16-column tile (128-byte row segments, row pitch 8 KiB) fastest?  GB/s = read + write bytes.
// Shapes: "first" = strided read, contiguous 128 KiB write;  "later" = strided read, strided write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

// tile index -> (transform g, tile j); ORDER 0: block b = g * 64 + j (adjacent tiles on different XCDs);
// ORDER 1: XCD x = b % 8 owns 8 ADJACENT tiles j = 8 x .. 8 x + 7 of each transform
__device__ __forceinline__ void tile_of(long long b, int order, long long& g, int& j) {
    if (order == 0) {
        g = b / 64;
        j = (int)(b % 64);
    } else if (order == 1) {
        const long long grp = b / 64;
        const int r = (int)(b % 64), x = r % 8, i = r / 8;
        g = grp;
        j = 8 * x + i;
    } else if (order == 2) {  // XCD x owns whole transforms: 8 transforms in flight, one per XCD
        const long long grp = b / 512;
        const int r = (int)(b % 512), x = r % 8, i = r / 8;
        g = grp * 8 + x;
        j = i;
    } else if (order == 3) {  // XCD x owns 16 adjacent tiles of each of two transforms
        const long long grp = b / 128;
        const int r = (int)(b % 128), x = r % 8, i = r / 8;  // i in 0..15
        g = grp * 2 + (x / 4);
        j = 16 * (x % 4) + i;
    } else {  // order 4: XCD x owns 4 adjacent tiles, two such groups per transform
        const long long grp = b / 64;
        const int r = (int)(b % 64), x = r % 8, i = r / 8;  // i in 0..7
        g = grp;
        j = 4 * x + (i % 4) + 32 * (i / 4);
    }
}

// NT threads, E = 16384 / NT values (float2) per thread; lanes walk across the 16 columns first
template <int NT, int WPS> __global__ __launch_bounds__(NT, WPS) void tile_v2(const v2* __restrict__ in, v2* __restrict__ out, int later, int order, int pitch, int persistent, long long ntiles) {
    constexpr int E = 16384 / NT, RS = NT / 16;
    const int f = threadIdx.x % 16, u = threadIdx.x / 16;
    extern __shared__ char smem[];
    if (pitch == 1 && threadIdx.x == 0) smem[0] = 1;
    v2 v[E];
    for (long long b = blockIdx.x; b < ntiles; b += (persistent ? gridDim.x : ntiles)) {
        long long g;
        int j;
        tile_of(b, order, g, j);
        const v2* src = in + g * (1024LL * pitch) + j * 16;
        v2* dst = out + g * (1024LL * pitch);
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = src[f + (size_t)(u + k * RS) * pitch];
        if (!persistent) __syncthreads();
        if (later) {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[j * 16 + f + (size_t)(u + k * RS) * pitch] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[(size_t)j * 16384 + threadIdx.x + k * NT] = v[k];
        }
    }
}
// float4 lanes: a lane moves two adjacent columns (8 lanes per 128-byte segment)
template <int NT, int WPS> __global__ __launch_bounds__(NT, WPS) void tile_v4(const v4* __restrict__ in, v4* __restrict__ out, int later, int order, int pitch4, int persistent, long long ntiles) {
    constexpr int E = 8192 / NT, RS = NT / 8;
    const int f = threadIdx.x % 8, u = threadIdx.x / 8;
    extern __shared__ char smem[];
    if (pitch4 == 1 && threadIdx.x == 0) smem[0] = 1;
    v4 v[E];
    for (long long b = blockIdx.x; b < ntiles; b += (persistent ? gridDim.x : ntiles)) {
        long long g;
        int j;
        tile_of(b, order, g, j);
        const v4* src = in + g * (1024LL * pitch4) + j * 8;
        v4* dst = out + g * (1024LL * pitch4);
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = src[f + (size_t)(u + k * RS) * pitch4];
        if (!persistent) __syncthreads();
        if (later) {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[j * 8 + f + (size_t)(u + k * RS) * pitch4] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[(size_t)j * 8192 + threadIdx.x + k * NT] = v[k];
        }
    }
}
// chunked: the tile goes through the registers in CH chunks (load chunk, store chunk): shorter bursts, same bytes
template <int NT, int WPS, int CH> __global__ __launch_bounds__(NT, WPS) void tile_chunked(const v2* __restrict__ in, v2* __restrict__ out, int later, int order, int pitch) {
    constexpr int E = 16384 / NT / CH, RS = NT / 16;
    const int f = threadIdx.x % 16, u = threadIdx.x / 16;
    extern __shared__ char smem[];
    if (pitch == 1 && threadIdx.x == 0) smem[0] = 1;
    long long g;
    int j;
    tile_of(blockIdx.x, order, g, j);
    const v2* src = in + g * (1024LL * pitch) + j * 16;
    v2* dst = out + g * (1024LL * pitch);
    for (int c = 0; c < CH; ++c) {
        v2 v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = src[f + (size_t)(u + (c * E + k) * RS) * pitch];
        if (later) {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[j * 16 + f + (size_t)(u + (c * E + k) * RS) * pitch] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) dst[(size_t)j * 16384 + threadIdx.x + (c * E + k) * NT] = v[k];
        }
    }
}

template <class K> float time_it(K&& launch, int reps = 6) {
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
    const size_t bytes = (size_t)4 << 30;  // 512 transforms of 2^20 complex<f32>
    void *a, *b;
    (void)hipMalloc(&a, bytes + ((size_t)1 << 30));
    (void)hipMalloc(&b, bytes + ((size_t)1 << 30));
    (void)hipMemset(a, 1, bytes + ((size_t)1 << 30));
    (void)hipMemset(b, 2, bytes + ((size_t)1 << 30));
    const long long ntr = 512, ntiles = ntr * 64;
    auto rw = [&](const char* name, float ms) { printf("%-78s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
    char nm[160];
#define OPTIN(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 150000)
    OPTIN((tile_v2<512, 4>)); OPTIN((tile_v2<1024, 8>)); OPTIN((tile_v2<256, 1>)); OPTIN((tile_v2<512, 1>)); OPTIN((tile_v4<512, 4>)); OPTIN((tile_v4<256, 2>)); OPTIN((tile_v4<1024, 8>));
    OPTIN((tile_chunked<512, 4, 2>)); OPTIN((tile_chunked<512, 4, 4>)); OPTIN((tile_chunked<512, 4, 8>)); OPTIN((tile_chunked<512, 1, 8>)); OPTIN((tile_chunked<256, 1, 16>));
    for (int later = 0; later < 2; ++later) {
        const char* sh = later ? "later" : "first";
        for (int order = 0; order < 5; ++order) {
            snprintf(nm, sizeof nm, "%s v2 512thr x32 2WG/CU order %d", sh, order);
            rw(nm, time_it([&] { tile_v2<512, 4><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, order, 1024, 0, ntiles); }));
        }
        snprintf(nm, sizeof nm, "%s v2 1024thr x16 2WG/CU", sh);
        rw(nm, time_it([&] { tile_v2<1024, 8><<<ntiles, 1024, 70000>>>((v2*)a, (v2*)b, later, 0, 1024, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v2 256thr x64 (no cap)", sh);
        rw(nm, time_it([&] { tile_v2<256, 1><<<ntiles, 256, 0>>>((v2*)a, (v2*)b, later, 0, 1024, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v2 512thr x32 no reg cap", sh);
        rw(nm, time_it([&] { tile_v2<512, 1><<<ntiles, 512, 0>>>((v2*)a, (v2*)b, later, 0, 1024, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v4 512thr x16 2WG/CU", sh);
        rw(nm, time_it([&] { tile_v4<512, 4><<<ntiles, 512, 70000>>>((v4*)a, (v4*)b, later, 0, 512, 0, ntiles); }));
        for (int order = 1; order < 5; ++order) {
            snprintf(nm, sizeof nm, "%s v4 512thr x16 2WG/CU order %d", sh, order);
            rw(nm, time_it([&] { tile_v4<512, 4><<<ntiles, 512, 70000>>>((v4*)a, (v4*)b, later, order, 512, 0, ntiles); }));
        }
        snprintf(nm, sizeof nm, "%s v2 1024thr x16 2WG/CU order 1", sh);
        rw(nm, time_it([&] { tile_v2<1024, 8><<<ntiles, 1024, 70000>>>((v2*)a, (v2*)b, later, 1, 1024, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v4 256thr x32 2WG/CU", sh);
        rw(nm, time_it([&] { tile_v4<256, 2><<<ntiles, 256, 70000>>>((v4*)a, (v4*)b, later, 0, 512, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v4 1024thr x8", sh);
        rw(nm, time_it([&] { tile_v4<1024, 8><<<ntiles, 1024, 70000>>>((v4*)a, (v4*)b, later, 0, 512, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v2 512thr x32 padded pitch 1024+16", sh);
        rw(nm, time_it([&] { tile_v2<512, 4><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1040, 0, ntiles); }));
        snprintf(nm, sizeof nm, "%s v2 512thr x32 padded pitch 1024+32", sh);
        rw(nm, time_it([&] { tile_v2<512, 4><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1056, 0, ntiles); }));
        for (int g : {512, 1024}) {
            snprintf(nm, sizeof nm, "%s v2 512thr x32 persistent grid %d", sh, g);
            rw(nm, time_it([&] { tile_v2<512, 4><<<g, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1024, 1, ntiles); }));
            snprintf(nm, sizeof nm, "%s v2 512thr x32 persistent grid %d order 1", sh, g);
            rw(nm, time_it([&] { tile_v2<512, 4><<<g, 512, 70000>>>((v2*)a, (v2*)b, later, 1, 1024, 1, ntiles); }));
            snprintf(nm, sizeof nm, "%s v4 512thr x16 persistent grid %d", sh, g);
            rw(nm, time_it([&] { tile_v4<512, 4><<<g, 512, 70000>>>((v4*)a, (v4*)b, later, 0, 512, 1, ntiles); }));
        }
        snprintf(nm, sizeof nm, "%s v2 512thr chunked x2 order 1", sh);
        rw(nm, time_it([&] { tile_chunked<512, 4, 2><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 1, 1024); }));
        snprintf(nm, sizeof nm, "%s v2 512thr chunked x2 (16 values in flight)", sh);
        rw(nm, time_it([&] { tile_chunked<512, 4, 2><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1024); }));
        snprintf(nm, sizeof nm, "%s v2 512thr chunked x4 (8 values in flight)", sh);
        rw(nm, time_it([&] { tile_chunked<512, 4, 4><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1024); }));
        snprintf(nm, sizeof nm, "%s v2 512thr chunked x8 (4 values in flight)", sh);
        rw(nm, time_it([&] { tile_chunked<512, 4, 8><<<ntiles, 512, 70000>>>((v2*)a, (v2*)b, later, 0, 1024); }));
        snprintf(nm, sizeof nm, "%s v2 512thr chunked x8, 4 WG/CU allowed", sh);
        rw(nm, time_it([&] { tile_chunked<512, 1, 8><<<ntiles, 512, 0>>>((v2*)a, (v2*)b, later, 0, 1024); }));
        snprintf(nm, sizeof nm, "%s v2 256thr chunked x16 (4 values in flight), 8 WG/CU", sh);
        rw(nm, time_it([&] { tile_chunked<256, 1, 16><<<ntiles, 256, 0>>>((v2*)a, (v2*)b, later, 0, 1024); }));
    }
    return 0;
}
