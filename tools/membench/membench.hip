// Ceiling probe for the FFT kernels' HBM access shapes (no arithmetic): tells how much of the gap to the
// copy ceiling is due to 8-byte-per-lane accesses and to the load-all / store-all phase structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <class V, int E, int NT> __global__ __launch_bounds__(NT) void blockcopy(const V* __restrict__ in, V* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (E * NT);
    V v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = in[base + threadIdx.x + k * NT];
#pragma unroll
    for (int k = 0; k < E; ++k) out[base + threadIdx.x + k * NT] = v[k];
}
template <class V, int E, int NT, int NTL, int NTS> __global__ __launch_bounds__(NT) void blockcopy_nt(const V* __restrict__ in, V* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (E * NT);
    V v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if (NTL)
            v[k] = __builtin_nontemporal_load(&in[base + threadIdx.x + k * NT]);
        else
            v[k] = in[base + threadIdx.x + k * NT];
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if (NTS)
            __builtin_nontemporal_store(v[k], &out[base + threadIdx.x + k * NT]);
        else
            out[base + threadIdx.x + k * NT] = v[k];
    }
}
// column-tile read (rows of F contiguous elements, row stride M elements), contiguous write (K2 first-pass shape)
template <class V, int F, int R, int NT> __global__ __launch_bounds__(NT) void tilecopy(const V* __restrict__ in, V* __restrict__ out, size_t M, int strided_out) {
    constexpr int E = F * R / NT;
    const size_t tiles = M / F;
    const size_t g = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const V* src = in + g * M * R + tile * F;
    V* dsto = out + g * M * R + tile * F;
    V* dstc = out + ((size_t)blockIdx.x) * (F * R);
    const int f = threadIdx.x % F, u = threadIdx.x / F;
    V v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = src[f + (size_t)(u + k * (NT / F)) * M];
    if (strided_out) {
#pragma unroll
        for (int k = 0; k < E; ++k) dsto[f + (size_t)(u + k * (NT / F)) * M] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) dstc[threadIdx.x + k * NT] = v[k];
    }
}
// same as tilecopy, but carrying `LDSB` bytes of dynamic LDS per workgroup to impose the occupancy the FFT tiles have
template <class V, int F, int R, int NT> __global__ __launch_bounds__(NT) void tilecopy_lds(const V* __restrict__ in, V* __restrict__ out, size_t M, int strided_out) {
    extern __shared__ char smem[];
    constexpr int E = F * R / NT;
    const size_t tiles = M / F;
    const size_t g = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const V* src = in + g * M * R + tile * F;
    V* dsto = out + g * M * R + tile * F;
    V* dstc = out + ((size_t)blockIdx.x) * (F * R);
    const int f = threadIdx.x % F, u = threadIdx.x / F;
    V v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = src[f + (size_t)(u + k * (NT / F)) * M];
    if (threadIdx.x == 0 && M == 1) smem[0] = 1;  // keep the allocation alive
    __syncthreads();
    if (strided_out) {
#pragma unroll
        for (int k = 0; k < E; ++k) dsto[f + (size_t)(u + k * (NT / F)) * M] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) dstc[threadIdx.x + k * NT] = v[k];
    }
}
// the 1024x16 tile with the FFT kernels' own access order: loads as sub-pass 0 issues them (radix 8, butterflies u + 32 m, rows
// b + 128 k), stores as the last sub-pass issues them (radix 16, rows b + 64 k); ORD bit 0: FFT load order, bit 1: FFT store order
template <int BARRIER, int ORD> __global__ __launch_bounds__(512, 4) void tile_fftorder(const float2* __restrict__ in, float2* __restrict__ out, size_t M, int strided_out) {
    extern __shared__ char smem[];
    const size_t tiles = M / 16;
    const size_t g = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const float2* src = in + g * M * 1024 + tile * 16;
    float2* dsto = out + g * M * 1024 + tile * 16;
    float2* dstc = out + ((size_t)blockIdx.x) * (16 * 1024);
    const int f = threadIdx.x % 16, u = threadIdx.x / 16;
    float2 v[32];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (ORD & 1) ? (u + 32 * m + 128 * k) : (u + 32 * (m * 8 + k));
            v[m * 8 + k] = src[f + (size_t)row * M];
        }
    if (threadIdx.x == 0 && M == 1) smem[0] = 1;
    if (BARRIER) __syncthreads();
    if (strided_out) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = (ORD & 2) ? (u + 32 * m + 64 * k) : (u + 32 * (m * 16 + k));
                dsto[f + (size_t)row * M] = v[m * 16 + k];
            }
    } else {
        const int uu = threadIdx.x % 32, ff = threadIdx.x / 32;  // lanes along the rows of one column, as MAP_EF stores
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = (ORD & 2) ? (uu + 32 * m + 64 * k) : (uu + 32 * (m * 16 + k));
                dstc[ff * 1024 + row] = v[m * 16 + k];
            }
    }
}

template <class K> float time_it(K&& launch, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    void *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    auto report = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms / 1e6); };
    report("blockcopy float2 x16 /256thr (a->b)", time_it([&] { blockcopy<float2, 16, 256><<<bytes / 8 / 4096, 256>>>((float2*)a, (float2*)b); }));
    report("blockcopy float2 x16 /256thr in-place", time_it([&] { blockcopy<float2, 16, 256><<<bytes / 8 / 4096, 256>>>((float2*)a, (float2*)a); }));
    report("blockcopy float4 x8 /256thr (a->b)", time_it([&] { blockcopy<float4, 8, 256><<<bytes / 16 / 2048, 256>>>((float4*)a, (float4*)b); }));
    report("blockcopy float4 x16 /256thr (a->b)", time_it([&] { blockcopy<float4, 16, 256><<<bytes / 16 / 4096, 256>>>((float4*)a, (float4*)b); }));
    report("blockcopy float4 x16 /256thr in-place", time_it([&] { blockcopy<float4, 16, 256><<<bytes / 16 / 4096, 256>>>((float4*)a, (float4*)a); }));
    report("blockcopy float4 x4 /256thr (a->b)", time_it([&] { blockcopy<float4, 4, 256><<<bytes / 16 / 1024, 256>>>((float4*)a, (float4*)b); }));
    report("blockcopy float2 x4 /256thr (a->b)", time_it([&] { blockcopy<float2, 4, 256><<<bytes / 8 / 1024, 256>>>((float2*)a, (float2*)b); }));
    typedef float v2 __attribute__((ext_vector_type(2)));
    report("blockcopy v2 x16 nt-load", time_it([&] { blockcopy_nt<v2, 16, 256, 1, 0><<<bytes / 8 / 4096, 256>>>((v2*)a, (v2*)b); }));
    report("blockcopy v2 x16 nt-store", time_it([&] { blockcopy_nt<v2, 16, 256, 0, 1><<<bytes / 8 / 4096, 256>>>((v2*)a, (v2*)b); }));
    report("blockcopy v2 x16 nt-load+store", time_it([&] { blockcopy_nt<v2, 16, 256, 1, 1><<<bytes / 8 / 4096, 256>>>((v2*)a, (v2*)b); }));
    report("blockcopy v2 x16 nt-load+store in-place", time_it([&] { blockcopy_nt<v2, 16, 256, 1, 1><<<bytes / 8 / 4096, 256>>>((v2*)a, (v2*)a); }));
    report("blockcopy v2 x16 plain (ref)", time_it([&] { blockcopy_nt<v2, 16, 256, 0, 0><<<bytes / 8 / 4096, 256>>>((v2*)a, (v2*)b); }));
    const size_t n = (size_t)1 << 20, batch = bytes / 8 / n;
    report("tile 1024x16 float2 rd-strided wr-contig", time_it([&] { tilecopy<float2, 16, 1024, 1024><<<batch * (1024 / 16), 1024>>>((float2*)a, (float2*)b, 1024, 0); }));
    report("tile 1024x16 float2 rd-strided wr-strided", time_it([&] { tilecopy<float2, 16, 1024, 1024><<<batch * (1024 / 16), 1024>>>((float2*)a, (float2*)b, 1024, 1); }));
    report("tile 1024x8 float2 rd-strided wr-contig", time_it([&] { tilecopy<float2, 8, 1024, 512><<<batch * (1024 / 8), 512>>>((float2*)a, (float2*)b, 1024, 0); }));
    report("tile 1024x8 float2 rd-strided wr-strided", time_it([&] { tilecopy<float2, 8, 1024, 512><<<batch * (1024 / 8), 512>>>((float2*)a, (float2*)b, 1024, 1); }));
    report("tile 512x16 float2 rd-strided wr-contig (M=2048)", time_it([&] { tilecopy<float2, 16, 512, 512><<<batch * (2048 / 16), 512>>>((float2*)a, (float2*)b, 2048, 0); }));
    report("tile 512x16 float2 rd-strided wr-strided (M=2048)", time_it([&] { tilecopy<float2, 16, 512, 512><<<batch * (2048 / 16), 512>>>((float2*)a, (float2*)b, 2048, 1); }));
    report("tile 256x32 float2 rd-strided wr-strided (M=4096)", time_it([&] { tilecopy<float2, 32, 256, 512><<<batch * (4096 / 32), 512>>>((float2*)a, (float2*)b, 4096, 1); }));
    report("tile 1024x8 float4 rd-strided wr-strided", time_it([&] { tilecopy<float4, 8, 1024, 512><<<(bytes / 16 / n) * (1024 / 8), 512>>>((float4*)a, (float4*)b, 1024, 1); }));
    for (int lds : {0, 70000, 140000}) {
        char nm[128];
        hipFuncSetAttribute((const void*)tilecopy_lds<float2, 16, 1024, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        snprintf(nm, sizeof nm, "tile 1024x16 512thr x32, LDS %d, wr-contig", lds);
        report(nm, time_it([&] { tilecopy_lds<float2, 16, 1024, 512><<<batch * 64, 512, lds>>>((float2*)a, (float2*)b, 1024, 0); }));
        snprintf(nm, sizeof nm, "tile 1024x16 512thr x32, LDS %d, wr-strided", lds);
        report(nm, time_it([&] { tilecopy_lds<float2, 16, 1024, 512><<<batch * 64, 512, lds>>>((float2*)a, (float2*)b, 1024, 1); }));
        hipFuncSetAttribute((const void*)tilecopy_lds<float2, 32, 256, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        snprintf(nm, sizeof nm, "tile 256x32 512thr x16, LDS %d, wr-strided (M=4096)", lds);
        report(nm, time_it([&] { tilecopy_lds<float2, 32, 256, 512><<<batch * 128, 512, lds>>>((float2*)a, (float2*)b, 4096, 1); }));
    }
#define FFTORD(B, O)                                                                                                          \
    hipFuncSetAttribute((const void*)tile_fftorder<B, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);                   \
    report("fft-order tile barrier=" #B " ord=" #O " wr-contig", time_it([&] { tile_fftorder<B, O><<<batch * 64, 512, 70000>>>((float2*)a, (float2*)b, 1024, 0); })); \
    report("fft-order tile barrier=" #B " ord=" #O " wr-strided", time_it([&] { tile_fftorder<B, O><<<batch * 64, 512, 70000>>>((float2*)a, (float2*)b, 1024, 1); }));
    FFTORD(1, 0)
    FFTORD(0, 0)
    FFTORD(0, 1)
    FFTORD(0, 2)
    FFTORD(0, 3)
    FFTORD(1, 3)
    hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice);
    report("hipMemcpy D2D", time_it([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }));
    return 0;
}
