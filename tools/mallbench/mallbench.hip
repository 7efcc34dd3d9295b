// Probes for the one-launch (fused two-pass) large-N design: how fast can a tile go HBM -> CU -> workspace -> CU -> HBM when
// the workspace is a small ring that should stay in the L2 / Infinity Cache, compared with a plain copy and with two
// separate kernels through a full-size workspace.  Also: copy-ceiling calibration (MI355X_MICROARCH.md quotes 6.29 TB/s
// for a float4 copy) and the blockIdx -> XCC_ID map.  No cross-workgroup waits anywhere (timing probes only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gridstride_copy(const v4* __restrict__ in, v4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        v4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        out[i] = a;
        out[i + stride] = b;
        out[i + 2 * stride] = c;
        out[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) out[i] = in[i];
}
template <int E> __global__ __launch_bounds__(256) void block_copy(const v4* __restrict__ in, v4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (E * 256);
    v4 v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = in[base + threadIdx.x + k * 256];
#pragma unroll
    for (int k = 0; k < E; ++k) out[base + threadIdx.x + k * 256] = v[k];
}
__global__ __launch_bounds__(256) void write_only(v4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = v4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void read_only(const v4* __restrict__ in, size_t n, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    v4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        v4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n; i += stride) acc += in[i];
    if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

// ---- pipeline emulation --------------------------------------------------------------------------------------------
// 512 persistent workgroups of 512 threads, tile = 128 KiB (16 float4 per thread).
// MODE 0: in -> out.   MODE 1: in -> ws ring slot; ws slot written LAG iterations ago by another workgroup of the same
// XCD -> out.   NTIN / NTOUT: non-temporal hint on the streaming side.
template <int MODE, int NTIN, int NTOUT> __global__ __launch_bounds__(512, 4) void pipe(const v4* __restrict__ in, v4* __restrict__ out, v4* ws, int tiles_per_wg, int ring, int lag, int shift) {
    constexpr int E = 16, TILE = E * 512;
    const int b = blockIdx.x, G = gridDim.x;
    const int b2 = (b + shift) % G;
    v4 v[E];
    for (int i = 0; i < tiles_per_wg; ++i) {
        const v4* src = in + ((size_t)i * G + b) * TILE;
        v4* dst = out + ((size_t)i * G + b) * TILE;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (NTIN)
                v[k] = __builtin_nontemporal_load(&src[threadIdx.x + k * 512]);
            else
                v[k] = src[threadIdx.x + k * 512];
        }
        if (MODE == 1) {
            v4* w = ws + ((size_t)(i % ring) * G + b) * TILE;
#pragma unroll
            for (int k = 0; k < E; ++k) w[threadIdx.x + k * 512] = v[k];
            __syncthreads();
            const int j = i - lag < 0 ? 0 : i - lag;
            const v4* r = ws + ((size_t)(j % ring) * G + b2) * TILE;
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = __builtin_nontemporal_load(&r[threadIdx.x + k * 512]) * 0.f + r[threadIdx.x + k * 512];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (NTOUT)
                __builtin_nontemporal_store(v[k], &dst[threadIdx.x + k * 512]);
            else
                dst[threadIdx.x + k * 512] = v[k];
        }
        __syncthreads();
    }
}
// plain variant of MODE 1 without the doubled load (the line above loads twice to defeat CSE; this one is the real probe)
template <int NTIN, int NTOUT, int SC1> __global__ __launch_bounds__(512, 4) void pipe2(const v4* __restrict__ in, v4* __restrict__ out, v4* ws, int tiles_per_wg, int ring, int lag, int shift) {
    constexpr int E = 16, TILE = E * 512;
    const int b = blockIdx.x, G = gridDim.x;
    const int b2 = (b + shift) % G;
    v4 v[E];
    for (int i = 0; i < tiles_per_wg; ++i) {
        const v4* src = in + ((size_t)i * G + b) * TILE;
        v4* dst = out + ((size_t)i * G + b) * TILE;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (NTIN)
                v[k] = __builtin_nontemporal_load(&src[threadIdx.x + k * 512]);
            else
                v[k] = src[threadIdx.x + k * 512];
        }
        v4* w = ws + ((size_t)(i % ring) * G + b) * TILE;
#pragma unroll
        for (int k = 0; k < E; ++k) w[threadIdx.x + k * 512] = v[k];
        __syncthreads();
        if (SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int j = i - lag < 0 ? 0 : i - lag;
        const v4* r = ws + ((size_t)(j % ring) * G + b2) * TILE;
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = *(volatile const v4*)&r[threadIdx.x + k * 512];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (NTOUT)
                __builtin_nontemporal_store(v[k], &dst[threadIdx.x + k * 512]);
            else
                dst[threadIdx.x + k * 512] = v[k];
        }
        __syncthreads();
    }
}


// generalised fused probe: NT threads, E float4 per thread per tile; grid-sized ring slots
template <int NT, int E> __global__ __launch_bounds__(NT) void pipe3(const v4* __restrict__ in, v4* __restrict__ out, v4* ws, int tiles_per_wg, int ring, int lag, int shift, int fused) {
    constexpr int TILE = E * NT;
    const int b = blockIdx.x, G = gridDim.x;
    const int b2 = (b + shift) % G;
    v4 v[E];
    for (int i = 0; i < tiles_per_wg; ++i) {
        const v4* src = in + ((size_t)i * G + b) * TILE;
        v4* dst = out + ((size_t)i * G + b) * TILE;
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = src[threadIdx.x + k * NT];
        if (fused) {
            v4* w = ws + ((size_t)(i % ring) * G + b) * TILE;
#pragma unroll
            for (int k = 0; k < E; ++k) w[threadIdx.x + k * NT] = v[k];
            __syncthreads();
            const int j = i - lag < 0 ? 0 : i - lag;
            const v4* r = ws + ((size_t)(j % ring) * G + b2) * TILE;
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = *(volatile const v4*)&r[threadIdx.x + k * NT];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) dst[threadIdx.x + k * NT] = v[k];
        __syncthreads();
    }
}
// footprint probes: every block sweeps `reps` 4 KiB chunks of a buffer of nchunks chunks (rotated start per block)
__global__ __launch_bounds__(256) void loop_read(const v4* __restrict__ in, unsigned nchunks, int reps, float* sink) {
    v4 acc = {0, 0, 0, 0};
    unsigned c = (blockIdx.x * 2654435761u) % nchunks;
#pragma unroll 8
    for (int r = 0; r < reps; ++r) {
        acc += in[(size_t)c * 256 + threadIdx.x];
        c += 977;
        if (c >= nchunks) c -= nchunks;
    }
    if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}
__global__ __launch_bounds__(256) void loop_write(v4* __restrict__ out, unsigned nchunks, int reps) {
    unsigned c = (blockIdx.x * 2654435761u) % nchunks;
#pragma unroll 8
    for (int r = 0; r < reps; ++r) {
        out[(size_t)c * 256 + threadIdx.x] = v4{1.f, 2.f, (float)r, 4.f};
        c += 977;
        if (c >= nchunks) c -= nchunks;
    }
}

__global__ void xcc_probe(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)x;
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
    void *a, *b, *w;
    float* sink;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&w, bytes);
    hipMalloc(&sink, 64);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    hipMemset(w, 3, bytes);
    const size_t n4 = bytes / 16;
    auto rw = [&](const char* name, float ms, double moved) { printf("%-64s %8.3f ms  %7.1f GB/s\n", name, ms, moved / ms / 1e6); fflush(stdout); };
    // ---- 1. copy ceiling
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride float4 copy, %d blocks x256 (r+w)", g);
        rw(nm, time_it([&] { gridstride_copy<<<g, 256>>>((v4*)a, (v4*)b, n4); }), 2.0 * bytes);
    }
    rw("block float4 x4 copy (r+w)", time_it([&] { block_copy<4><<<n4 / 1024, 256>>>((v4*)a, (v4*)b); }), 2.0 * bytes);
    rw("block float4 x2 copy (r+w)", time_it([&] { block_copy<2><<<n4 / 512, 256>>>((v4*)a, (v4*)b); }), 2.0 * bytes);
    rw("block float4 x1 copy (r+w)", time_it([&] { block_copy<1><<<n4 / 256, 256>>>((v4*)a, (v4*)b); }), 2.0 * bytes);
    rw("write only 4 GiB", time_it([&] { write_only<<<4096, 256>>>((v4*)b, n4); }), 1.0 * bytes);
    rw("read only 4 GiB", time_it([&] { read_only<<<4096, 256>>>((v4*)a, n4, sink); }), 1.0 * bytes);
    // ---- 2. write then read, by size (does the Infinity Cache keep written lines?); read then read
    for (size_t mib : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024}) {
        const size_t sz = mib << 20, m4 = sz / 16;
        hipEvent_t e0, e1, e2;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventCreate(&e2);
        float tw = 0, tr = 0, trr = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 1; ++r) {
            hipEventRecord(e0);
            write_only<<<2048, 256>>>((v4*)w, m4);
            hipEventRecord(e1);
            read_only<<<2048, 256>>>((v4*)w, m4, sink);
            hipEventRecord(e2);
            hipEventSynchronize(e2);
            float x, y;
            hipEventElapsedTime(&x, e0, e1);
            hipEventElapsedTime(&y, e1, e2);
            hipEventRecord(e0);
            read_only<<<2048, 256>>>((v4*)w, m4, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float z;
            hipEventElapsedTime(&z, e0, e1);
            if (r) tw += x, tr += y, trr += z;
        }
        printf("size %5zu MiB: write %7.1f GB/s, read-after-write %7.1f GB/s, read-after-read %7.1f GB/s\n", mib, sz / (tw / reps) / 1e6,
               sz / (tr / reps) / 1e6, sz / (trr / reps) / 1e6);
        fflush(stdout);
    }
    // ---- 3. pipeline emulation
    const int G = 512, TILE_B = 128 * 1024;
    const int tiles_per_wg = (int)(bytes / TILE_B / G);
    rw("pipe copy in->out (r+w of in-bytes)", time_it([&] { pipe<0, 0, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 1, 0, 0); }), 2.0 * bytes);
    rw("pipe copy in->out nt-in", time_it([&] { pipe<0, 1, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 1, 0, 0); }), 2.0 * bytes);
    rw("pipe copy in->out nt-in nt-out", time_it([&] { pipe<0, 1, 1><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 1, 0, 0); }), 2.0 * bytes);
    {
        // two kernels through a full-size workspace
        float ms = time_it([&] {
            pipe<0, 0, 0><<<G, 512>>>((v4*)a, (v4*)w, (v4*)w, tiles_per_wg, 1, 0, 0);
            pipe<0, 0, 0><<<G, 512>>>((v4*)w, (v4*)b, (v4*)w, tiles_per_wg, 1, 0, 0);
        });
        rw("two kernels in->ws(4 GiB)->out (r+w of in-bytes)", ms, 2.0 * bytes);
    }
    for (int ring : {1, 2, 3, 4, 8})
        for (int lag : {0, 1, 2}) {
            if (lag >= ring && !(ring == 1 && lag == 0)) continue;
            for (int shift : {0, 136}) {
                char nm[128];
                snprintf(nm, sizeof nm, "fused in->ws ring %d (%d MiB) lag %d shift %d ->out", ring, ring * 64, lag, shift);
                rw(nm, time_it([&] { pipe2<0, 0, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, ring, lag, shift); }), 2.0 * bytes);
            }
        }
    rw("fused ring 2 lag 1 shift 136, nt-in", time_it([&] { pipe2<1, 0, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 2, 1, 136); }), 2.0 * bytes);
    rw("fused ring 2 lag 1 shift 136, nt-in nt-out", time_it([&] { pipe2<1, 1, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 2, 1, 136); }), 2.0 * bytes);
    rw("fused ring 3 lag 1 shift 136, nt-in nt-out", time_it([&] { pipe2<1, 1, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 3, 1, 136); }), 2.0 * bytes);
    rw("fused ring 4 lag 2 shift 136, nt-in nt-out", time_it([&] { pipe2<1, 1, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 4, 2, 136); }), 2.0 * bytes);
    rw("fused ring 2 lag 1 shift 136 + acquire fence per tile", time_it([&] { pipe2<0, 0, 1><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 2, 1, 136); }), 2.0 * bytes);
    rw("fused ring 3 lag 1 shift 136 + acquire fence per tile", time_it([&] { pipe2<0, 0, 1><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 3, 1, 136); }), 2.0 * bytes);
    // cross-XCD reader (shift not a multiple of 8 -> different XCD under the observed b % 8 placement)
    rw("fused ring 2 lag 1, reader on another XCD (b+3)", time_it([&] { pipe2<0, 0, 0><<<G, 512>>>((v4*)a, (v4*)b, (v4*)w, tiles_per_wg, 2, 1, 3); }), 2.0 * bytes);

    // ---- 5. footprint probes: 16384 blocks x 64 chunks of 4 KiB = 4 GiB moved, from footprints of different sizes
    for (size_t mib : {2, 4, 16, 32, 64, 128, 192, 256, 1024, 4096}) {
        const unsigned nchunks = (unsigned)((mib << 20) / 4096);
        char nm[96];
        snprintf(nm, sizeof nm, "loop read, footprint %zu MiB", mib);
        rw(nm, time_it([&] { loop_read<<<16384, 256>>>((v4*)w, nchunks, 64, sink); }), 16384.0 * 64 * 4096);
        snprintf(nm, sizeof nm, "loop write, footprint %zu MiB", mib);
        rw(nm, time_it([&] { loop_write<<<16384, 256>>>((v4*)w, nchunks, 64); }), 16384.0 * 64 * 4096);
    }
    // ---- 6. fused probe by tile size / occupancy (GB/s = r+w of in-bytes; the fused forms move twice that through the CUs)
#define P3(NT, E, WGPCU)                                                                                                     \
    {                                                                                                                         \
        const int g = 256 * WGPCU, tb = NT * E * 16, tpw = (int)(bytes / tb / g);                                            \
        char nm[128];                                                                                                         \
        snprintf(nm, sizeof nm, "pipe3 %d thr x %d (%d KiB tile) %d WG/CU copy", NT, E, tb / 1024, WGPCU);                    \
        rw(nm, time_it([&] { pipe3<NT, E><<<g, NT>>>((v4*)a, (v4*)b, (v4*)w, tpw, 2, 1, 136, 0); }), 2.0 * tb * (double)g * tpw); \
        snprintf(nm, sizeof nm, "pipe3 %d thr x %d (%d KiB tile) %d WG/CU fused ring 2 (%d MiB)", NT, E, tb / 1024, WGPCU, 2 * g * (tb / 1024) / 1024); \
        rw(nm, time_it([&] { pipe3<NT, E><<<g, NT>>>((v4*)a, (v4*)b, (v4*)w, tpw, 2, 1, 136, 1); }), 2.0 * tb * (double)g * tpw); \
    }
    P3(512, 16, 2)
    P3(512, 16, 1)
    P3(512, 8, 2)
    P3(512, 8, 4)
    P3(512, 4, 4)
    P3(512, 2, 4)
    P3(512, 1, 4)
    P3(256, 16, 4)
    P3(256, 8, 4)
    P3(256, 8, 8)
    P3(256, 4, 8)
    P3(256, 2, 8)
    P3(256, 1, 8)
    P3(1024, 8, 2)
    P3(1024, 16, 1)
    P3(1024, 4, 2)
    // ---- 4. XCC map
    {
        int* d;
        hipMalloc(&d, 1024 * 4);
        xcc_probe<<<1024, 64>>>(d);
        std::vector<int> h(1024);
        hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
        int ok = 0;
        for (int i = 0; i < 1024; ++i) ok += (h[i] == i % 8);
        printf("xcc map: %d / 1024 blocks have XCC_ID == blockIdx %% 8; first 16:", ok);
        for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
        printf("\n");
    }
    return 0;
}
