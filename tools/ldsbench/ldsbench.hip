// LDS bank-model probe for gfx950: cycles per wave-instruction of ds_read_b32 / ds_read2_b32 / ds_write_b32 / ds_write2_b32 /
// ds_read_b64 / ds_write_b64 for per-lane address patterns given on the command line.  Used to pin down which lanes and which
// dwords of a *2_b32 instruction share an LDS cycle (the exchange layouts of engine.h are designed against that model).
//   ldsbench <waves_per_wg> : runs the built-in pattern table and prints cycles per instruction (per wave, all waves busy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Op { RD32 = 0, RD2 = 1, WR32 = 2, WR2 = 3, RD64 = 4, WR64 = 5 };

template <int OP, int OFF1> __device__ __forceinline__ void one(unsigned a, float& x, float& y) {
    if constexpr (OP == RD32) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(a));
    if constexpr (OP == RD2) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 r;
        asm volatile("ds_read2_b32 %0, %1 offset1:%2" : "=v"(r) : "v"(a), "i"(OFF1));
        x = r.x; y = r.y;
    }
    if constexpr (OP == WR32) asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(x));
    if constexpr (OP == WR2) asm volatile("ds_write2_b32 %0, %1, %2 offset1:%3" : : "v"(a), "v"(x), "v"(y), "i"(OFF1));
    if constexpr (OP == RD64) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 r;
        asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a));
        x = r.x; y = r.y;
    }
    if constexpr (OP == WR64) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 r = {x, y};
        asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(r));
    }
}

template <int OP, int OFF1> __global__ __launch_bounds__(1024) void probe(const unsigned* addr, unsigned long long* out, int iters) {
    extern __shared__ float lds[];
    const unsigned a = addr[threadIdx.x & 63] * 4u;  // byte address
    float x = threadIdx.x, y = 1.f;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) one<OP, OFF1>(a, x, y);
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 12345.f && y == 3.f) out[0] = 0;
}

typedef void (*kern_t)(const unsigned*, unsigned long long*, int);
template <int OP> kern_t pick(int off1) {
    switch (off1) {
#define C(N) case N: return probe<OP, N>;
        C(1) C(2) C(3) C(4) C(5) C(8) C(9) C(16) C(17) C(31) C(32) C(33) C(34) C(36) C(64) C(65) C(66) C(68) C(96) C(128) C(130) C(132) C(136) C(160)
#undef C
    }
    return nullptr;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 8;
    const int iters = 200;
    unsigned* daddr;
    unsigned long long* dout;
    CHECK(hipMalloc(&daddr, 64 * 4));
    CHECK(hipMalloc(&dout, 1024 * 8));
    struct Pat { const char* name; int op; int off1; std::vector<unsigned> a; };
    std::vector<Pat> pats;
    auto lin = [](int s) { std::vector<unsigned> v(64); for (int l = 0; l < 64; ++l) v[l] = l * s; return v; };
    auto tile = [](int P, int d) { std::vector<unsigned> v(64); for (int l = 0; l < 64; ++l) v[l] = (l % 16) * P + (l / 16) * d; return v; };
    const char* opn[] = {"ds_read_b32", "ds_read2_b32", "ds_write_b32", "ds_write2_b32", "ds_read_b64", "ds_write_b64"};
    for (int op : {RD32, WR32}) for (int s : {1, 2, 4, 8, 16, 32, 64}) pats.push_back({"lane*s", op, s, lin(s)});
    for (int op : {RD64, WR64}) for (int s : {2, 4, 8, 16, 32, 64}) pats.push_back({"lane*s", op, s, lin(s)});
    for (int op : {RD2, WR2}) {
        for (int off : {1, 2, 4, 8, 16, 32, 64, 128}) pats.push_back({"lane*1+off1", op, off, lin(1)});
        for (int off : {1, 32, 64, 65}) pats.push_back({"lane*2+off1", op, off, lin(2)});
        // the column-tile exchange: 16 columns at pitch P, the 4 row slots of a wave delta apart; off1 = distance of the instruction's two dwords
        for (int P : {1058, 1060, 1090, 1092, 1057, 1089}) for (int d : {1, 8, 9}) for (int off : {1, 4, 8, 9, 32, 33, 66, 132}) {
            static char names[512][48];
            static int ni = 0;
            snprintf(names[ni], 48, "tile P=%d d=%d", P, d);
            pats.push_back({names[ni++], op, off, tile(P, d)});
        }
    }
    // `ldsbench <waves> brief`: a short, labelled pattern list (one dispatch per line, in order) for a run under the SQ LDS
    // counters (tools/r4/lds_counter_probe.py): known conflict-free / 2-way / 4-way patterns and the column-tile exchange ones
    if (argc > 2 && !strcmp(argv[2], "brief")) {
        std::vector<Pat> b;
        for (int op : {RD32, WR32, RD64, WR64}) for (int s : {1, 2, 4, 8}) b.push_back({"lane*s", op, (op == RD64 || op == WR64) ? 2 * s : s, lin((op == RD64 || op == WR64) ? 2 * s : s)});
        for (int op : {RD2, WR2}) for (int d : {1, 8}) for (int off : {1, 8, 33}) {
            static char names[64][48];
            static int ni = 0;
            snprintf(names[ni], 48, "tile P=1058 d=%d", d);
            b.push_back({names[ni++], op, off, tile(1058, d)});
        }
        // the staged twiddle-table reads of the column tiles: the 16 lanes of a column group read ONE 8-byte entry (broadcast), the
        // wave's four row slots four consecutive entries (sub-pass 1: stride 1 entry) or entries 8 apart
        auto bc = [](int stride_dw) { std::vector<unsigned> v(64); for (int l = 0; l < 64; ++l) v[l] = (l / 16) * stride_dw; return v; };
        b.push_back({"bcast16 x4 +1e", RD64, 2, bc(2)});
        b.push_back({"bcast16 x4 +8e", RD64, 16, bc(16)});
        b.push_back({"bcast16 x4 +32e", RD64, 64, bc(64)});
        b.push_back({"bcast16 x4 b32", RD32, 1, bc(1)});
        b.push_back({"bcast64", RD64, 0, bc(0)});
        pats = b;
    }
    printf("%-16s %-22s %6s %10s\n", "op", "pattern", "off1/s", "cyc/instr");
    for (auto& p : pats) {
        kern_t k = nullptr;
        if (p.op == RD32) k = probe<RD32, 1>;
        else if (p.op == WR32) k = probe<WR32, 1>;
        else if (p.op == RD64) k = probe<RD64, 1>;
        else if (p.op == WR64) k = probe<WR64, 1>;
        else if (p.op == RD2) k = pick<RD2>(p.off1);
        else k = pick<WR2>(p.off1);
        if (!k) continue;
        CHECK(hipMemcpy(daddr, p.a.data(), 64 * 4, hipMemcpyHostToDevice));
        CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 160 * 1024, 0, daddr, dout, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256);
        CHECK(hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (auto v : h) s += (double)v;
        // cycles per wave-instruction of the CU's LDS: all waves of the workgroup issue concurrently
        printf("%-16s %-22s %6d %10.2f\n", opn[p.op], p.name, p.off1, s / 256 / (iters * 16.0) / waves);
    }
    return 0;
}
