// Probe behind the host-slice path's design (capi.cpp process_host): what does it cost to get a caller's pageable buffer to
// the GPU and back?  (a) blocking pageable copies; (b) hipHostRegister of the caller's buffer + asynchronous chunked copies in
// both directions at once; (c) staging through the library's own pinned buffers with a CPU memcpy on either side.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 1024) * (size_t)(1 << 20);
    const size_t chunk = (argc > 2 ? atol(argv[2]) : 64) * (size_t)(1 << 20);
    char* h = (char*)aligned_alloc(4096, bytes);
    memset(h, 1, bytes);
    char* d;
    CHECK(hipMalloc(&d, bytes));
    hipStream_t s[4];
    for (auto& x : s) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CHECK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        CHECK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        double t2 = now();
        printf("(a) pageable blocking: H2D %.1f GB/s, D2H %.1f GB/s, round trip payload %.1f GB/s\n", bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9, bytes / (t2 - t0) / 1e9);
    }
    {   // two host threads, blocking pageable copies in opposite directions on different halves
        char* d2;
        CHECK(hipMalloc(&d2, bytes));
        double t0 = now();
        std::thread th([&] { CHECK(hipMemcpyAsync(h, d2, bytes / 2, hipMemcpyDeviceToHost, s[1])); CHECK(hipStreamSynchronize(s[1])); });
        CHECK(hipMemcpyAsync(d, h + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        th.join();
        double t1 = now();
        printf("(a2) pageable, two threads, opposite directions at once: %.1f GB/s per direction\n", bytes / 2 / (t1 - t0) / 1e9);
        CHECK(hipFree(d2));
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CHECK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double t1 = now();
        const size_t nc = (bytes + chunk - 1) / chunk;
        for (size_t c = 0; c < nc; ++c) {  // chunk c: H2D then D2H on stream c % 3 (a kernel would sit in between)
            const size_t o = c * chunk, b = std::min(chunk, bytes - o);
            CHECK(hipMemcpyAsync(d + o, h + o, b, hipMemcpyHostToDevice, s[c % 3]));
            CHECK(hipMemcpyAsync(h + o, d + o, b, hipMemcpyDeviceToHost, s[c % 3]));
        }
        for (int i = 0; i < 3; ++i) CHECK(hipStreamSynchronize(s[i]));
        double t2 = now();
        CHECK(hipHostUnregister(h));
        double t3 = now();
        printf("(b) register %.1f ms (%.1f GB/s), pipelined both ways %.1f GB/s payload, unregister %.1f ms; whole call %.1f GB/s\n", (t1 - t0) * 1e3,
               bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, bytes / (t3 - t0) / 1e9);
    }
    {   // (c) pinned staging + CPU memcpy, double-buffered
        char* p[2];
        for (auto& x : p) CHECK(hipHostMalloc(&x, chunk, hipHostMallocDefault));
        double t0 = now();
        const size_t nc = (bytes + chunk - 1) / chunk;
        for (size_t c = 0; c < nc; ++c) {
            const size_t o = c * chunk, b = std::min(chunk, bytes - o);
            CHECK(hipStreamSynchronize(s[c % 2]));
            if (c >= 2) memcpy(h + (c - 2) * chunk, p[c % 2], chunk);
            memcpy(p[c % 2], h + o, b);
            CHECK(hipMemcpyAsync(d + o, p[c % 2], b, hipMemcpyHostToDevice, s[c % 2]));
            CHECK(hipMemcpyAsync(p[c % 2], d + o, b, hipMemcpyDeviceToHost, s[c % 2]));
        }
        for (size_t c = nc >= 2 ? nc - 2 : 0; c < nc; ++c) {
            CHECK(hipStreamSynchronize(s[c % 2]));
            memcpy(h + c * chunk, p[c % 2], std::min(chunk, bytes - c * chunk));
        }
        double t1 = now();
        printf("(c) pinned staging + one-thread memcpy, double-buffered: %.1f GB/s payload\n", bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
