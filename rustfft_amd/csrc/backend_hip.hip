// HIP implementation of the backend seam (product build).
#include <hip/hip_runtime.h>

#include <cctype>
#include <cstring>

#include "backend.h"

namespace mi355 {
namespace backend {
static thread_local std::string g_err;
static int fail(hipError_t e) {
    if (e == hipSuccess) return 0;
    g_err = hipGetErrorString(e);
    return (int)e;
}
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int gfx950 = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::string(p.gcnArchName).rfind("gfx950", 0) == 0) ++gfx950;
    }
    return gfx950;
}
int init(int device) {
    hipDeviceProp_t p;
    if (int rc = fail(hipGetDeviceProperties(&p, device))) return rc;
    if (std::string(p.gcnArchName).rfind("gfx950", 0) != 0) {
        g_err = std::string("device is not gfx950: ") + p.gcnArchName;
        return -1;
    }
    return fail(hipSetDevice(device));
}
int current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}
int set_device(int device) { return fail(hipSetDevice(device)); }
void* dmalloc(size_t bytes) {
    void* p = nullptr;
    if (fail(hipMalloc(&p, bytes ? bytes : 16))) return nullptr;
    return p;
}
void dfree(void* p) {
    if (p) (void)hipFree(p);
}
int h2d(void* d, const void* h, size_t bytes, void* s) { return fail(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); }
int d2h(void* h, const void* d, size_t bytes, void* s) { return fail(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); }
int d2d(void* dst, const void* src, size_t bytes, void* s) { return fail(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s)); }
int sync(void* s) { return fail(hipStreamSynchronize((hipStream_t)s)); }
// (a kernel, not hipMemsetAsync: under stream capture a memset node did not reliably order in front of the next kernel node -- a
// fused launch replayed from a HIP graph next to another one found its control block unzeroed and did nothing; kernel -> kernel
// dependencies are what every captured launch of this library already relies on.  tests: test_device_calls_capture_into_a_hip_graph)
__global__ void fill_words_kernel(unsigned* p, unsigned v, size_t words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) p[i] = v;
}
int memset_async(void* d, int value, size_t bytes, void* s) {
    if (bytes % 4 != 0 || ((size_t)d & 3) != 0) return fail(hipMemsetAsync(d, value, bytes, (hipStream_t)s));
    const unsigned b = (unsigned)value & 0xffu, v = b | (b << 8) | (b << 16) | (b << 24);
    const size_t words = bytes / 4;
    if (words == 0) return 0;
    fill_words_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)s>>>((unsigned*)d, v, words);
    return fail(hipGetLastError());
}
int memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* s) {
    if (dst_device == src_device) return fail(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return fail(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)s));
}
int sync_device() { return fail(hipDeviceSynchronize()); }
int check_launch() { return fail(hipGetLastError()); }
std::string last_error() { return g_err; }
void* event_create() {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}
void event_destroy(void* e) { (void)hipEventDestroy((hipEvent_t)e); }
void event_record(void* e, void* s) { (void)hipEventRecord((hipEvent_t)e, (hipStream_t)s); }
float event_elapsed_ms(void* a, void* b) {
    float ms = 0;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return ms;
}
int event_sync(void* e) { return fail(hipEventSynchronize((hipEvent_t)e)); }
int stream_wait_event(void* s, void* e) { return fail(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); }
void* event_create_notiming() {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)e;
}
void* stream_create() {
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return (void*)s;
}
void stream_destroy(void* s) {
    if (s) (void)hipStreamDestroy((hipStream_t)s);
}
void* host_word_alloc(void** device_ptr) {
    void* h = nullptr;
    if (fail(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent))) return nullptr;
    memset(h, 0, 64);
    void* d = nullptr;
    if (fail(hipHostGetDevicePointer(&d, h, 0))) {
        (void)hipHostFree(h);
        return nullptr;
    }
    *device_ptr = d;
    return h;
}
void host_word_free(void* host_ptr) {
    if (host_ptr) (void)hipHostFree(host_ptr);
}
int cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
}
int mem_info(size_t* free_bytes, size_t* total_bytes) { return fail(hipMemGetInfo(free_bytes, total_bytes)); }
std::string pci_bus_id(int device) {
    char buf[64] = {0};
    if (hipDeviceGetPCIBusId(buf, (int)sizeof buf, device) != hipSuccess) return "";
    std::string id(buf);
    for (auto& c : id) c = (char)tolower((unsigned char)c);
    return id;
}
typedef float copy_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_f4_kernel(const copy_v4* __restrict__ in, copy_v4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    out[i] = in[i];
}
double copy_ceiling_gbps(size_t bytes) {
    bytes = (bytes / 4096) * 4096;
    if (bytes == 0 || bytes / 4096 > 0x7fffffffULL) return 0.0;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess) return 0.0;
    if (hipMalloc(&b, bytes) != hipSuccess) {
        (void)hipFree(a);
        return 0.0;
    }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)(bytes / 4096);
    const int reps = 8;
    copy_f4_kernel<<<grid, 256>>>((const copy_v4*)a, (copy_v4*)b);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) copy_f4_kernel<<<grid, 256>>>((const copy_v4*)a, (copy_v4*)b);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return ms > 0 ? 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
}
}  // namespace backend
}  // namespace mi355
