// TEST INFRASTRUCTURE ONLY — host stand-in for the HIP runtime used by the kernel-body emulator.
// "Device" memory is host memory; launches run the kernel bodies thread by thread (launch.h, MI355_EMU).
// The product library (rustfft_amd/lib/libmi355fft.so) never contains this file.
//
// MI355_EMU_DEVICES=n pretends to be a node with n devices: the current device is per thread (as in HIP), every allocation
// remembers the device it was made on, and a copy whose device-side pointer belongs to another device than the calling
// thread's current one fails -- which is how the multi-device entry points (mi355fft_multi_*) are checked on the CPU: a
// worker that forgets to switch to its shard's device, or touches another shard's staging buffer, returns an error here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "backend.h"

namespace mi355 {
namespace backend {
namespace {
int fake_devices() {
    const char* e = getenv("MI355_EMU_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n > 0 ? n : 1;
}
thread_local int t_device = 0;
std::mutex g_m;
std::map<const char*, std::pair<size_t, int>> g_allocs;  // base -> (bytes, device)
// device that owns `p`, or -1 when p is not inside a tracked allocation (host memory)
int owner(const void* p) {
    std::lock_guard<std::mutex> g(g_m);
    auto it = g_allocs.upper_bound((const char*)p);
    if (it == g_allocs.begin()) return -1;
    --it;
    return ((const char*)p < it->first + it->second.first) ? it->second.second : -1;
}
bool foreign(const void* dev_ptr) {
    const int o = owner(dev_ptr);
    return o >= 0 && o != t_device;
}
}  // namespace
int device_count() { return fake_devices(); }
int init(int d) {
    if (d < 0 || d >= fake_devices()) return -1;
    t_device = d;
    return 0;
}
int current_device() { return t_device; }
int set_device(int d) {
    if (d < 0 || d >= fake_devices()) return -1;
    t_device = d;
    return 0;
}
void* dmalloc(size_t bytes) {
    // MI355_EMU_MAX_ALLOC=bytes: larger "device" allocations fail (tests of the out-of-memory reporting)
    if (const char* e = getenv("MI355_EMU_MAX_ALLOC"))
        if (bytes > (size_t)atoll(e)) return nullptr;
    void* p = nullptr;
    if (posix_memalign(&p, 64, bytes ? bytes : 64)) return nullptr;
    memset(p, 0x7f, bytes);  // poison
    std::lock_guard<std::mutex> g(g_m);
    g_allocs[(const char*)p] = {bytes ? bytes : 64, t_device};
    return p;
}
void dfree(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_m);
        g_allocs.erase((const char*)p);
    }
    free(p);
}
int h2d(void* d, const void* h, size_t b, void*) {
    if (foreign(d)) return -2;
    memcpy(d, h, b);
    return 0;
}
int d2h(void* h, const void* d, size_t b, void*) {
    if (foreign(d)) return -2;
    memcpy(h, d, b);
    return 0;
}
int d2d(void* dst, const void* src, size_t b, void*) {
    if (foreign(dst) || foreign(src)) return -2;
    memmove(dst, src, b);
    return 0;
}
int memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t b, void*) {
    const int od = owner(dst), os = owner(src);
    if ((od >= 0 && od != dst_device) || (os >= 0 && os != src_device)) return -2;
    memmove(dst, src, b);
    return 0;
}
int sync(void*) { return 0; }
int memset_async(void* d, int value, size_t bytes, void*) {
    if (foreign(d)) return -2;
    memset(d, value, bytes);
    return 0;
}
int sync_device() { return 0; }
int check_launch() { return 0; }
std::string last_error() { return "emu: a copy touched memory of another (fake) device than the calling thread's current one, or an allocation above MI355_EMU_MAX_ALLOC"; }
void* event_create() { return (void*)1; }
void event_destroy(void*) {}
void event_record(void*, void*) {}
float event_elapsed_ms(void*, void*) { return 0.f; }
int event_sync(void*) { return 0; }
int stream_wait_event(void*, void*) { return 0; }
void* event_create_notiming() { return (void*)1; }
void* stream_create() { return (void*)2; }
void stream_destroy(void*) {}
double copy_ceiling_gbps(size_t) { return 0.0; }
void* host_word_alloc(void** device_ptr) {
    void* p = calloc(64, 1);
    *device_ptr = p;
    return p;
}
void host_word_free(void* p) { free(p); }
int cu_count() { return 256; }
// fake node: device d sits at PCI address 0000:<d>1:00.0 (tests point MI355FFT_SYSFS_ROOT at a directory that describes it)
std::string pci_bus_id(int device) {
    char buf[32];
    snprintf(buf, sizeof buf, "0000:%x1:00.0", device & 15);
    return buf;
}
int mem_info(size_t* f, size_t* t) {
    *f = *t = (size_t)288 << 30;
    return 0;
}
}  // namespace backend
}  // namespace mi355
