// TEST INFRASTRUCTURE ONLY — host stand-in for the HIP runtime used by the kernel-body emulator.
// "Device" memory is host memory; launches run the kernel bodies thread by thread (launch.h, MI355_EMU).
// The product library (rustfft_amd/lib/libmi355fft.so) never contains this file.
#include <cstdlib>
#include <cstring>

#include "backend.h"

namespace mi355 {
namespace backend {
int device_count() { return 1; }
int init(int) { return 0; }
int current_device() { return 0; }
int set_device(int) { return 0; }
void* dmalloc(size_t bytes) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, bytes ? bytes : 64)) return nullptr;
    memset(p, 0x7f, bytes);  // poison
    return p;
}
void dfree(void* p) { free(p); }
int h2d(void* d, const void* h, size_t b, void*) { memcpy(d, h, b); return 0; }
int d2h(void* h, const void* d, size_t b, void*) { memcpy(h, d, b); return 0; }
int d2d(void* dst, const void* src, size_t b, void*) { memmove(dst, src, b); return 0; }
int sync(void*) { return 0; }
int sync_device() { return 0; }
int check_launch() { return 0; }
std::string last_error() { return "emu"; }
void* event_create() { return (void*)1; }
void event_destroy(void*) {}
void event_record(void*, void*) {}
float event_elapsed_ms(void*, void*) { return 0.f; }
int event_sync(void*) { return 0; }
void* stream_create() { return (void*)2; }
void stream_destroy(void*) {}
double copy_ceiling_gbps(size_t) { return 0.0; }
}  // namespace backend
}  // namespace mi355
