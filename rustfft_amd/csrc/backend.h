// Thin device-runtime seam used by the host planner: HIP in the product build
// (backend_hip.hip), plain host memory in the kernel-body emulator build (tests/emu).
#pragma once
#include <cstddef>
#include <string>

namespace mi355 {
namespace backend {
int device_count();
int init(int device);              // 0 on success
int current_device();              // HIP's current device of the calling thread, -1 on failure
int set_device(int device);        // 0 on success
void* dmalloc(size_t bytes);       // nullptr on failure
void dfree(void* p);
int h2d(void* d, const void* h, size_t bytes, void* stream);
int d2h(void* h, const void* d, size_t bytes, void* stream);
int d2d(void* dst, const void* src, size_t bytes, void* stream);
int sync(void* stream);
int memset_async(void* d, int value, size_t bytes, void* stream);
// device-to-device copy between two devices (xGMI on an MI355X node), asynchronous on `stream` (a stream of the current device)
int memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream);
int sync_device();                // drains every stream of the current device
int check_launch();                // last launch error -> 0 / nonzero
std::string last_error();
void* event_create();
void event_destroy(void* e);
void event_record(void* e, void* stream);
float event_elapsed_ms(void* a, void* b);  // synchronises on b
int event_sync(void* e);                   // blocks the calling thread until the event has completed
int stream_wait_event(void* stream, void* e);  // work enqueued on `stream` after this call waits for the event's last record
void* event_create_notiming();              // ordering-only event (cheaper to record / wait on than a timing event)
void* stream_create();                     // non-blocking stream on the current device (the host-slice path's own streams)
void stream_destroy(void* s);
// read + write GB/s of the fastest plain copy this chip does (one float4 per thread, huge grid): the measured data-movement
// ceiling bench.py quotes next to the 8 TB/s spec (MI355X_MICROARCH.md: 6.29 TB/s); 0 on failure
double copy_ceiling_gbps(size_t bytes);
// One cache line of PINNED host memory that kernels of the current device can write (the fused launch's sticky error word): the host
// reads it without synchronising anything.  *device_ptr = the address kernels use.  nullptr on failure.
void* host_word_alloc(void** device_ptr);
void host_word_free(void* host_ptr);
int cu_count();                                // compute units of the current device as HIP reports them (32 in CPX mode, 256 in SPX); 0 on failure
int mem_info(size_t* free_bytes, size_t* total_bytes);  // HBM free / total of the current device; 0 on success
// PCI address of a device as sysfs spells it ("0000:c1:00.0"), "" when unknown: /sys/bus/pci/devices/<id>/numa_node names the socket whose
// memory and cores are closest to that GPU's host link
std::string pci_bus_id(int device);
}  // namespace backend
}  // namespace mi355
