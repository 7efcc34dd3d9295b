// extern "C" surface declared in include/mi355fft.h.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <sched.h>
#include <vector>

#include "backend.h"
#include "plan.h"

using namespace mi355;

namespace {
thread_local std::string g_last_error;
bool g_initialised = false;
std::mutex g_init_mutex;

int set_err(int status, const std::string& msg) {
    g_last_error = msg;
    return status;
}
int hip_err(int status) { return set_err(status, backend::last_error()); }
// a failed execute(): the status in words + what plan.cpp recorded about the failing step (pass, kernel, sizes, the runtime's message)
int exec_err(int status) {
    const std::string& d = exec_detail();
    return set_err(status, std::string(mi355fft_strerror(status)) + ": " + (d.empty() ? std::string("execution failed") : d));
}

int ensure_init() {
    std::lock_guard<std::mutex> g(g_init_mutex);
    if (g_initialised) return MI355FFT_OK;
    if (backend::device_count() <= 0) return set_err(MI355FFT_ERR_NO_DEVICE, "no gfx950 (MI355X) device visible to HIP");
    g_initialised = true;
    return MI355FFT_OK;
}

// src/common.rs:13-104 — the panic texts, reproduced so the Rust shim can panic!() with them
int validation_error(size_t expected_len, size_t actual_in, size_t actual_out, bool two_buffers) {
    char buf[256];
    if (two_buffers && actual_in != actual_out) {
        snprintf(buf, sizeof buf,
                 "Provided FFT input buffer and output buffer must have the same length. Got input.len() = %zu, output.len() = %zu",
                 actual_in, actual_out);
        return set_err(MI355FFT_ERR_LENGTH_MISMATCH, buf);
    }
    if (actual_in < expected_len) {
        snprintf(buf, sizeof buf, "Provided FFT buffer was too small. Expected len = %zu, got len = %zu", expected_len, actual_in);
        return set_err(MI355FFT_ERR_BUFFER_TOO_SMALL, buf);
    }
    snprintf(buf, sizeof buf, "Input FFT buffer must be a multiple of FFT length. Expected multiple of %zu, got len = %zu",
             expected_len, actual_in);
    return set_err(MI355FFT_ERR_NOT_MULTIPLE, buf);
}

void* stage(Workspace& w, size_t bytes) {
    if (w.bytes < bytes) {
        backend::dfree(w.ptr);
        w.ptr = backend::dmalloc(bytes);
        w.bytes = w.ptr ? bytes : 0;
    }
    return w.ptr;
}

// One upload and one download at a time per process: concurrent blocking pageable copies in the SAME direction thrash (four
// threads at once: 29 - 37 GB/s of aggregate payload against 45 for one caller; tools/hostpath_bench.py), so the chunk copies of
// concurrent host-slice calls take turns per direction while their kernels and the opposite direction overlap.
// (per DEVICE: every GPU has its own host link, so the shards of a multi-device call copy concurrently)
constexpr int kMaxTurnDevices = 64;
std::mutex g_upload_turn[kMaxTurnDevices], g_download_turn[kMaxTurnDevices];
inline int turn_index(int device) { return device >= 0 ? device % kMaxTurnDevices : 0; }

// ---- NUMA placement of the library's own threads ----------------------------------------------------------------------------------
// A shard worker (and the download helper of the host-slice pipeline) drives blocking pageable copies over ONE GPU's host link; on a
// two-socket node the staging runs at the link's rate only from the socket the GPU hangs off (the other one adds an inter-socket hop to
// every copied page).  The threads this library CREATES are therefore bound to the cores of the GPU's NUMA node, read from sysfs:
// /sys/bus/pci/devices/<pci id>/numa_node -> /sys/devices/system/node/node<k>/cpulist.  Best effort: an unknown node (-1: one socket,
// a container without sysfs) leaves the thread where the scheduler puts it.  The CALLER's threads are never touched.
std::string sysfs_root() {
#if defined(MI355_TUNING) || defined(MI355_EMU)
    if (const char* e = getenv("MI355FFT_SYSFS_ROOT")) return e;  // tests: a directory that describes a fake two-socket node
#endif
    return "/sys";
}
std::string read_line(const std::string& path) {
    std::string out;
    if (FILE* f = fopen(path.c_str(), "r")) {
        char buf[4096];
        if (fgets(buf, sizeof buf, f)) out = buf;
        fclose(f);
    }
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}
// "0-63,128-191" of the NUMA node device `device` is attached to; "" when it cannot be told
std::string device_cpulist(int device) {
    const std::string id = backend::pci_bus_id(device);
    if (id.empty()) return "";
    const std::string node = read_line(sysfs_root() + "/bus/pci/devices/" + id + "/numa_node");
    if (node.empty() || node[0] == '-') return "";
    return read_line(sysfs_root() + "/devices/system/node/node" + node + "/cpulist");
}
bool pin_this_thread(const std::string& cpulist) {
    if (cpulist.empty()) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    const char* p = cpulist.c_str();
    while (*p) {
        char* end = nullptr;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0) {
                CPU_SET((int)c, &set);
                ++n;
            }
        if (*p == ',') ++p;
    }
    if (n == 0) return false;
    // never leave the mask the thread inherited from its creator (taskset, numactl, a job scheduler's or an MPI rank's binding): the node's
    // cores are intersected with it, and an empty intersection leaves the thread where it is
    cpu_set_t inherited, both;
    if (sched_getaffinity(0, sizeof inherited, &inherited) != 0) return false;
    CPU_AND(&both, &set, &inherited);
    if (CPU_COUNT(&both) == 0) return false;
    return sched_setaffinity(0, sizeof both, &both) == 0;
}

// A staging context of the plan's pool for the duration of one host-slice call.
struct HostLease {
    Plan& plan;
    Plan::HostCtx* ctx = nullptr;
    explicit HostLease(Plan& p) : plan(p) {
        std::lock_guard<std::mutex> g(plan.host_pool_mutex);
        if (plan.host_pool.empty()) {
            plan.host_busy.emplace_back(new Plan::HostCtx());
        } else {
            plan.host_busy.push_back(std::move(plan.host_pool.back()));
            plan.host_pool.pop_back();
        }
        ctx = plan.host_busy.back().get();
    }
    ~HostLease() {
        std::lock_guard<std::mutex> g(plan.host_pool_mutex);
        for (size_t i = 0; i < plan.host_busy.size(); ++i)
            if (plan.host_busy[i].get() == ctx) {
                plan.host_pool.push_back(std::move(plan.host_busy[i]));
                plan.host_busy.erase(plan.host_busy.begin() + (long)i);
                break;
            }
    }
};

// host-slice path shared by the three trait methods.  mode as in execute().
// Large calls run as a two-thread pipeline over row chunks: the calling thread uploads chunk c + 1 and enqueues its kernels
// while a helper thread downloads chunk c, so both directions of the host link carry data at once (measured on the MI355X
// box, tools/hostprobe: blocking pageable copies 56 GB/s up, 52 - 56 down, 27 - 28 GB/s of payload per round trip when one
// follows the other; both directions at once 33 - 37 GB/s per direction, whether the caller's pages are registered or not --
// the ceiling of this path; a CPU memcpy through pinned staging buffers reaches 15).
int process_host_impl(const mi355fft_plan* cplan, const void* in, size_t n_in, void* out, size_t n_out, size_t scratch_elems, int mode) {
    if (!cplan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    Plan& plan = const_cast<Plan&>(cplan->p);
    const size_t len = plan.len;
    if (len == 0) return MI355FFT_OK;  // fft_helper.rs:16-18
    (void)scratch_elems;               // required scratch is 0 for every mode: the check can never fail
    if (mode != 0 && n_in != n_out) return validation_error(len, n_in, n_out, true);
    const size_t batch = n_in / len, rem = n_in % len;
    if (batch > 0) {
        if (!in || !out) return set_err(MI355FFT_ERR_INVALID_ARG, "null buffer");
        const size_t esz = plan.prec == 32 ? 8 : 16;
        const size_t row = len * esz;
        // staging buffers, streams, copies and syncs all belong to the plan's device, whatever device is current in the
        // calling thread (a fresh thread defaults to device 0)
        DeviceGuard dev(plan.device);
        HostLease lease(plan);
        Plan::HostCtx& cx = *lease.ctx;
        if (!cx.stream_a) cx.stream_a = backend::stream_create();
        if (!cx.stream_b) cx.stream_b = backend::stream_create();
        if (!cx.stream_a || !cx.stream_b) return hip_err(MI355FFT_ERR_HIP);
        // chunks of about 64 MiB (whole rows), two staging slots; small calls are one chunk
        size_t kChunkBytes = (size_t)64 << 20;
#if defined(MI355_TUNING) || defined(MI355_EMU)
        if (const char* e = getenv("MI355FFT_HOST_CHUNK_KIB")) kChunkBytes = (size_t)atol(e) << 10;  // tests: a pipeline over small chunks
#endif
        size_t rows_per_chunk = std::max<size_t>(1, kChunkBytes / row);
        if (batch * row <= kChunkBytes + kChunkBytes / 2) rows_per_chunk = batch;
        const size_t nchunks = (batch + rows_per_chunk - 1) / rows_per_chunk;
        const size_t slot_bytes = rows_per_chunk * row, nslots = nchunks > 1 ? 2 : 1;
        char* d_in = (char*)stage(cx.in, slot_bytes * nslots);
        char* d_out = mode == 0 ? d_in : (char*)stage(cx.out, slot_bytes * nslots);
        if (!d_in || !d_out) return set_err(MI355FFT_ERR_OUT_OF_MEMORY, "device staging allocation failed");
        // Shared state of the two threads.  `Pipe`'s destructor is the only way out of this scope: it tells the helper to stop, joins it and
        // frees the chunk events, so an exception (std::string / std::vector allocation, std::thread start) between the helper's start and
        // its join unwinds into process_host()'s catch instead of destroying a joinable std::thread (std::terminate).
        struct Pipe {
            std::vector<void*> done;
            std::mutex m;
            std::condition_variable cv;
            size_t uploaded = 0, downloaded = 0;  // chunks whose kernels are enqueued / whose results are back on the host
            int rc_dl = MI355FFT_OK;
            std::string err_dl;
            bool abort_dl = false;
            bool gave_up = false;  // a fused launch of this call gave up a wait: later chunks run one launch per pass
            std::thread helper;
            explicit Pipe(size_t n) : done(n, nullptr) {}
            void stop_helper() {
                if (!helper.joinable()) return;
                {
                    std::lock_guard<std::mutex> lk(m);
                    abort_dl = true;
                }
                cv.notify_all();
                helper.join();
            }
            ~Pipe() {
                stop_helper();
                for (void* e : done)
                    if (e) backend::event_destroy(e);
            }
        } pipe(nchunks);
        int rc_main = MI355FFT_OK;
        const int device = plan.device;
        // the sticky error word of THIS call's stream slot (nullptr until a fused launch has run on it)
        auto gave_up_now = [&]() -> bool { return fused_check(plan, cx.stream_a, false) != 0; };
        auto download = [&](bool own_thread) {
            backend::set_device(device);
            if (own_thread) pin_this_thread(device_cpulist(device));  // the helper this library created, never the caller's thread
            for (size_t c = 0; c < nchunks; ++c) {
                {
                    std::unique_lock<std::mutex> lk(pipe.m);
                    pipe.cv.wait(lk, [&] { return pipe.uploaded > c || pipe.abort_dl; });
                    if (pipe.uploaded <= c) return;  // aborted before this chunk was enqueued
                }
                const size_t r0 = c * rows_per_chunk, rows = std::min(rows_per_chunk, batch - r0), slot = (c % nslots) * slot_bytes;
                int rc = backend::event_sync(pipe.done[c]);
                std::string why;
                // Chunk c's launches have completed.  If a fused launch of this call has given up a wait by now (chunk c's or a later one that
                // is already running: the word does not say which), chunk c's device results are not trusted: its rows are still intact in
                // the caller's `in` (nothing of chunk c has been copied back yet), so they are uploaded again and transformed with one launch
                // per pass on this thread's own stream.  Chunks checked clean at their own completion stay as they are.
                if (!rc && gave_up_now()) {
                    {
                        std::lock_guard<std::mutex> lk(pipe.m);
                        pipe.gave_up = true;
                    }
                    {
                        std::lock_guard<std::mutex> turn(g_upload_turn[turn_index(device)]);
                        rc = backend::h2d(d_in + slot, (const char*)in + r0 * row, rows * row, cx.stream_b) || backend::sync(cx.stream_b);
                    }
                    if (!rc) {
                        const int erc = execute(plan, d_in + slot, d_out + slot, rows, cx.stream_b, mode, nullptr, EXEC_NO_FUSE | EXEC_NO_STICKY_CHECK);
                        if (erc) {
                            rc = erc;
                            why = std::string(mi355fft_strerror(erc)) + ": " + exec_detail();
                        }
                    }
                }
                std::unique_lock<std::mutex> turn(g_download_turn[turn_index(device)]);
                if (!rc) rc = backend::d2h((char*)out + r0 * row, d_out + slot, rows * row, cx.stream_b);
                // the reference's out-of-place variant leaves `input` in an unspecified state; mirror the device buffer back so
                // host and device callers observe the same (clobbered) contents
                if (!rc && mode == 1) rc = backend::d2h((char*)const_cast<void*>(in) + r0 * row, d_in + slot, rows * row, cx.stream_b);
                if (!rc) rc = backend::sync(cx.stream_b);
                turn.unlock();
                std::lock_guard<std::mutex> lk(pipe.m);
                if (rc) {
                    pipe.rc_dl = MI355FFT_ERR_HIP;
                    pipe.err_dl = why.empty() ? backend::last_error() : why;
                }
                pipe.downloaded = c + 1;
                pipe.cv.notify_all();
                if (rc) return;
            }
        };
        if (nchunks > 1) pipe.helper = std::thread(download, true);
        for (size_t c = 0; c < nchunks && rc_main == MI355FFT_OK; ++c) {
            const size_t r0 = c * rows_per_chunk, rows = std::min(rows_per_chunk, batch - r0), slot = (c % nslots) * slot_bytes;
            bool unfused = false;
            {
                std::unique_lock<std::mutex> lk(pipe.m);
                if (c >= nslots) {  // the slot's previous chunk must be back on the host first
                    pipe.cv.wait(lk, [&] { return pipe.downloaded + nslots > c || pipe.rc_dl != MI355FFT_OK; });
                    if (pipe.rc_dl != MI355FFT_OK) break;
                }
                unfused = pipe.gave_up;
            }
            {
                std::lock_guard<std::mutex> turn(g_upload_turn[turn_index(device)]);
                if (backend::h2d(d_in + slot, (const char*)in + r0 * row, rows * row, cx.stream_a) || backend::sync(cx.stream_a)) {
                    rc_main = hip_err(MI355FFT_ERR_HIP);
                    break;
                }
            }
            int rc = execute(plan, d_in + slot, d_out + slot, rows, cx.stream_a, mode, nullptr, EXEC_NO_STICKY_CHECK | (unfused ? EXEC_NO_FUSE : 0));
            if (rc) {
                rc_main = exec_err(rc);
                break;
            }
            pipe.done[c] = backend::event_create();
            backend::event_record(pipe.done[c], cx.stream_a);
            {
                std::lock_guard<std::mutex> lk(pipe.m);
                pipe.uploaded = c + 1;
            }
            pipe.cv.notify_all();
        }
        if (nchunks > 1) {
            if (rc_main != MI355FFT_OK) {
                pipe.stop_helper();
            } else {
                pipe.helper.join();  // it ends by itself after the last chunk (or at its first error)
            }
        } else if (rc_main == MI355FFT_OK) {
            download(false);
        }
        backend::sync(cx.stream_a);
        // every launch of this call has completed and every chunk was checked at its own completion: the word has served its purpose
        if (pipe.gave_up || gave_up_now()) fused_check(plan, cx.stream_a, true);
        if (rc_main != MI355FFT_OK) return rc_main;
        if (pipe.rc_dl != MI355FFT_OK) return set_err(pipe.rc_dl, pipe.err_dl);
    }
    // a trailing partial chunk is reported after the complete chunks were transformed (array_utils.rs:164-176)
    if (rem != 0) return validation_error(len, n_in, n_out, false);
    return MI355FFT_OK;
}

// std::thread / std::vector construction can throw; nothing may unwind through the extern "C" boundary
int process_host(const mi355fft_plan* cplan, const void* in, size_t n_in, void* out, size_t n_out, size_t scratch_elems, int mode) {
    try {
        return process_host_impl(cplan, in, n_in, out, n_out, scratch_elems, mode);
    } catch (const std::bad_alloc&) {
        return set_err(MI355FFT_ERR_OUT_OF_MEMORY, "host allocation failed in the host-slice pipeline");
    } catch (const std::exception& e) {
        return set_err(MI355FFT_ERR_HIP, std::string("host-slice pipeline: ") + e.what());
    }
}

struct EventTracer : Tracer {
    std::vector<void*> ev0, ev1;
    std::vector<double> total_ms;
    std::vector<std::pair<void*, void*>> pending;
    std::vector<int> pending_pass;
    explicit EventTracer(int n) : total_ms(n, 0.0) {}
    void before(int pass, void* stream) override {
        void* a = backend::event_create();
        backend::event_record(a, stream);
        pending.push_back({a, nullptr});
        pending_pass.push_back(pass);
    }
    void after(int, void* stream) override {
        void* b = backend::event_create();
        backend::event_record(b, stream);
        pending.back().second = b;
    }
    void collect() {
        for (size_t i = 0; i < pending.size(); ++i) {
            total_ms[pending_pass[i]] += backend::event_elapsed_ms(pending[i].first, pending[i].second);
            backend::event_destroy(pending[i].first);
            backend::event_destroy(pending[i].second);
        }
        pending.clear();
        pending_pass.clear();
    }
};
}  // namespace

// a plan and the plans it runs through: the inner plan of the large Bluestein form, the balanced split a fused-resplit plan runs whenever a call
// does not fuse (plan.cpp build_plan) -- a setting must reach whichever of them executes
template <class Fn> static void for_each_subplan(Plan& p, Fn&& fn) {
    fn(p);
    if (p.inner) for_each_subplan(*p.inner, fn);
    if (p.unfused_alt) for_each_subplan(*p.unfused_alt, fn);
}
extern "C" {

int mi355fft_device_count(void) { return backend::device_count(); }

int mi355fft_init(int device) {
    if (backend::device_count() <= 0) return set_err(MI355FFT_ERR_NO_DEVICE, "no gfx950 (MI355X) device visible to HIP");
    if (backend::init(device)) return hip_err(MI355FFT_ERR_NO_DEVICE);
    std::lock_guard<std::mutex> g(g_init_mutex);
    g_initialised = true;
    return MI355FFT_OK;
}

int mi355fft_plan_create_ex(size_t len, int direction, int precision, const mi355fft_plan_options* opts, mi355fft_plan** out_plan) {
    if (!out_plan) return set_err(MI355FFT_ERR_INVALID_ARG, "out_plan is null");
    *out_plan = nullptr;
    if (precision != 32 && precision != 64) return set_err(MI355FFT_ERR_INVALID_ARG, "precision must be 32 or 64");
    if (direction != MI355FFT_FORWARD && direction != MI355FFT_INVERSE) return set_err(MI355FFT_ERR_INVALID_ARG, "bad direction");
    mi355fft_plan_options o{};
    if (opts) {
        if (opts->struct_size < sizeof(size_t) + sizeof(int) || opts->struct_size > sizeof(o))
            return set_err(MI355FFT_ERR_INVALID_ARG, "mi355fft_plan_options.struct_size does not match this library");
        memcpy(&o, opts, opts->struct_size);  // fields past the caller's struct_size stay zero (older callers)
        if (o.algorithm < MI355FFT_ALGO_AUTO || o.algorithm > MI355FFT_ALGO_MIXED_RADIX) return set_err(MI355FFT_ERR_INVALID_ARG, "unknown algorithm");
    }
    if (int rc = ensure_init()) return rc;
    mi355fft_plan* p = new mi355fft_plan();
    p->p.len = len;
    p->p.direction = direction;
    p->p.prec = precision;
    p->p.algorithm = o.algorithm;
    p->p.tw_fn = o.twiddle_fn;
    p->p.tw_ctx = o.twiddle_ctx;
    p->p.opt_rader = o.rader_inner_fft_data;
    p->p.opt_bs_tw = o.bluestein_twiddles;
    p->p.opt_bs_mul = o.bluestein_multiplier;
    p->p.opt_bs_inner = o.bluestein_inner_len;
    if (o.recipe_nodes) {
        if (!o.recipe || o.recipe_nodes > ((size_t)1 << 20)) {
            delete p;
            return set_err(MI355FFT_ERR_INVALID_ARG, "recipe_nodes without a recipe array (or an absurd node count)");
        }
        p->p.recipe.assign(o.recipe, o.recipe + o.recipe_nodes);
        const char* why = "";
        if (int rrc = apply_recipe(p->p, &why)) {
            delete p;
            return set_err(rrc, why);
        }
    }
    // the finished tables belong to one family: the one `algorithm` names or the recipe's root implies
    if (o.rader_inner_fft_data && p->p.algorithm != MI355FFT_ALGO_RADER) {
        delete p;
        return set_err(MI355FFT_ERR_INVALID_ARG, "rader_inner_fft_data needs algorithm = MI355FFT_ALGO_RADER");
    }
    if ((o.bluestein_twiddles || o.bluestein_multiplier) && p->p.algorithm != MI355FFT_ALGO_BLUESTEIN) {
        delete p;
        return set_err(MI355FFT_ERR_INVALID_ARG, "Bluestein tables need algorithm = MI355FFT_ALGO_BLUESTEIN");
    }
    int rc = build_plan(p->p);
    // the caller's tables and callback are only borrowed for the duration of this call
    p->p.tw_fn = nullptr;
    p->p.tw_ctx = nullptr;
    p->p.opt_rader = p->p.opt_bs_tw = p->p.opt_bs_mul = nullptr;
    if (rc != MI355FFT_OK) {
        delete p;
        if (rc == MI355FFT_ERR_UNSUPPORTED) {
            char buf[200];
            snprintf(buf, sizeof buf, "length %zu (precision %d, algorithm %d) has no GPU plan in this build", len, precision, o.algorithm);
            return set_err(rc, buf);
        }
        if (rc == MI355FFT_ERR_INVALID_ARG) return set_err(rc, "host tables do not fit the kernels of this length (see mi355fft_bluestein_inner_len)");
        return rc == MI355FFT_ERR_HIP ? hip_err(rc) : set_err(rc, "plan construction failed");
    }
    *out_plan = p;
    return MI355FFT_OK;
}
int mi355fft_plan_create(size_t len, int direction, int precision, mi355fft_plan** out_plan) {
    return mi355fft_plan_create_ex(len, direction, precision, nullptr, out_plan);
}
size_t mi355fft_bluestein_inner_len(size_t len, int precision) {
    if (precision != 32 && precision != 64) return 0;
    return bluestein_inner_len(len, precision);
}

int mi355fft_plan_destroy(mi355fft_plan* plan) {
    if (!plan) return MI355FFT_OK;
    unsigned word = 0;
    {
        // the last chance to report a fused launch that gave up a wait: drain the device (the destructor does the same), then look at every
        // stream slot's sticky word
        DeviceGuard dev(plan->p.device);
        backend::sync_device();
        word = fused_check(plan->p, nullptr, true, true);
    }
    delete plan;
    if (word)
        return set_err(MI355FFT_ERR_HIP, "a fused two-pass launch of the destroyed plan gave up waiting for a dependency and nobody asked: the results of that call are INVALID");
    return MI355FFT_OK;
}

size_t mi355fft_plan_len(const mi355fft_plan* plan) { return plan ? plan->p.len : 0; }
int mi355fft_plan_direction(const mi355fft_plan* plan) { return plan ? plan->p.direction : 0; }
int mi355fft_plan_precision(const mi355fft_plan* plan) { return plan ? plan->p.prec : 0; }
int mi355fft_plan_recipe_status(const mi355fft_plan* plan) { return plan ? plan->p.recipe_status : 0; }
size_t mi355fft_scratch_len(const mi355fft_plan*, int) { return 0; }

int mi355fft_plan_describe(const mi355fft_plan* plan, char* buf, size_t cap) {
    if (!plan || !buf || cap == 0) return set_err(MI355FFT_ERR_INVALID_ARG, "bad describe arguments");
    std::string s = plan->p.describe();
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return MI355FFT_OK;
}

int mi355fft_process_inplace_host(const mi355fft_plan* plan, void* buffer, size_t n_elems, void*, size_t scratch_elems) {
    return process_host(plan, buffer, n_elems, buffer, n_elems, scratch_elems, 0);
}
int mi355fft_process_outofplace_host(const mi355fft_plan* plan, void* input, size_t n_in, void* output, size_t n_out, void*,
                                     size_t scratch_elems) {
    return process_host(plan, input, n_in, output, n_out, scratch_elems, 1);
}
int mi355fft_process_immutable_host(const mi355fft_plan* plan, const void* input, size_t n_in, void* output, size_t n_out, void*,
                                    size_t scratch_elems) {
    return process_host(plan, input, n_in, output, n_out, scratch_elems, 2);
}

static int process_dev(const mi355fft_plan* plan, const void* in, void* out, size_t batch, void* stream, int mode) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    if (batch && plan->p.len && (!in || !out)) return set_err(MI355FFT_ERR_INVALID_ARG, "null device buffer");
    int rc = execute(const_cast<Plan&>(plan->p), in, out, batch, stream, mode, nullptr);
    if (rc) return exec_err(rc);
    return MI355FFT_OK;
}
int mi355fft_process_inplace_dev(const mi355fft_plan* plan, void* buffer, size_t batch, void* stream) {
    return process_dev(plan, buffer, buffer, batch, stream, 0);
}
int mi355fft_process_outofplace_dev(const mi355fft_plan* plan, void* input, void* output, size_t batch, void* stream) {
    return process_dev(plan, input, output, batch, stream, input == output ? 0 : 1);
}
int mi355fft_process_immutable_dev(const mi355fft_plan* plan, const void* input, void* output, size_t batch, void* stream) {
    return process_dev(plan, input, output, batch, stream, input == output ? 0 : 2);
}

int mi355fft_plan_num_kernels(const mi355fft_plan* plan) { return plan ? (int)plan->p.passes.size() : 0; }
const char* mi355fft_plan_kernel_name(const mi355fft_plan* plan, int index) {
    if (!plan || index < 0 || index >= (int)plan->p.passes.size()) return "";
    return plan->p.passes[index].k->name;
}
int mi355fft_profile_inplace_dev(const mi355fft_plan* plan, void* buffer, size_t batch, void* stream, int reps, float* ms, int n_kernels) {
    if (!plan || !ms || reps < 1) return set_err(MI355FFT_ERR_INVALID_ARG, "bad profile arguments");
    Plan& p = const_cast<Plan&>(plan->p);
    const int nk = (int)p.passes.size();
    if (n_kernels < nk) return set_err(MI355FFT_ERR_INVALID_ARG, "ms_per_kernel too short");
    EventTracer tr(nk);
    for (int r = 0; r < reps; ++r) {
        int rc = execute(p, buffer, buffer, batch, stream, 0, &tr);
        if (rc) return exec_err(rc);
        tr.collect();
    }
    for (int i = 0; i < nk; ++i) ms[i] = (float)(tr.total_ms[i] / reps);
    return MI355FFT_OK;
}
int mi355fft_measure_copy_ceiling(size_t bytes, double* gbps) {
    if (!gbps) return set_err(MI355FFT_ERR_INVALID_ARG, "gbps is null");
    if (int rc = ensure_init()) return rc;
    *gbps = backend::copy_ceiling_gbps(bytes);
    return *gbps > 0 ? MI355FFT_OK : set_err(MI355FFT_ERR_HIP, "copy measurement failed");
}
int mi355fft_plan_set_fused(mi355fft_plan* plan, int mode) {
    if (!plan || mode < -1 || mode > 1) return set_err(MI355FFT_ERR_INVALID_ARG, "bad fused mode");
    Plan& p = plan->p;
    p.fuse_on = p.fused && (mode == 1 || (mode == -1 && p.fuse_default));
    return MI355FFT_OK;
}
// Stream-ordered completion WITH the verdict (the single-device twin of mi355fft_multi_synchronize): waits for everything enqueued on
// `stream`, then reports a fused launch of this plan that gave up a dependency wait -- the one failure an asynchronous entry point cannot
// return (include/mi355fft.h).  The word is reported once.
int mi355fft_plan_synchronize(const mi355fft_plan* plan, void* stream) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    Plan& p = const_cast<Plan&>(plan->p);
    DeviceGuard dev(p.device);
    if (backend::sync(stream)) return hip_err(MI355FFT_ERR_HIP);
    if (fused_check(p, stream, true))
        return set_err(MI355FFT_ERR_HIP, "a fused two-pass launch on this stream gave up waiting for a dependency; the rows it touched carry NaN imaginary parts and the results of that call are INVALID");
    return MI355FFT_OK;
}
int mi355fft_plan_is_fused(const mi355fft_plan* plan) { return plan && plan->p.fuse_on && plan->p.fused ? 1 : 0; }
int mi355fft_plan_fused_status(const mi355fft_plan* plan, void* stream, unsigned* error_word) {
    if (!plan || !error_word) return set_err(MI355FFT_ERR_INVALID_ARG, "null argument");
    *error_word = 0;
    Plan& p = const_cast<Plan&>(plan->p);
    DeviceGuard dev(p.device);
    if (backend::sync(stream)) return hip_err(MI355FFT_ERR_HIP);  // every launch enqueued on the stream has completed: the word is final
    *error_word = fused_check(p, stream, true);                  // reported once: the word is cleared
    return MI355FFT_OK;
}
int mi355fft_plan_set_fused_wait_limit(mi355fft_plan* plan, int polls) {
    if (!plan || polls < -1) return set_err(MI355FFT_ERR_INVALID_ARG, "bad wait limit");
    for_each_subplan(plan->p, [&](Plan& q) { q.fuse_spin_limit = polls; });
    return MI355FFT_OK;
}
int mi355fft_plan_set_workspace_placement(mi355fft_plan* plan, int on) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    for_each_subplan(plan->p, [&](Plan& q) { q.place_workspace = on != 0; });
    return MI355FFT_OK;
}
int mi355fft_plan_set_chunk_batch(mi355fft_plan* plan, size_t chunk_batch) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    for_each_subplan(plan->p, [&](Plan& q) { q.chunk_batch = chunk_batch; });
    return MI355FFT_OK;
}

size_t mi355fft_plan_workspace_bytes(const mi355fft_plan* plan) { return plan ? const_cast<Plan&>(plan->p).workspace_bytes() : 0; }
int mi355fft_plan_trim_workspaces(mi355fft_plan* plan, size_t* freed) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    const size_t n = plan->p.trim_workspaces();
    if (freed) *freed = n;
    return MI355FFT_OK;
}

const char* mi355fft_strerror(int status) {
    switch (status) {
        case MI355FFT_OK: return "ok";
        case MI355FFT_ERR_NO_DEVICE: return "no gfx950 device";
        case MI355FFT_ERR_BUFFER_TOO_SMALL: return "Provided FFT buffer was too small";
        case MI355FFT_ERR_NOT_MULTIPLE: return "Input FFT buffer must be a multiple of FFT length";
        case MI355FFT_ERR_SCRATCH_TOO_SMALL: return "Not enough scratch space was provided";
        case MI355FFT_ERR_LENGTH_MISMATCH: return "Provided FFT input buffer and output buffer must have the same length";
        case MI355FFT_ERR_UNSUPPORTED: return "length/precision not supported by the GPU planner";
        case MI355FFT_ERR_INVALID_ARG: return "invalid argument";
        case MI355FFT_ERR_HIP: return "HIP runtime error";
        case MI355FFT_ERR_OUT_OF_MEMORY: return "out of device memory";
        default: return "unknown status";
    }
}
const char* mi355fft_last_error(void) { return g_last_error.c_str(); }
const char* mi355fft_version(void) { return "mi355fft 0.1 (gfx950)"; }
}  // extern "C"

// ---- one plan over several devices: batch rows sharded across GPUs (include/mi355fft.h, "one plan, every GPU of the node") ----
// The reference's batch loop (src/array_utils.rs:151-177) is the shard axis: rows [g ceil(batch / G), (g + 1) ceil(batch / G))
// go to shard g.  Every shard has its own replica plan (tables on its device), its own staging pool and streams, and a
// persistent worker thread bound to its device; a call hands every worker its rows and waits -- no collective, no data of one
// shard ever touches another device.  Only the optional scatter / gather edges move rows between devices (peer copies).
namespace {
struct ShardWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    bool pinned = false;
    explicit ShardWorker(int device) {
        th = std::thread([this, device] {
            backend::set_device(device);
            pinned = pin_this_thread(device_cpulist(device));  // staging copies from the socket the GPU's host link belongs to
            for (;;) {
                std::function<void()> job;
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    job = std::move(q.front());
                    q.pop_front();
                }
                job();
            }
        });
    }
    void submit(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(job));
        }
        cv.notify_one();
    }
    ~ShardWorker() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};
// completion of one multi-device call: every shard reports its status and message
struct ShardJoin {
    std::mutex m;
    std::condition_variable cv;
    int pending;
    int rc = MI355FFT_OK;
    std::string msg;
    explicit ShardJoin(int n) : pending(n) {}
    void done(int status) {
        std::lock_guard<std::mutex> lk(m);
        if (status != MI355FFT_OK && rc == MI355FFT_OK) {
            rc = status;
            msg = g_last_error;  // the worker thread's message
        }
        if (--pending == 0) cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return pending == 0; });
        if (rc != MI355FFT_OK) g_last_error = msg;
        return rc;
    }
};
void shard_rows(size_t batch, int n_shards, int shard, size_t* first, size_t* rows) {
    const size_t per = n_shards > 0 ? (batch + (size_t)n_shards - 1) / (size_t)n_shards : batch;
    const size_t lo = std::min(batch, (size_t)shard * per), hi = std::min(batch, ((size_t)shard + 1) * per);
    *first = lo;
    *rows = hi - lo;
}
}  // namespace

struct mi355fft_multi_plan {
    std::vector<int> devices;
    std::vector<mi355fft_plan*> replicas;
    std::vector<std::unique_ptr<ShardWorker>> workers;
    size_t len = 0;
    int prec = 32;
    ~mi355fft_multi_plan() {
        workers.clear();  // joins the threads (queued jobs run first)
        for (size_t g = 0; g < replicas.size(); ++g) {
            DeviceGuard dev(devices[g]);
            delete replicas[g];
        }
    }
};

namespace {
// runs fn(shard) on every shard's worker and returns the first failure
template <class Fn> int on_all_shards(const mi355fft_multi_plan* mp, Fn fn) {
    const int G = (int)mp->replicas.size();
    ShardJoin join(G);
    for (int g = 0; g < G; ++g)
        mp->workers[g]->submit([&join, &fn, g] {
            int rc;
            try {
                rc = fn(g);
            } catch (const std::bad_alloc&) {
                rc = set_err(MI355FFT_ERR_OUT_OF_MEMORY, "host allocation failed in a shard worker");
            } catch (const std::exception& e) {
                rc = set_err(MI355FFT_ERR_HIP, std::string("shard worker: ") + e.what());
            }
            join.done(rc);
        });
    return join.wait();
}
int multi_process_host(const mi355fft_multi_plan* mp, const void* in, size_t n_in, void* out, size_t n_out, int mode) {
    if (!mp) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    const size_t len = mp->len;
    if (len == 0) return MI355FFT_OK;  // fft_helper.rs:16-18
    if (mode != 0 && n_in != n_out) return validation_error(len, n_in, n_out, true);
    const size_t batch = n_in / len, rem = n_in % len, esz = mp->prec == 32 ? 8 : 16;
    if (batch > 0) {
        if (!in || !out) return set_err(MI355FFT_ERR_INVALID_ARG, "null buffer");
        const int G = (int)mp->replicas.size();
        int rc = on_all_shards(mp, [&](int g) -> int {
            size_t first, rows;
            shard_rows(batch, G, g, &first, &rows);
            if (rows == 0) return MI355FFT_OK;
            const char* sin = (const char*)in + first * len * esz;
            char* sout = (char*)out + first * len * esz;
            return process_host(mp->replicas[g], sin, rows * len, sout, rows * len, 0, mode);
        });
        if (rc) return rc;
    }
    // a trailing partial chunk is reported after the complete chunks were transformed (array_utils.rs:164-176)
    if (rem != 0) return validation_error(len, n_in, n_out, false);
    return MI355FFT_OK;
}
int multi_process_dev(const mi355fft_multi_plan* mp, const void* const* in, void* const* out, size_t batch, void* const* streams, int mode) {
    if (!mp) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    if (batch == 0 || mp->len == 0) return MI355FFT_OK;
    if (!in || !out) return set_err(MI355FFT_ERR_INVALID_ARG, "null shard buffer array");
    const int G = (int)mp->replicas.size();
    for (int g = 0; g < G; ++g) {
        size_t first, rows;
        shard_rows(batch, G, g, &first, &rows);
        if (rows && (!in[g] || !out[g])) return set_err(MI355FFT_ERR_INVALID_ARG, "null device buffer for a non-empty shard");
    }
    // the shards' launch sequences are enqueued concurrently (a fused or chunked plan issues many launches per call)
    return on_all_shards(mp, [&](int g) -> int {
        size_t first, rows;
        shard_rows(batch, G, g, &first, &rows);
        if (rows == 0) return MI355FFT_OK;
        const int m = (mode != 0 && in[g] == out[g]) ? 0 : mode;
        int rc = execute(mp->replicas[g]->p, in[g], out[g], rows, streams ? streams[g] : nullptr, m, nullptr);
        if (rc) return exec_err(rc);
        return MI355FFT_OK;
    });
}
int multi_edge(const mi355fft_multi_plan* mp, void* const* buffers, void* root, int root_device, size_t batch, void* const* streams, bool scatter) {
    if (!mp) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    if (batch == 0 || mp->len == 0) return MI355FFT_OK;
    if (!buffers || !root) return set_err(MI355FFT_ERR_INVALID_ARG, "null buffer");
    const int G = (int)mp->replicas.size();
    const size_t row = mp->len * (mp->prec == 32 ? 8 : 16);
    return on_all_shards(mp, [&](int g) -> int {
        size_t first, rows;
        shard_rows(batch, G, g, &first, &rows);
        if (rows == 0) return MI355FFT_OK;
        if (!buffers[g]) return set_err(MI355FFT_ERR_INVALID_ARG, "null device buffer for a non-empty shard");
        char* r = (char*)root + first * row;
        void* st = streams ? streams[g] : nullptr;
        const int rc = scatter ? backend::memcpy_peer(buffers[g], mp->devices[g], r, root_device, rows * row, st)
                               : backend::memcpy_peer(r, root_device, buffers[g], mp->devices[g], rows * row, st);
        return rc ? hip_err(MI355FFT_ERR_HIP) : MI355FFT_OK;
    });
}
}  // namespace

extern "C" {
int mi355fft_shard_rows(size_t batch, int n_shards, int shard, size_t* first_row, size_t* rows) {
    if (n_shards < 1 || shard < 0 || shard >= n_shards || !first_row || !rows) return set_err(MI355FFT_ERR_INVALID_ARG, "bad shard arguments");
    shard_rows(batch, n_shards, shard, first_row, rows);
    return MI355FFT_OK;
}
int mi355fft_multi_plan_create(size_t len, int direction, int precision, const mi355fft_plan_options* options, const int* devices, int n_devices,
                               mi355fft_multi_plan** out_plan) {
    if (!out_plan) return set_err(MI355FFT_ERR_INVALID_ARG, "out_plan is null");
    *out_plan = nullptr;
    if (n_devices < 0 || (n_devices > 0 && !devices)) return set_err(MI355FFT_ERR_INVALID_ARG, "bad device list");
    if (int rc = ensure_init()) return rc;
    const int visible = backend::device_count();
    std::vector<int> devs;
    if (n_devices == 0)
        for (int d = 0; d < visible; ++d) devs.push_back(d);
    else
        devs.assign(devices, devices + n_devices);
    for (int d : devs)
        if (d < 0 || d >= visible) return set_err(MI355FFT_ERR_INVALID_ARG, "device ordinal out of range");
    std::unique_ptr<mi355fft_multi_plan> mp;
    try {
        mp.reset(new mi355fft_multi_plan());
        mp->len = len;
        mp->prec = precision;
        mp->devices = devs;
        const int prev = backend::current_device();
        int rc = MI355FFT_OK;
        for (int d : devs) {  // a replica is bound to the device current at its creation
            mi355fft_plan* r = nullptr;
            if (backend::set_device(d)) {
                rc = hip_err(MI355FFT_ERR_HIP);
                break;
            }
            rc = mi355fft_plan_create_ex(len, direction, precision, options, &r);
            if (rc) break;
            mp->replicas.push_back(r);
        }
        if (prev >= 0) backend::set_device(prev);
        if (rc) {
            mp->devices.resize(mp->replicas.size());
            return rc;  // the message of the failing replica is in g_last_error
        }
        for (int d : devs) mp->workers.emplace_back(new ShardWorker(d));
    } catch (const std::exception& e) {
        return set_err(MI355FFT_ERR_OUT_OF_MEMORY, std::string("multi-plan construction: ") + e.what());
    }
    *out_plan = mp.release();
    return MI355FFT_OK;
}
int mi355fft_multi_plan_destroy(mi355fft_multi_plan* plan) {
    if (!plan) return MI355FFT_OK;
    // as mi355fft_plan_destroy: the last chance to report a fused launch of a replica that gave up a wait and that nobody asked about
    unsigned word = 0;
    for (size_t g = 0; g < plan->replicas.size(); ++g) {
        Plan& p = plan->replicas[g]->p;
        DeviceGuard dev(p.device);
        backend::sync_device();
        word |= fused_check(p, nullptr, true, true);
    }
    delete plan;
    if (word)
        return set_err(MI355FFT_ERR_HIP, "a fused two-pass launch of a replica of the destroyed multi-device plan gave up waiting for a dependency and nobody asked: the results of that call are INVALID");
    return MI355FFT_OK;
}
int mi355fft_multi_plan_shards(const mi355fft_multi_plan* plan) { return plan ? (int)plan->replicas.size() : 0; }
int mi355fft_device_cpulist(int device, char* buf, size_t cap) {
    if (!buf || cap == 0 || device < 0) return -1;
    const std::string l = device_cpulist(device);
    const size_t n = l.size() < cap - 1 ? l.size() : cap - 1;
    memcpy(buf, l.data(), n);
    buf[n] = 0;
    return (int)n;
}
int mi355fft_multi_plan_shard_pinned(const mi355fft_multi_plan* plan, int shard) {
    if (!plan || shard < 0 || shard >= (int)plan->workers.size()) return 0;
    // the worker binds itself first thing; an empty job through its queue orders this read behind that
    ShardJoin join(1);
    plan->workers[shard]->submit([&join] { join.done(MI355FFT_OK); });
    join.wait();
    return plan->workers[shard]->pinned ? 1 : 0;
}
int mi355fft_multi_plan_device(const mi355fft_multi_plan* plan, int shard) {
    return (plan && shard >= 0 && shard < (int)plan->devices.size()) ? plan->devices[shard] : -1;
}
const mi355fft_plan* mi355fft_multi_plan_replica(const mi355fft_multi_plan* plan, int shard) {
    return (plan && shard >= 0 && shard < (int)plan->replicas.size()) ? plan->replicas[shard] : nullptr;
}
int mi355fft_multi_process_inplace_host(const mi355fft_multi_plan* plan, void* buffer, size_t n_elems, void*, size_t) {
    return multi_process_host(plan, buffer, n_elems, buffer, n_elems, 0);
}
int mi355fft_multi_process_outofplace_host(const mi355fft_multi_plan* plan, void* input, size_t n_in, void* output, size_t n_out, void*, size_t) {
    return multi_process_host(plan, input, n_in, output, n_out, 1);
}
int mi355fft_multi_process_immutable_host(const mi355fft_multi_plan* plan, const void* input, size_t n_in, void* output, size_t n_out, void*, size_t) {
    return multi_process_host(plan, input, n_in, output, n_out, 2);
}
int mi355fft_multi_process_inplace_dev(const mi355fft_multi_plan* plan, void* const* buffers, size_t batch, void* const* streams) {
    return multi_process_dev(plan, (const void* const*)buffers, buffers, batch, streams, 0);
}
int mi355fft_multi_process_outofplace_dev(const mi355fft_multi_plan* plan, void* const* inputs, void* const* outputs, size_t batch, void* const* streams) {
    return multi_process_dev(plan, (const void* const*)inputs, outputs, batch, streams, 1);
}
int mi355fft_multi_process_immutable_dev(const mi355fft_multi_plan* plan, const void* const* inputs, void* const* outputs, size_t batch, void* const* streams) {
    return multi_process_dev(plan, inputs, outputs, batch, streams, 2);
}
int mi355fft_multi_synchronize(const mi355fft_multi_plan* plan, void* const* streams) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    return on_all_shards(plan, [&](int g) -> int {
        void* st = streams ? streams[g] : nullptr;
        if (backend::sync(st)) return hip_err(MI355FFT_ERR_HIP);
        // the shard's launches have completed: a fused launch that gave up a wait is reported here, once (plan.cpp fused_check)
        if (fused_check(plan->replicas[g]->p, st, true))
            return set_err(MI355FFT_ERR_HIP, "shard " + std::to_string(g) + ": a fused two-pass launch gave up waiting for a dependency; the results of that call are INVALID");
        return MI355FFT_OK;
    });
}
int mi355fft_multi_scatter_dev(const mi355fft_multi_plan* plan, const void* root_buffer, int root_device, void* const* buffers, size_t batch, void* const* streams) {
    return multi_edge(plan, buffers, const_cast<void*>(root_buffer), root_device, batch, streams, true);
}
int mi355fft_multi_gather_dev(const mi355fft_multi_plan* plan, void* const* buffers, void* root_buffer, int root_device, size_t batch, void* const* streams) {
    return multi_edge(plan, buffers, root_buffer, root_device, batch, streams, false);
}
}  // extern "C"
