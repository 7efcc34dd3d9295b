// extern "C" surface declared in include/mi355fft.h.
#include <cstdio>
#include <cstring>
#include <string>

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

// host-slice path shared by the three trait methods.  mode as in execute().
int process_host(const mi355fft_plan* cplan, const void* in, size_t n_in, void* out, size_t n_out, size_t scratch_elems, int mode) {
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
        const size_t bytes = batch * len * esz;
        // staging buffers, copies and the final sync all belong to the plan's device, whatever device is current in the
        // calling thread (a fresh thread defaults to device 0)
        DeviceGuard dev(plan.device);
        std::lock_guard<std::mutex> g(plan.host_mutex);
        void* d_in = stage(plan.stage_a, bytes);
        void* d_out = mode == 0 ? d_in : stage(plan.stage_b, bytes);
        if (!d_in || !d_out) return set_err(MI355FFT_ERR_OUT_OF_MEMORY, "device staging allocation failed");
        if (backend::h2d(d_in, in, bytes, nullptr)) return hip_err(MI355FFT_ERR_HIP);
        int rc = execute(plan, d_in, d_out, batch, nullptr, mode, nullptr);
        if (rc) return rc == MI355FFT_ERR_HIP ? hip_err(rc) : set_err(rc, "execution failed");
        if (backend::d2h(out, d_out, bytes, nullptr)) return hip_err(MI355FFT_ERR_HIP);
        if (mode == 1) {
            // the reference's out-of-place variant leaves `input` in an unspecified state; mirror the device
            // buffer back so host and device callers observe the same (clobbered) contents
            if (backend::d2h(const_cast<void*>(in), d_in, bytes, nullptr)) return hip_err(MI355FFT_ERR_HIP);
        }
        if (backend::sync(nullptr)) return hip_err(MI355FFT_ERR_HIP);
    }
    // a trailing partial chunk is reported after the complete chunks were transformed (array_utils.rs:164-176)
    if (rem != 0) return validation_error(len, n_in, n_out, false);
    return MI355FFT_OK;
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
        if (o.rader_inner_fft_data && o.algorithm != MI355FFT_ALGO_RADER)
            return set_err(MI355FFT_ERR_INVALID_ARG, "rader_inner_fft_data needs algorithm = MI355FFT_ALGO_RADER");
        if ((o.bluestein_twiddles || o.bluestein_multiplier) && o.algorithm != MI355FFT_ALGO_BLUESTEIN)
            return set_err(MI355FFT_ERR_INVALID_ARG, "Bluestein tables need algorithm = MI355FFT_ALGO_BLUESTEIN");
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
    delete plan;
    return MI355FFT_OK;
}

size_t mi355fft_plan_len(const mi355fft_plan* plan) { return plan ? plan->p.len : 0; }
int mi355fft_plan_direction(const mi355fft_plan* plan) { return plan ? plan->p.direction : 0; }
int mi355fft_plan_precision(const mi355fft_plan* plan) { return plan ? plan->p.prec : 0; }
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
    if (rc) return rc == MI355FFT_ERR_HIP ? hip_err(rc) : set_err(rc, "execution failed");
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
        if (rc) return rc == MI355FFT_ERR_HIP ? hip_err(rc) : set_err(rc, "execution failed");
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
int mi355fft_plan_set_chunk_batch(mi355fft_plan* plan, size_t chunk_batch) {
    if (!plan) return set_err(MI355FFT_ERR_INVALID_ARG, "null plan");
    plan->p.chunk_batch = chunk_batch;
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
