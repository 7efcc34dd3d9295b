// Plan object behind `mi355fft_plan` (include/mi355fft.h).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi355fft.h"
#include "dyn_engine.h"
#include "lsm.h"
#include "backend.h"
#include "registry.h"

namespace mi355 {

// RAII: make the plan's device current for the calling thread (HIP's current device is per thread and defaults to 0)
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want) {
        if (want < 0) return;
        prev = backend::current_device();
        if (prev != want && backend::set_device(want) == 0) switched = true;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) backend::set_device(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

enum PlanKind { PLAN_TRIVIAL = 0, PLAN_SINGLE = 1, PLAN_MACRO = 2, PLAN_BLUESTEIN = 3, PLAN_RADER = 4, PLAN_BLUESTEIN_LARGE = 5, PLAN_BLUESTEIN_FUSED = 6, PLAN_BLUESTEIN_2K = 7, PLAN_RADER_FUSED = 8 };

struct PassDesc {
    const KernelEntry* k;
    void* d_tw;   // sub-pass twiddles of the workgroup transform
    void* d_tw2;  // one-kernel Bluestein: the same for the reversed schedule of the second transform
    void* d_tlo;  // two-level inter-pass twiddle tables (macro passes after the first)
    void* d_thi;
    int hshift, lmask;
    long long m, s;  // M = N / R and S = product of the earlier macro radices
    void* d_aux1;  // Bluestein: chirp[n]        Rader: d[p-1]
    void* d_aux2;  // Bluestein: bf[M]
    void* d_perm_in;   // Rader: g^(j+1) mod p
    void* d_perm_out;  // Rader: g^-(j+1) mod p
    DynSched dyn;      // run-time schedule (KIND_DYN_K1 / KIND_DYN_RADER)
    // LDS stage machine (KIND_LSM, lsm.h): the program's device tables and launch geometry
    struct Lsm {
        void *d_stages = nullptr, *d_desc = nullptr, *d_ltab = nullptr, *d_gtab = nullptr, *d_ldperm = nullptr, *d_stperm = nullptr;
        int nstages = 0, ltab_n = 0, n = 0, f = 1, tab_off = 0, nt = 256, lds_bytes = 0;
        std::string desc;
    } lsm;
    long long row_n;   // transform length this pass belongs to when it is not the plan's own (fused Bluestein: M); 0 = plan.len
};

struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    bool placed = false;  // the placement tournament (plan.cpp place_workspace) has run for this allocation
};

// Per-(plan, stream) execution slot.  `launch_mutex` is held for the WHOLE enqueue sequence of a multi-pass call
// (workspace lookup, growth and every pass launch), so two host threads sharing one plan on one stream cannot
// interleave their passes; stream order then makes the shared workspace safe (A's passes all precede B's).  Callers
// on different streams get different slots and run concurrently.  Growth synchronises on the stream of the call that
// asks for it (a live handle by construction); the plan's destructor drains the whole device instead of the cached
// stream handles, which the caller may have destroyed by then.
// Chunk pipeline of the multi-pass plans (plan.cpp execute_pipelined): a small ring of intermediate buffers that stays in the
// Infinity Cache, side streams for every pass but the last, ordering-only events [pass * slots + slot].
struct PipeState {
    Workspace ring;
    std::vector<void*> side;
    std::vector<void*> ev;
    void* ev_fork = nullptr;
    void* ctrl = nullptr;  // fused two-pass kernel: control block (ticket, per-slot counters), zeroed in front of every launch
    size_t ctrl_bytes = 0;
    // fused two-pass kernel: the slot's STICKY error word -- pinned host memory a tile writes when one of its bounded waits gives up
    // (launch.h k2f_wait).  No launch clears it; the host reads it without synchronising (fused_check) and clears it when it reports it.
    volatile unsigned* err_host = nullptr;
    void* err_dev = nullptr;  // the same word as kernels address it
};
struct StreamSlot {
    std::mutex launch_mutex;
    Workspace ws;
    PipeState pipe;
};

// per-kernel event hooks for mi355fft_profile_inplace_dev
struct Tracer {
    virtual ~Tracer() {}
    virtual void before(int pass, void* stream) = 0;
    virtual void after(int pass, void* stream) = 0;
};

struct Plan {
    size_t len = 0;
    int direction = 0, prec = 32;
    int kind = PLAN_TRIVIAL;
    std::vector<PassDesc> passes;
    std::vector<void*> device_allocs;
    size_t chunk_batch = 0;
    int dbg = 0;
    bool place_workspace = false;  // opt-in workspace placement tournament (plan.cpp execute_t)
    // chunk pipeline (execute_pipelined): 0 = off (one full-size workspace), 1 = chunks through a cache-resident ring on the caller's
    // stream, 2 = the same with every pass but the last on side streams (passes of neighbouring chunks overlap)
    int pipe_mode = 0;
    size_t pipe_slot_bytes = 0;  // bytes of one ring slot (0 = default)
    int pipe_slots = 0;          // ring slots per intermediate buffer (0 = default)
    // fused two-pass kernel (launch.h k2f_kernel): the registry entry that fuses this plan's two passes (nullptr: none), whether
    // execute() uses it, its protocol mode (K2FusedParams::mode) and, for experiments, lag / ring slots (0 = derived)
    const KernelEntry* fused = nullptr;
    bool fuse_on = false;
    bool fuse_default = false;  // the planner's measured choice for this plan (mi355fft_plan_set_fused(-1) restores it)
    int fuse_mode = 3, fuse_lag = 0, fuse_slots = 0;  // mode 3: dependency counters + work items by ticket
    int fuse_resident = 0;           // workgroups of the fused kernel the plan's device holds at once: compute units x the runtime's occupancy
    int fuse_spin_limit = 1 << 21;   // polls (x s_sleep(8) ~ 0.5 us) before a dependency wait gives up: about a second (mi355fft_plan_set_fused_wait_limit)
    // mi355fft_plan_options (host planner in charge): algorithm family, twiddle source, finished tables
    int algorithm = 0;
    mi355fft_twiddle_fn tw_fn = nullptr;
    void* tw_ctx = nullptr;
    const void* opt_rader = nullptr;
    const void* opt_bs_tw = nullptr;
    const void* opt_bs_mul = nullptr;
    size_t opt_bs_inner = 0;
    // the host planner's Recipe tree (mi355fft_plan_options.recipe), copied; what apply_recipe() derived from it
    std::vector<mi355fft_recipe_node> recipe;
    std::vector<size_t> recipe_split;  // six-step pass heights a MixedRadix / GoodThomas root asks for (first pass first)
    size_t recipe_bs_inner = 0;        // inner length a Bluesteins root names
    int recipe_status = MI355FFT_RECIPE_STATUS_NONE;
    int device = -1;  // HIP device the tables live on (the device current at creation)
    std::mutex ws_mutex;
    std::map<void*, std::unique_ptr<StreamSlot>> slots;  // one execution slot (HBM workspace) per stream
    // Host-slice path (mi355fft_process_*_host): a pool of staging contexts -- device buffers + two private streams each.  A
    // calling thread takes one for the duration of its call (and makes a new one when all are busy), so host threads that share a
    // plan (examples/concurrency.rs:9-30) stage, transform and copy back concurrently instead of queueing on one mutex.
    struct HostCtx {
        Workspace in, out;
        void* stream_a = nullptr;  // uploads + kernels
        void* stream_b = nullptr;  // downloads (a second thread drives them, so both directions of the link are busy)
    };
    std::mutex host_pool_mutex;
    std::vector<std::unique_ptr<HostCtx>> host_pool;   // idle contexts
    std::vector<std::unique_ptr<HostCtx>> host_busy;   // contexts lent to a call (kept here so the destructor sees them all)
    std::unique_ptr<Plan> inner;  // PLAN_BLUESTEIN_LARGE: forward power-of-two plan of the padded length M
    // two-pass plans whose split was taken only for its fused kernel: the balanced split, run whenever the call does not fuse (plan.cpp build_plan)
    std::unique_ptr<Plan> unfused_alt;
    bool fused_resplit = false, no_fused_resplit = false;

    ~Plan();
    std::string describe() const;
    StreamSlot& slot_for(void* stream);
    // called with slot.launch_mutex held
    void* workspace_in(StreamSlot& slot, size_t bytes, void* stream);
    // Bytes of plan-owned HBM workspace currently cached / released by trim (mi355fft_plan_trim_workspaces)
    size_t workspace_bytes();
    size_t trim_workspaces();
};

int build_plan(Plan& plan);
// Validates plan.recipe against plan.len and derives family / split / inner length (0 or MI355FFT_ERR_INVALID_ARG; *why = reason)
int apply_recipe(Plan& plan, const char** why);
size_t bluestein_inner_len(size_t len, int prec);
// flags of execute()
enum { EXEC_NO_FUSE = 1,          // run a fused plan as one launch per pass for this call
       EXEC_NO_STICKY_CHECK = 2   // the caller looks after the slot's sticky error word itself (host-slice path: it re-runs rows instead of failing)
};
int execute(Plan& plan, const void* in, void* out, size_t batch, void* stream, int mode, Tracer* tr, int flags = 0);
// Sticky error word of the plan's slot for `stream` (launch.h k2f_wait): nonzero = a fused launch enqueued on that stream since the last report
// gave up a wait, its results are INVALID.  Reads pinned host memory, never blocks; `clear` resets the word (a launch still running may set
// it again -- the next check reports that).  all_streams: any slot of the plan.
unsigned fused_check(Plan& plan, void* stream, bool clear, bool all_streams = false);
// why the last execute() / build_plan() of the calling thread failed: pass, kernel, sizes, the runtime's own message ("" when it did not fail)
const std::string& exec_detail();

}  // namespace mi355

struct mi355fft_plan {
    mi355::Plan p;
};
