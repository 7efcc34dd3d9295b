/* mi355fft — C ABI of the MI355X (gfx950) engine behind RustFFT's `Fft<T>::process()` hot path.
 *
 * This header is the drop-in boundary.  RustFFT has no FFI today; the seam is the `Fft<T>` trait object
 * that `FftPlanner::plan_fft` returns (reference: src/lib.rs:184-278, src/plan.rs:72-126).  Each entry
 * point below names the reference interface it replaces; INTEGRATION.md shows the Rust `extern "C"` block
 * and the `impl Fft<T> for HipFft<T>` a maintainer would add next to src/avx/avx_planner.rs.
 *
 * Conventions (identical to the reference, src/lib.rs:81-89):
 *   - data is interleaved Complex<T> = {re, im} of float (precision 32) or double (precision 64);
 *   - a call transforms `batch = n_elems / len` independent sequences stored back to back;
 *   - natural order in and out, unnormalised in both directions;
 *   - direction 0 = FftDirection::Forward (exp(-2 pi i jk/N)), 1 = FftDirection::Inverse.
 * All functions return MI355FFT_OK (0) or a negative/positive status; mi355fft_strerror() maps it to
 * text, and for the validation failures the text is the reference's panic message
 * (src/common.rs:13-104) so the Rust shim can `panic!` with it verbatim.
 * A plan is immutable after creation: any number of host threads may call process_* on one plan
 * concurrently (reference contract: `Fft: Send + Sync`, examples/concurrency.rs:9-30).
 * There is NO CPU fallback inside this library: without a gfx950 device every call fails with
 * MI355FFT_ERR_NO_DEVICE.
 */
#ifndef MI355FFT_H
#define MI355FFT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355fft_plan mi355fft_plan;

enum {
    MI355FFT_OK = 0,
    MI355FFT_ERR_NO_DEVICE = 1,        /* no gfx950 GPU / HIP runtime failure at init            */
    MI355FFT_ERR_BUFFER_TOO_SMALL = 2, /* common.rs:19-24   "Provided FFT buffer was too small"  */
    MI355FFT_ERR_NOT_MULTIPLE = 3,     /* common.rs:25-31   "must be a multiple of FFT length"   */
    MI355FFT_ERR_SCRATCH_TOO_SMALL = 4,/* common.rs:32-37   "Not enough scratch space"           */
    MI355FFT_ERR_LENGTH_MISMATCH = 5,  /* common.rs:51      input.len() != output.len()          */
    MI355FFT_ERR_UNSUPPORTED = 6,      /* length/precision this build cannot plan on the GPU     */
    MI355FFT_ERR_INVALID_ARG = 7,
    MI355FFT_ERR_HIP = 8,              /* HIP runtime error, see mi355fft_last_error()           */
    MI355FFT_ERR_OUT_OF_MEMORY = 9
};

enum { MI355FFT_FORWARD = 0, MI355FFT_INVERSE = 1 };
enum { MI355FFT_SCRATCH_INPLACE = 0, MI355FFT_SCRATCH_OUTOFPLACE = 1, MI355FFT_SCRATCH_IMMUTABLE = 2 };

/* ---- device probe -----------------------------------------------------------------------------------
 * Replaces the ISA probe of a SIMD planner (`FftPlannerAvx::new() -> Result<Self, ()>`,
 * src/avx/avx_planner.rs:113-164; chooser chain src/plan.rs:72-94): `FftPlannerHip::new()` returns
 * Err(()) when mi355fft_device_count() == 0 or mi355fft_init() fails, and FftPlanner falls through to
 * the next back-end exactly as it does for a missing ISA. */
int mi355fft_device_count(void);
int mi355fft_init(int device);

/* ---- planning ---------------------------------------------------------------------------------------
 * Replaces `FftPlanner::plan_fft(len, direction) -> Arc<dyn Fft<T>>` (src/plan.rs:101-111, 289-295) for
 * the HIP back-end: builds the device twiddle/index tables and picks the kernel sequence.
 * precision: 32 (Complex<f32>) or 64 (Complex<f64>).  The plan is bound to the device current at
 * creation: its tables live there, and every process_* call runs there whatever device is current in
 * the calling thread (the call switches and restores it).  `Drop for HipFft<T>` calls
 * mi355fft_plan_destroy, which waits for the device to drain before freeing tables and workspaces. */
int mi355fft_plan_create(size_t len, int direction, int precision, mi355fft_plan** out_plan);
int mi355fft_plan_destroy(mi355fft_plan* plan);

/* Planning with the HOST planner in charge (north_star: "the planner/twiddle/cache host code stays in Rust").
 * The reference's planner decides a Recipe per length (src/plan.rs:134-188) and its algorithm constructors
 * compute the tables (twiddles src/twiddles.rs:6-23; Rader inner_fft_data src/algorithm/raders_algorithm.rs:87-113;
 * Bluestein twiddles + inner_fft_multiplier src/algorithm/bluesteins_algorithm.rs:63-98).  Every field is optional:
 *   algorithm    top-level family of the Recipe: AUTO lets the GPU planner choose (what plan_create does);
 *                RADER = Recipe::RadersAlgorithm (prime len whose len - 1 is 13-smooth), BLUESTEIN =
 *                Recipe::BluesteinsAlgorithm, MIXED_RADIX = Radix4 / RadixN / MixedRadix / butterfly recipes
 *                (direct Cooley-Tukey kernels).  A family the length cannot run fails with MI355FFT_ERR_UNSUPPORTED.
 *   twiddle_fn   the host's `twiddles::compute_twiddle(index, fft_len, direction)`: when set, EVERY table entry the
 *                library uploads (sub-pass twiddles, inter-pass two-level tables, chirps, Rader / Bluestein
 *                precomputation inputs) is obtained from it -- called with FORWARD direction only (the inverse
 *                transform runs as conj(FFT(conj x)) on the device); re/im are doubles carrying values already rounded
 *                to the plan's precision or wider.  Called on the planning thread, before plan_create_ex returns.
 *   rader_inner_fft_data / bluestein_twiddles / bluestein_multiplier
 *                the host's finished tables for a RADER / BLUESTEIN recipe, interleaved Complex<T> in the plan's
 *                precision and DIRECTION (exactly the Box<[Complex<T>]> the reference algorithm objects hold).
 *                rader: len - 1 entries, built with the smallest primitive root (src/math_utils.rs:3-20).
 *                bluestein: len chirp entries and inner_len multiplier entries; inner_len must be an inner length this
 *                build has a kernel for (query: mi355fft_bluestein_inner_len), else MI355FFT_ERR_INVALID_ARG.
 *   recipe / recipe_nodes
 *                the host planner's whole Recipe TREE (src/plan.rs:134-188), flattened: node 0 is the root, a child's index
 *                is larger than its parent's.  The library checks it the way the reference's constructors assert
 *                (root len == len; MixedRadix / GoodThomas: left.len * right.len == len, mixed_radix.rs:53-62; Raders:
 *                inner.len == len - 1, raders_algorithm.rs:68-78; Bluesteins: inner.len >= 2 len - 1,
 *                bluesteins_algorithm.rs:55-61 -- else MI355FFT_ERR_INVALID_ARG) and takes from it
 *                  - the top-level family (as `algorithm`, which it overrides when `algorithm` is AUTO),
 *                  - for a MixedRadix / GoodThomas root that does not fit one workgroup: the six-step SPLIT -- the
 *                    leaves of the MixedRadix / GoodThomas sub-tree, right (height: the transforms that run first,
 *                    mixed_radix.rs:128-158) before left (width), become the column-tile pass heights when every one of
 *                    them has a compiled tile,
 *                  - for a Bluesteins root: the inner length, when a kernel of that inner length exists.
 *                Everything below that (which butterflies, Radix4 vs RadixN inside a tile) encodes CPU cache behaviour
 *                and stays the GPU planner's choice; a split the build cannot realise falls back to the GPU planner's own
 *                within the same family.  mi355fft_plan_recipe_status tells which happened.
 * struct_size = sizeof(mi355fft_plan_options) as the CALLER was compiled (forward compatibility: fields beyond it are
 * taken as zero, so a binding written against an earlier header keeps working).  NULL options == mi355fft_plan_create. */
enum { MI355FFT_ALGO_AUTO = 0, MI355FFT_ALGO_RADER = 1, MI355FFT_ALGO_BLUESTEIN = 2, MI355FFT_ALGO_MIXED_RADIX = 3 };
/* Recipe kinds, one per variant of `enum Recipe` (src/plan.rs:134-188); BUTTERFLY stands for Butterfly2 .. Butterfly32
 * (the size is the node's len). */
enum {
    MI355FFT_RECIPE_DFT = 0,
    MI355FFT_RECIPE_MIXED_RADIX = 1,
    MI355FFT_RECIPE_GOOD_THOMAS = 2,
    MI355FFT_RECIPE_MIXED_RADIX_SMALL = 3,
    MI355FFT_RECIPE_GOOD_THOMAS_SMALL = 4,
    MI355FFT_RECIPE_RADERS = 5,
    MI355FFT_RECIPE_BLUESTEINS = 6,
    MI355FFT_RECIPE_RADIXN = 7,
    MI355FFT_RECIPE_RADIX4 = 8,
    MI355FFT_RECIPE_BUTTERFLY = 9
};
typedef struct mi355fft_recipe_node {
    int kind;   /* MI355FFT_RECIPE_*                                                              */
    int left;   /* node index of left_fft / inner_fft / base_fft, -1 when the variant has none     */
    int right;  /* node index of right_fft, -1 when the variant has none                          */
    size_t len; /* Recipe::len() of this node (src/plan.rs:190-230)                                */
} mi355fft_recipe_node;
typedef void (*mi355fft_twiddle_fn)(void* ctx, size_t index, size_t fft_len, double* re, double* im);
typedef struct mi355fft_plan_options {
    size_t struct_size;
    int algorithm;
    mi355fft_twiddle_fn twiddle_fn;
    void* twiddle_ctx;
    const void* rader_inner_fft_data;
    const void* bluestein_twiddles;
    const void* bluestein_multiplier;
    size_t bluestein_inner_len;
    const mi355fft_recipe_node* recipe;
    size_t recipe_nodes;
} mi355fft_plan_options;
int mi355fft_plan_create_ex(size_t len, int direction, int precision, const mi355fft_plan_options* options,
                            mi355fft_plan** out_plan);
/* Inner (padded) transform length the GPU Bluestein path uses for `len` (0 when `len` is not planned through
 * Bluestein under MI355FFT_ALGO_BLUESTEIN): what a host planner must size inner_fft_multiplier for. */
size_t mi355fft_bluestein_inner_len(size_t len, int precision);

/* What the plan took from options.recipe: NONE = no recipe was given; FAMILY = the top-level family only (the GPU
 * planner chose tilings / inner length itself); SPLIT = also the six-step split (pass heights) or the Bluestein inner
 * length the recipe names. */
enum { MI355FFT_RECIPE_STATUS_NONE = 0, MI355FFT_RECIPE_STATUS_FAMILY = 1, MI355FFT_RECIPE_STATUS_SPLIT = 2 };
int mi355fft_plan_recipe_status(const mi355fft_plan* plan);

/* `Length::len`, `Direction::fft_direction` (src/lib.rs:140-143, 174-177) */
size_t mi355fft_plan_len(const mi355fft_plan* plan);
int mi355fft_plan_direction(const mi355fft_plan* plan);
int mi355fft_plan_precision(const mi355fft_plan* plan);
/* `Fft::get_inplace_scratch_len / get_outofplace_scratch_len / get_immutable_scratch_len`
 * (src/lib.rs:262-277).  Host-side scratch is 0 for every mode: the workspace lives in HBM and is owned
 * by the plan (the reference allows these numbers to change between versions, src/lib.rs:259-261). */
size_t mi355fft_scratch_len(const mi355fft_plan* plan, int mode);
/* Human-readable kernel plan, e.g. "k2first<1024..>xF16 -> k2later<1024..>xF16" (diagnostics only). */
int mi355fft_plan_describe(const mi355fft_plan* plan, char* buf, size_t cap);

/* ---- host-slice entry points: the literal drop-ins for the three trait methods ------------------------
 * Replace `Fft::process_with_scratch` (src/lib.rs:211), `process_outofplace_with_scratch` (:231, may
 * clobber `input`) and `process_immutable_with_scratch` (:250).  Pointers are host memory; n_* are
 * element (Complex<T>) counts; scratch may be NULL when scratch_elems == 0.  Validation order and
 * outcomes follow src/fft_helper.rs:9-150 + src/array_utils.rs:151-327: len == 0 is a no-op; an empty
 * buffer is accepted; otherwise scratch-too-small is reported before any work, and a trailing partial
 * chunk is reported AFTER all complete chunks have been transformed.  Each call stages H2D, runs the
 * kernels and copies back before returning. */
int mi355fft_process_inplace_host(const mi355fft_plan* plan, void* buffer, size_t n_elems, void* scratch,
                                  size_t scratch_elems);
int mi355fft_process_outofplace_host(const mi355fft_plan* plan, void* input, size_t n_in, void* output, size_t n_out,
                                     void* scratch, size_t scratch_elems);
int mi355fft_process_immutable_host(const mi355fft_plan* plan, const void* input, size_t n_in, void* output,
                                    size_t n_out, void* scratch, size_t scratch_elems);

/* ---- device-resident entry points: the measured path ---------------------------------------------------
 * Same three modes on HBM-resident buffers (16-byte aligned device pointers), asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  `batch` = number of length-len sequences.
 * The large-N passes use a plan-owned HBM workspace of batch*len elements for in-place calls
 * (allocated on first use, grown on demand).  Concurrent callers: a multi-pass call holds a
 * per-(plan, stream) lock while it enqueues its passes, so calls that share a plan AND a stream run
 * their pass sequences back to back in stream order; calls on different streams proceed in parallel
 * with their own workspaces.  HIP graphs: the first call of a (plan, stream, batch size) may allocate (and synchronise the stream
 * once while it does); after that one warm-up call a sequence of these calls is plain stream work -- kernel launches only -- and
 * captures into a HIP graph (hipStreamBeginCapture ... EndCapture) that replays with the same results. */
int mi355fft_process_inplace_dev(const mi355fft_plan* plan, void* buffer, size_t batch, void* stream);
int mi355fft_process_outofplace_dev(const mi355fft_plan* plan, void* input, void* output, size_t batch, void* stream);
int mi355fft_process_immutable_dev(const mi355fft_plan* plan, const void* input, void* output, size_t batch,
                                   void* stream);

/* ---- one plan, every GPU of the node: batch rows sharded across devices ----------------------------------------
 * A batched call is `batch` independent transforms stored back to back -- the chunk loop of src/array_utils.rs:151-177
 * (validate_and_iter) -- so the rows shard across devices with no collective in the data path: shard g of G owns the rows
 *     [g * ceil(batch / G), min(batch, (g + 1) * ceil(batch / G)))                      (mi355fft_shard_rows)
 * with its own replica of the (small) plan tables.  This is what lets the LITERAL drop-in,
 * `Arc<dyn Fft<T>>::process(&mut [Complex<T>])` (src/lib.rs:195-255), use all eight GPUs of a node through an unchanged
 * call site: one host thread + one staging pool per device, eight host links and eight HBM stacks instead of one
 * (examples/concurrency.rs:9-30 is the reference's own multi-thread-over-one-plan pattern).
 * `devices`: HIP device ordinals, n_devices >= 1; NULL / 0 = every visible gfx950 device.  An ordinal may appear more
 * than once (each entry is one shard with its own replica and streams; `{0, 0}` exercises the whole path on one GPU).
 * `options`: as mi355fft_plan_create_ex (NULL = the GPU planner decides); every replica is built from the same options.
 * Thread safety as for a plan: any number of host threads may call mi355fft_multi_process_* on one object. */
typedef struct mi355fft_multi_plan mi355fft_multi_plan;
int mi355fft_multi_plan_create(size_t len, int direction, int precision, const mi355fft_plan_options* options,
                               const int* devices, int n_devices, mi355fft_multi_plan** out_plan);
int mi355fft_multi_plan_destroy(mi355fft_multi_plan* plan);
int mi355fft_multi_plan_shards(const mi355fft_multi_plan* plan);            /* G */
int mi355fft_multi_plan_device(const mi355fft_multi_plan* plan, int shard); /* device ordinal of shard g, -1 if out of range */
/* The replica of shard g (len / direction / describe / workspace queries; owned by the multi-plan). */
const mi355fft_plan* mi355fft_multi_plan_replica(const mi355fft_multi_plan* plan, int shard);
/* The sharding law above: rows [*first_row, *first_row + *rows) of a batch of `batch` belong to shard g of G. */
int mi355fft_shard_rows(size_t batch, int n_shards, int shard, size_t* first_row, size_t* rows);

/* Host slices: the three trait methods (same validation order, messages and partial-chunk behaviour as the one-device
 * entry points above).  The rows are split by mi355fft_shard_rows; every device stages, transforms and copies back its own
 * rows concurrently (one worker thread per shard inside the library); the call returns when all rows are back. */
int mi355fft_multi_process_inplace_host(const mi355fft_multi_plan* plan, void* buffer, size_t n_elems, void* scratch,
                                        size_t scratch_elems);
int mi355fft_multi_process_outofplace_host(const mi355fft_multi_plan* plan, void* input, size_t n_in, void* output,
                                           size_t n_out, void* scratch, size_t scratch_elems);
int mi355fft_multi_process_immutable_host(const mi355fft_multi_plan* plan, const void* input, size_t n_in, void* output,
                                          size_t n_out, void* scratch, size_t scratch_elems);

/* Device-resident shards (the measured multi-GPU path): buffers[g] points to shard g's rows -- mi355fft_shard_rows(batch,
 * G, g) rows of `len` elements, back to back -- in the memory of device g (16-byte aligned; ignored when the shard is
 * empty).  streams[g] is a hipStream_t of device g (streams == NULL or streams[g] == NULL: that device's default stream).
 * Asynchronous: the call returns when every shard's kernels are enqueued; mi355fft_multi_synchronize waits for them. */
int mi355fft_multi_process_inplace_dev(const mi355fft_multi_plan* plan, void* const* buffers, size_t batch,
                                       void* const* streams);
int mi355fft_multi_process_outofplace_dev(const mi355fft_multi_plan* plan, void* const* inputs, void* const* outputs,
                                          size_t batch, void* const* streams);
int mi355fft_multi_process_immutable_dev(const mi355fft_multi_plan* plan, const void* const* inputs, void* const* outputs,
                                         size_t batch, void* const* streams);
int mi355fft_multi_synchronize(const mi355fft_multi_plan* plan, void* const* streams);
/* NUMA placement.  Every shard's worker thread (and the download helper of a host-slice call) is bound to the cores of the NUMA node its GPU
 * is attached to (/sys/bus/pci/devices/<pci id>/numa_node -> /sys/devices/system/node/node<k>/cpulist): the blocking pageable copies of
 * eight staging pipelines on a two-socket node then never cross the socket interconnect.  Threads of the CALLER are never re-bound; an
 * unknown node (one socket, no sysfs) leaves the library's threads unbound.  mi355fft_device_cpulist writes that core list ("0-63,128-191",
 * "" when unknown) and returns its length, or -1 on a bad argument; mi355fft_multi_plan_shard_pinned says whether a shard's worker is bound. */
int mi355fft_device_cpulist(int device, char* buf, size_t cap);
int mi355fft_multi_plan_shard_pinned(const mi355fft_multi_plan* plan, int shard);
/* The optional edges when the whole batch lives on ONE device: peer copies (hipMemcpyPeerAsync over xGMI) of every shard's
 * rows from / to `root_buffer` (batch * len elements on device `root_device`), enqueued on the shards' streams.  The data
 * path itself has no collective; these are bounded by the root's links and are never part of a
 * timed transform. */
int mi355fft_multi_scatter_dev(const mi355fft_multi_plan* plan, const void* root_buffer, int root_device,
                               void* const* buffers, size_t batch, void* const* streams);
int mi355fft_multi_gather_dev(const mi355fft_multi_plan* plan, void* const* buffers, void* root_buffer, int root_device,
                              size_t batch, void* const* streams);

/* ---- measurement hooks (used by bench.py; not part of the reference surface) -----------------------------
 * Number of kernel launches one in-place transform of this plan issues, and their names. */
int mi355fft_plan_num_kernels(const mi355fft_plan* plan);
const char* mi355fft_plan_kernel_name(const mi355fft_plan* plan, int index);
/* Runs the in-place transform `reps` times on `stream` with every kernel launch bracketed by HIP events
 * recorded on that stream; ms_per_kernel[i] receives the mean duration of kernel i in milliseconds. */
int mi355fft_profile_inplace_dev(const mi355fft_plan* plan, void* buffer, size_t batch, void* stream, int reps,
                                 float* ms_per_kernel, int n_kernels);
/* Read + write GB/s of the fastest plain device copy of `bytes` bytes (one float4 per thread, huge grid -- the access
 * pattern that reaches the chip's measured 6.2 - 6.3 TB/s): the data-movement ceiling bench.py quotes next to the 8 TB/s
 * spec.  Allocates and frees two scratch buffers of that size. */
int mi355fft_measure_copy_ceiling(size_t bytes, double* gbps);
/* Fused two-pass launches.  A two-pass power-of-two plan (2^16 .. 2^22 in Complex<f32>, 2^15 .. 2^21 in Complex<f64>) can run BOTH
 * column-tile passes in one launch (a three-pass plan -- 2^23, 2^24 -- its first two, over units of a transform): the second pass of transform g - lag runs beside the first pass of transform g, and the intermediate goes
 * through a ring of a few transform-sized slots that stays in the Infinity Cache, so HBM sees one read and one write per
 * transform instead of two (the reference's own structure, for comparison: Radix4 / MixedRadix sweep the whole buffer once
 * per level, src/algorithm/radix4.rs:167-203).  The planner uses it where an on-device A/B measured a gain; this setter
 * overrides that: -1 = the planner's choice (default), 0 = never, 1 = whenever a fused kernel exists for the plan.
 * Batches with fewer transforms than the ring has slots always run as two launches.  Results are those of the two-launch
 * plan up to rounding (the same kernel bodies compiled into another kernel; at two lengths the later tile in another shape). */
int mi355fft_plan_set_fused(mi355fft_plan* plan, int mode);
/* 1 when process_* calls of this plan currently use a fused launch (for a large enough batch), else 0. */
int mi355fft_plan_is_fused(const mi355fft_plan* plan);
/* Waits between the workgroups of a fused launch are bounded; a wait that gave up (the device time-sliced between processes, a debugger
 * holding a workgroup) raises the plan's STICKY error word for that stream instead of hanging the GPU, and what that launch wrote is invalid.
 * What a caller can and cannot see (the reference's contract, src/lib.rs:184: an Fft is never silently wrong):
 *   - the DATA says so: a tile whose wait gave up multiplies the imaginary part of everything it stores by NaN, and every tile that reads such
 *     a value (the second pass of that transform, a tile of another transform whose ring slot was overwritten early) produces NaN in all of
 *     its outputs -- the rows the give-up touched come back as NaN, never as plausible numbers, whatever the caller does next;
 *   - host slices (mi355fft_process_*_host, mi355fft_multi_process_*_host): the affected rows are transformed again with one launch per
 *     pass before they are copied back -- the call succeeds with correct results;
 *   - device buffers: an asynchronous entry point returns before the launch has run, so the CALL cannot fail.  The way to learn the verdict
 *     is mi355fft_plan_synchronize(plan, stream) (multi-device plans: mi355fft_multi_synchronize): it waits for the stream and returns
 *     MI355FFT_ERR_HIP when a fused launch of the plan on that stream gave up.  A caller that synchronises the stream itself
 *     (hipStreamSynchronize, an event) must call mi355fft_plan_synchronize or mi355fft_plan_fused_status afterwards -- or look for NaN.  As a
 *     fast path the NEXT mi355fft_process_*_dev / mi355fft_multi_process_*_dev on that plan and stream looks at the word when it is entered and
 *     returns MI355FFT_ERR_HIP without running -- but it reads the word without waiting for the stream, so it only sees give-ups of launches
 *     that have COMPLETED by then; mi355fft_plan_destroy / mi355fft_multi_plan_destroy drain the device and report what nobody asked about.
 *     After a reported give-up the caller re-runs the failed call (mi355fft_plan_set_fused(plan, 0) avoids a repeat).
 * mi355fft_plan_fused_status synchronises `stream` and returns the word through *error_word (0 = every dependency of every fused launch since
 * the last report was met in time; also 0 when the plan never ran fused); reporting clears it. */
int mi355fft_plan_synchronize(const mi355fft_plan* plan, void* stream);
int mi355fft_plan_fused_status(const mi355fft_plan* plan, void* stream, unsigned* error_word);
/* The bound: polls of about half a microsecond before a dependency wait gives up (default 2^21: about a second).  0 makes every wait that is
 * not already satisfied give up -- how the tests exercise the paths above on a healthy device; -1 does the same and, in addition, raises the
 * word after EVERY fused launch whether or not a wait gave up (a deterministic test hook: at some sizes a healthy device meets every
 * dependency at the first poll nine launches out of ten). */
int mi355fft_plan_set_fused_wait_limit(mi355fft_plan* plan, int polls);
/* Workspace placement (off by default).  Identical multi-pass plans run up to 3.6 % apart depending on which device allocation
 * holds their in-place workspace (profiles/r3/ab_ws_placement.jsonl).  With on = 1, the FIRST in-place call of a (plan, stream)
 * whose workspace is 256 MiB or more tries up to three allocations, times the call's first pass into each and keeps the
 * fastest.  That call blocks the host until the measurement is done, holds up to 3x the workspace meanwhile and must not be
 * made under stream capture; later calls are asynchronous as documented above. */
int mi355fft_plan_set_workspace_placement(mi355fft_plan* plan, int on);
/* Tunables (0 = library default): transforms per workspace chunk of the multi-pass path. */
int mi355fft_plan_set_chunk_batch(mi355fft_plan* plan, size_t chunk_batch);
/* Plan-owned HBM workspaces (one per stream the plan was used on, kept for reuse): bytes currently held, and a
 * release of all of them (waits for the device first; returns the bytes freed through *freed, which may be NULL).
 * The reference's counterpart is the scratch the caller owns (src/lib.rs:259-277); here it lives in HBM, so a
 * long-lived plan used on many transient streams can hand the memory back without being destroyed. */
size_t mi355fft_plan_workspace_bytes(const mi355fft_plan* plan);
int mi355fft_plan_trim_workspaces(mi355fft_plan* plan, size_t* freed);

const char* mi355fft_strerror(int status);
/* Detailed message of the calling thread's most recent failure: the reference's panic text for the validation errors; for a failed
 * transform the status in words and the step that failed -- pass index and kernel name with its grid, the size of the allocation that was
 * refused with the device's free memory, the HIP runtime's own error string. */
const char* mi355fft_last_error(void);
const char* mi355fft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355FFT_H */
