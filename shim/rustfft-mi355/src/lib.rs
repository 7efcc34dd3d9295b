//! rustfft-mi355 — MI355X (gfx950) back-end for RustFFT's `Fft<T>` trait.
//!
//! `FftPlannerHip<T>` hands out `Arc<dyn rustfft::Fft<T>>` objects whose `process*()` methods run on the GPU through
//! `libmi355fft.so` (C ABI: include/mi355fft.h of the mi355fft repository).  The shape follows RustFFT's own ISA-gated
//! planners: `new()` fails when the "instruction set" (here: a gfx950 device) is missing, and a length the GPU build
//! cannot plan falls back, per length, to RustFFT's CPU planner.
//!
//! Semantics are the trait's: `buffer.len() / len()` transforms back to back, natural order, unnormalised in both
//! directions; validation failures panic with RustFFT's own messages (the library reproduces them).
//!
//! SOURCE ONLY: never compiled (no rustc on the build or GPU machines); tests/test_integration_doc.py checks every
//! declaration below against the header.
#![allow(clippy::missing_safety_doc)]

use rustfft::num_complex::Complex;
use rustfft::{Direction, Fft, FftDirection, FftNum, Length};
use std::any::TypeId;
use std::collections::HashMap;
use std::ffi::{c_int, c_void, CStr};
use std::marker::PhantomData;
use std::sync::Arc;

/// Raw bindings: one declaration per function of include/mi355fft.h.
pub mod ffi {
    use std::ffi::{c_int, c_uint, c_void};

    #[repr(C)]
    pub struct Mi355Plan {
        _private: [u8; 0],
    }
    /// `mi355fft_multi_plan`: one plan over several devices (batch rows sharded across them).
    #[repr(C)]
    pub struct Mi355MultiPlan {
        _private: [u8; 0],
    }

    /// `mi355fft_recipe_node`: one node of the host planner's flattened Recipe tree.
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct Mi355RecipeNode {
        pub kind: c_int,
        pub left: c_int,
        pub right: c_int,
        pub len: usize,
    }

    /// `mi355fft_plan_options`.
    #[repr(C)]
    pub struct Mi355PlanOptions {
        pub struct_size: usize,
        pub algorithm: c_int,
        pub twiddle_fn: Option<extern "C" fn(*mut c_void, usize, usize, *mut f64, *mut f64)>,
        pub twiddle_ctx: *mut c_void,
        pub rader_inner_fft_data: *const c_void,
        pub bluestein_twiddles: *const c_void,
        pub bluestein_multiplier: *const c_void,
        pub bluestein_inner_len: usize,
        pub recipe: *const Mi355RecipeNode,
        pub recipe_nodes: usize,
    }

    pub const ALGO_AUTO: c_int = 0;
    pub const ALGO_RADER: c_int = 1;
    pub const ALGO_BLUESTEIN: c_int = 2;
    pub const ALGO_MIXED_RADIX: c_int = 3;

    pub const RECIPE_DFT: c_int = 0;
    pub const RECIPE_MIXED_RADIX: c_int = 1;
    pub const RECIPE_GOOD_THOMAS: c_int = 2;
    pub const RECIPE_MIXED_RADIX_SMALL: c_int = 3;
    pub const RECIPE_GOOD_THOMAS_SMALL: c_int = 4;
    pub const RECIPE_RADERS: c_int = 5;
    pub const RECIPE_BLUESTEINS: c_int = 6;
    pub const RECIPE_RADIXN: c_int = 7;
    pub const RECIPE_RADIX4: c_int = 8;
    pub const RECIPE_BUTTERFLY: c_int = 9;

    pub const ERR_UNSUPPORTED: c_int = 6;

    #[cfg(feature = "link")]
    extern "C" {
        pub fn mi355fft_device_count() -> c_int;
        pub fn mi355fft_init(device: c_int) -> c_int;
        pub fn mi355fft_plan_create(len: usize, direction: c_int, precision: c_int, out_plan: *mut *mut Mi355Plan) -> c_int;
        pub fn mi355fft_plan_destroy(plan: *mut Mi355Plan) -> c_int;
        pub fn mi355fft_plan_create_ex(len: usize, direction: c_int, precision: c_int, options: *const Mi355PlanOptions, out_plan: *mut *mut Mi355Plan) -> c_int;
        pub fn mi355fft_bluestein_inner_len(len: usize, precision: c_int) -> usize;
        pub fn mi355fft_plan_recipe_status(plan: *const Mi355Plan) -> c_int;
        pub fn mi355fft_plan_len(plan: *const Mi355Plan) -> usize;
        pub fn mi355fft_plan_direction(plan: *const Mi355Plan) -> c_int;
        pub fn mi355fft_plan_precision(plan: *const Mi355Plan) -> c_int;
        pub fn mi355fft_scratch_len(plan: *const Mi355Plan, mode: c_int) -> usize;
        pub fn mi355fft_plan_describe(plan: *const Mi355Plan, buf: *mut std::ffi::c_char, cap: usize) -> c_int;
        pub fn mi355fft_process_inplace_host(plan: *const Mi355Plan, buffer: *mut c_void, n_elems: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_process_outofplace_host(plan: *const Mi355Plan, input: *mut c_void, n_in: usize, output: *mut c_void, n_out: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_process_immutable_host(plan: *const Mi355Plan, input: *const c_void, n_in: usize, output: *mut c_void, n_out: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_process_inplace_dev(plan: *const Mi355Plan, buffer: *mut c_void, batch: usize, stream: *mut c_void) -> c_int;
        pub fn mi355fft_process_outofplace_dev(plan: *const Mi355Plan, input: *mut c_void, output: *mut c_void, batch: usize, stream: *mut c_void) -> c_int;
        pub fn mi355fft_process_immutable_dev(plan: *const Mi355Plan, input: *const c_void, output: *mut c_void, batch: usize, stream: *mut c_void) -> c_int;
        pub fn mi355fft_plan_num_kernels(plan: *const Mi355Plan) -> c_int;
        pub fn mi355fft_plan_kernel_name(plan: *const Mi355Plan, index: c_int) -> *const std::ffi::c_char;
        pub fn mi355fft_profile_inplace_dev(plan: *const Mi355Plan, buffer: *mut c_void, batch: usize, stream: *mut c_void, reps: c_int, ms_per_kernel: *mut f32, n_kernels: c_int) -> c_int;
        pub fn mi355fft_multi_plan_create(len: usize, direction: c_int, precision: c_int, options: *const Mi355PlanOptions, devices: *const c_int, n_devices: c_int, out_plan: *mut *mut Mi355MultiPlan) -> c_int;
        pub fn mi355fft_multi_plan_destroy(plan: *mut Mi355MultiPlan) -> c_int;
        pub fn mi355fft_multi_plan_shards(plan: *const Mi355MultiPlan) -> c_int;
        pub fn mi355fft_multi_plan_device(plan: *const Mi355MultiPlan, shard: c_int) -> c_int;
        pub fn mi355fft_multi_plan_replica(plan: *const Mi355MultiPlan, shard: c_int) -> *const Mi355Plan;
        pub fn mi355fft_shard_rows(batch: usize, n_shards: c_int, shard: c_int, first_row: *mut usize, rows: *mut usize) -> c_int;
        pub fn mi355fft_multi_process_inplace_host(plan: *const Mi355MultiPlan, buffer: *mut c_void, n_elems: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_multi_process_outofplace_host(plan: *const Mi355MultiPlan, input: *mut c_void, n_in: usize, output: *mut c_void, n_out: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_multi_process_immutable_host(plan: *const Mi355MultiPlan, input: *const c_void, n_in: usize, output: *mut c_void, n_out: usize, scratch: *mut c_void, scratch_elems: usize) -> c_int;
        pub fn mi355fft_multi_process_inplace_dev(plan: *const Mi355MultiPlan, buffers: *const *mut c_void, batch: usize, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_multi_process_outofplace_dev(plan: *const Mi355MultiPlan, inputs: *const *mut c_void, outputs: *const *mut c_void, batch: usize, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_multi_process_immutable_dev(plan: *const Mi355MultiPlan, inputs: *const *const c_void, outputs: *const *mut c_void, batch: usize, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_multi_synchronize(plan: *const Mi355MultiPlan, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_device_cpulist(device: c_int, buf: *mut std::ffi::c_char, cap: usize) -> c_int;
        pub fn mi355fft_multi_plan_shard_pinned(plan: *const Mi355MultiPlan, shard: c_int) -> c_int;
        pub fn mi355fft_multi_scatter_dev(plan: *const Mi355MultiPlan, root_buffer: *const c_void, root_device: c_int, buffers: *const *mut c_void, batch: usize, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_multi_gather_dev(plan: *const Mi355MultiPlan, buffers: *const *mut c_void, root_buffer: *mut c_void, root_device: c_int, batch: usize, streams: *const *mut c_void) -> c_int;
        pub fn mi355fft_measure_copy_ceiling(bytes: usize, gbps: *mut f64) -> c_int;
        pub fn mi355fft_plan_set_fused(plan: *mut Mi355Plan, mode: c_int) -> c_int;
        pub fn mi355fft_plan_is_fused(plan: *const Mi355Plan) -> c_int;
        pub fn mi355fft_plan_fused_status(plan: *const Mi355Plan, stream: *mut c_void, error_word: *mut c_uint) -> c_int;
        pub fn mi355fft_plan_synchronize(plan: *const Mi355Plan, stream: *mut c_void) -> c_int;
        pub fn mi355fft_plan_set_fused_wait_limit(plan: *mut Mi355Plan, polls: c_int) -> c_int;
        pub fn mi355fft_plan_set_workspace_placement(plan: *mut Mi355Plan, on: c_int) -> c_int;
        pub fn mi355fft_plan_set_chunk_batch(plan: *mut Mi355Plan, chunk_batch: usize) -> c_int;
        pub fn mi355fft_plan_workspace_bytes(plan: *const Mi355Plan) -> usize;
        pub fn mi355fft_plan_trim_workspaces(plan: *mut Mi355Plan, freed: *mut usize) -> c_int;
        pub fn mi355fft_strerror(status: c_int) -> *const std::ffi::c_char;
        pub fn mi355fft_last_error() -> *const std::ffi::c_char;
        pub fn mi355fft_version() -> *const std::ffi::c_char;
    }
}

fn precision_of<T: 'static>() -> Option<c_int> {
    if TypeId::of::<T>() == TypeId::of::<f32>() {
        Some(32)
    } else if TypeId::of::<T>() == TypeId::of::<f64>() {
        Some(64)
    } else {
        None
    }
}

fn direction_code(direction: FftDirection) -> c_int {
    match direction {
        FftDirection::Forward => 0,
        FftDirection::Inverse => 1,
    }
}

/// A host planner's recipe, as far as a caller outside the rustfft crate can state it (rustfft's own `Recipe` is
/// crate-private; the in-tree arm of INTEGRATION.md §3 converts that enum node by node instead).
#[derive(Clone, Debug)]
pub enum RecipeTree {
    Dft(usize),
    Butterfly(usize),
    MixedRadix { left: Box<RecipeTree>, right: Box<RecipeTree> },
    GoodThomas { left: Box<RecipeTree>, right: Box<RecipeTree> },
    Raders { inner: Box<RecipeTree> },
    Bluesteins { len: usize, inner: Box<RecipeTree> },
    RadixN { factors: Vec<usize>, base: Box<RecipeTree> },
    Radix4 { k: u32, base: Box<RecipeTree> },
}

impl RecipeTree {
    pub fn len(&self) -> usize {
        match self {
            RecipeTree::Dft(n) | RecipeTree::Butterfly(n) => *n,
            RecipeTree::MixedRadix { left, right } | RecipeTree::GoodThomas { left, right } => left.len() * right.len(),
            RecipeTree::Raders { inner } => inner.len() + 1,
            RecipeTree::Bluesteins { len, .. } => *len,
            RecipeTree::RadixN { factors, base } => base.len() * factors.iter().product::<usize>(),
            RecipeTree::Radix4 { k, base } => base.len() << (2 * k),
        }
    }

    /// Root first, every child after its parent: the layout `mi355fft_plan_options.recipe` asks for.
    pub fn flatten(&self) -> Vec<ffi::Mi355RecipeNode> {
        let mut nodes: Vec<ffi::Mi355RecipeNode> = Vec::new();
        let mut todo: std::collections::VecDeque<(&RecipeTree, Option<(usize, bool)>)> = std::collections::VecDeque::new();
        todo.push_back((self, None));
        while let Some((tree, parent)) = todo.pop_front() {
            let index = nodes.len();
            if let Some((p, is_right)) = parent {
                if is_right {
                    nodes[p].right = index as c_int;
                } else {
                    nodes[p].left = index as c_int;
                }
            }
            let (kind, left, right): (c_int, Option<&RecipeTree>, Option<&RecipeTree>) = match tree {
                RecipeTree::Dft(_) => (ffi::RECIPE_DFT, None, None),
                RecipeTree::Butterfly(_) => (ffi::RECIPE_BUTTERFLY, None, None),
                RecipeTree::MixedRadix { left, right } => (ffi::RECIPE_MIXED_RADIX, Some(left), Some(right)),
                RecipeTree::GoodThomas { left, right } => (ffi::RECIPE_GOOD_THOMAS, Some(left), Some(right)),
                RecipeTree::Raders { inner } => (ffi::RECIPE_RADERS, Some(inner), None),
                RecipeTree::Bluesteins { inner, .. } => (ffi::RECIPE_BLUESTEINS, Some(inner), None),
                RecipeTree::RadixN { base, .. } => (ffi::RECIPE_RADIXN, Some(base), None),
                RecipeTree::Radix4 { base, .. } => (ffi::RECIPE_RADIX4, Some(base), None),
            };
            nodes.push(ffi::Mi355RecipeNode { kind, left: -1, right: -1, len: tree.len() });
            if let Some(l) = left {
                todo.push_back((l, Some((index, false))));
            }
            if let Some(r) = right {
                todo.push_back((r, Some((index, true))));
            }
        }
        nodes
    }
}

#[cfg(feature = "link")]
mod hip {
    use super::*;

    #[cold]
    fn hip_panic(rc: c_int) -> ! {
        // the library's text for the validation failures is RustFFT's own panic message (src/common.rs of the reference)
        let msg = unsafe { CStr::from_ptr(ffi::mi355fft_last_error()) }.to_string_lossy().into_owned();
        panic!("{} (mi355fft status {})", msg, rc)
    }

    /// One planned transform on the GPU.  Immutable after creation; `process*` may be called from many threads at once.
    pub struct HipFft<T> {
        plan: *mut ffi::Mi355Plan,
        len: usize,
        direction: FftDirection,
        _marker: PhantomData<T>,
    }
    // The plan is immutable; the library hands each calling thread its own staging context and serialises the use of
    // a stream's HBM workspace internally.
    unsafe impl<T> Send for HipFft<T> {}
    unsafe impl<T> Sync for HipFft<T> {}

    impl<T> Drop for HipFft<T> {
        fn drop(&mut self) {
            unsafe {
                ffi::mi355fft_plan_destroy(self.plan);
            }
        }
    }
    impl<T> Length for HipFft<T> {
        fn len(&self) -> usize {
            self.len
        }
    }
    impl<T> Direction for HipFft<T> {
        fn fft_direction(&self) -> FftDirection {
            self.direction
        }
    }

    impl<T> HipFft<T> {
        /// Kernel sequence of the plan (diagnostics).
        pub fn describe(&self) -> String {
            let mut buf = vec![0 as std::ffi::c_char; 1024];
            unsafe {
                ffi::mi355fft_plan_describe(self.plan, buf.as_mut_ptr(), buf.len());
                CStr::from_ptr(buf.as_ptr()).to_string_lossy().into_owned()
            }
        }
        /// What the plan took from the recipe it was created with: 0 none, 1 the family, 2 also the split / inner length.
        pub fn recipe_status(&self) -> i32 {
            unsafe { ffi::mi355fft_plan_recipe_status(self.plan) as i32 }
        }
        /// In-place transform of `batch` sequences that already live in HBM (16-byte aligned device pointer),
        /// asynchronous on `stream` (a `hipStream_t`; null = the default stream).
        pub unsafe fn process_device(&self, buffer: *mut c_void, batch: usize, stream: *mut c_void) {
            let rc = ffi::mi355fft_process_inplace_dev(self.plan, buffer, batch, stream);
            if rc != 0 {
                hip_panic(rc)
            }
        }
        /// Waits for everything enqueued on `stream` and panics if a fused launch of this plan gave up a dependency wait -- the one
        /// failure the asynchronous `process_*_device` calls cannot report themselves (src/lib.rs:184: never silently wrong).  Call it
        /// before reading results produced by the device entry points.
        pub unsafe fn synchronize(&self, stream: *mut c_void) {
            let rc = ffi::mi355fft_plan_synchronize(self.plan, stream);
            if rc != 0 {
                hip_panic(rc)
            }
        }
        /// Out-of-place on device buffers; `input` is left untouched.
        pub unsafe fn process_immutable_device(&self, input: *const c_void, output: *mut c_void, batch: usize, stream: *mut c_void) {
            let rc = ffi::mi355fft_process_immutable_dev(self.plan, input, output, batch, stream);
            if rc != 0 {
                hip_panic(rc)
            }
        }
        /// Out-of-place on device buffers; `input` may be used as workspace.
        pub unsafe fn process_outofplace_device(&self, input: *mut c_void, output: *mut c_void, batch: usize, stream: *mut c_void) {
            let rc = ffi::mi355fft_process_outofplace_dev(self.plan, input, output, batch, stream);
            if rc != 0 {
                hip_panic(rc)
            }
        }
        /// Bytes of HBM workspace the plan currently caches, and a release of all of them.
        pub fn workspace_bytes(&self) -> usize {
            unsafe { ffi::mi355fft_plan_workspace_bytes(self.plan) }
        }
        pub fn trim_workspaces(&self) -> usize {
            let mut freed = 0usize;
            unsafe {
                ffi::mi355fft_plan_trim_workspaces(self.plan, &mut freed);
            }
            freed
        }
    }

    impl<T: FftNum> Fft<T> for HipFft<T> {
        fn process_with_scratch(&self, buffer: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_process_inplace_host(self.plan, buffer.as_mut_ptr() as *mut c_void, buffer.len(), scratch.as_mut_ptr() as *mut c_void, scratch.len())
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn process_outofplace_with_scratch(&self, input: &mut [Complex<T>], output: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_process_outofplace_host(
                    self.plan,
                    input.as_mut_ptr() as *mut c_void,
                    input.len(),
                    output.as_mut_ptr() as *mut c_void,
                    output.len(),
                    scratch.as_mut_ptr() as *mut c_void,
                    scratch.len(),
                )
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn process_immutable_with_scratch(&self, input: &[Complex<T>], output: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_process_immutable_host(
                    self.plan,
                    input.as_ptr() as *const c_void,
                    input.len(),
                    output.as_mut_ptr() as *mut c_void,
                    output.len(),
                    scratch.as_mut_ptr() as *mut c_void,
                    scratch.len(),
                )
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn get_inplace_scratch_len(&self) -> usize {
            unsafe { ffi::mi355fft_scratch_len(self.plan, 0) }
        }
        fn get_outofplace_scratch_len(&self) -> usize {
            unsafe { ffi::mi355fft_scratch_len(self.plan, 1) }
        }
        fn get_immutable_scratch_len(&self) -> usize {
            unsafe { ffi::mi355fft_scratch_len(self.plan, 2) }
        }
    }

    /// One planned transform over SEVERAL GPUs: the rows of every `process*` call are sharded across the devices (the chunk loop
    /// of RustFFT's `validate_and_iter`, src/array_utils.rs:151-177, is the shard axis; no collective in the data path), each
    /// device staging, transforming and copying back its own rows concurrently.  The call site does not change: this is still an
    /// `Arc<dyn Fft<T>>`.
    pub struct HipFftMulti<T> {
        plan: *mut ffi::Mi355MultiPlan,
        len: usize,
        direction: FftDirection,
        _marker: PhantomData<T>,
    }
    unsafe impl<T> Send for HipFftMulti<T> {}
    unsafe impl<T> Sync for HipFftMulti<T> {}
    impl<T> Drop for HipFftMulti<T> {
        fn drop(&mut self) {
            unsafe {
                ffi::mi355fft_multi_plan_destroy(self.plan);
            }
        }
    }
    impl<T> Length for HipFftMulti<T> {
        fn len(&self) -> usize {
            self.len
        }
    }
    impl<T> Direction for HipFftMulti<T> {
        fn fft_direction(&self) -> FftDirection {
            self.direction
        }
    }
    impl<T> HipFftMulti<T> {
        /// Number of shards (= entries of the device list) and the device of each.
        pub fn shards(&self) -> usize {
            unsafe { ffi::mi355fft_multi_plan_shards(self.plan) as usize }
        }
        pub fn device(&self, shard: usize) -> i32 {
            unsafe { ffi::mi355fft_multi_plan_device(self.plan, shard as c_int) as i32 }
        }
        /// Rows `[first, first + rows)` of a batch of `batch` transforms that belong to `shard`.
        pub fn shard_rows(&self, batch: usize, shard: usize) -> (usize, usize) {
            let (mut first, mut rows) = (0usize, 0usize);
            unsafe {
                ffi::mi355fft_shard_rows(batch, self.shards() as c_int, shard as c_int, &mut first, &mut rows);
            }
            (first, rows)
        }
        /// In place on device-resident shards: `buffers[g]` holds shard g's rows in the memory of `device(g)`; asynchronous on
        /// `streams[g]` (null entries / an empty slice: the devices' default streams).  `synchronize` waits for all of them.
        pub unsafe fn process_device(&self, buffers: &[*mut c_void], batch: usize, streams: &[*mut c_void]) {
            assert_eq!(buffers.len(), self.shards());
            let st = if streams.is_empty() { std::ptr::null() } else { streams.as_ptr() };
            let rc = ffi::mi355fft_multi_process_inplace_dev(self.plan, buffers.as_ptr(), batch, st);
            if rc != 0 {
                hip_panic(rc)
            }
        }
        pub unsafe fn synchronize(&self, streams: &[*mut c_void]) {
            let st = if streams.is_empty() { std::ptr::null() } else { streams.as_ptr() };
            let rc = ffi::mi355fft_multi_synchronize(self.plan, st);
            if rc != 0 {
                hip_panic(rc)
            }
        }
    }
    impl<T: FftNum> Fft<T> for HipFftMulti<T> {
        fn process_with_scratch(&self, buffer: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_multi_process_inplace_host(self.plan, buffer.as_mut_ptr() as *mut c_void, buffer.len(), scratch.as_mut_ptr() as *mut c_void, scratch.len())
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn process_outofplace_with_scratch(&self, input: &mut [Complex<T>], output: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_multi_process_outofplace_host(
                    self.plan,
                    input.as_mut_ptr() as *mut c_void,
                    input.len(),
                    output.as_mut_ptr() as *mut c_void,
                    output.len(),
                    scratch.as_mut_ptr() as *mut c_void,
                    scratch.len(),
                )
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn process_immutable_with_scratch(&self, input: &[Complex<T>], output: &mut [Complex<T>], scratch: &mut [Complex<T>]) {
            let rc = unsafe {
                ffi::mi355fft_multi_process_immutable_host(
                    self.plan,
                    input.as_ptr() as *const c_void,
                    input.len(),
                    output.as_mut_ptr() as *mut c_void,
                    output.len(),
                    scratch.as_mut_ptr() as *mut c_void,
                    scratch.len(),
                )
            };
            if rc != 0 {
                hip_panic(rc)
            }
        }
        fn get_inplace_scratch_len(&self) -> usize {
            0
        }
        fn get_outofplace_scratch_len(&self) -> usize {
            0
        }
        fn get_immutable_scratch_len(&self) -> usize {
            0
        }
    }

    /// What the host planner keeps in charge of when it plans through `plan_fft_with`.
    pub struct HostPlannerOptions<'a, T> {
        /// the planner's whole recipe for this length; the GPU takes the family, the six-step split and the Bluestein
        /// inner length from it
        pub recipe: Option<&'a RecipeTree>,
        /// route every twiddle the library uploads through `rustfft`-side arithmetic (`twiddle(index, fft_len)` in the
        /// FORWARD direction, values rounded to `T`)
        pub twiddle: Option<fn(usize, usize) -> Complex<T>>,
        /// finished `RadersAlgorithm` table (`len - 1` entries, the plan's direction)
        pub rader_inner_fft_data: Option<&'a [Complex<T>]>,
        /// finished `BluesteinsAlgorithm` tables (`len` chirp entries / `inner_len` multiplier entries)
        pub bluestein_twiddles: Option<&'a [Complex<T>]>,
        pub bluestein_multiplier: Option<&'a [Complex<T>]>,
    }

    impl<'a, T> Default for HostPlannerOptions<'a, T> {
        fn default() -> Self {
            Self { recipe: None, twiddle: None, rader_inner_fft_data: None, bluestein_twiddles: None, bluestein_multiplier: None }
        }
    }

    extern "C" fn twiddle_thunk<T: FftNum>(ctx: *mut c_void, index: usize, fft_len: usize, re: *mut f64, im: *mut f64) {
        // ctx carries the fn pointer itself
        let f: fn(usize, usize) -> Complex<T> = unsafe { std::mem::transmute::<*mut c_void, fn(usize, usize) -> Complex<T>>(ctx) };
        let w = f(index, fft_len);
        unsafe {
            *re = w.re.to_f64().unwrap();
            *im = w.im.to_f64().unwrap();
        }
    }

    /// Same shape as RustFFT's ISA-gated planners: construction fails when the hardware is missing.
    pub struct FftPlannerHip<T: FftNum> {
        cache: HashMap<(usize, bool), Arc<dyn Fft<T>>>,
        precision: c_int,
        fallback: rustfft::FftPlanner<T>,
        /// devices the plans shard their batch rows over (more than one entry: `HipFftMulti`)
        devices: Vec<c_int>,
    }

    impl<T: FftNum> FftPlannerHip<T> {
        /// `Err(())` when `T` is neither f32 nor f64, no gfx950 device is visible, or the HIP runtime fails to start.
        pub fn new() -> Result<Self, ()> {
            let precision = precision_of::<T>().ok_or(())?;
            if unsafe { ffi::mi355fft_device_count() } <= 0 || unsafe { ffi::mi355fft_init(0) } != 0 {
                return Err(());
            }
            // ONE device, like every other back-end planner of the reference: a rank-per-GPU job must not create contexts, tables and
            // worker threads on its neighbours' GPUs, and a batch of one must not pay a worker-thread hop.  Sharding over the node is
            // opt-in: `new_multi()` / `with_devices()`.
            Ok(Self { cache: HashMap::new(), precision, fallback: rustfft::FftPlanner::new(), devices: vec![0] })
        }
        /// Every visible gfx950 device: plans shard their batch rows across all of them (`HipFftMulti`: one replica, one worker thread
        /// and one staging pool per device and per planned (len, direction) -- meant for a process that owns the whole node).
        pub fn new_multi() -> Result<Self, ()> {
            let mut p = Self::new()?;
            p.devices = (0..unsafe { ffi::mi355fft_device_count() }).collect();
            Ok(p)
        }
        /// The same over an explicit device list (an ordinal may repeat: each entry is one shard).
        pub fn with_devices(devices: &[i32]) -> Result<Self, ()> {
            let mut p = Self::new()?;
            let visible = unsafe { ffi::mi355fft_device_count() };
            if devices.is_empty() || devices.iter().any(|&d| d < 0 || d >= visible as i32) {
                return Err(());
            }
            p.devices = devices.iter().map(|&d| d as c_int).collect();
            Ok(p)
        }
        fn create_multi(&self, len: usize, direction: FftDirection) -> Result<HipFftMulti<T>, i32> {
            let mut plan: *mut ffi::Mi355MultiPlan = std::ptr::null_mut();
            let rc = unsafe {
                ffi::mi355fft_multi_plan_create(len, direction_code(direction), self.precision, std::ptr::null(), self.devices.as_ptr(), self.devices.len() as c_int, &mut plan)
            };
            if rc != 0 {
                return Err(rc as i32);
            }
            Ok(HipFftMulti { plan, len, direction, _marker: PhantomData })
        }

        /// One `Arc` per (len, direction), like RustFFT's planner cache.
        pub fn plan_fft(&mut self, len: usize, direction: FftDirection) -> Arc<dyn Fft<T>> {
            let key = (len, direction == FftDirection::Inverse);
            if let Some(fft) = self.cache.get(&key) {
                return Arc::clone(fft);
            }
            let planned: Result<Arc<dyn Fft<T>>, i32> = if self.devices.len() > 1 {
                self.create_multi(len, direction).map(|gpu| Arc::new(gpu) as Arc<dyn Fft<T>>)
            } else {
                self.create(len, direction, std::ptr::null()).map(|gpu| Arc::new(gpu) as Arc<dyn Fft<T>>)
            };
            let fft: Arc<dyn Fft<T>> = match planned {
                Ok(gpu) => gpu,
                // MI355FFT_ERR_UNSUPPORTED only: no GPU plan for this length in this build -> the CPU planner serves it
                Err(rc) if rc == ffi::ERR_UNSUPPORTED as i32 => self.fallback.plan_fft(len, direction),
                // anything else (out of memory, a HIP failure, an invalid argument) is a real error: a silent CPU plan would hide it
                Err(rc) => panic!("mi355fft_plan_create failed for len {}: {}", len, strerror(rc)),
            };
            self.cache.insert(key, Arc::clone(&fft));
            fft
        }
        pub fn plan_fft_forward(&mut self, len: usize) -> Arc<dyn Fft<T>> {
            self.plan_fft(len, FftDirection::Forward)
        }
        pub fn plan_fft_inverse(&mut self, len: usize) -> Arc<dyn Fft<T>> {
            self.plan_fft(len, FftDirection::Inverse)
        }

        /// The host planner in charge (`mi355fft_plan_create_ex`).  Not cached.  `Err(status)` when the GPU cannot run the
        /// requested family at this length (status 6) or the tables do not fit (status 7): the caller retries with
        /// less, or with `plan_fft`.
        pub fn plan_fft_with(&mut self, len: usize, direction: FftDirection, options: &HostPlannerOptions<T>) -> Result<Arc<HipFft<T>>, i32> {
            let nodes = options.recipe.map(|r| r.flatten()).unwrap_or_default();
            let as_ptr = |s: Option<&[Complex<T>]>| s.map_or(std::ptr::null(), |t| t.as_ptr() as *const c_void);
            let raw = ffi::Mi355PlanOptions {
                struct_size: std::mem::size_of::<ffi::Mi355PlanOptions>(),
                algorithm: if options.rader_inner_fft_data.is_some() {
                    ffi::ALGO_RADER
                } else if options.bluestein_multiplier.is_some() || options.bluestein_twiddles.is_some() {
                    ffi::ALGO_BLUESTEIN
                } else {
                    ffi::ALGO_AUTO // a recipe, when present, names the family
                },
                twiddle_fn: options.twiddle.map(|_| twiddle_thunk::<T> as extern "C" fn(*mut c_void, usize, usize, *mut f64, *mut f64)),
                twiddle_ctx: options.twiddle.map_or(std::ptr::null_mut(), |f| f as usize as *mut c_void),
                rader_inner_fft_data: as_ptr(options.rader_inner_fft_data),
                bluestein_twiddles: as_ptr(options.bluestein_twiddles),
                bluestein_multiplier: as_ptr(options.bluestein_multiplier),
                bluestein_inner_len: options.bluestein_multiplier.map_or(0, |t| t.len()),
                recipe: if nodes.is_empty() { std::ptr::null() } else { nodes.as_ptr() },
                recipe_nodes: nodes.len(),
            };
            self.create(len, direction, &raw).map(Arc::new)
        }

        /// Inner (padded) length the GPU's own Bluestein choice uses for `len`: what a host planner sizes its
        /// multiplier table for when it does not name an inner length itself.
        pub fn bluestein_inner_len(&self, len: usize) -> usize {
            unsafe { ffi::mi355fft_bluestein_inner_len(len, self.precision) }
        }

        fn create(&self, len: usize, direction: FftDirection, options: *const ffi::Mi355PlanOptions) -> Result<HipFft<T>, i32> {
            let mut plan: *mut ffi::Mi355Plan = std::ptr::null_mut();
            let rc = unsafe {
                if options.is_null() {
                    ffi::mi355fft_plan_create(len, direction_code(direction), self.precision, &mut plan)
                } else {
                    ffi::mi355fft_plan_create_ex(len, direction_code(direction), self.precision, options, &mut plan)
                }
            };
            if rc != 0 {
                return Err(rc as i32);
            }
            debug_assert_eq!(unsafe { ffi::mi355fft_plan_len(plan) }, len);
            debug_assert_eq!(unsafe { ffi::mi355fft_plan_direction(plan) }, direction_code(direction));
            debug_assert_eq!(unsafe { ffi::mi355fft_plan_precision(plan) }, self.precision);
            Ok(HipFft { plan, len, direction, _marker: PhantomData })
        }
    }

    /// Library version string and the text of a status code.
    pub fn version() -> String {
        unsafe { CStr::from_ptr(ffi::mi355fft_version()) }.to_string_lossy().into_owned()
    }
    pub fn strerror(status: i32) -> String {
        unsafe { CStr::from_ptr(ffi::mi355fft_strerror(status as c_int)) }.to_string_lossy().into_owned()
    }
}

#[cfg(feature = "link")]
pub use hip::{strerror, version, FftPlannerHip, HipFft, HipFftMulti, HostPlannerOptions};

/// Without the `link` feature the crate is the stub RustFFT uses for a disabled back-end: the planner type exists and
/// its constructor reports that the hardware is unavailable.
#[cfg(not(feature = "link"))]
pub struct FftPlannerHip<T: FftNum> {
    _marker: PhantomData<T>,
}
#[cfg(not(feature = "link"))]
impl<T: FftNum> FftPlannerHip<T> {
    pub fn new() -> Result<Self, ()> {
        Err(())
    }
    pub fn plan_fft(&mut self, _len: usize, _direction: FftDirection) -> Arc<dyn Fft<T>> {
        unreachable!()
    }
    pub fn plan_fft_forward(&mut self, _len: usize) -> Arc<dyn Fft<T>> {
        unreachable!()
    }
    pub fn plan_fft_inverse(&mut self, _len: usize) -> Arc<dyn Fft<T>> {
        unreachable!()
    }
}

/// `rustfft::FftPlanner`'s chooser with the GPU in front: the GPU planner when a gfx950 device is present, RustFFT's own
/// planner (AVX / SSE / NEON / scalar) otherwise.
pub enum FftPlanner<T: FftNum> {
    Hip(FftPlannerHip<T>),
    Cpu(rustfft::FftPlanner<T>),
}
impl<T: FftNum> FftPlanner<T> {
    pub fn new() -> Self {
        match FftPlannerHip::new() {
            Ok(p) => FftPlanner::Hip(p),
            Err(()) => FftPlanner::Cpu(rustfft::FftPlanner::new()),
        }
    }
    pub fn plan_fft(&mut self, len: usize, direction: FftDirection) -> Arc<dyn Fft<T>> {
        match self {
            FftPlanner::Hip(p) => p.plan_fft(len, direction),
            FftPlanner::Cpu(p) => p.plan_fft(len, direction),
        }
    }
    pub fn plan_fft_forward(&mut self, len: usize) -> Arc<dyn Fft<T>> {
        self.plan_fft(len, FftDirection::Forward)
    }
    pub fn plan_fft_inverse(&mut self, len: usize) -> Arc<dyn Fft<T>> {
        self.plan_fft(len, FftDirection::Inverse)
    }
}
impl<T: FftNum> Default for FftPlanner<T> {
    fn default() -> Self {
        Self::new()
    }
}
