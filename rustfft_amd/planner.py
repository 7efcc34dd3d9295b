"""Python mirror of the reference's `FftPlanner` / `Fft<T>` surface for the HIP back-end.

Names, argument meaning and error behaviour follow the reference so the parity tests read like the
reference's own tests:
  FftPlanner::new / plan_fft / plan_fft_forward / plan_fft_inverse      src/plan.rs:72-126
  Fft::process / process_with_scratch / process_outofplace_with_scratch /
       process_immutable_with_scratch / get_*_scratch_len               src/lib.rs:184-278
  Length::len, Direction::fft_direction                                 src/lib.rs:140-177
  panic messages                                                         src/common.rs:13-104  (-> FftPanic)
Buffers may be numpy arrays (host slices: the literal drop-in, staged through PCIe) or torch CUDA tensors
(HBM-resident: the measured path, asynchronous on torch's current stream).
"""
import ctypes
import enum

import numpy as np

from . import _native

SCRATCH_INPLACE, SCRATCH_OUTOFPLACE, SCRATCH_IMMUTABLE = 0, 1, 2
ALGO_AUTO, ALGO_RADER, ALGO_BLUESTEIN, ALGO_MIXED_RADIX = 0, 1, 2, 3  # top-level Recipe family (src/plan.rs:134-188)


RECIPE_STATUS_NONE, RECIPE_STATUS_FAMILY, RECIPE_STATUS_SPLIT = 0, 1, 2


class Recipe:
    """The host planner's `enum Recipe` (src/plan.rs:134-188) as a tree of (kind, len, children); `flatten()` gives the node
    array of mi355fft_plan_options.recipe (root first, children after their parent).  `Recipe.parse` reads the Debug-like
    notation `MixedRadix{<left>,<right>}`, `RadersAlgorithm{<inner>}`, `BluesteinsAlgorithm{len,<inner>}`,
    `RadixN{[f0,f1,..],<base>}`, `Radix4{k,<base>}`, `Dft(len)`, `ButterflyN`."""
    DFT, MIXED_RADIX, GOOD_THOMAS, MIXED_RADIX_SMALL, GOOD_THOMAS_SMALL, RADERS, BLUESTEINS, RADIXN, RADIX4, BUTTERFLY = range(10)
    _SPLITS = {"MixedRadix": 1, "GoodThomasAlgorithm": 2, "MixedRadixSmall": 3, "GoodThomasAlgorithmSmall": 4}

    def __init__(self, kind, len, left=None, right=None):
        self.kind, self.len, self.left, self.right = int(kind), int(len), left, right

    @staticmethod
    def dft(len):
        return Recipe(Recipe.DFT, len)

    @staticmethod
    def butterfly(len):
        return Recipe(Recipe.BUTTERFLY, len)

    @staticmethod
    def mixed_radix(left, right, kind=MIXED_RADIX):
        return Recipe(kind, left.len * right.len, left, right)

    @staticmethod
    def raders(inner):
        return Recipe(Recipe.RADERS, inner.len + 1, inner)

    @staticmethod
    def bluesteins(len, inner):
        return Recipe(Recipe.BLUESTEINS, len, inner)

    @staticmethod
    def radixn(factors, base):
        n = base.len
        for f in factors:
            n *= int(f)
        return Recipe(Recipe.RADIXN, n, base)

    @staticmethod
    def radix4(k, base):
        return Recipe(Recipe.RADIX4, base.len << (2 * int(k)), base)

    def flatten(self):
        nodes, todo = [], [(self, None, None)]
        while todo:
            r, parent, side = todo.pop(0)
            idx = len(nodes)
            nodes.append([r.kind, -1, -1, r.len])
            if parent is not None:
                nodes[parent][side] = idx
            if r.left is not None:
                todo.append((r.left, idx, 1))
            if r.right is not None:
                todo.append((r.right, idx, 2))
        arr = (_native.RecipeNode * len(nodes))()
        for i, (k, l, rr, n) in enumerate(nodes):
            arr[i].kind, arr[i].left, arr[i].right, arr[i].len = k, l, rr, n
        return arr

    @staticmethod
    def parse(text):
        text = text.replace(" ", "")
        pos = 0

        def number():
            nonlocal pos
            start = pos
            while pos < len(text) and text[pos].isdigit():
                pos += 1
            return int(text[start:pos])

        def expect(ch):
            nonlocal pos
            if text[pos] != ch:
                raise ValueError(f"recipe: expected {ch!r} at {pos} in {text!r}")
            pos += 1

        def node():
            nonlocal pos
            start = pos
            while pos < len(text) and text[pos].isalpha():
                pos += 1
            name = text[start:pos]
            if name == "Radix" and text[pos:pos + 1] == "4":
                name, pos = "Radix4", pos + 1
            if name == "Butterfly":
                return Recipe.butterfly(number())
            if name == "Dft":
                expect("(")
                n = number()
                expect(")")
                return Recipe.dft(n)
            expect("{")
            if name in Recipe._SPLITS:
                left = node()
                expect(",")
                right = node()
                r = Recipe.mixed_radix(left, right, Recipe._SPLITS[name])
            elif name == "RadersAlgorithm":
                r = Recipe.raders(node())
            elif name == "BluesteinsAlgorithm":
                n = number()
                expect(",")
                r = Recipe.bluesteins(n, node())
            elif name == "RadixN":
                expect("[")
                factors = [number()]
                while text[pos] == ",":
                    pos += 1
                    factors.append(number())
                expect("]")
                expect(",")
                r = Recipe.radixn(factors, node())
            elif name == "Radix4":
                k = number()
                expect(",")
                r = Recipe.radix4(k, node())
            else:
                raise ValueError(f"recipe: unknown variant {name!r}")
            expect("}")
            return r

        r = node()
        if pos != len(text):
            raise ValueError(f"recipe: trailing text at {pos} in {text!r}")
        return r


class FftDirection(enum.IntEnum):  # src/lib.rs:146-171
    Forward = 0
    Inverse = 1

    def opposite_direction(self):
        return FftDirection.Inverse if self == FftDirection.Forward else FftDirection.Forward


class FftPanic(RuntimeError):
    """A reference `panic!` (src/common.rs:13-104), carrying the reference's message."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _precision(dtype):
    dtype = np.dtype(dtype)
    if dtype in (np.dtype(np.complex64), np.dtype(np.float32)):
        return 32
    if dtype in (np.dtype(np.complex128), np.dtype(np.float64)):
        return 64
    raise TypeError("FftNum must be f32 or f64 (complex64 / complex128 buffers)")


def device_count(lib=None):
    return (lib or _native.load()).mi355fft_device_count()


def device_cpulist(device, lib=None):
    """Cores of the NUMA node GPU `device` is attached to ("0-63,128-191"; "" when sysfs does not say): where the library binds the
    threads it creates for that device (mi355fft_device_cpulist)."""
    buf = ctypes.create_string_buffer(4096)
    n = (lib or _native.load()).mi355fft_device_cpulist(int(device), buf, 4096)
    return buf.value.decode() if n >= 0 else ""


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Fft:
    """`Arc<dyn Fft<T>>` for the HIP back-end."""

    def __init__(self, lib, handle, dtype):
        self._lib = lib
        self._h = handle
        self.dtype = np.dtype(dtype)
        self._len = lib.mi355fft_plan_len(handle)
        self._dir = FftDirection(lib.mi355fft_plan_direction(handle))

    def __del__(self):
        try:
            self._lib.mi355fft_plan_destroy(self._h)
        except Exception:
            pass

    # Length / Direction
    def len(self):
        return self._len

    def fft_direction(self):
        return self._dir

    # scratch queries (src/lib.rs:262-277): 0 — the workspace lives in HBM and belongs to the plan
    def get_inplace_scratch_len(self):
        return self._lib.mi355fft_scratch_len(self._h, SCRATCH_INPLACE)

    def get_outofplace_scratch_len(self):
        return self._lib.mi355fft_scratch_len(self._h, SCRATCH_OUTOFPLACE)

    def get_immutable_scratch_len(self):
        return self._lib.mi355fft_scratch_len(self._h, SCRATCH_IMMUTABLE)

    def recipe_status(self):
        """What the plan took from the host planner's recipe (RECIPE_STATUS_NONE / _FAMILY / _SPLIT)."""
        return self._lib.mi355fft_plan_recipe_status(self._h)

    def describe(self):
        buf = ctypes.create_string_buffer(1024)
        self._lib.mi355fft_plan_describe(self._h, buf, 1024)
        return buf.value.decode()

    def kernel_names(self):
        n = self._lib.mi355fft_plan_num_kernels(self._h)
        return [self._lib.mi355fft_plan_kernel_name(self._h, i).decode() for i in range(n)]

    def set_workspace_placement(self, on=True):
        """Opt into the measured choice of the in-place workspace allocation (blocking first call, 3x transient memory)."""
        self._check(self._lib.mi355fft_plan_set_workspace_placement(self._h, 1 if on else 0))

    def set_chunk_batch(self, chunk_batch):
        self._check(self._lib.mi355fft_plan_set_chunk_batch(self._h, int(chunk_batch)))

    def set_fused(self, mode):
        """Fused two-pass launch: -1 = the planner's measured choice, 0 = never, 1 = whenever a fused kernel exists."""
        self._check(self._lib.mi355fft_plan_set_fused(self._h, int(mode)))

    def is_fused(self):
        return bool(self._lib.mi355fft_plan_is_fused(self._h))

    def set_fused_wait_limit(self, polls):
        """Polls (about half a microsecond each) before a dependency wait of a fused launch gives up; 0 = at once (tests)."""
        self._check(self._lib.mi355fft_plan_set_fused_wait_limit(self._h, int(polls)))

    def fused_status(self):
        """Sticky error word of the plan's fused two-pass launches on torch's current stream since the last report (synchronises the stream,
        clears the word); 0 = every dependency was met in time."""
        w = ctypes.c_uint(0)
        self._check(self._lib.mi355fft_plan_fused_status(self._h, self._stream(), ctypes.byref(w)))
        return int(w.value)

    def synchronize(self):
        """Waits for torch's current stream and raises if a fused launch of this plan on it gave up a dependency wait: the verdict an asynchronous
        device call cannot return itself (mi355fft_plan_synchronize; src/lib.rs:184)."""
        self._check(self._lib.mi355fft_plan_synchronize(self._h, self._stream()))

    def workspace_bytes(self):
        """HBM the plan currently holds as per-stream workspaces."""
        return int(self._lib.mi355fft_plan_workspace_bytes(self._h))

    def trim_workspaces(self):
        freed = ctypes.c_size_t(0)
        self._check(self._lib.mi355fft_plan_trim_workspaces(self._h, ctypes.byref(freed)))
        return int(freed.value)

    # ---- helpers -------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or self._lib.mi355fft_strerror(rc).decode())

    def _host(self, a, writable):
        if not isinstance(a, np.ndarray):
            raise TypeError("expected a numpy array or a torch CUDA tensor")
        if a.dtype != self.dtype:
            raise TypeError(f"buffer dtype {a.dtype} does not match the plan ({self.dtype})")
        if not a.flags.c_contiguous or (writable and not a.flags.writeable):
            raise ValueError("buffers must be C-contiguous (and writable where the trait takes &mut)")
        return ctypes.c_void_p(a.ctypes.data if a.size else 0), a.size

    def _dev(self, t):
        import torch

        want = torch.complex64 if self.dtype == np.complex64 else torch.complex128
        if t.dtype != want or not t.is_cuda or not t.is_contiguous():
            raise TypeError("device buffers must be contiguous CUDA tensors of the plan's complex dtype")
        return ctypes.c_void_p(t.data_ptr()), t.numel()

    @staticmethod
    def _stream():
        import torch

        if not torch.cuda.is_available():  # the kernel-body emulator of the CPU tests: no streams
            return ctypes.c_void_p(0)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _validate_dev(self, n_in, n_out=None):
        # same outcomes as the host path (src/fft_helper.rs + src/common.rs), raised before any launch
        n = self._len
        if n == 0:
            return 0
        if n_out is not None and n_in != n_out:
            raise FftPanic(5, "Provided FFT input buffer and output buffer must have the same length. "
                              f"Got input.len() = {n_in}, output.len() = {n_out}")
        return n_in // n

    def _tail_error(self, n_in):
        n = self._len
        if n and n_in % n:
            if n_in < n:
                raise FftPanic(2, f"Provided FFT buffer was too small. Expected len = {n}, got len = {n_in}")
            raise FftPanic(3, f"Input FFT buffer must be a multiple of FFT length. Expected multiple of {n}, got len = {n_in}")

    # ---- the trait methods ------------------------------------------------------------------------------
    def process(self, buffer):
        """Fft::process (src/lib.rs:195-198)."""
        self.process_with_scratch(buffer, None)

    def process_with_scratch(self, buffer, scratch=None):
        """Fft::process_with_scratch (src/lib.rs:211)."""
        if _is_torch(buffer):
            p, n = self._dev(buffer)
            batch = self._validate_dev(n)
            if batch:
                self._check(self._lib.mi355fft_process_inplace_dev(self._h, p, batch, self._stream()))
            self._tail_error(n)
            return
        p, n = self._host(buffer, True)
        sp, sn = (None, 0) if scratch is None else self._host(scratch, True)
        self._check(self._lib.mi355fft_process_inplace_host(self._h, p, n, sp, sn))

    def process_outofplace_with_scratch(self, input, output, scratch=None):
        """Fft::process_outofplace_with_scratch (src/lib.rs:231) — `input` may be clobbered."""
        if _is_torch(input):
            pi, ni = self._dev(input)
            po, no = self._dev(output)
            batch = self._validate_dev(ni, no)
            if batch:
                self._check(self._lib.mi355fft_process_outofplace_dev(self._h, pi, po, batch, self._stream()))
            self._tail_error(ni)
            return
        pi, ni = self._host(input, True)
        po, no = self._host(output, True)
        sp, sn = (None, 0) if scratch is None else self._host(scratch, True)
        self._check(self._lib.mi355fft_process_outofplace_host(self._h, pi, ni, po, no, sp, sn))

    def process_immutable_with_scratch(self, input, output, scratch=None):
        """Fft::process_immutable_with_scratch (src/lib.rs:250) — `input` is preserved."""
        if _is_torch(input):
            pi, ni = self._dev(input)
            po, no = self._dev(output)
            batch = self._validate_dev(ni, no)
            if batch:
                self._check(self._lib.mi355fft_process_immutable_dev(self._h, pi, po, batch, self._stream()))
            self._tail_error(ni)
            return
        pi, ni = self._host(input, False)
        po, no = self._host(output, True)
        sp, sn = (None, 0) if scratch is None else self._host(scratch, True)
        self._check(self._lib.mi355fft_process_immutable_host(self._h, pi, ni, po, no, sp, sn))

    # ---- measurement hook --------------------------------------------------------------------------------
    def profile_kernels(self, buffer, reps=5):
        """Mean milliseconds of each kernel of the in-place transform (HIP events on torch's current stream)."""
        p, n = self._dev(buffer)
        batch = n // self._len
        nk = self._lib.mi355fft_plan_num_kernels(self._h)
        ms = (ctypes.c_float * max(nk, 1))()
        self._check(self._lib.mi355fft_profile_inplace_dev(self._h, p, batch, self._stream(), reps, ms, nk))
        return [float(ms[i]) for i in range(nk)]


class FftPlannerHip:
    """`FftPlannerHip<T>`: the back-end planner a `ChosenFftPlanner::Hip` arm would hold (pattern:
    FftPlannerAvx, src/avx/avx_planner.rs:113-192).  Construction fails (the `Err(())` of the reference's
    SIMD planners) when no gfx950 device is visible."""

    def __init__(self, dtype=np.complex64, device=0, lib=None):
        self._lib = lib or _native.load()
        self.dtype = np.dtype(dtype)
        self._prec = _precision(dtype)
        rc = self._lib.mi355fft_init(device)
        if rc != 0:
            raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or "no gfx950 device")
        self._cache = {}  # src/fft_cache.rs:5-39 — one instance per (len, direction)

    def plan_fft(self, len, direction):
        direction = FftDirection(direction)
        key = (int(len), direction)
        if key not in self._cache:
            h = ctypes.c_void_p()
            rc = self._lib.mi355fft_plan_create(int(len), int(direction), self._prec, ctypes.byref(h))
            if rc != 0:
                raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or self._lib.mi355fft_strerror(rc).decode())
            self._cache[key] = Fft(self._lib, h, self.dtype)
        return self._cache[key]

    def plan_fft_with(self, len, direction, algorithm=ALGO_AUTO, twiddle_fn=None, rader_inner_fft_data=None,
                      bluestein_twiddles=None, bluestein_multiplier=None, recipe=None):
        """mi355fft_plan_create_ex: the HOST planner in charge -- it names the Recipe family (or hands over its whole `Recipe`
        tree: a `Recipe` or its Debug-like text) and may supply its own `compute_twiddle(index, fft_len) -> complex`
        (src/twiddles.rs:6-23, forward direction) and / or its finished Rader / Bluestein tables (numpy arrays of the plan's
        complex dtype, in the plan's direction).  Not cached."""
        direction = FftDirection(direction)
        o = _native.PlanOptions()
        o.struct_size = ctypes.sizeof(_native.PlanOptions)
        o.algorithm = int(algorithm)
        keep = []
        if recipe is not None:
            nodes = (Recipe.parse(recipe) if isinstance(recipe, str) else recipe).flatten()
            keep.append(nodes)
            o.recipe = ctypes.cast(nodes, ctypes.POINTER(_native.RecipeNode))
            o.recipe_nodes = nodes._length_
        if twiddle_fn is not None:
            def thunk(_ctx, index, fft_len, re, im):
                w = complex(twiddle_fn(index, fft_len))
                re[0], im[0] = w.real, w.imag

            cb = _native.TWIDDLE_FN(thunk)
            keep.append(cb)
            o.twiddle_fn = cb
        for name, arr in (("rader_inner_fft_data", rader_inner_fft_data), ("bluestein_twiddles", bluestein_twiddles),
                          ("bluestein_multiplier", bluestein_multiplier)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=self.dtype)
                keep.append(a)
                setattr(o, name, a.ctypes.data)
                if name == "bluestein_multiplier":
                    o.bluestein_inner_len = a.size
        h = ctypes.c_void_p()
        rc = self._lib.mi355fft_plan_create_ex(int(len), int(direction), self._prec, ctypes.byref(o), ctypes.byref(h))
        if rc != 0:
            raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or self._lib.mi355fft_strerror(rc).decode())
        return Fft(self._lib, h, self.dtype)

    def bluestein_inner_len(self, len):
        return int(self._lib.mi355fft_bluestein_inner_len(int(len), self._prec))

    def plan_fft_forward(self, len):
        return self.plan_fft(len, FftDirection.Forward)

    def plan_fft_inverse(self, len):
        return self.plan_fft(len, FftDirection.Inverse)


class FftMulti:
    """One `Arc<dyn Fft<T>>` over several GPUs (mi355fft_multi_plan): the rows of a batched call are sharded across the
    devices by `shard_rows` (the chunk loop of src/array_utils.rs:151-177 is the shard axis), no collective in the data
    path.  Host slices (numpy) go through the three trait methods unchanged; device-resident data is a LIST of per-device
    torch tensors, shard g holding rows `shard_rows(batch, g)` on `devices()[g]`."""

    def __init__(self, lib, handle, dtype):
        self._lib, self._h, self.dtype = lib, handle, np.dtype(dtype)
        self._replica = lib.mi355fft_multi_plan_replica(handle, 0)
        self._len = lib.mi355fft_plan_len(self._replica)
        self._dir = FftDirection(lib.mi355fft_plan_direction(self._replica))

    def __del__(self):
        try:
            self._lib.mi355fft_multi_plan_destroy(self._h)
        except Exception:
            pass

    def len(self):
        return self._len

    def fft_direction(self):
        return self._dir

    def get_inplace_scratch_len(self):
        return 0

    def get_outofplace_scratch_len(self):
        return 0

    def get_immutable_scratch_len(self):
        return 0

    def shards(self):
        return self._lib.mi355fft_multi_plan_shards(self._h)

    def devices(self):
        return [self._lib.mi355fft_multi_plan_device(self._h, g) for g in range(self.shards())]

    def shard_pinned(self, shard):
        """True when the shard's worker thread is bound to the cores of its GPU's NUMA node (mi355fft_multi_plan_shard_pinned)."""
        return bool(self._lib.mi355fft_multi_plan_shard_pinned(self._h, int(shard)))

    def shard_rows(self, batch, shard):
        first, rows = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self._lib.mi355fft_shard_rows(int(batch), self.shards(), int(shard), ctypes.byref(first), ctypes.byref(rows)))
        return int(first.value), int(rows.value)

    def describe(self):
        buf = ctypes.create_string_buffer(1024)
        self._lib.mi355fft_plan_describe(self._replica, buf, 1024)
        return "%d shards on devices %s: %s" % (self.shards(), self.devices(), buf.value.decode())

    def _check(self, rc):
        if rc != 0:
            raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or self._lib.mi355fft_strerror(rc).decode())

    _host = Fft._host

    # ---- the trait methods on host slices ----------------------------------------------------------------
    def process(self, buffer):
        self.process_with_scratch(buffer, None)

    def process_with_scratch(self, buffer, scratch=None):
        if isinstance(buffer, (list, tuple)):
            return self._dev_call(self._lib.mi355fft_multi_process_inplace_dev, buffer, None)
        p, n = self._host(buffer, True)
        self._check(self._lib.mi355fft_multi_process_inplace_host(self._h, p, n, None, 0))

    def process_outofplace_with_scratch(self, input, output, scratch=None):
        if isinstance(input, (list, tuple)):
            return self._dev_call(self._lib.mi355fft_multi_process_outofplace_dev, input, output)
        pi, ni = self._host(input, True)
        po, no = self._host(output, True)
        self._check(self._lib.mi355fft_multi_process_outofplace_host(self._h, pi, ni, po, no, None, 0))

    def process_immutable_with_scratch(self, input, output, scratch=None):
        if isinstance(input, (list, tuple)):
            return self._dev_call(self._lib.mi355fft_multi_process_immutable_dev, input, output)
        pi, ni = self._host(input, False)
        po, no = self._host(output, True)
        self._check(self._lib.mi355fft_multi_process_immutable_host(self._h, pi, ni, po, no, None, 0))

    # ---- device-resident shards -----------------------------------------------------------------------------
    def _ptrs(self, tensors):
        arr = (ctypes.c_void_p * self.shards())()
        for g, t in enumerate(tensors):
            arr[g] = t.data_ptr() if t is not None and t.numel() else None
        return arr

    def _streams(self):
        import torch

        arr = (ctypes.c_void_p * self.shards())()
        if torch.cuda.is_available():  # (the kernel-body emulator of the CPU tests has no streams: NULL = default)
            for g, d in enumerate(self.devices()):
                arr[g] = torch.cuda.current_stream(d).cuda_stream
        return arr

    def _dev_call(self, fn, a, b):
        if len(a) != self.shards() or (b is not None and len(b) != self.shards()):
            raise ValueError("one tensor per shard")
        batch = sum(t.numel() for t in a) // self._len if self._len else 0
        for g, t in enumerate(a):  # the tensors must hold exactly their shard's rows
            if t.numel() != self.shard_rows(batch, g)[1] * self._len:
                raise ValueError(f"shard {g}: expected {self.shard_rows(batch, g)[1]} rows of {self._len}")
        if b is None:
            self._check(fn(self._h, self._ptrs(a), batch, self._streams()))
        else:
            self._check(fn(self._h, self._ptrs(a), self._ptrs(b), batch, self._streams()))

    def synchronize(self):
        self._check(self._lib.mi355fft_multi_synchronize(self._h, self._streams()))

    def scatter(self, root, shards, root_device=None):
        """Peer copies of every shard's rows out of `root` (a CUDA tensor with the whole batch) into `shards`."""
        batch = root.numel() // self._len
        dev = root.device.index if root_device is None else root_device
        self._check(self._lib.mi355fft_multi_scatter_dev(self._h, ctypes.c_void_p(root.data_ptr()), dev, self._ptrs(shards), batch, self._streams()))

    def gather(self, shards, root, root_device=None):
        batch = root.numel() // self._len
        dev = root.device.index if root_device is None else root_device
        self._check(self._lib.mi355fft_multi_gather_dev(self._h, self._ptrs(shards), ctypes.c_void_p(root.data_ptr()), dev, batch, self._streams()))


class FftPlannerHipMulti:
    """`FftPlannerHip` over every (or the listed) gfx950 device of the node: `plan_fft*` return `FftMulti` objects whose
    process*() calls shard the batch rows across the devices behind the unchanged trait surface."""

    def __init__(self, dtype=np.complex64, devices=None, lib=None):
        self._lib = lib or _native.load()
        self.dtype = np.dtype(dtype)
        self._prec = _precision(dtype)
        if self._lib.mi355fft_device_count() <= 0:
            raise FftPanic(1, "no gfx950 device")
        self._devices = None if devices is None else [int(d) for d in devices]
        self._cache = {}

    def plan_fft(self, len, direction):
        direction = FftDirection(direction)
        key = (int(len), direction)
        if key not in self._cache:
            h = ctypes.c_void_p()
            if self._devices is None:
                devs, nd = None, 0
            else:
                devs, nd = (ctypes.c_int * len_(self._devices))(*self._devices), len_(self._devices)
            rc = self._lib.mi355fft_multi_plan_create(int(len), int(direction), self._prec, None, devs, nd, ctypes.byref(h))
            if rc != 0:
                raise FftPanic(rc, self._lib.mi355fft_last_error().decode() or self._lib.mi355fft_strerror(rc).decode())
            self._cache[key] = FftMulti(self._lib, h, self.dtype)
        return self._cache[key]

    def plan_fft_forward(self, len):
        return self.plan_fft(len, FftDirection.Forward)

    def plan_fft_inverse(self, len):
        return self.plan_fft(len, FftDirection.Inverse)


len_ = len  # the trait's `len` argument name shadows the builtin inside plan_fft


# `FftPlanner::new()` picks the best available back-end (src/plan.rs:72-94); here the only back-end is HIP.
FftPlanner = FftPlannerHip
