"""CPU checks (kernel-body emulator, tests/emu) of the round-4 host logic: the fused two-pass launch (work-item order,
dependency counters, ring slot reuse), the chunk pipeline, and the multi-device plan (row sharding, per-shard workers, the
trait's validation semantics through it).  The emulator runs the fused kernel's work items in index order and CHECKS every
dependency the device kernel would wait on: an item order that could deadlock or read an unwritten slot sets the error word."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import compare_vectors, numpy_fft, random_signal, rel_l2

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s"])
    from rustfft_amd import _native

    return _native.load(os.path.join(EMU_DIR, "libmi355fft_emu.so"))


def _planner(lib, dtype=np.complex64):
    import rustfft_amd

    return rustfft_amd.FftPlannerHip(dtype, lib=lib)


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("dtype,log2n,maxr,batch", [(np.complex64, 22, 256, 2), (np.complex64, 23, 0, 1), (np.complex128, 23, 0, 1)])
def test_fused_units_of_three_pass_plans(emu_lib, oracle, dtype, log2n, maxr, batch):
    """Three-pass plans: passes 0 and 1 in one fused launch over UNITS (closed groups of R1 first-pass tiles and F0 R0 / F1 second-pass
    tiles, kernels_params.h; a ring slot holds one unit in compact form, a tile reads and writes at different tile indices), the third
    pass as its own launch.  2^22 as 256 x 256 x 64 (two units per transform), 2^23 as 256 x 256 x 128 (four; Complex<f64>: eight):
    bit-identical to the three-launch plan in all three modes and both directions, the emulator's dependency check clean."""
    n = 1 << log2n

    def build(direction, fused):
        f = _with_env({"MI355FFT_MAXR": maxr} if maxr else {}, lambda: _planner(emu_lib, dtype).plan_fft(n, direction))
        f.set_fused(fused)
        return f

    x = random_signal(n * batch, dtype)
    for direction in (0, 1):
        ref, fus = build(direction, 0), build(direction, 1)
        assert fus.is_fused() and fus.describe().startswith("fused{") and "} -> k2later" in fus.describe() and not ref.is_fused(), fus.describe()
        a, b = x.copy(), x.copy()
        ref.process(a)
        fus.process(b)
        assert np.array_equal(a, b) and fus.fused_status() == 0, (log2n, direction)
        src, out = x.copy(), np.empty_like(x)
        fus.process_immutable_with_scratch(src, out)
        assert np.array_equal(out, a) and np.array_equal(src, x), (log2n, direction, "immutable")
        fus.process_outofplace_with_scratch(src, out)
        assert np.array_equal(out, a) and fus.fused_status() == 0, (log2n, direction, "out of place")
        if direction == 0 and log2n == 22:
            want = x[:n].copy()
            oracle.plan(dtype, n, 0).process(want)
            assert compare_vectors(want, b[:n])


def test_planner_takes_the_split_with_a_default_fused_kernel(emu_lib):
    """Two-pass plans whose balanced split has no default fused kernel while another split has one take that one (plan.cpp
    choose_macro_radices): Complex<f32> 2^17 = 256 x 512, 2^18 = 256 x 1024 and 2^21 = 1024 x 2048; three-pass 2^23 / 2^24 fuse their first two passes
    by default; Complex<f64> 2^15 and 2^19 (whose fused kernel runs the plan's 8-column later tile as 16 columns) are fused by default."""
    p32, p64 = _planner(emu_lib), _planner(emu_lib, np.complex128)
    assert p32.plan_fft_forward(1 << 17).describe() == "fused{k2first<256, 16, 16, 16>xF32 | k2later<512, 16, 8, 8, 8>xF32t}"
    assert p32.plan_fft_forward(1 << 18).describe() == "fused{k2first<256, 16, 16, 16>xF32 | k2later<1024, 32, 8, 8, 16>xF16t}"
    assert p32.plan_fft_forward(1 << 23).describe().startswith("fused{k2first<256, 16, 16, 16>xF32 | k2later<256, 16, 16, 16>xF32} -> ")
    assert p32.plan_fft_forward(1 << 24).describe().startswith("fused{")
    assert p32.plan_fft_forward(1 << 21).describe() == "fused{k2first<1024, 32, 8, 8, 16>xF16t | k2later<2048, 128, 8, 16, 16>xF16p2}"
    assert not p32.plan_fft_forward(1 << 22).is_fused() and not p32.plan_fft_forward(1 << 25).is_fused()
    big = p32.plan_fft_forward(1 << 26)  # 512 x 512 x 256: a fused kernel exists for the first two passes, the planner's choice is "no"
    for mode, want in ((1, True), (-1, False), (0, False)):
        big.set_fused(mode)
        assert big.is_fused() == want, mode
    mid = p32.plan_fft_forward(1 << 23)
    for mode, want in ((0, False), (-1, True)):
        mid.set_fused(mode)
        assert mid.is_fused() == want, mode
    for k in (15, 16, 17, 18, 19, 20, 21, 23, 24):
        assert p64.plan_fft_forward(1 << k).is_fused(), k
    assert not p64.plan_fft_forward(1 << 22).is_fused()
    x = random_signal((1 << 19) * 24, np.complex128)
    fus, two = p64.plan_fft_forward(1 << 19), _planner(emu_lib, np.complex128).plan_fft_forward(1 << 19)
    two.set_fused(0)
    a, b = x.copy(), x.copy()
    fus.process(a)
    two.process(b)
    assert np.array_equal(a, b) and fus.fused_status() == 0


@pytest.mark.parametrize("log2n,lag,slots,batch", [(16, 1, 2, 5), (16, 2, 5, 7), (17, 1, 3, 4), (18, 1, 2, 3), (19, 1, 2, 3), (21, 1, 2, 2)])
def test_fused_two_pass_launch_matches_two_launches(emu_lib, oracle, log2n, lag, slots, batch):
    """Every fused kernel against the two-launch plan of the same length, with a ring so small that slots are reused (a first-pass
    tile has to find its slot read, a second-pass tile its slot written): bit-identical results, error word 0.  (2^21: the fused
    kernel runs the later pass on 8-column tiles of 64 threads per column where the plan's own tile has 128 -- another chain of
    inter-pass factor products, so equal up to rounding only.)"""
    n = 1 << log2n

    def same(u, v):
        # 2^17, 2^18, 2^21: the split exists for its fused kernel only, and a plan that does not fuse runs the balanced split (Plan::unfused_alt,
        # round 5) -- other tile heights, another chain of inter-pass factor products: equal up to rounding
        return np.array_equal(u, v) if log2n in (16, 19, 20) else float(np.abs(u - v).max()) <= 4e-7 * float(np.abs(v).max())

    ref = _planner(emu_lib).plan_fft_forward(n)
    ref.set_fused(0)
    fus = _with_env({"MI355FFT_FUSE_LAG": lag, "MI355FFT_FUSE_SLOTS": slots}, lambda: _planner(emu_lib).plan_fft_forward(n))
    fus.set_fused(1)
    assert fus.is_fused() and "fused{" in fus.describe() and not ref.is_fused()
    x = random_signal(n * batch, np.complex64)
    a, b = x.copy(), x.copy()
    ref.process(a)
    fus.process(b)
    assert same(a, b)
    assert fus.fused_status() == 0
    want = x[:n].copy()
    oracle.plan(np.complex64, n, 0).process(want)
    assert compare_vectors(want, b[:n])
    # the other two API modes run the same launch: the input is never clobbered
    y = np.empty_like(x)
    fus.process_immutable_with_scratch(x, y)
    assert np.array_equal(y, b)
    x2, y2 = x.copy(), np.empty_like(x)
    fus.process_outofplace_with_scratch(x2, y2)
    assert np.array_equal(y2, b) and fus.fused_status() == 0


def test_fused_default_ring_and_small_batches(emu_lib):
    """The default lag / ring (from the number of resident workgroups) at config 2's length; a batch smaller than the ring runs
    as two launches (nothing to overlap), a larger one fused -- same results."""
    n = 1 << 20
    fus = _planner(emu_lib).plan_fft_forward(n)
    assert fus.is_fused()  # the planner's default at 2^20 (measured: profiles/r4)
    ref = _planner(emu_lib).plan_fft_forward(n)
    ref.set_fused(0)
    for batch in (1, 18):  # (the default ring at 2^20 has 16 slots)
        x = random_signal(n * batch, np.complex64)
        a, b = x.copy(), x.copy()
        ref.process(a)
        fus.process(b)
        assert np.array_equal(a, b) and fus.fused_status() == 0
    inv = _planner(emu_lib).plan_fft_inverse(n)
    x = random_signal(n * 17, np.complex64)
    y = x.copy()
    fus.process(y)
    inv.process(y)
    assert rel_l2(y / n, x) < 2e-6 and inv.fused_status() == 0


@pytest.mark.parametrize("mode", [1, 2])
def test_chunk_pipeline_matches_full_workspace(emu_lib, mode):
    """Chunks through a ring of intermediate buffers (tuning builds: MI355FFT_PIPE), two- and three-pass plans, all API modes."""
    for n, batch in ((1 << 16, 9), (1 << 23, 3)):
        ref = _planner(emu_lib).plan_fft_forward(n)
        ref.set_fused(0)
        pip = _with_env({"MI355FFT_PIPE": mode, "MI355FFT_PIPE_MIB": 1 if n < (1 << 20) else 128, "MI355FFT_FUSE": 8}, lambda: _planner(emu_lib).plan_fft_forward(n))
        x = random_signal(n * batch, np.complex64)
        a, b = x.copy(), x.copy()
        ref.process(a)
        pip.process(b)
        assert np.array_equal(a, b)
        y = np.empty_like(x)
        pip.process_immutable_with_scratch(x, y)
        assert np.array_equal(y, a)


# ---- multi-device plan -------------------------------------------------------------------------------------------------------
def _multi(lib, n, direction, devices, dtype=np.complex64):
    import rustfft_amd

    return rustfft_amd.FftPlannerHipMulti(dtype, devices=devices, lib=lib).plan_fft(n, direction)


def test_shard_rows_is_the_law_of_sharding_py(emu_lib):
    from rustfft_amd.sharding import shard_rows

    first, rows = ctypes.c_size_t(), ctypes.c_size_t()
    for batch in (0, 1, 2, 7, 8, 9, 1000, 1024, 8192):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                assert emu_lib.mi355fft_shard_rows(batch, world, r, ctypes.byref(first), ctypes.byref(rows)) == 0
                lo, hi = shard_rows(batch, world, r)
                assert (first.value, first.value + rows.value) == (lo, hi)
                covered += rows.value
            assert covered == batch
    assert emu_lib.mi355fft_shard_rows(4, 2, 2, ctypes.byref(first), ctypes.byref(rows)) != 0


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_multi_device_host_slices(emu_lib, oracle, dtype, monkeypatch):
    """The three trait methods through a plan over two (fake) devices and over the SAME device twice: results equal the
    one-device plan's, whatever the batch (ragged shards, an empty shard, a single row)."""
    from rustfft_amd import FftPanic

    monkeypatch.setenv("MI355_EMU_DEVICES", "3")
    for n in (1200 if dtype == np.complex128 else 1009, 1 << 16):
        one = _planner(emu_lib, dtype).plan_fft_forward(n)
        for devices in ([0, 1], [2, 2], [0, 1, 2]):
            multi = _multi(emu_lib, n, 0, devices, dtype)
            assert multi.shards() == len(devices) and multi.devices() == devices and multi.len() == n
            for batch in (1, 2, 5):
                x = random_signal(n * batch, dtype)
                want = x.copy()
                one.process(want)
                a = x.copy()
                multi.process(a)
                assert np.array_equal(a, want)
                y = np.empty_like(x)
                multi.process_immutable_with_scratch(x, y)
                assert np.array_equal(y, want)
                x2, y2 = x.copy(), np.empty_like(x)
                multi.process_outofplace_with_scratch(x2, y2)
                assert np.array_equal(y2, want)
            ref = x[:n].copy()
            oracle.plan(dtype, n, 0).process(ref)
            assert compare_vectors(ref, want[:n])
        # validation semantics of src/common.rs:13-104 through the multi-device entry points
        multi = _multi(emu_lib, n, 0, [0, 1], dtype)
        x = random_signal(n * 3 + 5, dtype)
        want = x.copy()
        with pytest.raises(FftPanic, match="multiple of FFT length"):
            one.process(want)
        got = x.copy()
        with pytest.raises(FftPanic, match="multiple of FFT length"):
            multi.process(got)
        assert np.array_equal(got, want)  # the complete chunks WERE transformed before the panic (array_utils.rs:164-176)
        with pytest.raises(FftPanic, match="too small"):
            multi.process(random_signal(n - 1, dtype))
        with pytest.raises(FftPanic, match="same length"):
            multi.process_outofplace_with_scratch(random_signal(n, dtype), np.empty(2 * n, dtype))
        multi.process(np.empty(0, dtype))  # an empty buffer is accepted
    with pytest.raises(FftPanic):
        _multi(emu_lib, 64, 0, [0, 7], dtype)  # no such device


class _Dev:
    """numpy array standing in for a device tensor (the emulator's device memory is host memory)."""

    def __init__(self, a):
        self.a = a

    def data_ptr(self):
        return self.a.ctypes.data

    def numel(self):
        return self.a.size


def test_multi_device_resident_shards_and_edges(emu_lib, monkeypatch):
    monkeypatch.setenv("MI355_EMU_DEVICES", "2")
    n, batch = 1 << 16, 5
    multi = _multi(emu_lib, n, 0, [0, 1])
    one = _planner(emu_lib).plan_fft_forward(n)
    x = random_signal(n * batch, np.complex64)
    want = x.copy()
    one.process(want)
    shards = []
    for g in range(2):
        lo, rows = multi.shard_rows(batch, g)
        shards.append(np.zeros(rows * n, np.complex64))
    root = _Dev(x.copy())
    root.device = None
    multi.scatter(root, [_Dev(s) for s in shards], root_device=0)
    assert np.array_equal(np.concatenate(shards), x)
    multi.process([_Dev(s) for s in shards])
    multi.synchronize()
    assert np.array_equal(np.concatenate(shards), want)
    out = _Dev(np.zeros_like(x))
    multi.gather([_Dev(s) for s in shards], out, root_device=0)
    assert np.array_equal(out.a, want)
    # out of place / immutable on device-resident shards
    ins = [x[: 3 * n].copy(), x[3 * n:].copy()]
    outs = [np.zeros(3 * n, np.complex64), np.zeros(2 * n, np.complex64)]
    multi.process_immutable_with_scratch([_Dev(a) for a in ins], [_Dev(a) for a in outs])
    assert np.array_equal(np.concatenate(outs), want) and np.array_equal(np.concatenate(ins), x)
    with pytest.raises(ValueError):
        multi.process([_Dev(shards[1]), _Dev(shards[0])])  # the shards must hold their own rows


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_general_split_prefers_full_tiles(emu_lib, oracle, dtype):
    """Round 4 planner rule (plan.cpp choose_general_radices): among the splits with the fewest passes, the one that wastes the
    fewest tile columns, not the most balanced one -- 4225 = 65 x 65 would be ONE 128-column tile per transform with 63 columns
    masked (Complex<f32> tile widths).  A sample of the lengths whose split changes, all API modes against the oracle."""
    from helpers import check_fft_algorithm

    planner = _planner(emu_lib, dtype)
    # (round 5: 13-smooth lengths up to 16384 -- Complex<f32>: most up to 32768 -- are whole-row kernels now, so the rule matters above:
    # 33124 = 2^2 7^2 13^2 is 182 x 182 balanced, i.e. 6 32-column tiles of which the last is a sixth full; 637 x 52 wastes nothing)
    for n in (32928, 33033, 33124, 33275, 33750, 33957, 34125, 34300):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k2gfirst<") and "->" in fft.describe(), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
    if dtype == np.complex64:
        assert planner.plan_fft_forward(33124).describe().startswith("k2gfirst<637,"), planner.plan_fft_forward(33124).describe()
        balanced = _with_env({"MI355FFT_SPLIT_BALANCED": 1}, lambda: _planner(emu_lib, dtype).plan_fft_forward(33124).describe())
        assert balanced.startswith("k2gfirst<182,"), balanced


# ---- round 5: a fused launch that gives up a wait cannot be missed; failures say why -----------------------------------------------
def _dev_call(fft, arr, batch):
    """mi355fft_process_inplace_dev on emulator "device" memory (= host memory), default stream."""
    return fft._lib.mi355fft_process_inplace_dev(fft._h, arr.ctypes.data_as(ctypes.c_void_p), batch, None)


def test_fused_giveup_fails_the_next_device_call_and_is_reported_once(emu_lib):
    """launch.h k2f_wait: a wait that gives up raises the (plan, stream) slot's STICKY word; no launch clears it.  The asynchronous entry
    points cannot know at enqueue time -- the NEXT device call on that plan and stream fails with the reason instead of running
    (include/mi355fft.h; the reference's contract: src/lib.rs:184, an Fft is never silently wrong).  MI355FFT_FUSED_GIVEUP makes the
    emulator leave the word a gave-up tile leaves."""
    n, batch = 1 << 16, 136  # (the default ring at 2^16 has 128 slots: smaller batches run as two launches)
    fus = _planner(emu_lib).plan_fft_forward(n)
    assert fus.is_fused()
    x = random_signal(n * batch, np.complex64)
    want = x.copy()
    ref = _planner(emu_lib).plan_fft_forward(n)
    ref.set_fused(0)
    ref.process(want)
    a = x.copy()
    assert _with_env({"MI355FFT_FUSED_GIVEUP": 1}, lambda: _dev_call(fus, a, batch)) == 0  # enqueued fine; the give-up happens "on the device"
    b = x.copy()
    rc = _dev_call(fus, b, batch)
    msg = emu_lib.mi355fft_last_error().decode()
    assert rc == 8 and "gave up waiting for a dependency" in msg and "INVALID" in msg, (rc, msg)
    assert np.array_equal(b, x), "the failing call must not have run"
    assert _dev_call(fus, b, batch) == 0 and np.array_equal(b, want)  # reported once: the plan works again
    assert fus.fused_status() == 0
    # the explicit query reports (and clears) it as well
    _with_env({"MI355FFT_FUSED_GIVEUP": 1}, lambda: _dev_call(fus, x.copy(), batch))
    assert fus.fused_status() == 1 and fus.fused_status() == 0
    # and so does destroying the plan when nobody asked
    h = ctypes.c_void_p()
    assert emu_lib.mi355fft_plan_create(n, 0, 32, ctypes.byref(h)) == 0
    c = x.copy()
    assert _with_env({"MI355FFT_FUSED_GIVEUP": 1}, lambda: emu_lib.mi355fft_process_inplace_dev(h, c.ctypes.data_as(ctypes.c_void_p), batch, None)) == 0
    assert emu_lib.mi355fft_plan_destroy(h) == 8 and "gave up" in emu_lib.mi355fft_last_error().decode()


@pytest.mark.parametrize("chunk_kib,batch", [(0, 160), (65536, 288)])
def test_fused_giveup_on_host_slices_reruns_the_rows(emu_lib, chunk_kib, batch):
    """The host-slice path (the literal drop-in for `&mut [Complex<T>]`) checks the word when a chunk's launches have completed and, when it
    is set, transforms the chunk's rows again from the caller's (still intact) input with one launch per pass: the call SUCCEEDS with
    correct results in all three API modes, one chunk or a pipeline of chunks, and leaves no stale word behind."""
    n = 1 << 16
    fus = _planner(emu_lib).plan_fft_forward(n)
    ref = _planner(emu_lib).plan_fft_forward(n)
    ref.set_fused(0)
    x = random_signal(n * batch, np.complex64)
    want = x.copy()
    ref.process(want)
    env = {"MI355FFT_FUSED_GIVEUP": 1}
    if chunk_kib:
        env["MI355FFT_HOST_CHUNK_KIB"] = chunk_kib  # 64 MiB: chunks of 128, 128 and 32 rows -- two fused (the ring has 128 slots), the last as two launches

    def run():
        a = x.copy()
        fus.process(a)
        assert np.array_equal(a, want), "in place"
        src, out = x.copy(), np.empty_like(x)
        fus.process_immutable_with_scratch(src, out)
        assert np.array_equal(out, want) and np.array_equal(src, x), "immutable"
        fus.process_outofplace_with_scratch(src, out)
        assert np.array_equal(out, want), "out of place"

    _with_env(env, run)
    assert fus.fused_status() == 0
    run()  # and without the injected give-up the same plan still runs fused and right


def test_multi_synchronize_reports_a_fused_giveup(emu_lib):
    n, batch = 1 << 16, 272
    mp = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, 0)
    assert emu_lib.mi355fft_multi_plan_create(n, 0, 32, None, devs, 2, ctypes.byref(mp)) == 0
    x = random_signal(n * batch, np.complex64)
    halves = [x[: n * 136].copy(), x[n * 136:].copy()]
    ptrs = (ctypes.c_void_p * 2)(*[h.ctypes.data_as(ctypes.c_void_p) for h in halves])
    assert _with_env({"MI355FFT_FUSED_GIVEUP": 1}, lambda: emu_lib.mi355fft_multi_process_inplace_dev(mp, ptrs, batch, None)) == 0
    assert emu_lib.mi355fft_multi_synchronize(mp, None) == 8 and "gave up" in emu_lib.mi355fft_last_error().decode()
    assert emu_lib.mi355fft_multi_synchronize(mp, None) == 0
    assert emu_lib.mi355fft_multi_plan_destroy(mp) == 0


def test_a_failed_transform_says_which_step_and_why(emu_lib):
    """`execution failed` dropped the status.  A refused workspace allocation now names the allocation, its size and what
    the device had left; the status in words leads the message."""
    n, batch = 1 << 17, 4
    fft = _planner(emu_lib).plan_fft_forward(n)
    fft.set_fused(0)
    x = random_signal(n * batch, np.complex64)
    rc = _with_env({"MI355_EMU_MAX_ALLOC": 1 << 20}, lambda: _dev_call(fft, x, batch))
    msg = emu_lib.mi355fft_last_error().decode()
    assert rc == 9 and msg.startswith("out of device memory: workspace of a multi-pass plan: device allocation of %d bytes failed" % (n * batch * 8)) and "free" in msg, (rc, msg)
    assert _dev_call(fft, x, batch) == 0  # and the plan is usable afterwards


def test_shard_workers_are_bound_to_their_gpus_numa_node(emu_lib, monkeypatch, tmp_path):
    """eight staging pipelines must not cross sockets.  A fake two-socket node in a sysfs tree (the emulator's device d
    sits at PCI 0000:<d>1:00.0): device 0 on node 0, device 1 on node 1, device 2 with no node (-1).  The worker of a shard binds ITSELF to
    its node's cores (here: whatever cores this machine really has, split in two), reports it, and still transforms its rows; the calling
    thread's affinity is untouched."""
    import rustfft_amd

    cpus = sorted(os.sched_getaffinity(0))
    halves = [cpus[: max(1, len(cpus) // 2)], cpus[max(1, len(cpus) // 2):] or cpus[:1]]
    for d, node in ((0, "0"), (1, "1"), (2, "-1")):
        pdir = tmp_path / "bus" / "pci" / "devices" / ("0000:%x1:00.0" % d)
        pdir.mkdir(parents=True)
        (pdir / "numa_node").write_text(node + "\n")
    for k in (0, 1):
        ndir = tmp_path / "devices" / "system" / "node" / ("node%d" % k)
        ndir.mkdir(parents=True)
        (ndir / "cpulist").write_text(",".join(str(c) for c in halves[k]) + "\n")
    monkeypatch.setenv("MI355_EMU_DEVICES", "3")
    monkeypatch.setenv("MI355FFT_SYSFS_ROOT", str(tmp_path))
    assert rustfft_amd.device_cpulist(0, lib=emu_lib) == ",".join(str(c) for c in halves[0])
    assert rustfft_amd.device_cpulist(1, lib=emu_lib) == ",".join(str(c) for c in halves[1])
    assert rustfft_amd.device_cpulist(2, lib=emu_lib) == ""
    before = os.sched_getaffinity(0)
    n, batch = 1200, 9
    multi = _multi(emu_lib, n, 0, [0, 1, 2], np.complex128)
    assert [multi.shard_pinned(g) for g in range(3)] == [True, True, False]
    x = random_signal(n * batch, np.complex128)
    want = x.copy()
    _planner(emu_lib, np.complex128).plan_fft_forward(n).process(want)
    multi.process(x)
    assert np.array_equal(x, want)
    assert os.sched_getaffinity(0) == before


def test_a_split_taken_for_its_fused_kernel_keeps_the_balanced_split_for_unfused_calls(emu_lib, oracle):
    """Complex<f32> 2^18 is planned as 256 x 1024 because that pair has a default fused kernel; whenever the fused launch cannot
    run (a batch below the ring's slot count -- the interactive case --, mi355fft_plan_set_fused(plan, 0)) the balanced 512 x 512 split is the
    faster two-launch plan (5.92 against 6.33 ms at 2^18 x 256).  The plan keeps both pass sets: fused calls run the re-split, every other
    call the balanced split; describe() says which."""
    n = 1 << 18
    fft = _planner(emu_lib).plan_fft_forward(n)
    assert fft.describe() == "fused{k2first<256, 16, 16, 16>xF32 | k2later<1024, 32, 8, 8, 16>xF16t}"
    x = random_signal(n * 2, np.complex64)
    want = x[:n].copy()
    oracle.plan(np.complex64, n, 0).process(want)
    small = x.copy()
    fft.process(small)  # two transforms: fewer than the ring has slots -> the balanced split as two launches
    assert compare_vectors(want, small[:n])
    fft.set_fused(0)
    assert fft.describe().startswith("k2first<512,") and "k2later<512," in fft.describe(), fft.describe()
    again = x.copy()
    fft.process(again)
    assert np.array_equal(again, small)  # the same kernels ran both times
    fft.set_fused(-1)
    assert fft.describe().startswith("fused{k2first<256,")
    for log2n in (17, 21):
        f = _planner(emu_lib).plan_fft_forward(1 << log2n)
        fused_desc = f.describe()
        f.set_fused(0)
        assert fused_desc.startswith("fused{") and not f.describe().startswith("fused{") and f.describe().replace(" -> ", " | ") != fused_desc[6:-1], (fused_desc, f.describe())
