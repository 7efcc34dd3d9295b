"""Parity tests proper (-m gpu): the HIP path, called through the C ABI (include/mi355fft.h) via the
rustfft_amd host mirror, against the oracle (oracle/rustfft_scalar.hpp, RustFFT's scalar path) on the same
seeded inputs.  Tolerance = the reference's own (tests/accuracy.rs:30-37): mean |a - b| < 0.1 on inputs
re, im ~ U[0,10); in addition a relative-L2 bound vs a float64 reference is asserted (SURVEY App. C:
O(eps log2 N): 5e-6 for f32, 1e-13 for f64)."""
import os
import threading

import numpy as np
import pytest

from helpers import check_fft_algorithm, compare_vectors, mean_abs_err, numpy_fft, random_signal, rel_l2, zero_mean_signal

pytestmark = pytest.mark.gpu

REL = {np.dtype(np.complex64): 5e-6, np.dtype(np.complex128): 1e-13}


@pytest.fixture(scope="module")
def planners():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    import rustfft_amd

    return {np.dtype(np.complex64): rustfft_amd.FftPlanner(np.complex64), np.dtype(np.complex128): rustfft_amd.FftPlanner(np.complex128)}


def _native_loaded():
    return any("libmi355fft.so" in line for line in open("/proc/self/maps"))


def test_native_library_is_the_one_running(planners):
    assert _native_loaded(), "the HIP extension must be loaded in-process (no fallback path exists)"
    import rustfft_amd

    assert rustfft_amd.device_count() >= 1


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_check_fft_algorithm_pow2(planners, oracle, dtype):
    """src/test_utils.rs:70-209 on every planned power of two up to 2^18 (host-slice entry points:
    process, process_with_scratch, process_outofplace_with_scratch, process_immutable_with_scratch)."""
    planner = planners[np.dtype(dtype)]
    for n in [0, 1] + [1 << p for p in range(1, 19)]:
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3 if n < (1 << 16) else 2)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_relative_error_vs_float64(planners, dtype):
    planner = planners[np.dtype(dtype)]
    for p in (4, 7, 10, 12, 13, 16, 20, 21, 22):
        n = 1 << p
        for d in (0, 1):
            x = zero_mean_signal(n * 2, dtype)
            y = x.copy()
            planner.plan_fft(n, d).process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d)


def test_config1_n1024_plumbing(planners, oracle):
    """BASELINE config 1: single forward FFT, N = 1024, Complex<f32>."""
    x = random_signal(1024, np.complex64)
    y = x.copy()
    planners[np.dtype(np.complex64)].plan_fft_forward(1024).process(y)
    want = x.copy()
    oracle.plan(np.complex64, 1024, 0).process(want)
    assert compare_vectors(want, y) and mean_abs_err(want, y) < 1e-3


def test_config2_device_resident_2p20(planners, oracle):
    """BASELINE config 2 shape (N = 2^20 f32, forward + inverse) on HBM-resident data: 8 sampled rows vs the
    oracle's Radix4 + numpy c128, and the round trip ifft(fft(x)) = N x on every row of a 64-row batch."""
    import torch

    n, batch = 1 << 20, 64
    planner = planners[np.dtype(np.complex64)]
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 2)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    y = x.clone()
    fwd.process(y)
    torch.cuda.synchronize()
    ref = oracle.plan(np.complex64, n, 0)
    for row in (0, 1, 7, 13, 31, 32, 62, 63):
        xr = x[row * n:(row + 1) * n].cpu().numpy()
        got = y[row * n:(row + 1) * n].cpu().numpy()
        want = xr.copy()
        ref.process(want)
        assert compare_vectors(want, got), row
        assert rel_l2(got, numpy_fft(xr, n, False)) < REL[np.dtype(np.complex64)], row
    inv.process(y)
    torch.cuda.synchronize()
    err = (y / n - x).abs().mean().item()
    assert err < 1e-4, err
    # out-of-place and immutable device entry points agree bit-for-bit with the in-place one
    a = x.clone()
    out = torch.empty_like(x)
    fwd.process_immutable_with_scratch(a, out)
    assert torch.equal(a, x)
    z = x.clone()
    fwd.process(z)
    assert torch.equal(out, z)
    out2 = torch.empty_like(x)
    fwd.process_outofplace_with_scratch(a, out2)
    assert torch.equal(out2, z)


def test_full_size_properties_2p22(planners):
    """N = 2^22 (config 5's length, two 2048-row passes): impulse -> complex exponential, constant -> delta,
    Parseval, on a 16-row batch."""
    import torch

    n, batch = 1 << 22, 16
    planner = planners[np.dtype(np.complex64)]
    fwd = planner.plan_fft_forward(n)
    x = torch.zeros(batch * n, dtype=torch.complex64, device="cuda")
    xs = x.view(batch, n)
    xs[0, 5] = 1.0
    xs[1, :] = 1.0
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    torch.view_as_real(xs[2:]).uniform_(-1.0, 1.0, generator=g)
    energy_in = (xs[2:].abs().double() ** 2).sum(dim=1)
    fwd.process(x)
    torch.cuda.synchronize()
    k = torch.arange(n, device="cuda", dtype=torch.float64)
    want = torch.exp(-2j * np.pi * 5 * k / n)
    assert (xs[0].to(torch.complex128) - want).abs().max().item() < 1e-5
    assert abs(xs[1, 0].item() - n) < 1e-1 * 1e-3 * n and xs[1, 1:].abs().max().item() < 1e-2
    energy_out = (xs[2:].abs().double() ** 2).sum(dim=1) / n
    assert torch.allclose(energy_in, energy_out, rtol=1e-5)


def test_chunked_workspace_identical(planners):
    import torch

    n, batch = 1 << 17, 37
    fft = planners[np.dtype(np.complex64)].plan_fft_forward(n)
    x = torch.from_numpy(random_signal(n * batch, np.complex64)).cuda()
    a = x.clone()
    fft.process(a)
    fft.set_chunk_batch(5)
    b = x.clone()
    fft.process(b)
    fft.set_chunk_batch(0)
    assert torch.equal(a, b)


def test_error_paths(planners):
    import rustfft_amd

    f = planners[np.dtype(np.complex64)].plan_fft_forward(256)
    with pytest.raises(rustfft_amd.FftPanic, match="Provided FFT buffer was too small. Expected len = 256, got len = 10"):
        f.process(np.zeros(10, np.complex64))
    with pytest.raises(rustfft_amd.FftPanic, match="must be a multiple of FFT length"):
        f.process(np.zeros(300, np.complex64))
    with pytest.raises(rustfft_amd.FftPanic, match="must have the same length"):
        f.process_immutable_with_scratch(np.zeros(256, np.complex64), np.zeros(512, np.complex64))
    f.process(np.zeros(0, np.complex64))
    import torch

    with pytest.raises(rustfft_amd.FftPanic, match="must be a multiple of FFT length"):
        f.process(torch.zeros(300, dtype=torch.complex64, device="cuda"))
    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        planners[np.dtype(np.complex64)].plan_fft_forward((1 << 30) + 1)  # would need a 2^31-point inner transform


def test_concurrent_process_on_one_plan(planners, oracle):
    """examples/concurrency.rs:9-30: one Arc<dyn Fft> shared by several threads, each with its own buffer."""
    n = 4096
    fft = planners[np.dtype(np.complex64)].plan_fft_forward(n)
    ref = oracle.plan(np.complex64, n, 0)
    results, inputs = {}, {t: random_signal(n * 3, np.complex64, seed=100 + t) for t in range(4)}

    def work(t):
        buf = inputs[t].copy()
        for _ in range(3):
            b2 = inputs[t].copy()
            fft.process(b2)
            buf = b2
        results[t] = buf

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for t in range(4):
        want = inputs[t].copy()
        ref.process(want)
        assert compare_vectors(want, results[t])


@pytest.mark.parametrize("log2n,same_stream", [(17, True), (17, False), (20, True), (20, False)])
def test_concurrent_device_calls_share_a_multipass_plan(planners, oracle, log2n, same_stream):
    """examples/concurrency.rs:9-30 on the DEVICE path: four host threads share one two-pass plan (plan-owned HBM
    workspace) with their own HBM buffers and DIFFERENT batch sizes -- once all on torch's default stream (their pass
    sequences must not interleave and a workspace growth must not free a buffer another call still uses), once on
    per-thread streams (separate workspaces, true overlap).  Every row is compared with the oracle."""
    import torch

    n = 1 << log2n
    fft = planners[np.dtype(np.complex64)].plan_fft_forward(n)
    assert len(fft.kernel_names()) >= 2, fft.describe()
    fft.trim_workspaces()  # start from no workspace so that the growth path is exercised under contention
    ref = oracle.plan(np.complex64, n, 0)
    batches = [1, 3, 2, 5]
    inputs = {t: random_signal(n * batches[t], np.complex64, seed=500 + t) for t in range(4)}
    results, errors = {}, []
    start = threading.Barrier(4)

    def work(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.current_stream() if same_stream else torch.cuda.Stream()
            with torch.cuda.stream(stream):
                x = torch.from_numpy(inputs[t]).cuda()
                start.wait()
                last = None
                for _ in range(6):
                    y = x.clone()
                    fft.process(y)
                    last = y
                stream.synchronize()
                results[t] = last.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for t in range(4):
        want = inputs[t].copy()
        ref.process(want)
        assert compare_vectors(want, results[t]), (t, mean_abs_err(want, results[t]))
        assert rel_l2(results[t], numpy_fft(inputs[t], n, False)) < REL[np.dtype(np.complex64)], t
    assert fft.workspace_bytes() >= n * 8
    assert fft.trim_workspaces() >= n * 8 and fft.workspace_bytes() == 0


def test_config5_rows_vs_oracle_2p22(planners, oracle):
    """BASELINE config 5's per-GPU length (N = 2^22 f32, two 2048-row passes): sampled rows of a 6-row HBM-resident
    batch against the oracle's Radix4 and numpy complex128, both directions."""
    import torch

    n, batch = 1 << 22, 6
    planner = planners[np.dtype(np.complex64)]
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 5)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    for d in (0, 1):
        fft = planner.plan_fft(n, d)
        assert fft.describe().count("k2") == 2, fft.describe()
        y = x.clone()
        fft.process(y)
        torch.cuda.synchronize()
        ref = oracle.plan(np.complex64, n, d)
        for row in (0, 2, 3, 5):
            xr = x[row * n:(row + 1) * n].cpu().numpy()
            got = y[row * n:(row + 1) * n].cpu().numpy()
            want = xr.copy()
            ref.process(want)
            assert compare_vectors(want, got), (d, row)
            assert rel_l2(got, numpy_fft(xr, n, d == 1)) < REL[np.dtype(np.complex64)], (d, row)


def test_config2_full_batch_1024(planners, oracle):
    """BASELINE config 2 at its FULL size (N = 2^20 f32, batch 1024 = 8 GiB in HBM, forward then inverse in place):
    rows sampled across the whole batch against the oracle, and the round trip ifft(fft(x)) = N x on every row."""
    import torch

    n, batch = 1 << 20, 1024
    planner = planners[np.dtype(np.complex64)]
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 22)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    rows = (0, 1, 511, 512, 777, 1023)
    keep = {r: x[r * n:(r + 1) * n].cpu().numpy() for r in rows}
    checksum = torch.view_as_real(x).view(batch, -1).sum(dim=1, dtype=torch.float64)
    fwd.process(x)
    torch.cuda.synchronize()
    ref = oracle.plan(np.complex64, n, 0)
    for r in rows:
        got = x[r * n:(r + 1) * n].cpu().numpy()
        want = keep[r].copy()
        ref.process(want)
        assert compare_vectors(want, got), r
        assert rel_l2(got, numpy_fft(keep[r], n, False)) < REL[np.dtype(np.complex64)], r
    inv.process(x)
    torch.cuda.synchronize()
    x.mul_(1.0 / n)
    for r in rows:
        assert rel_l2(x[r * n:(r + 1) * n].cpu().numpy(), keep[r]) < 2e-6, r
    # every row: its element sum survives the round trip (a mis-addressed or skipped tile anywhere in the batch breaks it)
    after = torch.view_as_real(x).view(batch, -1).sum(dim=1, dtype=torch.float64)
    assert ((after - checksum).abs() / checksum.abs()).max().item() < 1e-5


def test_environment_cannot_change_results(planners, oracle):
    """The shipped library reads no environment variable (tuning knobs and ablation kernels exist only in
    -DMI355_TUNING builds): plans created under MI355FFT_VARIANT / MI355FFT_DBG still give the reference's results."""
    import rustfft_amd

    n = 1 << 20
    x = random_signal(n, np.complex64)
    want = x.copy()
    oracle.plan(np.complex64, n, 0).process(want)
    for var, val in (("MI355FFT_VARIANT", "5"), ("MI355FFT_VARIANT", "8"), ("MI355FFT_DBG", "1"), ("MI355FFT_MAXR", "256")):
        os.environ[var] = val
        try:
            fft = rustfft_amd.FftPlanner(np.complex64).plan_fft_forward(n)  # a fresh planner: no cached plan
            y = x.copy()
            fft.process(y)
            assert "abl" not in fft.describe() and compare_vectors(want, y), (var, val, fft.describe())
            assert rel_l2(y, numpy_fft(x, n, False)) < REL[np.dtype(np.complex64)], (var, val)
        finally:
            del os.environ[var]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("direction", [0, 1])
def test_accuracy_1_to_1000(planners, oracle, dtype, direction):
    """tests/accuracy.rs:128-187 on the GPU: planner output for every length 1..1000 vs the reference's control
    (BluesteinsAlgorithm over Radix4, tests/accuracy.rs:98-122) through the in-place, out-of-place and
    immutable entry points (tests/accuracy.rs:39-82), tolerance tests/accuracy.rs:30-37."""
    planner = planners[np.dtype(dtype)]
    for n in range(1, 1001):
        inner_len = 1
        while inner_len < 2 * n - 1:
            inner_len *= 2
        control = oracle.bluesteins(n, oracle.radix4(dtype, inner_len, direction))
        fft = planner.plan_fft(n, direction)
        assert fft.len() == n and int(fft.fft_direction()) == direction
        x = random_signal(n, dtype)
        ctrl = x.copy()
        control.process_with_scratch(ctrl, np.zeros(control.get_inplace_scratch_len(), dtype=dtype))
        a = x.copy()
        fft.process_with_scratch(a, np.zeros(fft.get_inplace_scratch_len(), dtype=dtype))
        i2, b = x.copy(), x.copy()
        fft.process_outofplace_with_scratch(i2, b, np.zeros(0, dtype=dtype))
        c = x.copy()
        fft.process_immutable_with_scratch(x, c, np.zeros(0, dtype=dtype))
        assert compare_vectors(ctrl, a) and compare_vectors(ctrl, b) and compare_vectors(ctrl, c), n


def test_config3_n1200_f64(planners, oracle):
    """BASELINE config 3: N = 1200 Complex<f64>, batch 65536 on HBM-resident data; all four API modes on a
    3-row slice (mirrors check_fft_algorithm) and 64 sampled rows of the full batch vs the oracle's
    RadixN{[5,5,2], Butterfly24} recipe."""
    import torch

    n, batch = 1200, 65536
    planner = planners[np.dtype(np.complex128)]
    fft = planner.plan_fft_forward(n)
    assert "k1<1200" in fft.describe()
    check_fft_algorithm(fft, n, 0, reference=oracle.plan(np.complex128, n, 0))
    check_fft_algorithm(planner.plan_fft_inverse(n), n, 1, reference=oracle.plan(np.complex128, n, 1))
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 3)
    x = torch.empty(batch * n, dtype=torch.complex128, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    y = x.clone()
    fft.process(y)
    torch.cuda.synchronize()
    ref = oracle.plan(np.complex128, n, 0)
    rows = np.random.default_rng(3).choice(batch, 64, replace=False)
    for r in rows:
        xr = x[r * n:(r + 1) * n].cpu().numpy()
        want = xr.copy()
        ref.process(want)
        got = y[r * n:(r + 1) * n].cpu().numpy()
        assert compare_vectors(want, got) and rel_l2(got, want) < 1e-13, r
    planner.plan_fft_inverse(n).process(y)
    assert ((y / n - x).abs().max().item()) < 1e-10


@pytest.mark.parametrize("n,tag", [(1009, "rader<1008"), (1019, "bluestein<2048")])
def test_config4_prime_sizes_f32(planners, oracle, n, tag):
    """BASELINE config 4: N = 1009 (Rader; the reference plans RadersAlgorithm over RadixN{[7,6],B24}) and the
    complementary Bluestein prime 1019, Complex<f32>, batch 2^20 on HBM-resident data; 64 sampled rows vs the
    oracle, plus the round trip on all rows."""
    import torch

    batch = 1 << 20
    planner = planners[np.dtype(np.complex64)]
    fft = planner.plan_fft_forward(n)
    assert tag in fft.describe()
    check_fft_algorithm(fft, n, 0, reference=oracle.plan(np.complex64, n, 0), n=5)
    check_fft_algorithm(planner.plan_fft_inverse(n), n, 1, reference=oracle.plan(np.complex64, n, 1), n=5)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 4)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    y = x.clone()
    fft.process(y)
    torch.cuda.synchronize()
    ref = oracle.plan(np.complex64, n, 0)
    rows = np.random.default_rng(4).choice(batch, 64, replace=False)
    for r in rows:
        xr = x[r * n:(r + 1) * n].cpu().numpy()
        want = xr.copy()
        ref.process(want)
        got = y[r * n:(r + 1) * n].cpu().numpy()
        assert compare_vectors(want, got), r
        assert rel_l2(got, numpy_fft(xr, n, False)) < REL[np.dtype(np.complex64)], r
    planner.plan_fft_inverse(n).process(y)
    assert (y / n - x).abs().mean().item() < 1e-4


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_lengths_beyond_one_workgroup(planners, oracle, dtype):
    """Non-powers of two above 4096 (multi-kernel Bluestein): primes the reference plans as Rader / Bluestein, a
    smooth composite (RadixN on the CPU), a product of two large primes (MixedRadix on the CPU)."""
    planner = planners[np.dtype(dtype)]
    for n in (4097, 5000, 5759, 10007, 101 * 103, 100003, 3 * (1 << 15)):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
            x = zero_mean_signal(n * 2, dtype)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_general_column_tile_passes(planners, oracle, dtype):
    """7-smooth lengths above one workgroup run as 2-4 general column-tile passes (k2g kernels): tile heights that
    do not divide the strides, ragged last tiles, powers of 3 and 5, ragged batches; vs the oracle's planner choice
    (RadixN / MixedRadix, src/plan.rs:430-560) up to 10^5, vs numpy complex128 beyond."""
    planner = planners[np.dtype(dtype)]
    for n in (36864, 39366, 50000, 44100, 78125, 98304, 100000, 117649, 150000, 1000000, 1536000, 3 << 20, 5 << 21, 7 << 20):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k2gfirst"), (n, fft.describe())
            if n <= 100000:
                check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
            batch = 3 if n < 200000 else 2
            x = zero_mean_signal(n * batch, dtype, seed=n)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d, fft.describe())
    # round trip at a production-sized batch (size-independent property)
    import torch

    n, rows = 100000, 256
    rt = torch.complex64 if dtype == np.complex64 else torch.complex128
    x = torch.randn(rows * n, dtype=rt, device="cuda")
    y = x.clone()
    planner.plan_fft_forward(n).process(y)
    assert abs((y.abs().pow(2).sum() / x.abs().pow(2).sum()).item() / n - 1) < 1e-4  # Parseval
    planner.plan_fft_inverse(n).process(y)
    assert ((y / n - x).abs().mean().item()) < (1e-5 if dtype == np.complex64 else 1e-13)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_single_kernel_above_4096(planners, oracle, dtype):
    """2^13 .. 2^15 (f32; 2^14 in f64) and every generated 7-smooth length in (4096, 16384] -- round 5: and every 13-smooth one with a factor
    11 / 13 (kernels_smooth4_*: 264 lengths in f32, 173 in f64, two general column-tile passes until then) -- run as one split-exchange kernel
    (and the kernels_smooth5_* lengths: prime radices 17 .. 31, 877 f32 / 1102 f64 lengths up to 16384, f64 below 4096 through the plain exchange):
    vs the oracle's plan (Radix4 / RadixN, src/plan.rs:508-607) under the reference tolerance and vs numpy in float64."""
    import glob
    import re

    planner = planners[np.dtype(dtype)]
    tag = "f32" if dtype == np.complex64 else "f64"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sizes = [8192, 16384] + ([32768] if dtype == np.complex64 else [])
    for f in glob.glob(os.path.join(root, "rustfft_amd", "csrc", "kernels_smooth2_%s_*.hip" % tag)) + glob.glob(os.path.join(root, "rustfft_amd", "csrc", "kernels_smooth4_%s_*.hip" % tag)):
        sizes += [int(m) for m in re.findall(r'MI_K1X?\(\w+, \d+, 1, true, (?:\d+, "\w*", )?(\d+),', open(f).read())]
    # round 5, late: the lengths with a prime factor 17 .. 31 above the smooth3 limits (kernels_smooth5_*: f32 (4096, 16384], f64 (2048, 16384]; Bluestein until then)
    for f in glob.glob(os.path.join(root, "rustfft_amd", "csrc", "kernels_smooth5_%s_*.hip" % tag)):
        sizes += [int(m) for m in re.findall(r'MI_K1X?\(\w+, \d+, 1, (?:true|false), (?:\d+, "\w*", )?(\d+),', open(f).read())]
    assert len(sizes) > 280 + (877 if tag == "f32" else 1102) and 5005 in sizes and 13312 in sizes and 4352 in sizes and 8184 in sizes and 16337 in sizes
    for n in sorted(sizes):
        d = n % 2
        fft = planner.plan_fft(n, d)
        assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
        x = random_signal(n * 3, dtype, seed=n)
        y = x.copy()
        fft.process(y)
        if n % 7 == 0 or n in (8192, 16384, 32768, 10000):  # the oracle is the slow side: a subset against it, all against numpy
            want = x.copy()
            oracle.plan(dtype, n, d).process(want)
            assert compare_vectors(want, y), n
        assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], n


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_runtime_scheduled_kernels(planners, oracle, dtype):
    """Lengths the reference plans as RadixN / MixedRadix / RadersAlgorithm beyond the compiled single-kernel set: 13-smooth
    lengths above one workgroup, lengths with a prime factor 17 .. 31 (run-time scheduled HEAVY kernel), and the Rader family
    through the host-planner entry point, vs the oracle's planner choice, all four API modes."""
    planner = planners[np.dtype(dtype)]
    import rustfft_amd

    for n in [36608, 40898, 45056] + ([16731, 20449] if dtype == np.complex128 else []):  # 13-smooth above the whole-row kernels (f32: 32768, f64: 16384): column-tile passes with 11 / 13 in the tile heights
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert "k2gfirst" in fft.describe(), (n, fft.describe())
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
    for n in [4352, 4836, 6448, 9248, 16337]:  # a prime factor 17 .. 31 above 4096: compiled whole-row schedules since round 5 (kernels_smooth5_*; rounds 2 - 4: AUTO took the
        for d in (0, 1):          # one-kernel Bluestein, a host planner's MixedRadix recipe the run-time scheduled HEAVY kernel); both entry points get them
            assert planner.plan_fft(n, d).describe().startswith("k1<%d," % n)
            fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_MIXED_RADIX)
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
    for n in [17408, 18496]:  # ... and above 16384 Bluestein (fused multi-kernel: the inner length exceeds one workgroup)
        assert "bluestein" in planner.plan_fft(n, 0).describe()
        check_fft_algorithm(planner.plan_fft(n, 0), n, 0, reference=oracle.plan(dtype, n, 0), n=3)
    if True:  # compiled prime-radix schedules: both precisions up to 8192 since round 5 (before: 4096 in f32, 2048 in f64)
        for n in [2057, 2108, 3553, 3910, 4048, 4092]:
            for d in (0, 1):
                fft = planner.plan_fft(n, d)
                assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
                check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
    for n in [1088, 3553]:  # ... which a host planner can also ask for below 4096 (AUTO prefers compiled schedules / Bluestein there)
        fft = planner.plan_fft_with(n, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX)
        assert "dyn_k1" in fft.describe() or fft.describe().startswith("k1<"), (n, fft.describe())
        check_fft_algorithm(fft, n, 0, reference=oracle.plan(dtype, n, 0), n=3)

    # run-time scheduled Rader: what a host planner gets when its Recipe says RadersAlgorithm (mi355fft_plan_create_ex,
    # MI355FFT_ALGO_RADER) for a prime without a compiled body; AUTO plans these primes through other kernels
    for p in [5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 53, 61, 67, 71, 73, 79, 89, 97, 127, 211, 257, 331, 1201, 2311, 3001]:
        for d in (0, 1):
            fft = planner.plan_fft_with(p, d, algorithm=rustfft_amd.ALGO_RADER)
            assert "rader" in fft.describe(), (p, fft.describe())
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=3)
    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        planner.plan_fft_with(1000, 0, algorithm=rustfft_amd.ALGO_RADER)  # not prime (raders_algorithm.rs:68)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_random_lengths_vs_float64(planners, dtype):
    """Fuzz over the planner: 150 random lengths in [2, 300000] (every plan kind: single kernel, multi-pass, run-time
    scheduled mixed radix, Rader, one-kernel and multi-kernel Bluestein) with random ragged batches, forward and inverse,
    against numpy.fft in complex128."""
    rng = np.random.default_rng(20260924)
    planner = planners[np.dtype(dtype)]
    lengths = sorted(set(int(v) for v in np.exp(rng.uniform(np.log(2), np.log(300000), 150))) | {8192, 1 << 16, 1 << 17, 1009, 5000, 100000, 4620, 4836, 6448, 10007})
    seen = set()
    for n in lengths:
        batch = int(rng.integers(1, max(2, min(40, 400000 // n))))
        d = int(rng.integers(0, 2))
        fft = planner.plan_fft(n, d)
        seen.add(fft.describe().replace("fused{", "").split("<")[0].split("(")[0])  # (a fused two-pass plan is still the k2 kernel family)
        x = zero_mean_signal(n * batch, dtype, seed=n)
        y = x.copy()
        fft.process(y)
        assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, batch, d, fft.describe())
    want_kinds = {"k1", "k2first", "k2gfirst", "rader", "bluestein", "bluestein_large"}
    if dtype == np.complex64:  # f64: every padded length that fits one workgroup is a one-kernel plan since round 2
        want_kinds.add("bluestein2_first")
    assert want_kinds <= seen, seen


def _thirteen_smooth(limit):
    s = {1}
    for p in (2, 3, 5, 7, 11, 13):
        s = {v * p**k for v in s for k in range(0, 13) if v * p**k <= limit}
    return sorted(v for v in s if v > 2 and (v & (v - 1)))


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_compiled_smooth_schedules(planners, oracle, dtype):
    """Every 13-smooth length in [3, 4096] runs its own compiled schedule (the reference plans these as RadixN,
    src/plan.rs:508-607): vs the oracle's recipe under the reference tolerance and vs numpy in float64."""
    planner = planners[np.dtype(dtype)]
    for n in _thirteen_smooth(4096):
        d = n % 2
        fft = planner.plan_fft(n, d)
        assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
        x = random_signal(n * 3, dtype, seed=n)
        y = x.copy()
        fft.process(y)
        want = x.copy()
        oracle.plan(dtype, n, d).process(want)
        assert compare_vectors(want, y), n
        assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], n


def test_cpp_host_mirror(planners):
    """The C++17 mirror of FftPlanner / Fft (rustfft_amd/host/mi355fft.hpp): tests/cpp/mirror_check.cpp restates
    check_fft_algorithm (src/test_utils.rs:70-209) in C++ over the C ABI -- four entry points, dirty scratch, planner cache,
    panic text -- for ten lengths x two directions x two precisions against the O(n^2) definition."""
    import subprocess

    from helpers import build_cpp_mirror_check

    exe = build_cpp_mirror_check()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("ok n=") == 40


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_every_prime_below_1000_and_prime_radices(planners, oracle, dtype):
    """Every prime < 1000 through the planner's own choice (butterflies 2 .. 31, compiled Rader bodies where p - 1 is
    13-smooth, Bluestein elsewhere -- raders_algorithm.rs:302-322 and bluesteins_algorithm.rs:210-215 test the same families)
    and a spread of lengths with a prime factor 17 .. 31 (butterflies.rs:6414-6433), against the oracle."""
    planner = planners[np.dtype(dtype)]
    primes = [p for p in range(2, 1000) if all(p % q for q in range(2, int(p**0.5) + 1))]
    families = set()
    for p in primes:
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            families.add(fft.describe().split("<")[0])
            x = random_signal(3 * p, dtype)
            y = x.copy()
            fft.process(y)
            want = x.copy()
            oracle.plan(dtype, p, d).process(want)
            assert compare_vectors(want, y), (p, d, fft.describe())
            assert rel_l2(y, numpy_fft(x, p, d == 1)) < REL[np.dtype(dtype)], (p, d)
    assert {"k1", "rader", "bluestein"} <= families, families
    limit = 2048 if dtype == np.complex64 else 1024
    for n in (34, 289, 323, 437, 527, 899, 961, 992, 1023, 1088, 1445, 1734, 2046):
        fft = planner.plan_fft(n, 0)
        if n <= limit:
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
        check_fft_algorithm(fft, n, 0, reference=oracle.plan(dtype, n, 0), n=3)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_rader_bodies_of_the_31_smooth_primes_on_the_device(planners, oracle, dtype):
    """Round 5: every prime <= 4096 that moved from the one-kernel Bluestein to a compiled Rader body with prime-radix sub-passes
    (tools/gen_rader_kernels.py EXTRA31_R5: 89 Complex<f32> / 77 Complex<f64>), both directions, one full workgroup of rows and a ragged
    one, against numpy in float64; every fourth prime also element-wise against the reference's plan (tests/accuracy.rs bar)."""
    import re
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_rader_kernels as gen

    planner = planners[np.dtype(dtype)]
    prec = 32 if dtype == np.complex64 else 64
    new = sorted(p for (pr, p) in gen.EXTRA31_R5 if pr == prec and (pr, p) not in gen.EXTRA31_R2)
    assert len(new) == (89 if prec == 32 else 77)
    worst = 0.0
    for i, p in enumerate(new):
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            assert fft.describe().startswith("rader<%d," % (p - 1)), (p, fft.describe())
            rows = int(re.search(r"xF(\d+)", fft.describe()).group(1)) + 3
            x = random_signal(rows * p, dtype, seed=p)
            y = x.copy()
            fft.process(y)
            err = rel_l2(y, numpy_fft(x, p, d == 1))
            worst = max(worst, err)
            assert err < REL[np.dtype(dtype)], (p, d, err, fft.describe())
            if i % 4 == 0:
                want = x.copy()
                oracle.plan(dtype, p, d).process(want)
                assert compare_vectors(want, y), (p, d, fft.describe())
    print(f"31-smooth Rader primes {np.dtype(dtype).name}: {len(new)} primes x 2 directions, worst rel L2 vs numpy c128 {worst:.2e}")


def _moved_to_the_stage_machine(planner, lo, hi):
    """Lengths in [lo, hi] that AUTO plans as the LDS stage machine (plan.cpp try_lsm: the calibrated choice)."""
    out = []
    for n in range(lo, hi + 1):
        small = n
        for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
            while small % q == 0:
                small //= q
        if small == 1:
            continue  # 31-smooth: a compiled whole-row schedule
        if planner.plan_fft(n, 0).describe().startswith("lsm<"):
            out.append(n)
    return out


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_stage_machine_every_moved_length_up_to_4096(planners, oracle, dtype):
    """Round 6: EVERY length <= 4096 that left whole-length Bluestein for the LDS stage machine (the reference's MixedRadix over the smooth
    part and one Rader per large prime factor, src/plan.rs:412-425, 474-506; lsm.h), both directions, two full workgroups of rows and a ragged
    one, against numpy in float64; every seventh length also element-wise against the oracle's plan of the length (tests/accuracy.rs bar)."""
    import re

    planner = planners[np.dtype(dtype)]
    moved = _moved_to_the_stage_machine(planner, 38, 4096)
    assert len(moved) >= (1200 if dtype == np.complex64 else 1000), len(moved)
    worst = 0.0
    for i, n in enumerate(moved):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            rows = 2 * int(re.search(r"sF(\d+)$", fft.describe()).group(1)) + 1
            x = random_signal(rows * n, dtype, seed=n)
            y = x.copy()
            fft.process(y)
            err = rel_l2(y, numpy_fft(x, n, d == 1))
            worst = max(worst, err)
            assert err < REL[np.dtype(dtype)], (n, d, err, fft.describe())
            if i % 7 == 0:
                want = x.copy()
                oracle.plan(dtype, n, d).process(want)
                assert compare_vectors(want, y), (n, d, fft.describe())
    print(f"stage machine {np.dtype(dtype).name}: {len(moved)} lengths <= 4096 x 2 directions, worst rel L2 vs numpy c128 {worst:.2e}")


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_stage_machine_above_4096_and_host_planner_trees(planners, oracle, dtype):
    """The stage machine above 4096 (every 29th moved length up to 16384: 512- and 1024-thread programs, six-step tables in global memory) and
    the trees AUTO does not take but a host planner may ask for (MixedRadix / Rader requests: two Rader factors, Rader over MixedRadix over
    Rader), device-resident at FULL occupancy (2 x 256 x F rows), all three API modes, against numpy float64 and the oracle."""
    import re

    import torch

    import rustfft_amd

    planner = planners[np.dtype(dtype)]
    tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
    cases = [(n, None) for n in _moved_to_the_stage_machine(planner, 4097, 16384 if dtype == np.complex64 else 8192)[::29]]
    assert len(cases) >= (20 if dtype == np.complex64 else 5), len(cases)
    cases += [(1369, rustfft_amd.ALGO_MIXED_RADIX), (1517, rustfft_amd.ALGO_MIXED_RADIX), (3034, rustfft_amd.ALGO_MIXED_RADIX), (167, rustfft_amd.ALGO_RADER),
              (1283, rustfft_amd.ALGO_RADER), (3067, rustfft_amd.ALGO_RADER)]
    for n, algo in cases:
        for d in (0, 1):
            fft = planner.plan_fft(n, d) if algo is None else planner.plan_fft_with(n, d, algorithm=algo)
            assert fft.describe().startswith("lsm<"), (n, fft.describe())
            F = int(re.search(r"sF(\d+)$", fft.describe()).group(1))
            rows = min(2 * 256 * F + 1, max(3, (1 << 27) // (n * np.dtype(dtype).itemsize)))
            x = random_signal(rows * n, dtype, seed=n + d)
            want = numpy_fft(x[: 64 * n], n, d == 1)
            dx = torch.from_numpy(x).cuda()
            y = dx.clone()
            fft.process(y)  # in place, device-resident
            out = torch.empty_like(dx)
            fft.process_immutable_with_scratch(dx, out)
            assert torch.equal(torch.view_as_real(out), torch.view_as_real(y)), (n, "immutable")
            src = dx.clone()
            out2 = torch.empty_like(dx)
            fft.process_outofplace_with_scratch(src, out2)
            assert torch.equal(torch.view_as_real(out2), torch.view_as_real(y)), (n, "out of place")
            got = y.cpu().numpy()
            assert rel_l2(got[: 64 * n], want) < REL[np.dtype(dtype)], (n, d, fft.describe())
            # every row against the first block's rows is not possible (random rows): check the LAST rows too (the ragged workgroup)
            tail = x[-3 * n:]
            assert rel_l2(got[-3 * n:], numpy_fft(tail, n, d == 1)) < REL[np.dtype(dtype)], (n, d, "tail")
            ref = x[: 2 * n].copy()
            oracle.plan(dtype, n, d).process(ref)
            assert compare_vectors(ref, got[: 2 * n]), (n, d, fft.describe())


def test_repeatability_bit_for_bit(planners):
    """A transform is a pure function of its input: five runs of every kernel family on the same HBM-resident input must
    agree bit for bit.  A write-write or read-write race between threads (round 2: the Rader X[0] slot) shows up here as
    run-to-run differences long before it shows up as a tolerance failure."""
    import torch

    lengths = [17, 127, 257, 541, 911, 1009, 1201, 2311, 4051,     # Rader family (all three body forms)
               59, 1013, 1117, 2053, 4093,                           # Rader over prime-radix sub-passes (round 5: the 31-smooth primes)
               719, 1019, 4091, 4099, 7919, 10007, 65537,            # Bluestein: one kernel, split one kernel, fused multi-kernel
               289, 899, 1200, 4096, 5000, 1 << 14, 25000,           # compiled schedules incl. prime radices, whole-row split kernels
               4836, 20449, 44100, 1 << 17, 1 << 20, 1 << 22,        # run-time scheduled, general and power-of-two column tiles
               74, 592, 1110, 2368, 4070, 4218, 8144, 12210]         # round 6: the LDS stage machine (64 .. 1024 threads, in-place stages)
    for dtype, tdtype in ((np.complex64, torch.complex64), (np.complex128, torch.complex128)):
        planner = planners[np.dtype(dtype)]
        for n in lengths:
            batch = max(3, min(4096, (1 << 22) // n))
            x = torch.from_numpy(random_signal(n * batch, dtype, seed=n)).cuda()
            fft = planner.plan_fft_forward(n)
            first = None
            for _ in range(5):
                y = x.clone()
                fft.process(y)
                torch.cuda.synchronize()
                if first is None:
                    first = y
                else:
                    assert torch.equal(torch.view_as_real(first), torch.view_as_real(y)), (n, np.dtype(dtype).name, fft.describe())


@pytest.mark.parametrize("dtype,log2n", [(np.complex64, 23), (np.complex64, 24), (np.complex128, 23)])
def test_three_pass_pow2_plans(planners, oracle, dtype, log2n):
    """The three-kernel power-of-two plans (2^23, 2^24; north_star's range ends at 2^24, SURVEY section 8 a15 names "2^24 k=10"):
    both directions, two HBM-resident rows, against the oracle's Radix4 (src/algorithm/radix4.rs:167-203) under the
    reference tolerance and against numpy complex128."""
    import torch

    n, batch = 1 << log2n, 2
    planner = planners[np.dtype(dtype)]
    x = random_signal(n * batch, dtype, seed=log2n)
    for d in (0, 1):
        fft = planner.plan_fft(n, d)
        assert fft.describe().count("k2") == 3, fft.describe()
        y = torch.from_numpy(x).cuda()
        fft.process(y)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        want = x.copy()
        oracle.plan(dtype, n, d).process(want)
        assert compare_vectors(want, got), (log2n, d)
        assert rel_l2(got, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (log2n, d)
        # the out-of-place device entry point runs the same three passes through other buffers: bit-identical
        a = torch.from_numpy(x).cuda()
        out = torch.empty_like(a)
        fft.process_immutable_with_scratch(a, out)
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(y)), (log2n, d)


def test_host_planner_options_on_device(planners, oracle):
    """mi355fft_plan_create_ex on the real device -- the "planner / twiddle host code stays in Rust" half of the boundary: the
    oracle plays the Rust planner and supplies compute_twiddle (src/twiddles.rs:6-23), RadersAlgorithm::new's inner_fft_data
    (raders_algorithm.rs:87-113) and BluesteinsAlgorithm::new's twiddles + multiplier (bluesteins_algorithm.rs:63-98)."""
    from helpers import check_host_planner_options

    check_host_planner_options(planners[np.dtype(np.complex64)], oracle)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_every_compiled_rader_prime(planners, dtype):
    """Every prime with a compiled Rader body (tools/gen_rader_kernels.py: 141 f32 / 148 f64 primes up to 4096, body form per
    prime by measurement -- rows loop, rows side by side, side by side with the register hand-over between the two inner
    transforms): both directions, a ragged five rows, against numpy in complex128 under the reference's gate
    (tests/accuracy.rs:30-37) and a relative-L2 bound."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_rader_kernels as gen

    planner = planners[np.dtype(dtype)]
    prec = 32 if dtype == np.complex64 else 64
    s13 = set(gen.g.smooth(4096, [2, 3, 5, 7, 11, 13]))
    # (17 .. 31 are prime-radix butterflies, 1009 is the hand-tuned body of kernels_np2_*.hip)
    primes = sorted({p for p in range(37, 4097) if gen.is_prime(p) and (p - 1) in s13} | {p for (pr, p) in gen.EXTRA31 if pr == prec})
    assert len(primes) >= 136 and 1009 in primes
    forms = set()
    tol = 5e-6 if dtype == np.complex64 else 1e-13
    for p in primes:
        x = random_signal(5 * p, dtype, seed=p)
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            assert fft.describe().startswith("rader<%d," % (p - 1)), (p, fft.describe())
            forms.add(fft.describe().rsplit("m", 1)[1])
            y = x.copy()
            fft.process(y)
            want = numpy_fft(x, p, d == 1)
            assert compare_vectors(want.astype(dtype), y), (p, d, fft.describe())
            assert rel_l2(y, want) < tol, (p, d, fft.describe(), rel_l2(y, want))
    assert {"1", "5"} <= forms and (forms & {"2", "3", "4"}), forms


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_host_planner_recipe_on_device(planners, oracle, dtype):
    """mi355fft_plan_options.recipe on the real device: the oracle's restatement of FftPlannerScalar::design_fft_for_len
    (src/plan.rs:312-323, 412-665) plays the Rust planner and hands its whole Recipe tree over; six-step splits it names become
    the column-tile pass heights, its Bluestein inner length is used, malformed trees are rejected."""
    from helpers import check_host_planner_recipe

    check_host_planner_recipe(planners[np.dtype(dtype)], oracle, dtype, big=True)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_large_primes_vs_oracle(planners, oracle, dtype):
    """Primes above one workgroup's reach with a smooth p - 1 (the reference plans RadersAlgorithm for them, src/plan.rs:636-665;
    raders_algorithm.rs:302-322 tests 112501 / 216569 / 417623 the same way): all four API modes against the oracle's plan and
    numpy complex128, both directions."""
    planner = planners[np.dtype(dtype)]
    for p in (12289, 40961, 41959, 65537, 112501):  # 41959 - 1 = 2 * 3^4 * 7 * 37: a PRIME tile height (37) between the fused passes
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            # round 3: multi-kernel Rader (gather / spectrum multiply / scatter fused into the column-tile passes of the two
            # inner transforms of length p - 1) instead of the fused Bluestein over M >= 2p - 1
            assert fft.describe().startswith("rader_large(p-1=%d fused: k2gfirst_gather<" % (p - 1)) and "k2glast_scatter<" in fft.describe(), fft.describe()
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=2)
            # 19 rows: two complete groups of eight transforms (tiles of transform g on XCD g % 8 in the gather / scatter passes) + 3
            x = zero_mean_signal(p * 19, dtype, seed=p)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, p, d == 1)) < REL[np.dtype(dtype)], (p, d, fft.describe())


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_prime_tile_heights_vs_oracle(planners, oracle, dtype):
    """Composite lengths whose prime factors exceed 31 (the reference: MixedRadix over Rader inner FFTs, src/plan.rs:474-506,
    mixed_radix.rs:53-158) as column-tile passes with PRIME tile heights -- Rader inside the tile (k2r_body): 101 x 103, a prime
    tile beside a smooth one, three passes, ragged columns; 37 x 41 and 59 x 61 through a host planner's MixedRadix recipe (AUTO
    keeps the one-kernel Bluestein at or below 4096).  Against the oracle's plan and numpy complex128, both directions."""
    import rustfft_amd

    planner = planners[np.dtype(dtype)]
    for n in (101 * 103, 64 * 131, 37 * 41 * 43, 47 * 229, 89 * 97, 251 * 631):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            # (round 6: the LDS stage machine goes ahead of the prime-tile passes only up to 8192; these lengths keep the passes)
            assert ("k2rfirst<" in fft.describe() or "k2rlater<" in fft.describe()) and "bluestein" not in fft.describe(), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
            x = zero_mean_signal(n * 3, dtype, seed=n)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d, fft.describe())
    # at or below 4096: until round 5 AUTO kept the one-kernel Bluestein and a MixedRadix request got two prime-tile passes through HBM; round 6: the
    # reference's tree in ONE kernel (the LDS stage machine: seven stages, within AUTO's calibrated limit in both precisions)
    for n in (37 * 41, 59 * 61):
        auto = planner.plan_fft(n, 0).describe()
        # (59 x 61: 58 = 2 x 29 puts a second Rader inside the first -- a longer program than AUTO takes; on request it runs all the same)
        assert auto.startswith("lsm<mixed{rader") if n == 37 * 41 else "bluestein" in auto, auto
        for d in (0, 1):
            fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_MIXED_RADIX)
            assert fft.describe().startswith("lsm<mixed{rader"), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)


def test_host_slices_pipeline_and_shared_plan_threads(planners, oracle):
    """The literal drop-in path on the device: a 384 MiB host slice (six staging chunks: upload + kernels on the calling thread,
    download on the helper thread) in all three API modes against the one-chunk result of the same rows, and four host threads
    sharing one two-pass plan (examples/concurrency.rs:9-30), each through its own staging context, every row against the oracle."""
    import threading

    planner = planners[np.dtype(np.complex64)]
    n, batch = 1 << 16, 768
    fft = planner.plan_fft_forward(n)
    x = zero_mean_signal(n * batch, np.complex64, seed=16)
    y = x.copy()
    fft.process(y)
    for r in (0, 1, 127, 128, 383, 767):
        want = x[r * n:(r + 1) * n].copy()
        oracle.plan(np.complex64, n, 0).process(want)
        assert rel_l2(y[r * n:(r + 1) * n], want) < REL[np.dtype(np.complex64)], r
    small = x[: 8 * n].copy()
    fft.process(small)  # one chunk
    # (2^16 runs the fused two-pass launch for batches that fill its ring and two launches below that: the same kernel bodies
    # compiled into two kernels, equal up to the rounding of differently contracted multiply-adds)
    assert np.array_equal(small, y[: 8 * n]) if not fft.is_fused() else rel_l2(small, y[: 8 * n]) < 2e-7
    out = np.zeros_like(x)
    fft.process_immutable_with_scratch(x, out)
    assert np.array_equal(out, y)
    src, out2 = x.copy(), np.zeros_like(x)
    fft.process_outofplace_with_scratch(src, out2)
    assert np.array_equal(out2, y)
    xs = [zero_mean_signal(n * 40, np.complex64, seed=100 + i) for i in range(4)]
    errs = []

    def work(i):
        try:
            for _ in range(2):
                z = xs[i].copy()
                fft.process(z)
                for r in (0, 39):
                    want = xs[i][r * n:(r + 1) * n].copy()
                    oracle.plan(np.complex64, n, 0).process(want)
                    assert rel_l2(z[r * n:(r + 1) * n], want) < REL[np.dtype(np.complex64)]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_config2_full_batch_every_row(planners):
    """BASELINE config 2 at its full size, every row: per-row Parseval (a mis-twiddled row keeps its element sum but not its
    energy spectrum's consistency with a second check) and 64 rows drawn at random over the whole batch against numpy
    complex128 -- forward and inverse."""
    import torch

    n, batch = 1 << 20, 1024
    planner = planners[np.dtype(np.complex64)]
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 222)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0, generator=g)
    rows = sorted(int(r) for r in np.random.default_rng(22).choice(batch, 64, replace=False))
    keep = {r: x[r * n:(r + 1) * n].cpu().numpy() for r in rows}

    def energy(t):
        v = torch.view_as_real(t).view(batch, -1)
        return torch.cat([(v[r0:r0 + 64].double() ** 2).sum(dim=1) for r0 in range(0, batch, 64)])

    for d in (0, 1):
        y = x.clone()
        e_in = energy(y)
        planner.plan_fft(n, d).process(y)
        torch.cuda.synchronize()
        rel = ((energy(y) / n - e_in).abs() / e_in).max().item()
        assert rel < 1e-5, (d, rel)
        for r in rows:
            got = y[r * n:(r + 1) * n].cpu().numpy()
            assert rel_l2(got, numpy_fft(keep[r], n, d == 1)) < REL[np.dtype(np.complex64)], (d, r)
        del y


# ---- round 4 ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p", [216569, 417623])
def test_reference_overflow_regression_primes(planners, oracle, p):
    """raders_algorithm.rs:312-322: the reference's regression test for 32-bit overflow in the index arithmetic runs 112501,
    216569 and 417623.  216568 = 2^3 * 11 * 23 * 107 and 417622 = 2 * 208811 have prime factors above 31, so the GPU plans them
    through the fused multi-kernel Bluestein (element indices up to batch * M, chirp exponents i^2 mod 2p up to 1.7e11 computed
    on the host in 128 bits): both directions, against the oracle's plan and numpy complex128."""
    for dtype in (np.complex64, np.complex128):
        planner = planners[np.dtype(dtype)]
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            x = zero_mean_signal(p * 3, dtype, seed=p + d)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, p, d == 1)) < REL[np.dtype(dtype)], (p, d, fft.describe())
            want = x[:p].copy()
            oracle.plan(dtype, p, d).process(want)
            assert compare_vectors(want, y[:p]), (p, d)


def test_pow2_above_2p24(planners):
    """One length above north_star's range: 2^25 Complex<f32> (three passes, 256 MiB per row), two rows, against numpy
    complex128 -- the claim "three column-tile passes up to 2^30" is tested at least one step past 2^24."""
    import torch

    n = 1 << 25
    fft = planners[np.dtype(np.complex64)].plan_fft_forward(n)
    assert fft.describe().count("k2") == 3, fft.describe()
    x = zero_mean_signal(n * 2, np.complex64, seed=25)
    y = torch.from_numpy(x).cuda()
    fft.process(y)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu().numpy(), numpy_fft(x, n, False)) < REL[np.dtype(np.complex64)]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_host_slices_need_only_element_alignment_on_device(planners, oracle, dtype):
    """A Rust `&mut [Complex<T>]` guarantees align_of::<T>() only (4 bytes for Complex<f32>, 8 for Complex<f64>; SURVEY section
    8(b) layout rule): host buffers that start one float past a 16-byte boundary, all three trait methods, single-kernel,
    fused and multi-pass plans, on the real device (the staging copies are plain byte copies)."""
    real = np.float32 if dtype == np.complex64 else np.float64
    planner = planners[np.dtype(dtype)]
    for n, rows in ((1009, 3), (1200, 3), (1 << 16, 3), (1 << 20, 12)):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            x = random_signal(rows * n, dtype, seed=n + d)
            want = x[:n].copy()
            oracle.plan(dtype, n, d).process(want)

            def skewed(values=None):
                raw = np.zeros(2 * rows * n + 8, dtype=real)
                start = ((-raw.ctypes.data) % 16) // raw.itemsize + 1
                view = raw[start:start + 2 * rows * n].view(dtype)
                assert view.ctypes.data % 16 == raw.itemsize and view.flags.c_contiguous
                if values is not None:
                    view[:] = values
                return view

            a = skewed(x)
            fft.process(a)
            assert compare_vectors(want, a[:n]), (n, d, "in place")
            assert rel_l2(a, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d)
            src, dst = skewed(x), skewed()
            fft.process_outofplace_with_scratch(src, dst)
            assert rel_l2(dst, a) < 1e-6, (n, d, "out of place")
            src, dst = skewed(x), skewed()
            fft.process_immutable_with_scratch(src, dst)
            assert rel_l2(dst, a) < 1e-6 and np.array_equal(src, x), (n, d, "immutable")


@pytest.mark.parametrize("dtype,log2n", [(np.complex64, k) for k in (16, 17, 18, 19, 20, 21, 22, 23, 24)] + [(np.complex128, k) for k in (15, 16, 17, 18, 19, 20, 21, 23, 24)])
def test_fused_two_pass_kernel_vs_oracle(planners, oracle, dtype, log2n):
    """The fused two-pass launch (one kernel, second pass of transform g - lag beside the first pass of transform g, the
    intermediate through a cache-resident ring; launch.h k2f_kernel) for every length that has one, both precisions -- 2^23 and 2^24
    are THREE-pass plans whose first two passes run fused over units of a transform (kernels_params.h), the third as a launch of its
    own: rows across
    the whole batch against the oracle's Radix4 (src/algorithm/radix4.rs:167-203), the whole batch against the two-launch plan of
    the same kernels, the dependency error word, all three device entry points, both directions.  (Complex<f64> stores the ring
    with an inline-asm 16-byte write-through store: its missing hazard pad produced wrong results on the device at 2^17, 2^18 and
    2^20 while every emulator test passed -- this test is what guards it.)"""
    import torch

    import rustfft_amd

    n = 1 << log2n
    tdt, esz = (torch.complex64, 8) if dtype == np.complex64 else (torch.complex128, 16)
    batch = max(24, (1 << 31) // (n * esz))  # 2 GiB of rows (at least 24 transforms)
    planner = rustfft_amd.FftPlanner(dtype)  # own planner: the plans' fused setting is changed below
    x = torch.empty(batch * n, dtype=tdt, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + log2n)
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    rows = sorted({0, 1, batch // 2, batch - 1})
    for d in (0, 1):
        fus, two = planner.plan_fft(n, d), rustfft_amd.FftPlanner(dtype).plan_fft(n, d)
        fus.set_fused(1)
        two.set_fused(0)
        assert fus.is_fused() and fus.describe().startswith("fused{") and not two.is_fused(), fus.describe()
        a, b = x.clone(), x.clone()
        two.process(a)
        fus.process(b)
        assert fus.fused_status() == 0
        # the same kernel bodies, compiled into another kernel: equal up to the rounding of differently contracted multiply-adds
        err = (torch.view_as_real(a) - torch.view_as_real(b)).abs().max().item()
        assert err <= (4e-7 if dtype == np.complex64 else 1e-15) * torch.view_as_real(a).abs().max().item(), (log2n, d, err)
        ref = oracle.plan(dtype, n, d)
        for r in rows:
            want = x[r * n:(r + 1) * n].cpu().numpy()
            ref.process(want)
            assert compare_vectors(want, b[r * n:(r + 1) * n].cpu().numpy()), (log2n, d, r)
        out = torch.empty_like(x)
        fus.process_immutable_with_scratch(x, out)
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(b)), (log2n, d, "immutable")
        src = x.clone()
        fus.process_outofplace_with_scratch(src, out)
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(b)) and fus.fused_status() == 0, (log2n, d, "out of place")
        # a batch smaller than the ring runs as two launches: identical to the two-launch plan (three-pass plans: two transforms are
        # 8 - 32 units, more than the ring has slots -- the fused launch runs, equal up to rounding as above)
        c, e = x[: 2 * n].clone(), x[: 2 * n].clone()
        fus.process(c)
        two.process(e)
        if log2n < 23:
            assert torch.equal(torch.view_as_real(c), torch.view_as_real(e)), (log2n, d, "small batch")
        else:
            err = (torch.view_as_real(c) - torch.view_as_real(e)).abs().max().item()
            assert err <= (4e-7 if dtype == np.complex64 else 1e-15) * torch.view_as_real(e).abs().max().item() and fus.fused_status() == 0, (log2n, d, "small batch", err)


@pytest.mark.parametrize("log2n,batch", [(20, 256), (21, 128), (23, 32)])
def test_fused_kernel_repeatable_under_load_and_across_streams(planners, log2n, batch):
    """A stale read of the ring (a missing acquire, a slot rewritten too early) shows up as a run-to-run difference long before it
    breaks a tolerance: 30 fused transforms of one input, odd ones beside a copy stream that hammers HBM, must agree bit for
    bit; then two host threads drive the same plan on their own streams (own rings) concurrently.  2^20: the two-pass launch; 2^21:
    the re-split plan with the narrow later tile; 2^23: units of a three-pass plan (ring of 4 slots, each reused 32 times per call)."""
    import torch

    import rustfft_amd

    n = 1 << log2n
    fft = rustfft_amd.FftPlanner(np.complex64).plan_fft_forward(n)
    assert fft.is_fused()
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    side = torch.cuda.Stream()
    first = None
    for r in range(30):
        y = x.clone()
        if r % 2:
            with torch.cuda.stream(side):
                z = x.clone()  # noqa: F841
        fft.process(y)
        assert fft.fused_status() == 0
        if first is None:
            first = y
        else:
            assert torch.equal(torch.view_as_real(first), torch.view_as_real(y)), r
    results, errors = [None, None], []

    def worker(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                y = x.clone()
                for _ in range(3):
                    y.copy_(x)
                    fft.process(y)
                assert fft.fused_status() == 0
                s.synchronize()
                results[i] = y
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for y in results:
        assert torch.equal(torch.view_as_real(first), torch.view_as_real(y))


@pytest.mark.parametrize("n,batch", [(1024, 512), (1009, 256), (4099, 64), (45056, 32), (1 << 16, 96), (1 << 20, 40), (1 << 22, 8), (1 << 23, 8)])
def test_device_calls_capture_into_a_hip_graph(planners, n, batch):
    """The device entry points are stream-ordered and make no host-side synchronisation once a plan's workspaces exist (after one
    warm-up call on the stream): a forward + inverse pair captured into a HIP graph (hipStreamBeginCapture through torch.cuda.graph)
    replays with the results of the direct calls -- whole-row kernel, Rader, Bluestein, general passes, the fused launch (a memset node
    and a kernel node), two launches through a workspace, and the unit launch of a three-pass plan."""
    import torch

    import rustfft_amd

    planner = rustfft_amd.FftPlanner(np.complex64)
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    want = x.clone()
    fwd.process(want)
    torch.cuda.synchronize()
    want_pair = want.clone()
    inv.process(want_pair)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    buf, mid = x.clone(), torch.empty_like(x)
    with torch.cuda.stream(s):
        tmp = x.clone()
        fwd.process(tmp)  # warm-up on the capture stream: workspaces and rings are allocated (and synchronised) here, not under capture
        inv.process(tmp)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fwd.process(buf)
        mid.copy_(buf)
        inv.process(buf)
    for _ in range(3):
        buf.copy_(x)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(mid), torch.view_as_real(want)), (n, "forward")
        assert torch.equal(torch.view_as_real(buf), torch.view_as_real(want_pair)), (n, "pair")
    if fwd.is_fused():
        assert fwd.fused_status() == 0 and inv.fused_status() == 0


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_multi_device_plan_on_one_gpu(planners, oracle, dtype):
    """mi355fft_multi_plan with the device list [0, 0] (two shards, two replicas, two worker threads and staging pools on the one
    GPU of this box): the three trait methods on host slices and on device-resident shards against the oracle, ragged batches,
    the scatter / gather edges, the validation semantics."""
    import torch

    import rustfft_amd
    from rustfft_amd import FftPanic

    mp = rustfft_amd.FftPlannerHipMulti(dtype, devices=[0, 0])
    for n, batch in ((1009, 7), (1200, 5), (1 << 16, 5), (1 << 20, 25)):
        for d in (0, 1):
            multi = mp.plan_fft(n, d)
            assert multi.shards() == 2 and multi.devices() == [0, 0]
            x = random_signal(n * batch, dtype, seed=n + d)
            ref = oracle.plan(dtype, n, d)
            a = x.copy()
            multi.process(a)
            for r in (0, batch // 2, batch - 1):
                want = x[r * n:(r + 1) * n].copy()
                ref.process(want)
                assert compare_vectors(want, a[r * n:(r + 1) * n]), (n, d, r)
            assert rel_l2(a, numpy_fft(x, n, d == 1)) < REL[np.dtype(dtype)], (n, d)
            y = np.empty_like(x)
            multi.process_immutable_with_scratch(x, y)
            assert rel_l2(y, a) < 1e-6
            # device-resident shards + the edges
            root = torch.from_numpy(x).cuda()
            shards = [torch.empty(multi.shard_rows(batch, g)[1] * n, dtype=root.dtype, device="cuda") for g in range(2)]
            multi.scatter(root, shards)
            multi.process(shards)
            out = torch.empty_like(root)
            multi.gather(shards, out)
            multi.synchronize()
            assert rel_l2(out.cpu().numpy(), a) < 1e-6, (n, d, "device shards")
    multi = mp.plan_fft(1024, 0)
    # NUMA placement (round 5): the workers are bound to their GPU's node exactly when sysfs names one for the device (real hardware: the
    # list is the node's cores; a one-socket box or a container without the node files: unbound); the calling thread is never re-bound
    cpus = rustfft_amd.device_cpulist(0)
    node = set()
    for part in filter(None, cpus.split(",")):
        lo, _, hi = part.partition("-")
        node.update(range(int(lo), int(hi or lo) + 1))
    expect = bool(node & os.sched_getaffinity(0))  # (a cpuset that excludes the whole node leaves nothing to bind to)
    assert [multi.shard_pinned(g) for g in range(2)] == [expect] * 2, (cpus, [multi.shard_pinned(g) for g in range(2)])
    print("device 0 sits on the NUMA node with CPUs", repr(cpus))
    with pytest.raises(FftPanic, match="multiple of FFT length"):
        multi.process(random_signal(1024 * 3 + 1, dtype))
    with pytest.raises(FftPanic, match="same length"):
        multi.process_outofplace_with_scratch(random_signal(1024, dtype), np.empty(2048, dtype))


def test_bench_two_ranks_on_one_gpu():
    """Multi-GPU readiness on a one-GPU box: `bench.py --gpus 2` self-spawns two ranks (torchrun's environment contract) that
    share cuda:0 and rendezvous over gloo; barrier + MAX-over-ranks timing, the per-rank checks MAX-reduced, the nested
    config-5 measurement and the scatter / gather edges all run end to end.  (The RCCL path differs only in the backend name.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--one-device", "--dist-backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "64", "--edges", "--no-pmc", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "FAILED" not in line["check"] and line["check"]["roundtrip_rel_l2"] < 5e-6
    assert line["config5"]["parseval_max_rel_err_over_ranks"] < 1e-4
    assert line["edges"]["scatter_s"] > 0


def test_bench_via_cabi_two_shards_on_one_gpu():
    """The multi-GPU path behind the boundary, end to end: `bench.py --via-cabi --gpus 2 --one-device` -- ONE process, one
    mi355fft_multi_plan over the device list [0, 0], device-resident shards, mi355fft_multi_process_inplace_dev +
    mi355fft_multi_synchronize around the timed region; the line must carry a passing round-trip check."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--via-cabi", "--gpus", "2", "--one-device", "--steps", "3", "--warmup", "1", "--batch", "128"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "FAILED" not in line["check"] and line["config"]["finite"]


# ---- round 5 ---------------------------------------------------------------------------------------------------------------------
def _lengths_file(name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return [int(v) for v in open(os.path.join(root, "tools", "r4", name)).read().replace(",", " ").split()]


def test_soak_300_mixed_lengths_in_one_process(planners):
    """two round-4 sweeps died with `execution failed` at n = 603680 (three general passes) and n = 1176121 (multi-kernel
    Rader) after ~175 / ~87 lengths in ONE process.  The reference never fails a planned length (src/plan.rs:289-295).  The same 300
    lengths (general two- to four-pass plans, prime tile heights, multi-kernel Rader primes), each planned and run in place over a 1 GiB
    buffer in this one process with the workspaces trimmed in between as the sweeps did, two rows of every length against numpy
    complex128; the device's free memory is watched (a leak is what would make a later allocation fail)."""
    import torch

    import rustfft_amd

    lengths = _lengths_file("general_f32_lengths.txt") + _lengths_file("prime_tile_rader_large_lengths.txt") + [4875, 45056, 9000, 17325, 4199, 1 << 17, 1 << 22, 3 << 19]
    lengths = list(dict.fromkeys(lengths))
    assert len(lengths) >= 300 and 603680 in lengths and 1176121 in lengths
    planner = rustfft_amd.FftPlanner(np.complex64)
    x = torch.empty((1 << 30) // 8, dtype=torch.complex64, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    rng = np.random.default_rng(5)
    worst = (0.0, 0)
    for i, n in enumerate(lengths):
        batch = x.numel() // n
        buf = x[: batch * n]
        torch.view_as_real(buf).uniform_(-1.0, 1.0)
        rows = [0, int(rng.integers(0, batch))]
        want = [np.fft.fft(buf[r * n:(r + 1) * n].cpu().numpy().astype(np.complex128)) for r in rows]
        fft = planner.plan_fft_forward(n)
        try:
            fft.process(buf)
            torch.cuda.synchronize()
        except rustfft_amd.FftPanic as e:  # the message must say which step failed and why
            raise AssertionError(f"length {n} (#{i}, {fft.describe()}): {e}; free device memory {torch.cuda.mem_get_info()[0]} of {free0} at the start") from e
        for r, w in zip(rows, want):
            err = rel_l2(buf[r * n:(r + 1) * n].cpu().numpy(), w)
            worst = max(worst, (err, n))
            assert err < 5e-6, (n, r, err, fft.describe())
        fft.trim_workspaces()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost < (8 << 30), f"{lost} bytes of device memory gone after 300 trimmed plans (tables only are expected)"
    print("soak: worst rel L2", worst, "device memory held by 300 plans' tables", lost)


@pytest.mark.parametrize("dtype,log2n,batch", [(np.complex64, 16, 512), (np.complex64, 20, 64), (np.complex128, 19, 48), (np.complex64, 23, 16)])
def test_fused_giveup_cannot_be_missed_on_the_device(planners, dtype, log2n, batch):
    """VERDICT r4 weak 2 / ADVICE r4: a fused launch whose wait gave up used to return success with wrong data on the `_dev` path.  With the
    wait limit at 0 every dependency that is not already met gives up (include/mi355fft.h mi355fft_plan_set_fused_wait_limit) -- a healthy
    device then produces real give-ups: the sticky word is raised, the NEXT device call on the plan and stream fails without running, the
    explicit status reports and clears it, host slices come back CORRECT (the rows are re-run as one launch per pass; 2^16 x 512: four
    chunks of 128 rows, each a fused launch -- the longer rows' 64 MiB chunks hold fewer rows than the ring has slots and run as two
    launches anyway), and with the limit restored the plan is as good as new."""
    import torch

    import rustfft_amd

    n = 1 << log2n
    tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
    fft = rustfft_amd.FftPlanner(dtype).plan_fft_forward(n)
    ref = rustfft_amd.FftPlanner(dtype).plan_fft_forward(n)
    ref.set_fused(0)
    assert fft.is_fused() and not ref.is_fused()
    x = torch.empty(batch * n, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    want = x.clone()
    fft.process(want)  # the fused launch's own result (bit-repeatable; equal to the two-launch plan's up to rounding: other compilation units)
    assert fft.fused_status() == 0
    want_two = x.clone()
    ref.process(want_two)
    tol = (4e-7 if dtype == np.complex64 else 1e-15) * float(want.abs().max())
    assert float((want - want_two).abs().max()) <= tol
    fft.set_fused_wait_limit(0)
    gave_up = 0
    for it in range(12):
        if it == 6 and gave_up == 0:
            fft.set_fused_wait_limit(-1)  # this size meets its dependencies at the first poll almost always: raise the word on every launch instead
        y = x.clone()
        fft.process(y)  # enqueues fine whatever happens on the device
        torch.cuda.synchronize()
        z = x.clone()
        try:
            fft.process(z)
        except rustfft_amd.FftPanic as e:
            gave_up += 1
            assert e.status == 8 and "gave up waiting for a dependency" in str(e) and "INVALID" in str(e), str(e)
            assert torch.equal(torch.view_as_real(z), torch.view_as_real(x)), "the failing call must not have run"
        else:  # no wait of the first call gave up: then its results are right
            assert torch.equal(torch.view_as_real(y), torch.view_as_real(want))
        fft.fused_status()  # (clears whatever the second call left)
    assert gave_up >= 1, "neither a wait limit of 0 nor the raise-always hook produced a reported give-up"
    fft.set_fused_wait_limit(-1)  # (from here on deterministic: every fused launch leaves the word behind)
    y = x.clone()
    fft.process(y)
    assert fft.fused_status() == 1 and fft.fused_status() == 0  # reported once
    # host slices: the call succeeds with correct rows although its fused launches gave up
    # (rows of chunks whose launch gave up come from the two-launch kernels, the others from the fused one: equal up to rounding)
    hx = x.cpu().numpy()
    hwant = want.cpu().numpy()
    a = hx.copy()
    fft.process(a)
    assert float(np.abs(a - hwant).max()) <= tol, "in place"
    out = np.empty_like(hx)
    fft.process_immutable_with_scratch(hx, out)
    assert float(np.abs(out - hwant).max()) <= tol, "immutable"
    src = hx.copy()
    fft.process_outofplace_with_scratch(src, out)
    assert float(np.abs(out - hwant).max()) <= tol, "out of place"
    assert fft.fused_status() == 0
    fft.set_fused_wait_limit(1 << 21)
    for _ in range(3):
        y = x.clone()
        fft.process(y)
        assert torch.equal(torch.view_as_real(y), torch.view_as_real(want))
    assert fft.fused_status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,log2n,batch", [(np.complex64, 16, 512), (np.complex64, 20, 96), (np.complex128, 19, 64)])
def test_fused_giveup_is_visible_without_a_second_call(dtype, log2n, batch):
    """Round 6 (the review's asynchronous hole): `process_dev` -> the caller synchronises the stream ITSELF -> reads the result -> never calls
    the plan again.  The verdict must still be there: (a) the data says so -- every row a give-up touched carries NaN, and every row WITHOUT
    NaN is the right answer; (b) `synchronize()` (mi355fft_plan_synchronize) raises, once.  Real give-ups through a wait limit of 0."""
    import torch

    import rustfft_amd

    n = 1 << log2n
    tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
    fft = rustfft_amd.FftPlanner(dtype).plan_fft_forward(n)
    assert fft.is_fused()
    x = torch.empty(batch * n, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    want = x.clone()
    fft.process(want)
    fft.synchronize()  # healthy launch: no error
    stream = torch.cuda.current_stream()
    fft.set_fused_wait_limit(0)
    seen = 0
    for it in range(16):
        y = x.clone()
        fft.process(y)  # enqueues fine whatever happens on the device
        stream.synchronize()  # the caller's own synchronisation: nothing of the library is asked
        rows = torch.view_as_real(y).reshape(batch, -1)
        bad = torch.isnan(rows).any(dim=1)
        good = ~bad
        # (a) rows without NaN are right, bit for bit (the fused launch is bit-repeatable)
        assert torch.equal(rows[good], torch.view_as_real(want).reshape(batch, -1)[good]), (it, int(bad.sum()))
        # (b) the reporting synchronize: raises exactly when a wait gave up, and only once
        try:
            fft.synchronize()
            gave_up = False
        except rustfft_amd.FftPanic as e:
            gave_up = True
            assert e.status == 8 and "gave up waiting for a dependency" in str(e), str(e)
        assert gave_up == bool(bad.any()) or gave_up, "NaN rows without a reported give-up"
        if gave_up:
            seen += 1
            assert bool(bad.any()), "a wait gave up but no row carries NaN: the tile's output could be mistaken for data"
        fft.synchronize()  # reported once
    assert seen >= 1 or log2n < 19, "a wait limit of 0 produced no give-up in 16 launches"  # (2^16 meets its dependencies at the first poll almost always)
    fft.set_fused_wait_limit(1 << 21)
    y = x.clone()
    fft.process(y)
    fft.synchronize()
    assert torch.equal(torch.view_as_real(y), torch.view_as_real(want))


_HOG = r"""
import random, sys, time, torch
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)   # 1 GiB each: HBM traffic
c = torch.rand(1 << 24, device="cuda")                                                  # CU-bound transcendental chains
print("ready", flush=True)
t_end = time.time() + float(sys.argv[1])
rnd = random.Random(7)
while time.time() < t_end:
    for _ in range(rnd.randint(1, 6)):
        b.copy_(a)
    for _ in range(rnd.randint(0, 4)):
        c = torch.sin(c) * 1.0001 + 0.1
    if rnd.random() < 0.5:
        torch.cuda.synchronize()
        time.sleep(rnd.uniform(0.0, 0.004))
torch.cuda.synchronize()
"""


def test_fused_launch_under_a_second_process(planners):
    """The guide's protocol for cross-workgroup hand-offs (MI355X_MICROARCH.md, "inter-workgroup visibility"): test under UNEVEN load with the
    consumer's L1 warm and compare every word.  A second PROCESS hogs HBM and the CUs in bursts with random sleeps while this one runs
    >= 200 fused launches per size, alternating between two inputs (the ring a launch reads was filled with the OTHER input's intermediate
    by the launch before: a consumer CU that served a stale L1 line, or a slot rewritten too early, changes words), every output word
    compared with the two-launch plan's, the sticky error word 0 throughout.  2^16, 2^20, 2^21 (narrow later tile) and the 2^23 unit launch,
    both precisions."""
    import subprocess
    import sys

    import torch

    import rustfft_amd

    hog = subprocess.Popen([sys.executable, "-c", _HOG, "150"], stdout=subprocess.PIPE, text=True)
    try:
        assert hog.stdout.readline().strip() == "ready"
        launches = 0
        for dtype, tdt in ((np.complex64, torch.complex64), (np.complex128, torch.complex128)):
            for log2n in (16, 20, 21, 23):
                n = 1 << log2n
                batch = max(24, (1 << 28 if dtype == np.complex64 else 1 << 27) // n // (8 if dtype == np.complex64 else 16))
                fft = rustfft_amd.FftPlanner(dtype).plan_fft_forward(n)
                ref = rustfft_amd.FftPlanner(dtype).plan_fft_forward(n)
                ref.set_fused(0)
                assert fft.is_fused(), (dtype, log2n)
                xs, wants = [], []
                for seed in (1, 2):
                    x = torch.empty(batch * n, dtype=tdt, device="cuda")
                    torch.view_as_real(x).uniform_(-1.0, 1.0)
                    w = x.clone()
                    ref.process(w)
                    xs.append(x)
                    wants.append(w)
                y = torch.empty_like(xs[0])
                # the fused kernel is another compilation of the same bodies (f32 2^21: another later tile): equal to the two-launch plan up to
                # rounding, and bit-identical from launch to launch -- every word of every launch is compared with the first launch's
                firsts = [None, None]
                for r in range(200):
                    i = r & 1
                    fft.process_immutable_with_scratch(xs[i], y)
                    if firsts[i] is None:
                        firsts[i] = y.clone()
                        # (2^17, 2^18, 2^21 in Complex<f32>: the two-launch plan is the BALANCED split, other tile heights than the fused pair's)
                        assert float((y - wants[i]).abs().max()) <= (1.5e-6 if dtype == np.complex64 else 4e-15) * float(wants[i].abs().max()), (dtype, log2n)
                    else:
                        assert torch.equal(torch.view_as_real(y), torch.view_as_real(firsts[i])), (dtype, log2n, r)
                    if r % 50 == 49:
                        assert fft.fused_status() == 0, (dtype, log2n, r)
                    launches += 1
                assert hog.poll() is None, "the hog process must still be running beside the fused launches"
        print("fused launches compared under a second process:", launches)
    finally:
        hog.kill()
        hog.wait()
