"""Kernel-body checks without a GPU.  tests/emu compiles the SAME kernel bodies (rustfft_amd/csrc/*.h,
*.hip) for the host with an executor that runs every GPU thread of a workgroup phase by phase
(launch.h, MI355_EMU), behind the same C ABI.  This validates all index arithmetic, LDS exchange
patterns, twiddle tables, buffer rotation and the API semantics against the oracle; it says nothing about
performance and is never loaded by the product."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import check_fft_algorithm, compare_vectors, numpy_fft, random_signal, rel_l2, zero_mean_signal

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu_planner():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s"])
    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(EMU_DIR, "libmi355fft_emu.so"))
    return lambda dtype: rustfft_amd.FftPlannerHip(dtype, lib=lib)


@pytest.fixture(scope="module")
def emu_tuning_planner():
    """The small tuning build of the emulator (`make -C tests/emu tuning`: power-of-two, Rader and Bluestein units with their
    tuning variants) for the tests that select a variant by number; every other test runs the SHIPPED registry."""
    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s", "tuning"])
    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(EMU_DIR, "libmi355fft_emu_tuning.so"))
    return lambda dtype: rustfft_amd.FftPlannerHip(dtype, lib=lib)


POW2_SINGLE = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
POW2_MULTI = [1 << 16, 1 << 17, 1 << 18, 1 << 19]  # 2^13 .. 2^15 are single split-exchange kernels (test_single_kernel_above_4096)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_single_kernel_sizes_all_api_modes(emu_planner, oracle, dtype):
    planner = emu_planner(dtype)
    for n in [0, 1] + POW2_SINGLE:
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            ref = oracle.plan(dtype, n, d)  # the reference's scalar path (Radix4 over Butterfly8/16 etc.)
            check_fft_algorithm(fft, n, d, reference=ref)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_multi_pass_sizes_all_api_modes(emu_planner, oracle, dtype):
    planner = emu_planner(dtype)
    for n in POW2_MULTI:
        d = n.bit_length() % 2
        fft = planner.plan_fft(n, d)
        assert "k2first" in fft.describe() and "k2later" in fft.describe()
        check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)


def test_config2_and_three_pass_shapes(emu_planner, oracle):
    """BASELINE config 2 (N = 2^20, f32, forward + inverse), config 5's N = 2^22 (two passes of 2048-row tiles) and a
    three-pass length (2^23), small batch."""
    planner = emu_planner(np.complex64)
    for n, batch in ((1 << 20, 2), (1 << 22, 1), (1 << 23, 1)):
        x = zero_mean_signal(n * batch, np.complex64)
        fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
        y = x.copy()
        fwd.process(y)
        want = x.copy()
        oracle.plan(np.complex64, n, 0).process(want)
        assert compare_vectors(want, y)
        assert rel_l2(y, numpy_fft(x, n, False)) < 2e-6
        inv.process(y)  # round trip: ifft(fft(x)) == N x
        assert rel_l2(y / n, x) < 2e-6
    assert planner.plan_fft_forward(1 << 22).describe().count("k2") == 2  # 2048 x 2048
    assert planner.plan_fft_forward(1 << 23).describe().count("k2") == 3


def test_chunked_workspace_matches_unchunked(emu_planner):
    planner = emu_planner(np.complex64)
    n, batch = 1 << 13, 7
    x = random_signal(n * batch, np.complex64)
    fft = planner.plan_fft_forward(n)
    a = x.copy()
    fft.process(a)
    fft.set_chunk_batch(2)
    b = x.copy()
    fft.process(b)
    fft.set_chunk_batch(0)
    assert np.array_equal(a, b)


def test_host_slices_chunk_pipeline_and_threads(emu_planner, oracle):
    """The host-slice path (capi.cpp process_host): a call larger than one staging chunk runs as a two-thread pipeline over row
    chunks (upload + kernels on the calling thread, download on a helper) through a staging context of the plan's pool; four
    threads sharing one plan (examples/concurrency.rs:9-30) each take their own context.  All three API modes, ragged last
    chunk, a multi-pass plan (workspace per context stream), results identical to the one-chunk path."""
    import threading

    os.environ["MI355FFT_HOST_CHUNK_KIB"] = "64"
    try:
        planner = emu_planner(np.complex64)
        for n, batch in ((1024, 37), (1 << 16, 5), (1009, 23)):
            fft = planner.plan_fft_forward(n)
            x = zero_mean_signal(n * batch, np.complex64, seed=n)
            want = x.copy()
            oracle.plan(np.complex64, n, 0).process(want)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, want) < 2e-6
            out = np.zeros_like(x)
            fft.process_immutable_with_scratch(x, out)
            assert np.array_equal(out, y)
            src, out2 = x.copy(), np.zeros_like(x)
            fft.process_outofplace_with_scratch(src, out2)
            assert np.array_equal(out2, y)
        fft = planner.plan_fft_forward(1024)
        xs = [zero_mean_signal(1024 * 29, np.complex64, seed=s) for s in range(4)]
        wants = []
        for x in xs:
            w = x.copy()
            oracle.plan(np.complex64, 1024, 0).process(w)
            wants.append(w)
        errs = []

        def work(i):
            try:
                for _ in range(3):
                    y = xs[i].copy()
                    fft.process(y)
                    assert rel_l2(y, wants[i]) < 2e-6
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
    finally:
        del os.environ["MI355FFT_HOST_CHUNK_KIB"]


def test_validation_semantics(emu_planner):
    """src/common.rs:13-104 messages; partial trailing chunk reported AFTER the complete chunks ran
    (src/array_utils.rs:164-176); empty buffer accepted; len 0 is a no-op (src/fft_helper.rs:16-18)."""
    import rustfft_amd

    planner = emu_planner(np.complex64)
    f = planner.plan_fft_forward(64)
    with pytest.raises(rustfft_amd.FftPanic, match="Provided FFT buffer was too small. Expected len = 64, got len = 10"):
        f.process(np.zeros(10, np.complex64))
    x = random_signal(64 * 2 + 5, np.complex64)
    y = x.copy()
    with pytest.raises(rustfft_amd.FftPanic, match="must be a multiple of FFT length. Expected multiple of 64, got len = 133"):
        f.process(y)
    assert compare_vectors(y[:128], numpy_fft(x[:128], 64, False)) and np.array_equal(y[128:], x[128:])
    with pytest.raises(rustfft_amd.FftPanic, match=r"Got input.len\(\) = 64, output.len\(\) = 128"):
        f.process_outofplace_with_scratch(np.zeros(64, np.complex64), np.zeros(128, np.complex64))
    f.process(np.zeros(0, np.complex64))
    planner.plan_fft_forward(0).process(np.zeros(0, np.complex64))
    one = random_signal(3, np.complex64)
    o2 = one.copy()
    planner.plan_fft_forward(1).process(o2)
    assert np.array_equal(one, o2)
    assert f.get_inplace_scratch_len() == 0 and f.get_outofplace_scratch_len() == 0 and f.get_immutable_scratch_len() == 0
    assert f.len() == 64 and f.fft_direction() == rustfft_amd.FftDirection.Forward
    assert planner.plan_fft_forward(64) is f and planner.plan_fft_inverse(64) is not f  # fft_cache.rs:5-39


def test_unsupported_length_fails_loudly(emu_planner):
    import rustfft_amd

    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        emu_planner(np.complex64).plan_fft_forward((1 << 30) + 1)  # would need a 2^31-point inner transform


def test_linearity_and_shift(emu_planner):
    planner = emu_planner(np.complex128)
    n = 1 << 14
    fft = planner.plan_fft_forward(n)
    a, b = zero_mean_signal(n, np.complex128, 1), zero_mean_signal(n, np.complex128, 2)
    fa, fb, fc = a.copy(), b.copy(), (2.5 * a - 1j * b)
    fft.process(fa)
    fft.process(fb)
    fft.process(fc)
    assert rel_l2(fc, 2.5 * fa - 1j * fb) < 1e-13
    imp = np.zeros(n, np.complex128)
    imp[3] = 1.0
    fft.process(imp)
    assert rel_l2(imp, np.exp(-2j * np.pi * 3 * np.arange(n) / n)) < 1e-13


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("direction", [0, 1])
def test_accuracy_1_to_1000(emu_planner, oracle, dtype, direction):
    """tests/accuracy.rs:128-187 restated against the kernel bodies: every length 1..1000 (Bluestein bodies for
    the non-powers of two, Rader for 1009-class primes, native mixed radix where compiled) vs the reference's
    control = BluesteinsAlgorithm over Radix4 (tests/accuracy.rs:98-122), via the three API paths."""
    planner = emu_planner(dtype)
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


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_baseline_configs_3_and_4(emu_planner, oracle, dtype):
    """Config 3: N = 1200 (native mixed radix 10 x 10 x 12); config 4: N = 1009 (Rader over 16 x 9 x 7) and its
    Bluestein companion 1019 (M = 2048): all four API modes, ragged batch (tail workgroup partially filled)."""
    planner = emu_planner(dtype)
    for n, tag in ((1200, "k1<1200"), (1009, "rader<1008"), (1019, "bluestein<2048"), (719, "bluestein<1536")):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert tag in fft.describe(), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=5)
            x = zero_mean_signal(n * 3, dtype)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < (2e-6 if dtype == np.complex64 else 1e-13)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_bluestein_inner_length_ladder(emu_planner, oracle, dtype):
    """The one-kernel Bluestein bodies are compiled for M = 2^k, 3 * 2^k, 5 * 2^k and 7 * 2^k; the planner takes the
    smallest M >= 2n - 1 (never more than 1.25x of padding from M = 256 on; the reference pads to 2^k or 3 * 2^k,
    src/plan.rs:649-657).  One length per new M, both directions, against the reference's plan for that length."""
    import rustfft_amd

    planner = emu_planner(dtype)
    for n, M in ((150, 320), (200, 448), (300, 640), (401, 896), (601, 1280), (881, 1792), (1201, 2560), (1789, 3584),
                 (2311, 5120), (3581, 7168), (5003, 10240), (7001, 14336)):
        assert planner.bluestein_inner_len(n) == M, (n, planner.bluestein_inner_len(n))
        for d in (0, 1):
            fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_BLUESTEIN)
            assert fft.describe().startswith("bluestein<%d," % M), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
    # padding stays below 1.25x of the minimum from n = 129 up to the one-kernel limit
    for n in range(129, 7169):
        assert planner.bluestein_inner_len(n) * 4 < (2 * n - 1) * 5, n


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_lengths_beyond_one_workgroup(emu_planner, oracle, dtype):
    """Non-powers of two above 4096: a prime the reference plans as Rader (10007), a 'difficult' prime it plans as
    Bluestein (5759), a product of two large primes (101 * 103 -> MixedRadix): two-kernel Bluestein here while the padded
    length fits one workgroup, the fused multi-kernel Bluestein beyond (16411, 20011); and a smooth composite
    (5000 -> RadixN there, one split-exchange kernel here)."""
    import rustfft_amd

    planner = emu_planner(dtype)
    two_kernel_limit = 16384 if dtype == np.complex64 else 8192  # padded length <= 32768 (f32) / 16384 (f64) fits one workgroup
    for n in (4097, 5000, 5759, 10007, 101 * 103, 16411, 20011):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            if n == 5000:
                assert "k1<5000" in fft.describe()
            elif n == 101 * 103:  # round 3: two prime-tile passes (Rader inside the tile) instead of the two-kernel Bluestein
                assert fft.describe().startswith("k2rfirst<102,") and " -> k2rlater<100," in fft.describe(), fft.describe()
            elif n == 4097:  # round 6: 17 x 241 as the reference's MixedRadix over two Rader factors, one kernel (the LDS stage machine)
                assert fft.describe().startswith("lsm<mixed{rader241["), fft.describe()
            elif n == 5759:  # round 6: 13 x 443 (442 = 17 x 26): ten stages -- within the calibrated limit of both precisions above 4096
                assert fft.describe().startswith("lsm<mixed{rader443["), fft.describe()
            elif n <= 8192:  # round 2: ONE kernel -- split exchange, the spectrum handed over in registers (padded length <= 16384)
                assert fft.describe().startswith("bluestein<") and fft.describe().endswith(("s", "st1")), fft.describe()  # split exchange (+ staged tables)
            elif n <= two_kernel_limit:
                assert fft.describe().startswith("bluestein2_first<") and "bluestein2_second<" in fft.describe(), fft.describe()
            else:
                assert "bluestein_large" in fft.describe() and " fused: k2gfirst_chirp" in fft.describe(), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
    # three passes per inner transform (M = 640000), ragged batch
    fft = planner.plan_fft(300007, 0)
    assert fft.describe().count("->") == 4, fft.describe()
    x = zero_mean_signal(300007 * 2, dtype)
    y = x.copy()
    fft.process(y)
    assert rel_l2(y, numpy_fft(x, 300007, False)) < (2e-6 if dtype == np.complex64 else 1e-13)
    # the unfused fallback (separate chirp / multiply kernels around an inner plan) stays covered
    import os

    os.environ["MI355FFT_BLUESTEIN_UNFUSED"] = "1"
    try:
        fresh = emu_planner(dtype)
        for n in (8209, 10007):  # (lengths up to 8192 are one-kernel plans since round 2)
            fft = fresh.plan_fft_with(n, 1, algorithm=rustfft_amd.ALGO_BLUESTEIN)  # (round 6: AUTO plans 8209 as a Rader tree in the LDS stage machine)
            assert "bluestein_large" in fft.describe() and "fused" not in fft.describe()
            check_fft_algorithm(fft, n, 1, reference=oracle.plan(dtype, n, 1), n=2)
    finally:
        del os.environ["MI355FFT_BLUESTEIN_UNFUSED"]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_large_prime_rader(emu_planner, oracle, dtype):
    """Primes beyond one workgroup whose p - 1 factors into general tile heights (the reference plans RadersAlgorithm for them,
    src/plan.rs:636-665): the multi-kernel Rader -- g^j gather on the first load of the first inner transform, spectrum multiply
    + x[0] / X[0] step on its last store, g^-j scatter on the last store of the second (k2g_body FUSE 4 / 5 / 6).  4481 lies below
    AUTO's threshold (one-kernel Bluestein there) and is asked for as a host planner would; 65537 has a power-of-two inner length;
    also the host planner's own inner_fft_data (raders_algorithm.rs:87-113) and a ragged batch."""
    import rustfft_amd

    planner = emu_planner(dtype)
    for p, algo in ((4481, rustfft_amd.ALGO_RADER), (12289, rustfft_amd.ALGO_AUTO), (40961, rustfft_amd.ALGO_AUTO), (41959, rustfft_amd.ALGO_AUTO), (65537, rustfft_amd.ALGO_RADER)):
        # (41959 - 1 = 2 * 3^4 * 7 * 37: a PRIME tile height, Rader inside the tile, between the fused gather / multiply passes)
        for d in (0, 1):
            fft = planner.plan_fft_with(p, d, algorithm=algo)
            assert fft.describe().startswith("rader_large(p-1=%d fused: k2gfirst_gather<" % (p - 1)), fft.describe()
            assert "k2glast_rmul<" in fft.describe() and "k2glast_scatter<" in fft.describe(), fft.describe()
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=2 if p < 20000 else 1)
    # AUTO below the threshold: Bluestein until round 5; round 6: one Rader over a 4480-point leaf in ONE kernel (the LDS stage machine, seven stages)
    # (Complex<f64>: the row, the leaf's tables and D do not fit the 160 KiB of a workgroup: Bluestein as before)
    auto = planner.plan_fft(4481, 0).describe()
    assert auto.startswith("lsm<rader4481[leaf4480(") if dtype == np.complex64 else "bluestein" in auto, auto
    # eleven rows: the gather / scatter passes run the tiles of transform g on XCD g % 8 for complete groups of eight transforms
    # and in the plain order for the rest -- both index maps in one call
    for p, algo in ((4481, rustfft_amd.ALGO_RADER), (12289, rustfft_amd.ALGO_AUTO)):
        x = zero_mean_signal(p * 11, dtype, seed=11)
        for d in (0, 1):
            y = x.copy()
            planner.plan_fft_with(p, d, algorithm=algo).process(y)
            assert rel_l2(y, numpy_fft(x, p, d == 1)) < (2e-6 if dtype == np.complex64 else 1e-13)
    p = 12289
    x = zero_mean_signal(p * 3, dtype, seed=5)
    for d in (0, 1):
        want = numpy_fft(x, p, d == 1)
        g = oracle.primitive_root(p)
        ginv, t = pow(g, p - 2, p), 1
        data = np.zeros(p - 1, dtype=dtype)
        for j in range(p - 1):
            data[j] = oracle.compute_twiddle(dtype, t, p, d) / (p - 1)
            t = t * ginv % p
        oracle.plan(dtype, p - 1, d).process(data)
        fft = planner.plan_fft_with(p, d, algorithm=rustfft_amd.ALGO_RADER, rader_inner_fft_data=data)
        y = x.copy()
        fft.process(y)
        assert rel_l2(y, want) < (2e-6 if dtype == np.complex64 else 1e-13)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_prime_tile_heights(emu_planner, oracle, dtype):
    """Composite lengths with prime factors above 31 (the reference plans MixedRadix over Rader / Bluestein inner FFTs for them,
    src/plan.rs:474-506, src/algorithm/mixed_radix.rs:53-158): column-tile passes whose tile height is the prime, Rader inside
    the tile (kernels.h k2r_body).  101 x 103 (both passes prime tiles, 102 = 2 * 3 * 17: a prime-radix inner schedule), a prime
    tile next to a smooth one (64 x 131), three passes (37 * 41 * 43), a ragged batch of columns (47 x 229: 229 columns are no
    multiple of the tile width), and 37 x 41 the way a host planner asks for it (AUTO keeps the one-kernel Bluestein there)."""
    import rustfft_amd

    planner = emu_planner(dtype)
    for n in (101 * 103, 64 * 131, 37 * 41 * 43, 47 * 229, 89 * 97):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            # (round 6: the LDS stage machine goes ahead of the prime-tile passes only up to 8192 -- 5328 = 144 x 37 below; these lengths keep the passes)
            assert "k2rfirst<" in fft.describe() or "k2rlater<" in fft.describe(), fft.describe()
            assert "bluestein" not in fft.describe(), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
    # 37 x 41, at or below 4096: until round 5 AUTO kept the one-kernel Bluestein and a host planner's MixedRadix request got two prime-tile passes
    # through HBM; round 6: both get the reference's tree in ONE kernel (the LDS stage machine: seven stages -- within AUTO's calibrated limit
    # in both precisions); the prime-tile passes at this size stay reachable through an explicit six-step recipe
    n = 37 * 41
    auto = planner.plan_fft(n, 0).describe()
    assert auto.startswith("lsm<mixed{rader41["), auto
    for d in (0, 1):
        fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_MIXED_RADIX)
        assert fft.describe().startswith("lsm<mixed{rader41[leaf40("), fft.describe()
        check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_general_column_tile_passes(emu_planner, oracle, dtype):
    """7-smooth lengths above one workgroup (kernels.h k2g_body): 2, 3 and 4 passes, tile heights that do not divide
    the strides (per-column b mod s), ragged last tiles (M not a multiple of F), pure powers of 3 and 5.  The
    reference plans these as RadixN / MixedRadix (src/plan.rs:430-560)."""
    planner = emu_planner(dtype)
    # (round 2: tile heights with the factors 11 and 13 too -- 40898 = 2 11^2 13^2, 45056 = 11 * 2^12, 36608 = 2^8 11 13; round 5: such lengths up to
    # 16384 (Complex<f32>: most up to 32768) are whole-row kernels, test_single_kernel_above_4096)
    for n, npass in ((36864, 2), (39366, 2), (50000, 2), (44100, 2), (78125, 2), (98304, 2), (100000, 2), (1000000, 3), (3686400, 3),
                     (40898, 2), (45056, 2), (36608, 2), (157300, 2)):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            desc = fft.describe()
            want = 2 if (n >= 1000000 and dtype == np.complex128) else npass  # f64 has the tall split tiles: 1000 x 1000
            assert desc.startswith("k2gfirst") and desc.count("->") == want - 1, desc
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d) if n <= 100000 else None, n=2 if n <= 100000 else 1)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_single_kernel_above_4096(emu_planner, oracle, dtype):
    """2^13 .. 2^15 and the 7-smooth lengths up to 16384 run as ONE kernel whose LDS exchange moves the real and the imaginary
    plane one after the other (engine.h SPLIT): all API modes vs the oracle's plan, every radix mix of the generated list."""
    planner = emu_planner(dtype)
    sizes = [4116, 4375, 5000, 6561, 8192, 10000, 12288, 16384] + ([14406, 15625, 16200, 19683, 25000, 32768] if dtype == np.complex64 else [])
    sizes += [4125, 4459, 5005, 9009, 13312] + ([15015, 16380, 16562, 20449, 26325] if dtype == np.complex64 else [])  # round 5: factors 11 / 13 (kernels_smooth4_*; f32 also with 32 values per thread, up to 32768)
    # round 5, late: lengths with a prime factor 17 .. 31 above the smooth3 limits (kernels_smooth5_*: every prime radix, 32 values per thread in
    # f32, f64 through the plain exchange up to 4096 and the split one above); Bluestein until then
    sizes += [4352, 4495, 5239, 6800, 7429, 8160, 8184] + ([2052, 2185, 3400, 3553, 4080] if dtype == np.complex128 else []) + [8704, 9248, 12121, 16337, 16368]  # (the last five: the tier up to 16384)
    for n in sizes:
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_runtime_scheduled_kernels(emu_planner, oracle, dtype):
    """dyn_engine.h: 13-smooth lengths through the run-time scheduled mixed-radix kernel (the RadixN analogue,
    src/algorithm/radixn.rs:497-541 covers factors 2..7 over small bases; here every compiled radix appears) and
    primes with 13-smooth p - 1 through the run-time scheduled Rader (raders_algorithm.rs:302-309: primes < 100)."""
    planner = emu_planner(dtype)
    for n in [4368, 4459, 4620, 5005]:  # (round 1 planned these through the run-time scheduled kernel, rounds 2 - 4 as column-tile plans; whole-row kernels now)
        assert planner.plan_fft(n, 0).describe().startswith("k1<%d," % n)
    primes = [p for p in range(5, 100) if all(p % q for q in range(2, int(p**0.5) + 1))] + [127, 211, 257, 331, 1201, 2311, 3001]
    import rustfft_amd

    def smooth13(v):  # (31-smooth since the run-time scheduled kernels know the prime radices 17 .. 31)
        for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
            while v % q == 0:
                v //= q
        return v == 1

    for p in primes:  # MI355FFT_ALGO_RADER: the host planner's Recipe::RadersAlgorithm -> run-time scheduled Rader
        for d in (0, 1):
            fft = planner.plan_fft_with(p, d, algorithm=rustfft_amd.ALGO_RADER)
            if not smooth13(p - 1):  # round 6: Rader over MixedRadix over Rader, the reference's recursion (src/plan.rs:636-665), in the LDS stage machine
                assert fft.describe().startswith("lsm<rader%d[mixed{rader" % p), (p, fft.describe())
            assert "rader" in fft.describe(), (p, fft.describe())
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=3)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_compiled_rader_bodies_every_form(emu_planner, oracle, dtype):
    """tools/gen_rader_kernels.py picks a body form per prime (rows side by side, or the rows loop with / without prefetch at
    two register budgets; by measurement where the forms compete): primes of every form, including the ones whose row pitch
    leaves no spare slot past the exchange span (1297, 2003, 2081: the rows loop sizes its own buffer), the primes below 800
    that take a wider schedule to reach 64 threads per row, and f64 rows-loop bodies; both directions, ragged batch, against
    the reference's plan."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_rader_kernels as gen  # the per-prime choices are measured ones (its MODE1_BACK / F64_ROWS / WIDE_ROWS ... lists)

    planner = emu_planner(dtype)
    f32 = dtype == np.complex64
    prec = 32 if f32 else 64
    primes = ([97, 127, 193, 257, 271, 449, 541, 769, 811, 1009, 1201, 727, 883, 1297, 2003, 2081, 4051, 4057, 137, 647, 683, 2143] if f32 else
              [97, 127, 193, 257, 541, 727, 811, 883, 937, 1409, 911, 1201, 1297, 2081, 2801, 3697, 4057, 613, 2053, 3911])  # (the last ones: p - 1 has a factor 17 .. 31)

    def form(p):
        mode = gen.choose(p, prec)[1]
        if f32 and p in gen.MODE3_F32 and mode in (2, 4):
            return "m3"  # the Complex<f32> rows loops that measured faster without the prefetch (and without the SLP vectoriser)
        return "m5" if mode == 1 and (prec, p) in gen.MODE5 else "m%d" % mode

    want = {p: form(p) for p in primes}
    assert {"m1", "m2", "m3", "m4", "m5"} <= set(want.values()) if f32 else {"m1", "m3", "m5"} <= set(want.values()), want
    for p, form in want.items():
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            assert fft.describe().startswith("rader<%d," % (p - 1)) and fft.describe().endswith(form), (p, fft.describe())
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=11)  # 11 rows: one full group of 8 and a ragged one


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_rader_bodies_of_the_31_smooth_primes(emu_planner, oracle, dtype):
    """Round 5: the primes <= 4096 whose p - 1 has a prime factor 17 .. 31 and that measured faster through a compiled Rader body (prime-radix
    sub-passes in the inner transforms) than through the one-kernel Bluestein: 89 Complex<f32> / 77 Complex<f64> more than rounds 2 - 4 had
    (tools/gen_rader_kernels.py EXTRA31_R5).  The reference's planner takes Rader for none of them (src/plan.rs:636-665: Bluestein when p - 1 has
    a factor above 23 ... its own inner-length rule) -- the result must still be the reference's, so a sample of every form (side by side, with
    the register hand-over, compiled with / without the SLP vectoriser) runs both directions, ragged batch, against the reference's plan; the
    three primes with no side-by-side layout (929, 1217, 3041) stay with Bluestein."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_rader_kernels as gen

    planner = emu_planner(dtype)
    prec = 32 if dtype == np.complex64 else 64
    new = sorted(p for (pr, p) in gen.EXTRA31_R5 if pr == prec and (pr, p) not in gen.EXTRA31_R2)
    assert len(new) == (89 if prec == 32 else 77)
    for p in new:  # every one of them plans as a Rader body of its measured form
        want = "m5" if (prec, p) in gen.MODE5 else "m1"
        d = planner.plan_fft(p, 0).describe()
        assert d.startswith("rader<%d," % (p - 1)) and d.endswith(want), (p, d)
    for p in (929, 1217, 3041):
        assert planner.plan_fft(p, 0).describe().startswith("bluestein<"), p
    sample = ([47, 59, 103, 523, 1013, 1021, 1117, 1451, 2053, 3469, 4093] if prec == 32 else [47, 137, 191, 457, 1013, 1103, 2143, 3313, 4093])
    assert set(sample) <= set(new)
    for p in sample:
        for d in (0, 1):
            fft = planner.plan_fft(p, d)
            rows = int(re.search(r"xF(\d+)", fft.describe()).group(1))  # rows side by side in one workgroup
            check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=rows + 3)  # one full workgroup and a ragged one


def _thirteen_smooth(limit):
    s = {1}
    for p in (2, 3, 5, 7, 11, 13):
        s = {v * p**k for v in s for k in range(0, 13) if v * p**k <= limit}
    return sorted(v for v in s if v > 2 and (v & (v - 1)))


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_compiled_smooth_schedules(emu_planner, dtype):
    """Every 13-smooth length in [3, 4096] has its own compiled schedule (tools/gen_smooth_kernels.py — the lengths the
    reference plans as RadixN, src/plan.rs:508-607): forward and inverse, ragged batch, vs numpy in float64."""
    planner = emu_planner(dtype)
    tol = 1e-6 if dtype == np.complex64 else 1e-14
    for n in _thirteen_smooth(4096):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
        x = zero_mean_signal(n * 3, dtype, seed=n)
        y = x.copy()
        planner.plan_fft(n, n % 2).process(y)
        assert rel_l2(y, numpy_fft(x, n, n % 2 == 1)) < tol, n


def test_host_planner_options(emu_planner, oracle):
    """mi355fft_plan_create_ex on the kernel-body emulator: see helpers.check_host_planner_options (the same checks run on the
    device in tests/test_gpu_parity.py::test_host_planner_options_on_device)."""
    from helpers import check_host_planner_options

    check_host_planner_options(emu_planner(np.complex64), oracle)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_host_planner_recipe(emu_planner, oracle, dtype):
    """mi355fft_plan_options.recipe on the kernel-body emulator: see helpers.check_host_planner_recipe (the same checks, plus
    the large lengths, run on the device in tests/test_gpu_parity.py::test_host_planner_recipe_on_device)."""
    from helpers import check_host_planner_recipe

    check_host_planner_recipe(emu_planner(dtype), oracle, dtype, big=False)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_host_slices_need_only_element_alignment(emu_planner, oracle, dtype):
    """A Rust `&mut [Complex<T>]` guarantees align_of::<T>() only -- 4 bytes for Complex<f32>, 8 for Complex<f64> (SURVEY
    section 8(b) layout rule): the host-slice entry points stage through plain byte copies, so a buffer that starts one
    float past a 16-byte boundary works in all three modes (single-kernel and multi-pass plans, multi-row batches)."""
    real = np.float32 if dtype == np.complex64 else np.float64
    planner = emu_planner(dtype)
    for n in (1009, 1200, 1 << 16):
        rows = 3
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            x = random_signal(rows * n, dtype, seed=n + d)
            want = x.copy()
            oracle.plan(dtype, n, d).process(want)

            def skewed(values=None):
                raw = np.zeros(2 * rows * n + 8, dtype=real)
                base = raw.ctypes.data
                start = ((-base) % 16) // raw.itemsize + 1  # element index just past a 16-byte boundary
                view = raw[start:start + 2 * rows * n].view(dtype)
                assert view.ctypes.data % 16 == raw.itemsize and view.flags.c_contiguous
                if values is not None:
                    view[:] = values
                return view

            a = skewed(x)
            fft.process(a)
            assert compare_vectors(want, a), (n, d, "in place")
            src, dst = skewed(x), skewed()
            fft.process_outofplace_with_scratch(src, dst, np.zeros(fft.get_outofplace_scratch_len(), dtype=dtype))
            assert np.array_equal(dst, a), (n, d, "out of place")
            src, dst = skewed(x), skewed()
            fft.process_immutable_with_scratch(src, dst, np.zeros(fft.get_immutable_scratch_len(), dtype=dtype))
            assert np.array_equal(dst, a) and np.array_equal(src, x), (n, d, "immutable")


def test_random_recipe_trees(emu_planner):
    """Random VALID recipe trees (random factorisation trees of random lengths, random split kinds, Rader / Bluestein roots where
    the arithmetic allows) never produce a wrong transform: every accepted plan matches numpy in complex128, a refused one is
    refused with UNSUPPORTED (no kernel for that family) -- and random CORRUPTIONS of a valid tree (a length or a child index
    changed) are refused with INVALID_ARG or still compute the right transform of the plan's own length."""
    import ctypes

    import rustfft_amd
    from rustfft_amd import Recipe, _native

    rng = np.random.default_rng(20240917)
    planner = emu_planner(np.complex64)

    def is_prime(n):
        return n > 1 and all(n % q for q in range(2, int(n**0.5) + 1))

    def tree(n, depth=0):
        divs = [q for q in range(2, int(n**0.5) + 1) if n % q == 0]
        if divs and depth < 3 and rng.random() < 0.75:
            q = int(rng.choice(divs))
            kind = int(rng.choice([Recipe.MIXED_RADIX, Recipe.GOOD_THOMAS, Recipe.MIXED_RADIX_SMALL, Recipe.GOOD_THOMAS_SMALL]))
            return Recipe.mixed_radix(tree(q, depth + 1), tree(n // q, depth + 1), kind)
        if is_prime(n) and n > 3 and rng.random() < 0.5:
            return Recipe.raders(tree(n - 1, depth + 1))
        if n > 2 and rng.random() < 0.15:
            m = 1 << int(np.ceil(np.log2(2 * n - 1)))
            if m // 4 * 3 >= 2 * n - 1 and rng.random() < 0.5:
                m = m // 4 * 3  # the reference's other inner-length family, 3 * 2^k (plan.rs:649-657)
            return Recipe.bluesteins(n, Recipe.dft(m))
        return Recipe.dft(n) if rng.random() < 0.5 else Recipe.butterfly(n)

    def smooth_length():
        n = 1
        while n < 40:
            n *= int(rng.choice([2, 2, 2, 3, 3, 5, 7, 11, 13]))
        while n < 30000 and rng.random() < 0.6:
            n *= int(rng.choice([2, 2, 3, 5, 7, 11, 13, 37, 41, 101]))
        return n

    lengths = ([int(v) for v in rng.integers(2, 20000, 15)] + [smooth_length() for _ in range(30)] +
               [1517, 10403, 4096 * 3, 1 << 14, 1 << 16, 37 * 64, 1009, 4099, 719 * 2])
    accepted = refused = 0
    for n in lengths:
        t = tree(n)
        assert t.len == n
        try:
            fft = planner.plan_fft_with(n, n % 2, recipe=t)
        except rustfft_amd.FftPanic as e:
            assert e.status == 6, (n, e)  # MI355FFT_ERR_UNSUPPORTED: the family has no kernel at this length; never INVALID_ARG for a valid tree
            refused += 1
            continue
        accepted += 1
        x = random_signal(2 * n, np.complex64, seed=n)
        y = x.copy()
        fft.process(y)
        assert rel_l2(y, numpy_fft(x, n, n % 2 == 1)) < 5e-6, (n, fft.describe())
    assert accepted >= 25 and refused >= 1, (accepted, refused)

    # corruptions of a valid tree
    lib, prec = planner._lib, planner._prec
    for trial in range(200):
        n = int(rng.choice([1517, 10403, 1 << 14, 3 * 4096, 1200, 1009]))
        nodes = tree(n).flatten()
        k = len(nodes)
        i = int(rng.integers(0, k))
        what = int(rng.integers(0, 4))
        if what == 0:
            nodes[i].len = int(rng.integers(0, 3 * n))
        elif what == 1:
            nodes[i].left = int(rng.integers(-3, k + 3))
        elif what == 2:
            nodes[i].right = int(rng.integers(-3, k + 3))
        else:
            nodes[i].kind = int(rng.integers(-2, 14))
        o = _native.PlanOptions()
        o.struct_size = ctypes.sizeof(_native.PlanOptions)
        o.recipe = ctypes.cast(nodes, ctypes.POINTER(_native.RecipeNode))
        o.recipe_nodes = k
        h = ctypes.c_void_p()
        rc = lib.mi355fft_plan_create_ex(n, 0, prec, ctypes.byref(o), ctypes.byref(h))
        assert rc in (0, 6, 7), (rc, trial)
        if rc == 0:  # the corruption left a valid tree (or hit a field the variant ignores): the transform is still the length-n one
            fft = rustfft_amd.Fft(lib, h, np.complex64)
            x = random_signal(n, np.complex64, seed=trial)
            y = x.copy()
            fft.process(y)
            assert rel_l2(y, numpy_fft(x, n, False)) < 5e-6, (n, trial, fft.describe())


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_prime_radices_17_to_31(emu_planner, oracle, dtype):
    """The reference's Butterfly17 .. Butterfly31 (src/algorithm/butterflies.rs:1582-6241) as in-register prime radices:
    lengths with such a factor run as mixed-radix kernels -- compiled schedules up to 16384 (rounds 2 - 4: 4096, f64 2048; the
    one-kernel Bluestein above), the run-time scheduled HEAVY kernel on a host planner's request -- and no longer through Bluestein."""
    planner = emu_planner(dtype)
    for n in (17, 19, 23, 29, 31, 34, 51, 93, 289, 323, 437, 899, 961, 992, 1023):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
    import rustfft_amd

    for n in (1088, 1734, 2465, 3553, 4092, 4352, 6448):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("k1<%d," % n), (n, fft.describe())  # compiled up to 16384 (round 5: kernels_smooth5_*)
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
    for n in (17408, 18496):  # 2^10 x 17, 2^6 x 17^2: above the compiled range (16384): Bluestein (the run-time scheduled HEAVY kernel ends at 8192 and measured slower than it there)
        fft = planner.plan_fft(n, 0)
        assert "bluestein" in fft.describe(), (n, fft.describe())
        check_fft_algorithm(fft, n, 0, reference=oracle.plan(dtype, n, 0), n=2)
    fft = planner.plan_fft_with(6448, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX)  # a host planner's MixedRadix request: a mixed-radix kernel (compiled now; the run-time scheduled one before)
    assert fft.describe().startswith("k1<6448,") or "dyn_k1" in fft.describe(), fft.describe()
    check_fft_algorithm(fft, 6448, 0, reference=oracle.plan(dtype, 6448, 0), n=2)
    fft = planner.plan_fft(4913, 0)  # 17^3: three 17-point sub-passes on 289 threads per row (Bluestein until round 5)
    assert fft.describe().startswith("k1<4913, 289, 17, 17, 17>"), fft.describe()
    check_fft_algorithm(fft, 4913, 0, reference=oracle.plan(dtype, 4913, 0), n=2)


def test_pair_fused_column_tiles(emu_tuning_planner, oracle):
    """engine.h pair-fused sub-passes (first LDS exchange replaced by a lane-pair register exchange): the emulator's
    executor swaps the register slots of threads t and t + 32 exactly as v_permlane32_swap does.  Selected here through the
    tuning variant number (the emulator's tuning build)."""
    os.environ["MI355FFT_VARIANT"] = "12"
    try:
        planner = emu_tuning_planner(np.complex64)
        for n in (1 << 18, 1 << 19, 1 << 20):
            for d in (0, 1):
                fft = planner.plan_fft(n, d)
                assert "p" in fft.describe().split("->")[-1].split("xF")[-1], fft.describe()
                check_fft_algorithm(fft, n, d, reference=oracle.plan(np.complex64, n, d), n=2)
    finally:
        del os.environ["MI355FFT_VARIANT"]


@pytest.mark.parametrize("order", ["forward", "reverse"])
def test_rader_register_handover(emu_tuning_planner, oracle, order):
    """kernels.h rader_body MODE 5: the side-by-side Rader body whose second inner transform runs the reversed schedule, so the
    d[] multiply and the x[0] / X[0] step (raders_algorithm.rs:256-262) happen in registers and the spectrum never goes
    through LDS.  Primes of every schedule shape (two to four sub-passes, padded and unpadded layouts, different row pitches
    of the two schedules), both precisions and directions, ragged batches, also with the threads of every phase in reverse
    order (the x[0] / X[0] slots live behind all rows because the two schedules' exchange spans differ)."""
    if order == "reverse":
        os.environ["MI355_EMU_ORDER"] = "reverse"
    try:
        # variant 5: the side-by-side body (MODE 5); variant 6: the rows loop with the same hand-over (MODE 6, a tuning variant: the
        # reversed schedule's factors in a second register block, the g^-j targets taken from ITS last pass)
        for variant, primes in (("5", (97, 193, 271, 1009, 4057)), ("6", (193, 541, 1009, 4051))):
            os.environ["MI355FFT_VARIANT"] = variant
            for dtype in (np.complex64, np.complex128):
                planner = emu_tuning_planner(dtype)
                for p in primes:
                    for d in (0, 1):
                        fft = planner.plan_fft(p, d)
                        assert fft.describe().startswith("rader<%d," % (p - 1)) and fft.describe().endswith("m%sv%s" % (variant, variant)), (p, fft.describe())
                        check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=5)
                        x = random_signal(37 * p, dtype)
                        y = x.copy()
                        fft.process(y)
                        assert rel_l2(y, numpy_fft(x, p, d == 1)) < (5e-6 if dtype == np.complex64 else 1e-13), (p, d)
    finally:
        os.environ.pop("MI355FFT_VARIANT", None)
        os.environ.pop("MI355_EMU_ORDER", None)


@pytest.mark.parametrize("order", ["forward", "reverse"])
def test_two_columns_per_lane_tiles(emu_tuning_planner, oracle, order):
    """Tuning variants 50 - 53 of the column tiles (launch.h DevExecPair, kernels.h k2_body ABL bit 4096): two virtual threads --
    adjacent columns of the same rows -- per physical thread, the even one moving both columns' rows as 16-byte accesses.  The
    emulator runs the virtual threads as ordinary threads (their register arrays are adjacent exactly as on the device), so this
    checks the schedules with 16 values per virtual thread, the separate load / store phases and their index arithmetic, in
    both thread orders; the pairing itself is the four lines of DevExecPair."""
    if order == "reverse":
        os.environ["MI355_EMU_ORDER"] = "reverse"
    try:
        for variant, n, tag in (("50", 1 << 20, "1024, 64, 8, 8, 16"), ("51", 1 << 20, "1024, 64, 16, 8, 8"), ("52", 1 << 22, "2048, 128, 8, 16, 16"),
                                ("53", 1 << 21, "2048, 128, 16, 16, 8")):
            os.environ["MI355FFT_VARIANT"] = variant
            planner = emu_tuning_planner(np.complex64)  # (a planner caches its plans per length: one per variant)
            for d in (0, 1):
                fft = planner.plan_fft(n, d)
                assert fft.describe().count(tag) >= 1 and "abl4" in fft.describe(), (variant, fft.describe())
                x = random_signal(2 * n, np.complex64, seed=int(variant))
                y = x.copy()
                fft.process(y)
                assert rel_l2(y, numpy_fft(x, n, d == 1)) < 5e-6, (variant, d, fft.describe())
                z = np.zeros_like(x)
                fft.process_immutable_with_scratch(x, z, np.zeros(0, dtype=np.complex64))
                assert np.array_equal(z, y), (variant, d, "immutable")
    finally:
        os.environ.pop("MI355FFT_VARIANT", None)
        os.environ.pop("MI355_EMU_ORDER", None)


@pytest.mark.parametrize("order", ["", "reverse"])
def test_every_generated_kernel_in_both_thread_orders(emu_planner, order):
    """Every GENERATED kernel -- the compiled whole-row schedules (kernels_smooth*_*: 2504 Complex<f32> / 2200 Complex<f64> lengths up to 32768 /
    16384, among them the 877 / 1102 prime-radix schedules of kernels_smooth5_* that the last third of round 5 added) and the compiled Rader bodies
    (kernels_rader_*: 231 / 225 primes, 89 / 77 of them new with a 31-smooth p - 1) -- runs its body on the emulator, in thread order and in
    reverse thread order (a race between threads of one phase shows up as a difference), two rows each, against numpy in float64.  (The GPU suite
    runs the same lengths on the device; the emulator takes about ten seconds for all of them.)"""
    import glob
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_rader_kernels as gr
    import gen_smooth_kernels as gs

    if order:
        os.environ["MI355_EMU_ORDER"] = order
    try:
        for dtype, prec, ty, tag, tol in ((np.complex64, 32, "float", "f32", 5e-6), (np.complex128, 64, "double", "f64", 1e-13)):
            planner = emu_planner(dtype)
            whole = set()
            for f in glob.glob(os.path.join(root, "rustfft_amd", "csrc", "kernels_smooth*_%s_*.hip" % tag)):
                whole |= {int(m) for m in re.findall(r'MI_K1X?\(%s, \d+, \d+, (?:true|false), (?:\d+, "\w*", )?(\d+),' % ty, open(f).read())}
            primes = set()
            for f in glob.glob(os.path.join(root, "rustfft_amd", "csrc", "kernels_rader_%s_*.hip" % tag)):
                primes |= {int(m) for m in re.findall(r"// p = (\d+)", open(f).read())}
            late = set(gs.big31_sizes(prec)) | {p for (pr, p) in gr.EXTRA31_R5 if pr == prec and (pr, p) not in gr.EXTRA31_R2}
            assert len(late) == (877 + 89 if prec == 32 else 1102 + 77) and late <= (whole | primes)
            assert len(whole) == (2504 if prec == 32 else 2200) and len(primes) == (231 if prec == 32 else 225), (len(whole), len(primes))
            for n in sorted(whole | primes):
                d = n % 2
                fft = planner.plan_fft(n, d)
                # (17 .. 31 have both a Rader body and a prime butterfly: AUTO takes the butterfly)
                assert fft.describe().startswith("rader<%d," % (n - 1) if n in primes and n not in whole else "k1<%d," % n), (n, fft.describe())
                x = random_signal(2 * n, dtype, seed=n)
                y = x.copy()
                fft.process(y)
                assert rel_l2(y, numpy_fft(x, n, d == 1)) < tol, (n, d, order, fft.describe())
    finally:
        if order:
            del os.environ["MI355_EMU_ORDER"]


def test_random_lengths_in_reverse_thread_order(emu_planner):
    """400 lengths per precision drawn log-uniformly from 2 .. 120000 through AUTO (whole-row kernels, Rader bodies, one- / two-kernel and fused
    Bluestein, general and prime column tiles, the multi-kernel Rader), the emulator running every phase from the last thread to the first,
    two rows each against numpy in float64: the planner's choice for an arbitrary length must not depend on the order threads run in."""
    rng = np.random.default_rng(20260925)
    families = set()
    os.environ["MI355_EMU_ORDER"] = "reverse"
    try:
        for dtype, tol in ((np.complex64, 5e-6), (np.complex128, 1e-13)):
            planner = emu_planner(dtype)
            for n in (int(v) for v in np.exp(rng.uniform(np.log(2), np.log(120000), 400))):
                d = n % 2
                fft = planner.plan_fft(n, d)
                families.add(re.split(r"[<(]", fft.describe())[0])
                x = random_signal(2 * n, dtype, seed=n)
                y = x.copy()
                fft.process(y)
                assert rel_l2(y, numpy_fft(x, n, d == 1)) < tol, (n, d, fft.describe())
                fft.trim_workspaces()
    finally:
        del os.environ["MI355_EMU_ORDER"]
    assert {"k1", "rader", "bluestein", "bluestein_large", "k2gfirst", "k2rfirst"} <= families, families


def test_thread_order_independence(emu_planner, oracle):
    """The emulator runs the threads of a phase one after another, so a race between threads of one phase is invisible to it
    unless the order changes: MI355_EMU_ORDER=reverse runs every phase from the last thread to the first.  Results must not
    depend on the order (round 2: the Rader bodies kept X[0] in a slot that the output with g^-(j+1) = p - 1 also wrote
    whenever the schedule's LDS layout is unpadded -- invisible in thread order, wrong in reverse order, a coin toss on the GPU)."""
    lengths = [541, 911, 1009, 127, 257, 1201, 2311, 1297, 2003, 2081, 727, 2801, 613, 683, 2143, 2053, 719, 1019, 1200, 1281, 2311 + 2, 4096, 1 << 13, 1 << 16, 44100, 289, 992,
               # round 5: the Bluestein bodies with staged / prefetched sub-pass factors (2560, 3072, 3584, 6144, 8192) and whole-row kernels with the factors 11 / 13
               1279, 1523, 1789, 3067, 4091, 5005, 9009,
               # round 5, late: Rader bodies over prime-radix sub-passes (plain / hand-over) and the prime-radix whole-row kernels up to 16384
               47, 1013, 1117, 2143, 4093, 3400, 4352, 7429, 16337]
    os.environ["MI355_EMU_ORDER"] = "reverse"
    try:
        for dtype in (np.complex64, np.complex128):
            planner = emu_planner(dtype)
            for n in lengths:
                for d in (0, 1):
                    fft = planner.plan_fft(n, d)
                    x = random_signal(2 * n, dtype)
                    y = x.copy()
                    fft.process(y)
                    want = x.copy()
                    oracle.plan(dtype, n, d).process(want)
                    assert compare_vectors(want, y), (n, d, fft.describe())
                    assert rel_l2(y, numpy_fft(x, n, d == 1)) < (5e-6 if dtype == np.complex64 else 1e-13), (n, d, fft.describe())
    finally:
        del os.environ["MI355_EMU_ORDER"]
