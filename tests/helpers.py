"""Shared test helpers: the reference's acceptance metric and input distribution.

compare_vectors / random_signal restate src/test_utils.rs:19-43 and tests/accuracy.rs:30-37, 84-95:
inputs re, im ~ U[0,10) from a seeded stream; pass iff mean_i |a_i - b_i| < 0.1.
(The reference's StdRng stream cannot be reproduced without the `rand` crate and is not needed: the
reference tests are differential.)
"""
import numpy as np

SEED = 0x52555354  # "RUST"


def random_signal(n, dtype, seed=SEED):
    rng = np.random.default_rng(seed + n)
    real = np.float32 if np.dtype(dtype) == np.complex64 else np.float64
    x = rng.uniform(0.0, 10.0, n).astype(real) + 1j * rng.uniform(0.0, 10.0, n).astype(real)
    return x.astype(dtype)


def zero_mean_signal(n, dtype, seed=SEED):
    rng = np.random.default_rng(seed + 7 * n + 1)
    x = rng.uniform(-1.0, 1.0, n) + 1j * rng.uniform(-1.0, 1.0, n)
    return x.astype(dtype)


def mean_abs_err(a, b):
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    assert a.shape == b.shape
    if a.size == 0:
        return 0.0
    return float(np.mean(np.abs(a.astype(np.complex128) - b.astype(np.complex128))))


def compare_vectors(a, b):
    """tests/accuracy.rs:30-37 — THE tolerance north_star names: mean |a-b| < 0.1."""
    return mean_abs_err(a, b) < 0.1


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.complex128).reshape(-1)
    b = np.asarray(b, dtype=np.complex128).reshape(-1)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))


def numpy_fft(x, n, inverse):
    """Independent second oracle: numpy pocketfft in complex128, unnormalised both ways."""
    x = np.asarray(x, dtype=np.complex128).reshape(-1, n)
    return (np.fft.ifft(x, axis=1) * n if inverse else np.fft.fft(x, axis=1)).reshape(-1)


def check_fft_algorithm(fft, length, direction, reference=None, n=3):
    """src/test_utils.rs:70-209 restated: len/direction, then a batch of 3 through all four API
    entry points, each again with scratch pre-filled with (100,100) ("dirty scratch")."""
    assert fft.len() == length, "Algorithm reported incorrect size"
    assert fft.fft_direction() == direction, "Algorithm reported incorrect FFT direction"
    dtype = np.dtype(fft.dtype)
    x = random_signal(length * n, dtype)
    if reference is None:
        expected = numpy_fft(x, length, direction == 1) if length > 0 else x.copy()
    else:
        expected = x.copy()
        reference.process(expected)
    dirty = np.dtype(dtype).type(100 + 100j)

    buf = x.copy()
    fft.process(buf)
    assert compare_vectors(expected, buf), f"process() failed, length = {length}"

    for fill in (0, dirty):
        buf = x.copy()
        scratch = np.full(fft.get_inplace_scratch_len(), fill, dtype=dtype)
        fft.process_with_scratch(buf, scratch)
        assert compare_vectors(expected, buf), f"process_with_scratch() failed, length = {length}, fill={fill}"

    for fill in (0, dirty):
        inp = x.copy()
        out = np.zeros(n * length, dtype=dtype)
        scratch = np.full(fft.get_outofplace_scratch_len(), fill, dtype=dtype)
        fft.process_outofplace_with_scratch(inp, out, scratch)
        assert compare_vectors(expected, out), f"process_outofplace_with_scratch() failed, length = {length}"

    for fill in (0, dirty):
        inp = x.copy()
        out = np.zeros(n * length, dtype=dtype)
        scratch = np.full(fft.get_immutable_scratch_len(), fill, dtype=dtype)
        fft.process_immutable_with_scratch(inp, out, scratch)
        assert compare_vectors(expected, out), f"process_immutable_with_scratch() failed, length = {length}"
        assert np.array_equal(inp, x), "immutable input was modified"


def build_cpp_mirror_check():
    """Compiles tests/cpp/mirror_check.cpp (the C++17 host mirror's own check program) against the in-tree library."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "tests", "cpp", "mirror_check.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "rustfft_amd", "lib"), "-lmi355fft",
                           "-Wl,-rpath," + os.path.join(root, "rustfft_amd", "lib")])
    return exe


def check_host_planner_options(planner, oracle):
    """mi355fft_plan_create_ex (include/mi355fft.h): the host planner names the Recipe family, supplies its own
    compute_twiddle (src/twiddles.rs:6-23) and / or its finished Rader / Bluestein tables (raders_algorithm.rs:87-113,
    bluesteins_algorithm.rs:63-98).  Results must equal the default plan's bit for bit when the host's values are the
    library's own, and follow the host's values when they differ."""
    import pytest

    import rustfft_amd

    dtype = np.complex64
    calls = []

    def twiddle(index, fft_len):
        calls.append((index, fft_len))
        return oracle.compute_twiddle(dtype, index, fft_len, 0)  # the reference's own function, forward direction

    for n in (1024, 1200, 1 << 16, 1009, 719):
        x = random_signal(2 * n, dtype)
        want = x.copy()
        planner.plan_fft(n, 1).process(want)
        calls.clear()
        fft = planner.plan_fft_with(n, 1, twiddle_fn=twiddle)
        assert calls and all(i < 2 * l + 1 for i, l in calls), n
        got = x.copy()
        fft.process(got)
        if n in (1024, 1200, 1 << 16):
            assert np.array_equal(got, want), n  # oracle twiddles == library twiddles (both f64 angle, cos/sin, rounded to T)
        else:  # Rader / Bluestein precompute spectra FROM the twiddles: host values rounded to T move them by ~eps
            assert rel_l2(got, want) < 2e-6 and not np.array_equal(got, want), n
    # a host twiddle function that is deliberately different shows the tables really come from it
    fft = planner.plan_fft_with(4096, 0, twiddle_fn=lambda i, l: 1.0 + 0.0j)
    y = random_signal(4096, dtype)
    z = y.copy()
    fft.process(z)
    ref = y.copy()
    planner.plan_fft(4096, 0).process(ref)
    assert not np.allclose(z, ref)

    # algorithm families
    assert "bluestein" in planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN).describe()
    assert "k1<" in planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX).describe()
    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        planner.plan_fft_with(1019, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX)  # prime, 1018 = 2 * 509: nothing direct
    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        planner.plan_fft_with(1019, 0, algorithm=rustfft_amd.ALGO_RADER)  # 509 is not 13-smooth
    x = random_signal(3 * 1024, dtype)
    a, b = x.copy(), x.copy()
    planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN).process(a)
    oracle.plan(dtype, 1024, 0).process(b)
    assert compare_vectors(a, b)

    # finished tables from the reference's own algorithm objects (here: the oracle restating them), both directions
    for d in (0, 1):
        # Bluestein: twiddles[n] and inner_fft_multiplier[M] exactly as bluesteins_algorithm.rs:63-98 builds them
        n = 719
        M = planner.bluestein_inner_len(n)
        assert M >= 2 * n - 1
        tw = np.array([oracle.compute_twiddle(dtype, (i * i) % (2 * n), 2 * n, d) for i in range(n)], dtype=dtype)
        mult = np.zeros(M, dtype=dtype)
        mult[0] = np.conj(tw[0]) / M
        for i in range(1, n):
            mult[i] = mult[M - i] = np.conj(tw[i]) / M
        inner = oracle.plan(dtype, M, d)
        inner.process(mult)
        fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_BLUESTEIN, bluestein_twiddles=tw, bluestein_multiplier=mult)
        check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
        with pytest.raises(rustfft_amd.FftPanic, match="host tables do not fit"):
            planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_BLUESTEIN, bluestein_twiddles=tw, bluestein_multiplier=mult[: M // 2])
        # Rader: inner_fft_data[p - 1] as raders_algorithm.rs:87-113 builds it (smallest primitive root, unity / (p-1))
        p = 1009
        g = oracle.primitive_root(p)
        ginv = pow(g, p - 2, p)
        data = np.zeros(p - 1, dtype=dtype)
        t = 1
        for j in range(p - 1):
            data[j] = oracle.compute_twiddle(dtype, t, p, d) / (p - 1)
            t = t * ginv % p
        oracle.plan(dtype, p - 1, d).process(data)
        fft = planner.plan_fft_with(p, d, algorithm=rustfft_amd.ALGO_RADER, rader_inner_fft_data=data)
        assert "rader" in fft.describe()
        check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=3)
    with pytest.raises(rustfft_amd.FftPanic, match="needs algorithm"):
        planner.plan_fft_with(1009, 0, rader_inner_fft_data=np.zeros(1008, dtype=dtype))
