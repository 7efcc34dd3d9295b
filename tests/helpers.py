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
