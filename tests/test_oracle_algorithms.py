"""Differential unit tests of the oracle, restating each algorithm file's `mod unit_tests` of the
reference: every algorithm vs the naive Dft through all four API entry points with
clean and dirty scratch (src/test_utils.rs:70-209).  CPU only."""
import numpy as np
import pytest

from helpers import check_fft_algorithm, compare_vectors, numpy_fft, random_signal, rel_l2

BUTTERFLIES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 19, 23, 24, 27, 29, 31, 32]
BASES = [1, 2, 3, 4, 5, 6, 7, 8, 9]  # test_utils.rs:278-293 construct_base (1..9)


@pytest.mark.parametrize("n", BUTTERFLIES)
def test_butterflies(oracle, n):
    # butterflies.rs:6395-6434 (f32, both directions)
    for d in (0, 1):
        check_fft_algorithm(oracle.butterfly(np.complex64, n, d), n, d, reference=oracle.dft(np.complex64, n, d))
        y = oracle.butterfly(np.complex128, n, d).transform(random_signal(n, np.complex128))
        assert rel_l2(y, numpy_fft(random_signal(n, np.complex128), n, d == 1)) < 1e-14


def test_radix4_with_length(oracle):
    # radix4.rs:213-223
    for p in range(0, 8):
        n = 1 << p
        for d in (0, 1):
            check_fft_algorithm(oracle.radix4(np.complex64, n, d), n, d, reference=oracle.dft(np.complex64, n, d))


def test_radix4_with_base(oracle):
    # radix4.rs:225-243 (f64)
    for base in BASES:
        for d in (0, 1):
            b = oracle.butterfly(np.complex128, base, d)
            for k in range(0, 4):
                n = base * 4**k
                check_fft_algorithm(oracle.radix4_with_base(k, b), n, d)


def test_radixn(oracle):
    # radixn.rs:497-541: bases 1..6 x 0/1/2 factors from {2..7}
    for base in range(1, 7):
        for d in (0, 1):
            b = oracle.butterfly(np.complex128, base, d)
            check_fft_algorithm(oracle.radixn([], b), base, d)
            for f1 in range(2, 8):
                check_fft_algorithm(oracle.radixn([f1], b), base * f1, d)
                for f2 in range(2, 8):
                    check_fft_algorithm(oracle.radixn([f1, f2], b), base * f1 * f2, d)


@pytest.mark.parametrize("small", [False, True])
def test_mixed_radix(oracle, small):
    # mixed_radix.rs:416-451: W,H in 1..6 over Dft inners (f32)
    for w in range(1, 7):
        for h in range(1, 7):
            for d in (0, 1):
                wf, hf = oracle.dft(np.complex64, w, d), oracle.dft(np.complex64, h, d)
                if small:
                    wf, hf = oracle.butterfly(np.complex64, w, d), oracle.butterfly(np.complex64, h, d)
                check_fft_algorithm(oracle.mixed_radix(wf, hf, small), w * h, d)


@pytest.mark.parametrize("small", [False, True])
def test_good_thomas(oracle, small):
    # good_thomas_algorithm.rs:529-569: coprime pairs
    from math import gcd

    for w in range(1, 12):
        for h in range(1, 12):
            if gcd(w, h) != 1:
                continue
            for d in (0, 1):
                if small:
                    if w not in BUTTERFLIES or h not in BUTTERFLIES:
                        continue
                    wf, hf = oracle.butterfly(np.complex64, w, d), oracle.butterfly(np.complex64, h, d)
                else:
                    wf, hf = oracle.dft(np.complex64, w, d), oracle.dft(np.complex64, h, d)
                check_fft_algorithm(oracle.good_thomas(wf, hf, small), w * h, d)


def test_good_thomas_rejects_non_coprime(oracle):
    # good_thomas_algorithm.rs:74 / :378-381
    with pytest.raises(oracle.OraclePanic, match="Inputs must be coprime"):
        oracle.good_thomas(oracle.butterfly(np.complex64, 4), oracle.butterfly(np.complex64, 6), True)


def test_raders(oracle):
    # raders_algorithm.rs:302-309 and :324-329: primes < 100 over a Dft inner (f32)
    primes = [p for p in range(3, 100) if all(p % q for q in range(2, int(p**0.5) + 1))]
    for p in primes:
        for d in (0, 1):
            check_fft_algorithm(oracle.raders(oracle.dft(np.complex64, p - 1, d)), p, d, reference=oracle.dft(np.complex64, p, d))
    with pytest.raises(oracle.OraclePanic, match="must be prime"):
        oracle.raders(oracle.dft(np.complex64, 8))


@pytest.mark.parametrize("p", [112501, 216569, 417623])
def test_raders_32bit_overflow(oracle, p):
    # raders_algorithm.rs:311-322: large primes must not overflow the index arithmetic
    f = oracle.raders(oracle.plan(np.complex64, p - 1, 0))
    data = np.zeros(p, dtype=np.complex64)
    f.process(data)
    assert not np.any(data)
    x = random_signal(p, np.complex64)
    assert compare_vectors(f.transform(x), numpy_fft(x, p, False))


def test_bluesteins(oracle):
    # bluesteins_algorithm.rs:210-225: 3,5,7,11,13 over Dft inners of several lengths
    for n in (3, 5, 7, 11, 13):
        for inner in (2 * n - 1, 2 * n, 2 * n + 5):
            for d in (0, 1):
                f = oracle.bluesteins(n, oracle.dft(np.complex64, inner, d))
                check_fft_algorithm(f, n, d, reference=oracle.dft(np.complex64, n, d))
    with pytest.raises(oracle.OraclePanic, match="Bluestein's algorithm requires"):
        oracle.bluesteins(10, oracle.dft(np.complex64, 18))


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("direction", [0, 1])
def test_accuracy_1_to_1000(oracle, dtype, direction):
    """tests/accuracy.rs:128-187: planner output vs the control = Bluestein-over-Radix4 for len 1..1000,
    via in-place, out-of-place and immutable APIs (tests/accuracy.rs:39-82)."""
    for n in range(1, 1001):
        inner_len = 1
        while inner_len < 2 * n - 1:
            inner_len *= 2
        control = oracle.bluesteins(n, oracle.radix4(dtype, inner_len, direction))
        fft = oracle.plan(dtype, n, direction)
        assert fft.len() == n and fft.fft_direction() == direction
        x = random_signal(n, dtype)
        smax = max(control.get_inplace_scratch_len(), fft.get_inplace_scratch_len(), fft.get_outofplace_scratch_len(),
                   fft.get_immutable_scratch_len())
        scratch = np.zeros(smax, dtype=dtype)
        ctrl = x.copy()
        control.process_with_scratch(ctrl, scratch)
        a = x.copy()
        fft.process_with_scratch(a, scratch)
        i2, b = x.copy(), x.copy()
        fft.process_outofplace_with_scratch(i2, b, scratch)
        c = x.copy()
        fft.process_immutable_with_scratch(x, c, scratch)
        assert compare_vectors(ctrl, a) and compare_vectors(ctrl, b) and compare_vectors(ctrl, c), n


def test_error_messages(oracle):
    # common.rs:13-104 — the panic texts are part of the boundary contract
    f = oracle.plan(np.complex64, 16)
    with pytest.raises(oracle.OraclePanic, match="Provided FFT buffer was too small. Expected len = 16, got len = 5"):
        f.process(np.zeros(5, np.complex64))
    with pytest.raises(oracle.OraclePanic, match="Input FFT buffer must be a multiple of FFT length. Expected multiple of 16, got len = 40"):
        f.process(np.zeros(40, np.complex64))
    g = oracle.plan(np.complex64, 1024)
    with pytest.raises(oracle.OraclePanic, match="Not enough scratch space was provided. Expected scratch len >= 1024, got scratch len = 10"):
        g.process_with_scratch(np.zeros(1024, np.complex64), np.zeros(10, np.complex64))
    with pytest.raises(oracle.OraclePanic, match="must have the same length. Got input.len\\(\\) = 1024, output.len\\(\\) = 2048"):
        g.process_outofplace_with_scratch(np.zeros(1024, np.complex64), np.zeros(2048, np.complex64), np.zeros(0, np.complex64))
    # an empty buffer with len > 0 is accepted silently (array_utils.rs:164-176)
    g.process(np.zeros(0, np.complex64))


def test_scratch_lengths_match_reference_rules(oracle):
    # SURVEY Appendix B at the BASELINE configs
    for n, exp in [(1024, (1024, 0, 0)), (1 << 20, (1 << 20, 0, 0)), (1200, (1200, 0, 0)), (1009, (1008, 0, 2016))]:
        f = oracle.plan(np.complex64, n)
        assert (f.get_inplace_scratch_len(), f.get_outofplace_scratch_len(), f.get_immutable_scratch_len()) == exp
    f = oracle.plan(np.complex64, 1019)  # Bluestein M = 2048 + Radix4 inplace 2048
    assert f.get_inplace_scratch_len() == 4096
