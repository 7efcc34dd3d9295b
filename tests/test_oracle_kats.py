"""Pins the oracle (oracle/rustfft_scalar.hpp) to every known-answer test the reference holds for the
hot path: the literal Dft spectra (src/algorithm/dft.rs:283-398), the integer KATs of math_utils
(src/math_utils.rs:495-588, 590-682), the twiddle identities (src/twiddles.rs:76-98) and the planner
shape tests (src/plan.rs:700-883).  CPU only."""
import json
import os

import numpy as np
import pytest

from helpers import compare_vectors, numpy_fft, random_signal, rel_l2

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _c(v, dtype):
    return np.array([complex(a, b) for a, b in v], dtype=dtype)


@pytest.mark.parametrize("case", KATS["dft"], ids=lambda c: f"len{len(c['signal'])}")
def test_dft_known_answers(oracle, case):
    # dft.rs:270-281 test_dft_correct: Dft vs the literal spectrum under compare_vectors
    sig, spec = _c(case["signal"], np.complex64), _c(case["spectrum"], np.complex64)
    out = oracle.dft(np.complex64, len(sig)).transform(sig)
    assert compare_vectors(spec, out)
    # the reference's literals are hand-rounded (len 6: -8.16 for -8.196...), so the per-element bound is loose
    tol = 6e-2 if len(sig) == 6 else 1e-5
    assert np.max(np.abs(out - spec)) < tol
    # the planner's choice for the same length must agree with the same literal
    out2 = oracle.plan(np.complex64, len(sig)).transform(sig)
    assert np.max(np.abs(out2 - spec)) < tol


def test_dft_matches_textbook_definition(oracle):
    # dft.rs:92-203: Dft vs an in-test textbook DFT, lens 1..19, batches of 4... (here numpy c128)
    for n in range(1, 20):
        for d in (0, 1):
            x = random_signal(4 * n, np.complex128)
            f = oracle.dft(np.complex128, n, d)
            y = x.copy()
            f.process(y)
            assert rel_l2(y, numpy_fft(x, n, d == 1)) < 1e-13


def test_modular_exponent(oracle):
    for b, e, m, exp in KATS["modpow"]:
        assert oracle.modular_exponent(b, e, m) == exp


def test_primitive_root(oracle):
    for p, root in KATS["primitive_roots"]:
        assert oracle.primitive_root(p) == root
    # config C4: p = 1009 -> g = 11
    assert oracle.primitive_root(1009) == 11


def test_distinct_prime_factors(oracle):
    # math_utils.rs:525-538
    for n, exp in [(46, [2, 23]), (2, [2]), (3, [3]), (162, [2, 3])]:
        assert oracle.distinct_prime_factors(n) == exp


def test_prime_factors(oracle):
    # math_utils.rs:590-682
    for n, factors, total, distinct, is_prime in KATS["prime_factors"]:
        f = oracle.prime_factors(n)
        got = {}
        if f["power_two"]:
            got["2"] = f["power_two"]
        if f["power_three"]:
            got["3"] = f["power_three"]
        for v, c in f["other"]:
            got[str(v)] = c
        assert got == factors and f["n"] == n
        assert f["total"] == total and f["distinct"] == distinct and (f["total"] == 1) == is_prime


def test_partition_factors_products(oracle):
    # math_utils.rs:725-916: both halves multiply back to n, neither is 1
    for n in [4, 9, 16, 36, 100, 37 * 41, 11 * 13 * 17, 2 * 3 * 5 * 7 * 11, 121, 1331, 44100, 53 * 53 * 59]:
        l, r = oracle.partition_factors(n)
        assert l * r == n and l > 1 and r > 1, (n, l, r)


def test_rotate90_equals_quarter_twiddle(oracle):
    # twiddles.rs:76-98: rotate_90 == multiply by twiddle(1,4)
    for d, expect in ((0, -1j), (1, 1j)):
        t = oracle.compute_twiddle(np.complex128, 1, 4, d)
        assert abs(t - expect) < 1e-15
    # twiddle table formula pinned: cos/sin of an f64 angle, rounded to T (twiddles.rs:11-16)
    for n, k in [(1024, 3), (1200, 777), (1009, 500), (1 << 20, 123457)]:
        ang = (-2.0 * np.pi / n) * k
        t64 = oracle.compute_twiddle(np.complex128, k, n, 0)
        t32 = oracle.compute_twiddle(np.complex64, k, n, 0)
        assert t64 == complex(np.cos(ang), np.sin(ang))
        assert t32 == complex(np.float32(np.cos(ang)), np.float32(np.sin(ang)))
        assert oracle.compute_twiddle(np.complex128, k, n, 1) == t64.conjugate()


# ---- planner shape tests (plan.rs:700-830) ----------------------------------------------------
def test_plan_trivial(oracle):
    assert oracle.recipe(0) == "Dft(0)" and oracle.recipe(1) == "Dft(1)"


def test_plan_large_power_of_two(oracle):
    for p in range(6, 32):
        assert oracle.recipe(1 << p).startswith("Radix4{")


def test_plan_butterflies(oracle):
    for n in KATS["planner_shapes"]["butterflies"]:
        assert oracle.recipe(n) == f"Butterfly{n}"


def test_plan_radixn(oracle):
    for a in range(2, 5):
        for b in range(2, 5):
            for c in range(2, 5):
                for d in range(2, 5):
                    assert oracle.recipe(2**a * 3**b * 5**c * 7**d).startswith("RadixN{")


def test_plan_small_composites(oracle):
    for n in KATS["planner_shapes"]["mixedradixsmall"]:
        assert oracle.recipe(n).startswith("MixedRadixSmall{")
    for n in KATS["planner_shapes"]["goodthomassmall"]:
        assert oracle.recipe(n).startswith("GoodThomasAlgorithmSmall{")


def test_plan_bluestein_vs_rader(oracle):
    for n in KATS["planner_shapes"]["bluestein_primes"]:
        assert oracle.recipe(n).startswith("BluesteinsAlgorithm{")
    for n in KATS["planner_shapes"]["rader_primes"]:
        assert oracle.recipe(n).startswith("RadersAlgorithm{")


def test_plan_baseline_configs(oracle):
    # the recipes of the BASELINE.json configs
    assert oracle.recipe(1024) == "Radix4{3,Butterfly16}"
    assert oracle.recipe(1 << 20) == "Radix4{8,Butterfly16}"
    assert oracle.recipe(1 << 22) == "Radix4{9,Butterfly16}"
    assert oracle.recipe(1 << 11) == "Radix4{4,Butterfly8}"
    assert oracle.recipe(1200) == "RadixN{[5,5,2],Butterfly24}"
    assert oracle.recipe(1009) == "RadersAlgorithm{RadixN{[7,6],Butterfly24}}"
    assert oracle.recipe(1019) == "BluesteinsAlgorithm{1019,Radix4{4,Butterfly8}}"
    assert oracle.recipe(719) == "BluesteinsAlgorithm{719,Radix4{3,Butterfly24}}"
    assert oracle.recipe(2018) == "RadixN{[2],RadersAlgorithm{RadixN{[7,6],Butterfly24}}}"
    assert oracle.recipe(37 * 41).startswith("MixedRadix{")


def test_fft_cache_identity(oracle):
    # plan.rs:832-870
    a = oracle.plan(np.complex128, 1234, 0)
    b = oracle.plan(np.complex128, 1234, 0)
    c = oracle.plan(np.complex128, 1234, 1)
    assert oracle.same_instance(a, b) and not oracle.same_instance(a, c)


def test_plan_zero_does_not_explode(oracle):
    # plan.rs:873-882
    for dt in (np.complex64, np.complex128):
        oracle.plan(dt, 0).process(np.zeros(0, dtype=dt))
