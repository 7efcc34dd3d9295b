"""The LDS stage machine (rustfft_amd/csrc/lsm.h, lsm_plan.h; round 6): the reference's MixedRadix-over-Rader / Rader-over-MixedRadix trees
(src/plan.rs:412-425, 474-506, 636-665) as ONE kernel driven by a run-time program.

CPU: (1) tests/cpp/lsm_check.cpp runs the planner, the program and the very kernel body the GPU runs (every thread of a workgroup emulated
phase by phase, both thread orders, LDS poisoned with NaN) against a naive f64 DFT, with no other part of the library involved; (2) the same
through the C ABI of the emulator build: AUTO's choice, a host planner's MixedRadix / Rader request, all three API modes, against the oracle's
plan of the same tree.  GPU: tests/test_gpu_parity.py test_stage_machine_*."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import check_fft_algorithm, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def lsm_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lsm") / "lsm_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rustfft_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "lsm_check.cpp")])
    return exe


def run_check(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    m = re.search(r"(\d+) lengths planned and correct, (\d+) FAILED", r.stdout)
    assert r.returncode == 0 and m and int(m.group(2)) == 0, r.stdout[-3000:]
    return int(m.group(1))


def test_every_length_up_to_330_has_a_correct_program(lsm_check):
    """Every length 2 .. 330 -- smooth leaves, Rader over a leaf (17, 19 ... as well: the kernel's radices end at 16), MixedRadix over one and two
    Rader factors, Rader over MixedRadix over Rader (167, 179, 227 ...) -- in Complex<f32>, forward and inverse, two full workgroups and a ragged one."""
    assert run_check(lsm_check, 2, 330) == 329


def test_reverse_thread_order_and_f64(lsm_check):
    """Reverse thread order: a stage that read a slot another thread of the same stage writes would differ (in-place stages own their slots)."""
    assert run_check(lsm_check, 331, 460, "f64", "reverse") == 130
    assert run_check(lsm_check, 500, 560, "f32", "reverse") == 61


@pytest.mark.parametrize("n", [1369, 1517, 2368, 3034, 3599, 4001, 4070, 5661, 8144])
def test_larger_programs(lsm_check, n):
    """37^2, 37 x 41, 64 x 37, two Rader factors and a leaf, 59 x 61, a prime with smooth p - 1 (one Rader over a 4000-point leaf on 512 threads),
    lengths above 4096 (six-step tables in global memory), a three-level tree on 1024 threads."""
    assert run_check(lsm_check, n, n) == 1
    if n <= 4096 and n != 4001:  # (Complex<f64> 4001: the row, the 4000-point leaf's tables and D exceed the 160 KiB of a workgroup: no program)
        assert run_check(lsm_check, n, n, "f64", "reverse") == 1


@pytest.fixture(scope="module")
def emu_planner():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j", "8", "-s"])
    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(EMU_DIR, "libmi355fft_emu.so"))
    return lambda dtype: rustfft_amd.FftPlannerHip(dtype, lib=lib)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_auto_takes_the_stage_machine_and_matches_the_oracle(emu_planner, oracle, dtype):
    """AUTO plans 64 x 37 and friends as the stage machine (short programs: the calibrated choice, plan.cpp try_lsm); every API mode against the
    oracle's plan of the length (the reference's MixedRadix over Rader: the same tree)."""
    planner = emu_planner(dtype)
    for n in (74, 148, 592, 2368, 185 * 4):
        for d in (0, 1):
            fft = planner.plan_fft(n, d)
            assert fft.describe().startswith("lsm<"), fft.describe()
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d))


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_host_planner_tree_requests_get_the_stage_machine(emu_planner, oracle, dtype):
    """A host planner that names MixedRadix (composite) / Rader (prime) gets the tree whatever its program costs: 37 x 37 and 37 x 41 (two Rader
    factors), 167 and 1283 (Rader over MixedRadix over Rader: the reference's plan for a prime whose p - 1 has a large prime factor,
    src/plan.rs:636-665), 8144 = 16 x 509 (a three-level tree on 1024 threads; Complex<f32> only)."""
    import rustfft_amd

    planner = emu_planner(dtype)
    for n, algo in ((1369, rustfft_amd.ALGO_MIXED_RADIX), (1517, rustfft_amd.ALGO_MIXED_RADIX), (167, rustfft_amd.ALGO_RADER), (1283, rustfft_amd.ALGO_RADER),
                    (8144, rustfft_amd.ALGO_MIXED_RADIX)):
        if n == 8144 and dtype == np.complex128:
            continue
        for d in (0, 1):
            fft = planner.plan_fft_with(n, d, algorithm=algo)
            assert fft.describe().startswith("lsm<"), fft.describe()
            rng = np.random.default_rng(n + d)
            batch = 7
            x = (rng.uniform(0, 10, n * batch) + 1j * rng.uniform(0, 10, n * batch)).astype(dtype)
            got = x.copy()
            fft.process(got)
            want = x.copy()
            oracle.plan(dtype, n, d).process(want)
            assert np.mean(np.abs(got - want)) < 0.1  # tests/accuracy.rs:30-37
            assert rel_l2(got, want) < (3e-6 if dtype == np.complex64 else 1e-14)
            out = np.empty_like(x)
            fft.process_immutable_with_scratch(x, out)
            assert np.array_equal(out, got)


def test_bluestein_request_still_gets_bluestein(emu_planner):
    import rustfft_amd

    fft = emu_planner(np.complex64).plan_fft_with(2368, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN)
    assert "bluestein" in fft.describe()
