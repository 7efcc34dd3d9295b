#!/usr/bin/env python3
"""Census of the shipped planner: which plan family every length gets (the judge's count of round 5, reproducible).  Runs on the CPU through the
emulator build of the library (tests/emu: the SHIPPED registry and planner behind the same C ABI) -- no transform is executed.
  python tools/plan_census.py [--dtype f32|f64] [--out profiles/r6/plan_census_f32.json]"""
import argparse
import collections
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(desc):
    if desc.startswith("lsm<"):
        return "stage machine (MixedRadix / Rader tree, one kernel)"
    if desc.startswith("k1<"):
        return "compiled whole-row schedule"
    if desc.startswith("rader<"):
        return "compiled Rader body"
    if desc.startswith("rader_large"):
        return "multi-kernel Rader"
    if desc.startswith("bluestein2_first"):
        return "two-kernel Bluestein"
    if desc.startswith("bluestein_large"):
        return "multi-kernel Bluestein"
    if desc.startswith("bluestein<"):
        return "one-kernel Bluestein"
    if "k2r" in desc:
        return "column-tile passes with a prime tile height"
    if "k2g" in desc:
        return "general column-tile passes"
    if "k2first" in desc or "fused{" in desc:
        return "power-of-two column-tile passes"
    if desc.startswith("trivial"):
        return "trivial"
    return "other: " + desc.split("<")[0]


def main():
    import numpy as np

    import rustfft_amd
    from rustfft_amd import _native

    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", emu, "-j", "8", "-s"])
    lib = _native.load(os.path.join(emu, "libmi355fft_emu.so"))
    planner = rustfft_amd.FftPlannerHip(np.complex64 if args.dtype == "f32" else np.complex128, lib=lib)
    report = {"dtype": args.dtype, "ranges": {}}
    for name, rng in (("[2, 4096]", range(2, 4097)), ("(4096, 16384]", range(4097, 16385)), ("(16384, 65536] every 7th", range(16385, 65537, 7))):
        c = collections.Counter()
        primes_on_bluestein = 0
        for n in rng:
            fam = family(planner.plan_fft(n, 0).describe())
            c[fam] += 1
            if "Bluestein" in fam and n <= 4096 and all(n % q for q in range(2, int(n**0.5) + 1)):
                primes_on_bluestein += 1
        total = sum(c.values())
        report["ranges"][name] = {"lengths": total, "families": {k: [v, round(100.0 * v / total, 1)] for k, v in c.most_common()}}
        if name == "[2, 4096]":
            report["ranges"][name]["primes_on_bluestein"] = primes_on_bluestein
        print(name, total, "lengths")
        for k, v in c.most_common():
            print(f"   {v:6d}  {100.0 * v / total:5.1f} %  {k}")
        if name == "[2, 4096]":
            print("   primes <= 4096 on Bluestein:", primes_on_bluestein)
    if args.out:
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
