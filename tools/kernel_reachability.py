#!/usr/bin/env python3
"""Which compiled kernels can a plan reach?  Plans every length in a range (and the structured lengths above it: powers of two,
13-smooth lengths, primes with a smooth p - 1) through AUTO and through the three host-requested families, on the kernel-body
EMULATOR (tests/emu/libmi355fft_emu.so: same planner, same registry, no GPU needed), collects the kernel names the plans
use and compares them with the names in the shipped library.  Prints one JSON object: counts per kernel kind of compiled /
reached-by-AUTO / reached-only-on-request / never reached, and the never-reached names.  (Lengths above --exhaustive that are
not structured can only reach the fused Bluestein passes, whose tile heights the structured sweep already covers.)"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exhaustive", type=int, default=17000, help="every length up to here through AUTO (covers every whole-row kernel: the two-kernel Bluestein ends at 16384)")
    ap.add_argument("--asked-max", type=int, default=9000, help="every length up to here also through the three requested families")
    ap.add_argument("--structured-max", type=int, default=1 << 20, help="13-smooth lengths, primes with a smooth p - 1 and prime-tile composites up to here (2^24 takes most of an hour: the host builds a Rader table per large prime)")
    args = ap.parse_args()
    import numpy as np

    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(ROOT, "tests", "emu", "libmi355fft_emu.so"))
    pat = re.compile(r"^(rader<|k1<|k1bs|k2first<|k2later<|k2g|k2r|bluestein<|bluestein2_|dyn_|bluestein_pointwise)")
    shipped = set(l for l in subprocess.check_output(["strings", "-n", "6", os.path.join(ROOT, "rustfft_amd", "lib", "libmi355fft.so")]).decode(errors="ignore").splitlines() if pat.match(l))

    def smooth13(limit):
        s = {1}
        for q in (2, 3, 5, 7, 11, 13):
            s = {v * q**k for v in s for k in range(0, 31) if v * q**k <= limit}
        return s

    def is_prime(n):
        if n < 2 or n % 2 == 0:
            return n == 2
        d = 3
        while d * d <= n:
            if n % d == 0:
                return False
            d += 2
        return True

    sm = smooth13(args.structured_max)
    structured = sorted(v for v in sm if v > args.exhaustive)
    structured += [v for v in (1 << k for k in range(10, 31)) if v > args.exhaustive and v not in structured]  # every power of two up to 2^30
    structured += [v for v in (3 << 20, 5 << 20, 9 << 19, 3 << 22, 7 << 21) if v not in structured]                 # and a few large smooth lengths
    structured += sorted(p for p in (v + 1 for v in sm if v + 1 > args.exhaustive) if is_prime(p))
    # composites of two or three factors from the prime-tile range (a sample: every pair of tile primes)
    tile_primes = [p for p in range(37, 640) if is_prime(p)]
    structured += sorted({a * b for a in tile_primes for b in tile_primes if a * b > args.exhaustive})[::7]
    auto, asked = set(), set()
    h = ctypes.c_void_p()
    for prec in (32, 64):
        def names(n, algo):
            o = _native.PlanOptions()
            o.struct_size = ctypes.sizeof(_native.PlanOptions)
            o.algorithm = algo
            rc = lib.mi355fft_plan_create_ex(n, 0, prec, ctypes.byref(o), ctypes.byref(h))
            if rc != 0:
                return []
            out = [lib.mi355fft_plan_kernel_name(h, i).decode() for i in range(lib.mi355fft_plan_num_kernels(h))]
            # the one-kernel Bluestein names get their form suffix at run time (launch.h bs_name: t1 / p<N>); the library's strings hold the stem
            out = [re.sub(r"(>xF\d+s?)(?:t1)?(?:p\d+)?$", r"\1", v) if v.startswith("bluestein<") else v for v in out]
            lib.mi355fft_plan_destroy(h)
            return out

        for n in range(2, args.exhaustive + 1):
            auto.update(names(n, 0))
            if n <= args.asked_max:
                for algo in (1, 2, 3):
                    asked.update(names(n, algo))
        for n in structured:
            auto.update(names(n, 0))
    asked -= auto

    def kind(name):
        return re.match(r"[a-z0-9_]+", name).group(0)

    kinds = sorted({kind(n) for n in shipped})
    report = {"what": f"plans of every length 2 .. {args.exhaustive} (AUTO; up to {args.asked_max} also the three requested families) and {len(structured)} structured lengths up to 2^30 (AUTO), f32 + f64, on the emulator",
              "compiled_names": len(shipped), "reached_by_auto": len(shipped & auto), "reached_only_on_request": len(shipped & asked),
              "never_reached": len(shipped - auto - asked), "per_kind": {}}
    for k in kinds:
        names_k = {n for n in shipped if kind(n) == k}
        report["per_kind"][k] = {"compiled": len(names_k), "auto": len(names_k & auto), "on_request": len(names_k & asked), "never": len(names_k - auto - asked)}
    report["never_reached_names"] = sorted(shipped - auto - asked)
    report["reached_but_not_in_strings"] = sorted((auto | asked) - shipped)[:20]
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
