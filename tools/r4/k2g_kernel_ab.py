#!/usr/bin/env python3
"""Per-KERNEL times of the general column-tile passes in two builds (the shipped library and one whose k2g units are compiled without the
SLP vectoriser): many two- and three-pass lengths, mi355fft_profile_inplace_dev on both, one JSON line per (length, kernel)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import rustfft_amd
from rustfft_amd import _native


def smooth(limit, primes):
    s = {1}
    for q in primes:
        s = {v * q**k for v in s for k in range(0, 40) if v * q**k <= limit}
    return sorted(s)


def main():
    a, b, count, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    libs = [_native.load(os.path.join(ROOT, "rustfft_amd", "lib", p)) for p in (a, b)]
    pl = [rustfft_amd.FftPlannerHip(np.complex64, lib=l) for l in libs]
    random.seed(seed)
    sizes = [x for x in smooth(410000, [2, 3, 5, 7, 11, 13]) if x > 4096 and (x & (x - 1))]
    sizes = sorted(random.sample(sizes, min(count, len(sizes))))
    x = torch.empty((1 << 29) // 8, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    for n in sizes:
        ffts = [p.plan_fft_forward(n) for p in pl]
        if "k2g" not in ffts[0].describe():
            continue
        batch = x.numel() // n
        buf = x[: batch * n]
        ms = [None, None]
        for rep in range(2):  # interleaved, the second pass is the one kept
            for i in (0, 1):
                ms[i] = ffts[i].profile_kernels(buf, reps=3)
            torch.view_as_real(buf).uniform_(-1.0, 1.0)
        names = ffts[0].kernel_names()
        for k, nm in enumerate(names):
            print(json.dumps({"n": n, "kernel": nm, "pos": k, "of": len(names), "ms_a": ms[0][k], "ms_b": ms[1][k]}), flush=True)
        for f in ffts:
            f.trim_workspaces()  # (a planner keeps its plans: without this 300 plans hold 300 half-GiB workspaces)


main()
