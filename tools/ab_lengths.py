#!/usr/bin/env python3
"""Interleaved A/B of two builds of the library over a set of lengths, in ONE process: for every length both builds plan it,
both are warmed, then timed alternately (A B A B ...), and the minimum per build is kept.  Separate processes on one box drift
by several per cent (the first process after the box is acquired is the slowest), so generator choices are confirmed with this
tool.  Prints one JSON line per length: {"n", "a_TBps", "b_TBps", "b_over_a", "plan_a", "plan_b"}.

  python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt.so --set smooth13 [--dtype f64] [--sizes 1200,1500]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    import rustfft_amd
    from rustfft_amd import _native

    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="libmi355fft.so")
    ap.add_argument("--b", required=True)
    ap.add_argument("--set", default="smooth13", choices=["smooth13", "primes"])
    ap.add_argument("--sizes", default="")
    ap.add_argument("--sizes-file", default="", help="file with comma- or whitespace-separated lengths")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--gib", type=float, default=0.5)
    ap.add_argument("--check", action="store_true", help="also compare the two builds' results on four rows (relative L2 of b against a)")
    ap.add_argument("--no-trim", action="store_true", help="keep every plan's workspace (what this tool did before round 4's last hour: a sweep over hundreds of multi-pass lengths then runs the device out of memory)")
    ap.add_argument("--a-algo", default="", choices=["", "bluestein"], help="plan side a through the host-planner entry point with this algorithm (same library on both sides: AUTO's choice against Bluestein)")
    ap.add_argument("--b-algo", default="", choices=["", "tree"], help="plan side b through the host-planner entry point as the reference's tree: Rader for a prime, MixedRadix otherwise (round 6: the LDS stage machine whatever its program costs)")
    ap.add_argument("--all", action="store_true", help="time a length even when both builds describe the same plan (a changed kernel body keeps its name)")
    args = ap.parse_args()
    dt, tdt, esz = (np.complex64, torch.complex64, 8) if args.dtype == "f32" else (np.complex128, torch.complex128, 16)
    pl = [rustfft_amd.FftPlannerHip(dt, lib=_native.load(os.path.join(ROOT, "rustfft_amd", "lib", p))) for p in (args.a, args.b)]

    def smooth(v):
        for q in (2, 3, 5, 7, 11, 13):
            while v % q == 0:
                v //= q
        return v == 1

    if args.sizes_file:
        sizes = [int(v) for v in open(args.sizes_file).read().replace(",", " ").split()]
    elif args.sizes:
        sizes = [int(s) for s in args.sizes.split(",")]
    elif args.set == "smooth13":
        sizes = [v for v in range(3, 4097) if smooth(v) and (v & (v - 1))]
    else:
        sizes = [p for p in range(2, 4097) if all(p % q for q in range(2, int(p**0.5) + 1))]
    x = torch.empty(int(args.gib * (1 << 30)) // esz, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for n in sizes:
        batch = x.numel() // n
        buf = x[: batch * n]
        ffts = [p.plan_fft_forward(n) for p in pl]
        if args.a_algo == "bluestein":
            ffts[0] = pl[0].plan_fft_with(n, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN)
        if args.b_algo == "tree":
            prime = n > 3 and all(n % q for q in range(2, int(n**0.5) + 1))
            try:
                ffts[1] = pl[1].plan_fft_with(n, 0, algorithm=rustfft_amd.ALGO_RADER if prime else rustfft_amd.ALGO_MIXED_RADIX)
            except Exception:
                continue  # no tree within the machine's limits
        if ffts[0].describe() == ffts[1].describe() and not args.all:
            continue
        diff = None
        if args.check:
            rows = min(batch, 4)
            src = torch.view_as_complex(torch.rand(rows * n, 2, dtype=torch.float32 if esz == 8 else torch.float64, device="cuda") * 10.0)
            outs = []
            for f in ffts:
                y = src.clone()
                f.process(y)
                outs.append(y.to(torch.complex128))
            torch.cuda.synchronize()
            diff = float((outs[1] - outs[0]).norm() / outs[0].norm())
        for f in ffts:
            f.process(buf)
        best = [1e9, 1e9]
        for _ in range(3):
            for i, f in enumerate(ffts):
                e0.record()
                f.process(buf)
                f.process(buf)
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 2)
            buf.mul_(1e-4)
        tb = [batch * 2 * n * esz / (t * 1e-3) / 1e12 for t in best]
        print(json.dumps({"n": n, "a_TBps": round(tb[0], 3), "b_TBps": round(tb[1], 3), "b_over_a": round(tb[1] / tb[0], 3), "rel_l2_b_vs_a": diff,
                          "plan_a": ffts[0].describe(), "plan_b": ffts[1].describe()}), flush=True)
        for f in (() if args.no_trim else ffts):
            f.trim_workspaces()  # a planner keeps its plans: hundreds of lengths would otherwise hold hundreds of workspaces


if __name__ == "__main__":
    main()
