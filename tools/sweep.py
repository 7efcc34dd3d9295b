#!/usr/bin/env python3
"""Size sweep on one GPU: forward in-place transforms, HBM-resident, per-kernel HIP-event timings.
Prints one JSON line per size: GFLOP/s (5 N log2 N), algorithmic GB/s per kernel and for the whole transform."""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=10)
    ap.add_argument("--max", type=int, default=24)
    ap.add_argument("--bytes", type=float, default=2.0, help="buffer size in GiB")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--sizes", default="", help="comma-separated explicit lengths (overrides --min/--max)")
    ap.add_argument("--check", action="store_true", help="relative L2 error of row 0 against numpy complex128")
    ap.add_argument("--fused", type=int, default=-1, help="fused two-pass launch: -1 the planner's choice, 0 never, 1 whenever compiled")
    ap.add_argument("--lib", default="", help="A/B runs: load this build of libmi355fft.so instead of rustfft_amd/lib's")
    args = ap.parse_args()
    import numpy as np
    import torch

    import rustfft_amd

    dt, tdt, esz = (np.complex64, torch.complex64, 8) if args.dtype == "f32" else (np.complex128, torch.complex128, 16)
    if args.lib:
        from rustfft_amd import _native

        planner = rustfft_amd.FftPlannerHip(dt, lib=_native.load(args.lib))
    else:
        planner = rustfft_amd.FftPlanner(dt)
    sizes = [int(v) for v in args.sizes.split(",")] if args.sizes else [1 << q for q in range(args.min, args.max + 1)]
    for n in sizes:
        p = math.log2(n)
        batch = max(1, int(args.bytes * 2**30) // (n * esz))
        x = torch.empty(batch * n, dtype=tdt, device="cuda")
        torch.view_as_real(x).uniform_(-1.0, 1.0)
        fft = planner.plan_fft_forward(n)
        fft.set_chunk_batch(args.chunk)
        if args.fused >= 0:
            fft.set_fused(args.fused)
        err = None
        if args.check:
            x0 = x[:n].cpu().numpy()
        fft.process(x)
        torch.cuda.synchronize()
        if args.check:
            want = np.fft.fft(x0.astype(np.complex128))
            err = float(np.linalg.norm(x[:n].cpu().numpy() - want) / np.linalg.norm(want))
        for _ in range(8 if fft.is_fused() else 2):  # (a fused plan reaches its steady rate after ~10 launches of a fresh process, ~5 of a new plan: profiles/r4/fused_warmup_*.jsonl)
            torch.view_as_real(x).uniform_(-1.0, 1.0)
            fft.process(x)
        torch.view_as_real(x).uniform_(-1.0, 1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fft.process(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        torch.view_as_real(x).uniform_(-1.0, 1.0)
        kms = fft.profile_kernels(x, reps=args.reps)
        alg = batch * 2 * n * esz
        # Two clocks (they disagreed by 17 % at 2^15): `ms` = HIP events around `reps` back-to-back calls -- the launch's
        # average duration in a stream that is kept busy, the number rocprofv3's per-kernel average agrees with and every rate in the docs is
        # quoted from; the profiling hook brackets EVERY launch with its own event pair, so a short launch is timed from an idle chip (the
        # event's own latency and the ramp of a cold start are inside the bracket).  For a one-kernel plan the launch IS the call: kernel_ms is
        # `ms`, and the bracketed figure is kept beside it under its own name.
        bracketed = None
        if len(kms) == 1:
            bracketed, kms = kms, [ms]
        print(json.dumps({"n": n, "log2n": round(p, 3), "batch": batch, "ms": round(ms, 4), "gflops": round(batch * 5.0 * n * p / ms / 1e6, 1),
                          "alg_GBps": round(alg / ms / 1e6, 1), "kernel_ms": [round(k, 4) for k in kms],
                          **({"kernel_ms_individually_bracketed": [round(k, 4) for k in bracketed]} if bracketed else {}),
                          "kernel_GBps": [round(alg / k / 1e6, 1) if k > 0 else None for k in kms], "plan": fft.describe(), "fused": fft.is_fused(),
                          # a fused plan runs ONE launch (`ms`); kernel_ms are its two passes as separate launches (event-bracketed), for comparison
                          **({"fused_per_pass_equivalent_GBps": round(2 * alg / ms / 1e6, 1), "fused_error_word": fft.fused_status()} if fft.is_fused() else {}),
                          **({"rel_l2": err} if err is not None else {})}), flush=True)
        del x


if __name__ == "__main__":
    main()
