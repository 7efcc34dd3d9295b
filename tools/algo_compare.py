#!/usr/bin/env python3
"""One box, one process: the AUTO plan of each length against the Bluestein plan a host planner could force
(mi355fft_plan_create_ex, MI355FFT_ALGO_BLUESTEIN) -- the data behind the planner's Rader / mixed-radix / Bluestein order.
Prints one JSON line per (length, dtype): algorithmic TB/s of both plans and the relative L2 error of AUTO vs numpy c128."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", required=True)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--bytes", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mixed", action="store_true", help="extra arm: the plan a host planner gets with MI355FFT_ALGO_MIXED_RADIX (column-tile passes)")
    ap.add_argument("--rader", action="store_true", help="third arm: the plan a host planner gets with MI355FFT_ALGO_RADER (primes only)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import rustfft_amd

    dt, tdt, esz = (np.complex64, torch.complex64, 8) if args.dtype == "f32" else (np.complex128, torch.complex128, 16)
    planner = rustfft_amd.FftPlanner(dt)
    for n in [int(v) for v in args.sizes.split(",")]:
        batch = max(1, int(args.bytes * 2**30) // (n * esz))
        x = torch.empty(batch * n, dtype=tdt, device="cuda")
        out = {"n": n, "dtype": args.dtype, "batch": batch}
        arms = [("auto", rustfft_amd.ALGO_AUTO), ("bluestein", rustfft_amd.ALGO_BLUESTEIN)] + ([("rader", rustfft_amd.ALGO_RADER)] if args.rader else []) + ([("mixed", rustfft_amd.ALGO_MIXED_RADIX)] if args.mixed else [])
        for name, algo in arms:
            try:
                fft = planner.plan_fft_with(n, 0, algorithm=algo)
            except rustfft_amd.FftPanic as e:
                out[name] = {"error": str(e)[:80]}
                continue
            torch.view_as_real(x).uniform_(-1.0, 1.0)
            x0 = x[:n].cpu().numpy()
            fft.process(x)
            torch.cuda.synchronize()
            want = np.fft.fft(x0.astype(np.complex128))
            err = float(np.linalg.norm(x[:n].cpu().numpy() - want) / np.linalg.norm(want))
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                torch.view_as_real(x).uniform_(-1.0, 1.0)
                e0.record()
                for _ in range(args.reps):
                    fft.process(x)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / args.reps)
            ms = sorted(ts)[1]
            out[name] = {"TBps": round(batch * 2 * n * esz / ms / 1e9, 3), "rel_l2": err, "plan": fft.describe()[:60]}
        out["auto_over_bluestein"] = round(out["auto"]["TBps"] / out["bluestein"]["TBps"], 2)
        if args.mixed and "TBps" in out.get("mixed", {}):
            out["mixed_over_bluestein"] = round(out["mixed"]["TBps"] / out["bluestein"]["TBps"], 2)
        if args.rader and "TBps" in out.get("rader", {}):
            out["rader_over_bluestein"] = round(out["rader"]["TBps"] / out["bluestein"]["TBps"], 2)
        print(json.dumps(out), flush=True)
        del x


if __name__ == "__main__":
    main()
