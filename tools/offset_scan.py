#!/usr/bin/env python3
"""How much does the RELATIVE placement of a pass's input and output buffers matter?  (Round 3: two plans with identical
kernels differed by 4.5 % in one process -- the only difference was where the allocator had put their workspaces.)

Out-of-place forward (x -> y) + inverse (y -> x) of N = 2^log2n, with y = big[off : off + batch * n] for a scan of byte
offsets `off`; x and `big` are allocated once, so the only variable is (address of y - address of x).  Prints one JSON line
per offset: pair milliseconds (median of --reps) and GB/s per kernel-pass (algorithmic bytes)."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--offsets", default="", help="comma-separated byte offsets (multiples of 16); default: a scan")
    args = ap.parse_args()
    import numpy as np
    import torch

    import rustfft_amd

    n, batch = 1 << args.log2n, args.batch
    planner = rustfft_amd.FftPlanner(np.complex64)
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    if args.offsets:
        offs = [int(v) for v in args.offsets.split(",")]
    else:
        offs = [0] + [1 << k for k in range(7, 27)] + [3 << k for k in range(7, 25, 2)] + [(1 << 21) + (1 << k) for k in range(8, 20, 3)]
    slack = max(offs) + (1 << 20)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    big = torch.empty(batch * n + slack // 8, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    base_delta = big.data_ptr() - x.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nk = len(fwd.kernel_names())
    alg = batch * n * 16
    rows = []
    for rep in range(args.reps):
        for i, off in enumerate(offs):
            y = big[off // 8: off // 8 + batch * n]
            torch.view_as_real(x).uniform_(-1.0, 1.0)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                fwd.process_outofplace_with_scratch(x, y)
                inv.process_outofplace_with_scratch(y, x)
            e1.record()
            torch.cuda.synchronize()
            if rep == 0:
                rows.append({"offset": off, "delta_mod_64MiB": (base_delta + off) % (1 << 26), "ms": []})
            rows[i]["ms"].append(e0.elapsed_time(e1) / args.iters)
    for r in rows:
        med = statistics.median(r["ms"])
        print(json.dumps({"offset": r["offset"], "delta_mod_64MiB": r["delta_mod_64MiB"], "pair_ms": round(med, 4), "min_ms": round(min(r["ms"]), 4),
                          "GBps_per_pass": round(alg * 2 * nk / (med * 1e-3) / 1e9)}), flush=True)
    print(json.dumps({"plan": fwd.describe(), "x_ptr": x.data_ptr(), "big_ptr": big.data_ptr()}))


if __name__ == "__main__":
    main()
