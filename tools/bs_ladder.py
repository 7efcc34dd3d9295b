#!/usr/bin/env python3
"""Per-row cost of every one-kernel Bluestein body: n = M / 2 (the largest length that plans onto inner length M), forced
ALGO_BLUESTEIN, 1 GiB of rows.  One JSON line per (dtype, M): ns per row and algorithmic TB/s -- a body earns its place
in the ladder when its row time is below the next compiled M's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch

    import rustfft_amd

    for dt, tdt, esz, name in ((np.complex64, torch.complex64, 8, "f32"), (np.complex128, torch.complex128, 16, "f64")):
        if "--lib" in sys.argv:  # a tuning build (MI355FFT_VARIANT selects alternative bodies there)
            from rustfft_amd import _native

            planner = rustfft_amd.FftPlannerHip(dt, lib=_native.load(sys.argv[sys.argv.index("--lib") + 1]))
        else:
            planner = rustfft_amd.FftPlanner(dt)
        x = torch.empty((1 << 30) // esz, dtype=tdt, device="cuda")
        torch.view_as_real(x).uniform_(-1.0, 1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        Ms = sorted(set(m << k for m in (4, 5, 6, 7) for k in range(5, 13) if 256 <= (m << k) <= 16384))
        for M in Ms:
            n = M // 2
            if planner.bluestein_inner_len(n) != M:
                continue
            fft = planner.plan_fft_with(n, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN)
            batch = x.numel() // n
            buf = x[: batch * n]
            fft.process(buf)
            torch.cuda.synchronize()
            ts = []
            for _ in range(2):
                e0.record()
                for _ in range(3):
                    fft.process(buf)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
                buf.mul_(1e-3)
            t = min(ts) * 1e-3
            print(json.dumps({"dtype": name, "M": M, "n": n, "ns_per_row": round(t / batch * 1e9, 2), "TBps": round(batch * 2 * n * esz / t / 1e12, 3),
                              "ns_per_row_per_M": round(t / batch * 1e9 / M, 5), "plan": fft.describe()[:48]}), flush=True)


if __name__ == "__main__":
    main()
