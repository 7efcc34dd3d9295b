#!/usr/bin/env python3
"""Every prime p <= 4096 through the planner's AUTO choice, forward, HBM-resident, 1 GiB of rows each: algorithmic TB/s
(2 * p * sizeof(C) per transform / time) and the fraction of 8 TB/s, grouped by plan family.  Prints ONE JSON object:
{"primes": [[p, TBps, family], ...], "summary": {family: {"count", "min", "median", "max"}}}.  --dtype f64 for Complex<f64>;
--set smooth13 sweeps the 13-smooth non-power-of-two lengths <= 4096 instead (family = rows per workgroup); --lib <path> loads another build."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch

    import rustfft_amd

    f64 = "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "f64"
    dt, tdt, esz = (np.complex128, torch.complex128, 16) if f64 else (np.complex64, torch.complex64, 8)
    if "--lib" in sys.argv:  # an alternative build of the library (A/B of generator choices)
        from rustfft_amd import _native

        planner = rustfft_amd.FftPlannerHip(dt, lib=_native.load(sys.argv[sys.argv.index("--lib") + 1]))
    else:
        planner = rustfft_amd.FftPlanner(dt)
    primes = [p for p in range(2, 4097) if all(p % q for q in range(2, int(p**0.5) + 1))]
    smooth_set = "--set" in sys.argv and sys.argv[sys.argv.index("--set") + 1] == "smooth13"
    if smooth_set:  # every 13-smooth non-power-of-two length instead
        def smooth(v):
            for q in (2, 3, 5, 7, 11, 13):
                while v % q == 0:
                    v //= q
            return v == 1

        primes = [v for v in range(3, 4097) if smooth(v) and (v & (v - 1))]
    rows = []
    x = torch.empty((1 << 30) // esz, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for p in primes:
        batch = x.numel() // p
        buf = x[: batch * p]
        fft = planner.plan_fft_forward(p)
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
            buf.mul_(1e-3)  # keep magnitudes bounded (unnormalised transforms grow by sqrt(p) per call)
        d = fft.describe()
        fam = (d.split(">")[1] if smooth_set else "butterfly/mixed-radix") if d.startswith("k1<") else "rader" if d.startswith("rader") or "dyn_rader" in d else "bluestein" if "bluestein" in d else "butterfly/mixed-radix"
        rows.append([p, round(batch * 2 * p * esz / min(ts) / 1e9, 3), fam])
    summary = {}
    for fam in sorted(set(r[2] for r in rows)):
        v = [r[1] for r in rows if r[2] == fam]
        summary[fam] = {"count": len(v), "min_TBps": min(v), "median_TBps": statistics.median(v), "max_TBps": max(v),
                        "min_frac_of_8TBps": round(min(v) / 8, 3), "median_frac_of_8TBps": round(statistics.median(v) / 8, 3)}
    print(json.dumps({"what": "every prime <= 4096, Complex<%s>, forward, 1 GiB of rows, AUTO plan" % ("f64" if f64 else "f32"), "summary": summary, "primes": rows}))


if __name__ == "__main__":
    main()
