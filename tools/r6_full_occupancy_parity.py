#!/usr/bin/env python3
"""Round 6: a device-resident, FULL-OCCUPANCY parity pass over a sample of the kernel families compiled in round 5 and the
stage machine of round 6 -- batch = 2 x 256 x F + 1 rows (every CU holds workgroups of the kernel, the last workgroup is ragged), every row
against numpy in float64.  One JSON line per length; exit status 1 on any failure.
Families: whole-row schedules with a factor 11 / 13 above 4096 (smooth4), with a prime radix 17 .. 31 above 4096 / 2048 (smooth5), the compiled
Rader bodies over prime-radix sub-passes, the LDS stage machine."""
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import rustfft_amd

    rng = np.random.default_rng(66)
    bad = 0
    for dt, tdt, tol in ((np.complex64, torch.complex64, 5e-6), (np.complex128, torch.complex128, 1e-13)):
        planner = rustfft_amd.FftPlanner(dt)

        def pf(v):
            out = []
            d = 2
            while d * d <= v:
                while v % d == 0:
                    out.append(d)
                    v //= d
                d += 1
            if v > 1:
                out.append(v)
            return out

        cand = {"smooth4": [], "smooth5": [], "rader31": [], "lsm": []}
        for n in rng.permutation(np.arange(38, 16385)):
            n = int(n)
            f = pf(n)
            d = None
            if len(f) == 1:
                if any(q in (17, 19, 23, 29, 31) for q in pf(n - 1)) and max(pf(n - 1)) <= 31 and n <= 4096 and len(cand["rader31"]) < 25:
                    d = planner.plan_fft_forward(n).describe()
                    if d.startswith("rader<"):
                        cand["rader31"].append(n)
                continue
            if max(f) <= 13 and max(f) >= 11 and n > 4096 and len(cand["smooth4"]) < 25:
                cand["smooth4"].append(n)
            elif 17 <= max(f) <= 31 and n > (4096 if dt == np.complex64 else 2048) and len(cand["smooth5"]) < 25:
                cand["smooth5"].append(n)
            elif max(f) > 31 and len(cand["lsm"]) < 25:
                d = planner.plan_fft_forward(n).describe()
                if d.startswith("lsm<"):
                    cand["lsm"].append(n)
        for fam, ns in cand.items():
            for n in ns:
                fft = planner.plan_fft_forward(n)
                d = fft.describe()
                m = re.search(r"F(\d+)", d)
                F = int(m.group(1)) if m else 1
                rows = min(2 * 256 * F + 1, max(8, (1 << 28) // (n * np.dtype(dt).itemsize)))
                x = (rng.uniform(-1, 1, rows * n) + 1j * rng.uniform(-1, 1, rows * n)).astype(dt)
                t = torch.from_numpy(x).cuda()
                fft.process(t)
                got = t.cpu().numpy().reshape(rows, n).astype(np.complex128)
                want = np.fft.fft(x.reshape(rows, n).astype(np.complex128), axis=1)
                per_row = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
                ok = bool(per_row.max() < tol)
                bad += 0 if ok else 1
                print(json.dumps({"family": fam, "dtype": np.dtype(dt).name, "n": n, "rows": rows, "worst_row_rel_l2": float(per_row.max()), "ok": ok, "plan": d[:90]}), flush=True)
    print(json.dumps({"failures": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
