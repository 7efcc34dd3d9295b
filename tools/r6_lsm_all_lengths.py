#!/usr/bin/env python3
"""Round 6: EVERY length in [38, 16384] that AUTO plans as the LDS stage machine, both precisions, both directions, on the device against numpy
float64 (a workgroup's worth of rows + 1).  The -m gpu test covers every moved length up to 4096 and every 29th above; this is the exhaustive pass,
kept as a log (profiles/r6/lsm_all_lengths.json): counts, worst relative L2 error per precision, any failure by length."""
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import rustfft_amd

    out = {}
    for dt, tol in ((np.complex64, 5e-6), (np.complex128, 1e-13)):
        planner = rustfft_amd.FftPlanner(dt)
        rng = np.random.default_rng(6)
        count, worst, worst_n, failures = 0, 0.0, 0, []
        by_stages = {}
        for n in range(38, 16385):
            m = n
            for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
                while m % q == 0:
                    m //= q
            if m == 1:
                continue
            for d in (0, 1):
                fft = planner.plan_fft(n, d)
                desc = fft.describe()
                if not desc.startswith("lsm<"):
                    break
                F = int(re.search(r"sF(\d+)$", desc).group(1))
                stg = int(re.search(r"t(\d+)sF", desc).group(1))
                rows = F + 1
                x = (rng.uniform(-1, 1, rows * n) + 1j * rng.uniform(-1, 1, rows * n)).astype(dt)
                t = torch.from_numpy(x).cuda()
                fft.process(t)
                got = t.cpu().numpy().reshape(rows, n).astype(np.complex128)
                xx = x.reshape(rows, n).astype(np.complex128)
                want = np.fft.fft(xx, axis=1) if d == 0 else np.fft.ifft(xx, axis=1) * n
                err = float(np.linalg.norm(got - want) / np.linalg.norm(want))
                if not (err < tol):
                    failures.append([n, d, err, desc])
                if err > worst:
                    worst, worst_n = err, n
                if d == 0:
                    count += 1
                    by_stages[stg] = by_stages.get(stg, 0) + 1
        out[np.dtype(dt).name] = {"lengths": count, "directions": 2, "worst_rel_l2": worst, "worst_at": worst_n, "failures": failures, "lengths_by_stages": dict(sorted(by_stages.items()))}
        print(np.dtype(dt).name, out[np.dtype(dt).name], flush=True)
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "lsm_all_lengths.json", "w"), indent=1)
    sys.exit(1 if any(v["failures"] for v in out.values()) else 0)


if __name__ == "__main__":
    main()
