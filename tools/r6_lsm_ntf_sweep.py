#!/usr/bin/env python3
"""Round 6: block size x rows-per-workgroup sweep of the LDS stage machine on the device -- what the planner's cost model (lsm_plan.h) is
calibrated against.  Needs a library whose planner reads MI355FFT_LSM_NT / MI355FFT_LSM_F (plan.cpp built with -DMI355_TUNING:
libmi355fft_lsmtune.so).  One JSON line per (length, nt, f): TB/s algorithmic, plus the AUTO choice of the same library."""
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

    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    sizes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "74,122,246,370,592,678,710,938,1110,1351,1582,1834,2368,2892,3297,4070").split(",")]
    dt, tdt, esz = (np.complex64, torch.complex64, 8) if dtype == "f32" else (np.complex128, torch.complex128, 16)
    lib = _native.load(os.path.join(ROOT, "rustfft_amd", "lib", "libmi355fft_lsmtune.so"))
    x = torch.empty((1 << 28) // esz, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def rate(fft, n):
        batch = x.numel() // n
        buf = x[: batch * n]
        fft.process(buf)
        best = 1e9
        for _ in range(3):
            e0.record()
            fft.process(buf)
            fft.process(buf)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 2)
            buf.mul_(1e-4)
        return batch * 2 * n * esz / (best * 1e-3) / 1e12

    for n in sizes:
        prime = n > 3 and all(n % q for q in range(2, int(n**0.5) + 1))
        algo = rustfft_amd.ALGO_RADER if prime else rustfft_amd.ALGO_MIXED_RADIX
        os.environ.pop("MI355FFT_LSM_NT", None)
        os.environ.pop("MI355FFT_LSM_F", None)
        auto = rustfft_amd.FftPlannerHip(dt, lib=lib).plan_fft_with(n, 0, algorithm=algo)
        print(json.dumps({"n": n, "auto": auto.describe(), "TBps": round(rate(auto, n), 3)}), flush=True)
        for nt in (64, 128, 256, 512):
            seen = set()
            for f in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                os.environ["MI355FFT_LSM_NT"] = str(nt)
                os.environ["MI355FFT_LSM_F"] = str(f)
                try:
                    fft = rustfft_amd.FftPlannerHip(dt, lib=lib).plan_fft_with(n, 0, algorithm=algo)
                except Exception:
                    break
                d = fft.describe()
                if not d.startswith("lsm<") or d in seen:
                    break
                seen.add(d)
                print(json.dumps({"n": n, "nt": nt, "f": f, "plan": d[-14:], "TBps": round(rate(fft, n), 3)}), flush=True)


if __name__ == "__main__":
    main()
