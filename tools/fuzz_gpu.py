#!/usr/bin/env python3
"""One-off robustness sweep on the GPU: random lengths (log-uniform in [2, 3e6]) x random ragged batches x both precisions
x both directions x the three API modes, against numpy.fft in complex128.  Prints one line per failure and a summary.
Usage: python tools/fuzz_gpu.py [count] [seed] [device]   (third argument "device": HBM-resident torch tensors through the
_dev entry points instead of host slices)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfft_amd  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    on_device = len(sys.argv) > 3 and sys.argv[3] == "device"
    if on_device:
        import torch
    rng = np.random.default_rng(seed)
    planners = {np.complex64: rustfft_amd.FftPlanner(np.complex64), np.complex128: rustfft_amd.FftPlanner(np.complex128)}
    tol = {np.complex64: 5e-6, np.complex128: 1e-13}
    kinds, bad = {}, 0
    for it in range(count):
        n = int(np.exp(rng.uniform(np.log(2), np.log(3e6))))
        if it % 2 == 1:  # smooth lengths: products of small primes (the compiled schedules and the column-tile passes)
            n = 1
            while True:
                p = int(rng.choice([2, 2, 2, 3, 3, 5, 5, 7, 11, 13]))
                if n * p > 3_000_000 or (n > 16 and rng.uniform() < 0.12):
                    break
                n *= p
            n = max(n, 2)
        if os.environ.get("FUZZ_ROUND6") == "1":
            # round-6 plans: lengths up to 16384 with a prime factor above 31 -- the LDS stage machine where the planner takes it (short trees),
            # Bluestein where it does not
            while True:
                n = int(np.exp(rng.uniform(np.log(38), np.log(16384))))
                m = n
                for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
                    while m % q == 0:
                        m //= q
                if m > 1:
                    break
        if os.environ.get("FUZZ_ROUND3") == "1":
            # round-3 plans: composites with prime factors 37 .. 631 (prime tile heights), primes above 8192 with 13-smooth p - 1
            # (multi-kernel Rader), the small primes served by the side-by-side Rader bodies (batched loads)
            k2r = [p for p in range(37, 640) if all(p % q for q in range(2, int(p**0.5) + 1))]
            if it % 3 == 0:
                n = int(rng.choice(k2r)) * int(rng.choice(k2r + [25, 32, 49, 64, 100, 128, 243, 360, 512, 625]))
                if n <= 4096 or rng.uniform() < 0.25:
                    n *= int(rng.choice([27, 32, 35, 64, 100]))
            elif it % 3 == 1:
                while True:
                    n = 1
                    while n < 8192:
                        n *= int(rng.choice([2, 2, 2, 3, 3, 5, 7, 11, 13]))
                    n += 1
                    if n < 1_500_000 and all(n % q for q in range(2, int(n**0.5) + 1)):
                        break
            else:
                n = int(rng.choice([p for p in range(17, 1300) if all(p % q for q in range(2, int(p**0.5) + 1))]))
        elif it % 7 == 0:  # edge neighbourhoods of the planner's thresholds
            n = int(rng.choice([4096, 4097, 8191, 8193, 16383, 16384, 16385, 32768, 32769, 2 * 16384 - 1, 409600, 409601])) + int(rng.integers(0, 2))
        dt = np.complex64 if rng.integers(0, 2) else np.complex128
        d = int(rng.integers(0, 2))
        batch = int(rng.integers(1, max(2, min(50, 3_000_000 // n))))
        mode = int(rng.integers(0, 3))
        x = (rng.uniform(-1, 1, n * batch) + 1j * rng.uniform(-1, 1, n * batch)).astype(dt)
        fft = planners[dt].plan_fft(n, d)
        kinds[fft.describe().split("<")[0].split("(")[0]] = kinds.get(fft.describe().split("<")[0].split("(")[0], 0) + 1
        if on_device:
            tx = torch.from_numpy(x).cuda()
            if mode == 0:
                ty = tx.clone()
                fft.process(ty)
            elif mode == 1:
                ty = torch.zeros_like(tx)
                fft.process_outofplace_with_scratch(tx.clone(), ty)
            else:
                ty = torch.zeros_like(tx)
                keep = tx.clone()
                fft.process_immutable_with_scratch(tx, ty)
                assert torch.equal(keep, tx)
            y = ty.cpu().numpy()
        elif mode == 0:
            y = x.copy()
            fft.process(y)
        elif mode == 1:
            y = np.zeros_like(x)
            xin = x.copy()
            fft.process_outofplace_with_scratch(xin, y, np.zeros(fft.get_outofplace_scratch_len(), dtype=dt))
        else:
            y = np.zeros_like(x)
            xin = x.copy()
            fft.process_immutable_with_scratch(xin, y, np.zeros(fft.get_immutable_scratch_len(), dtype=dt))
            assert np.array_equal(xin, x)
        X = x.reshape(batch, n).astype(np.complex128)
        ref = np.fft.fft(X, axis=1) if d == 0 else np.fft.ifft(X, axis=1) * n
        err = np.linalg.norm(y.reshape(batch, n) - ref) / np.linalg.norm(ref)
        if not (err < tol[dt]):
            bad += 1
            print("FAIL", n, dt.__name__, d, batch, mode, f"{err:.3e}", fft.describe(), flush=True)
    print("done:", count, "cases,", bad, "failures; plan kinds:", kinds)


if __name__ == "__main__":
    main()
