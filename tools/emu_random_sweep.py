#!/usr/bin/env python3
"""Random lengths through AUTO on the kernel-body EMULATOR (tests/emu/libmi355fft_emu.so: the same planner, registry and kernel bodies compiled
for the CPU, no GPU needed), each against numpy in float64.  --order reverse runs every phase from the last thread to the first (a race between
threads of one phase shows up as a wrong result).  Prints one line per precision and the plan families met.
    python tools/emu_random_sweep.py --seed 2 --count 1200 --lo 2 --hi 120000 --rows 2 --order reverse"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--count", type=int, default=1200)
    ap.add_argument("--lo", type=int, default=2)
    ap.add_argument("--hi", type=int, default=120000)
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--order", default="reverse", choices=["", "reverse"])
    args = ap.parse_args()
    if args.order:
        os.environ["MI355_EMU_ORDER"] = args.order
    import numpy as np

    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(ROOT, "tests", "emu", "libmi355fft_emu.so"))
    rng = np.random.default_rng(args.seed)
    families = {}
    t0 = time.time()
    for dt, tol in ((np.complex64, 5e-6), (np.complex128, 1e-13)):
        planner = rustfft_amd.FftPlannerHip(dt, lib=lib)
        worst = 0.0
        for n in (int(v) for v in np.exp(rng.uniform(np.log(args.lo), np.log(args.hi), args.count))):
            d = n % 2
            x = (rng.uniform(-1, 1, args.rows * n) + 1j * rng.uniform(-1, 1, args.rows * n)).astype(dt)
            y = x.copy()
            fft = planner.plan_fft(n, d)
            k = fft.describe().split("<")[0].split("(")[0]
            families[k] = families.get(k, 0) + 1
            fft.process(y)
            X = x.astype(np.complex128).reshape(args.rows, n)
            want = (np.fft.ifft(X, axis=1) * n if d else np.fft.fft(X, axis=1)).reshape(-1)
            err = float(np.linalg.norm(y - want) / np.linalg.norm(want))
            worst = max(worst, err)
            if err >= tol:
                print("FAIL", n, d, err, fft.describe(), flush=True)
                sys.exit(1)
            fft.trim_workspaces()
        print(f"{np.dtype(dt).name}: {args.count} lengths log-uniform in [{args.lo}, {args.hi}], seed {args.seed}, {args.rows} row(s), thread order "
              f"{args.order or 'forward'}: worst relative L2 against numpy complex128 {worst:.2e} (bar {tol:g}), {time.time() - t0:.0f} s", flush=True)
    print("plan families:", dict(sorted(families.items())))


if __name__ == "__main__":
    main()
