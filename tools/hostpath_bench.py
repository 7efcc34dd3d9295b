#!/usr/bin/env python3
"""Host-slice path (mi355fft_process_*_host: the literal drop-in for a Rust caller's `&mut [Complex<T>]`): payload GB/s of one
call (bytes of the slice / wall time of the call, which moves them to the GPU and back) for one thread, and for four threads
sharing ONE plan (examples/concurrency.rs:9-30) -- each takes its own staging context, so the calls overlap.
Prints one JSON line per workload."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rustfft_amd


def main():
    pl = rustfft_amd.FftPlanner(np.complex64)
    for n, batch in ((1024, 1 << 17), (1 << 20, 128), (1009, 100000), (1 << 20, 1024)):
        x = (np.random.default_rng(0).uniform(0, 1, n * batch) + 0j).astype(np.complex64)
        fft = pl.plan_fft_forward(n)
        fft.process(x)  # first call: staging buffers, streams
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fft.process(x)
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[1]
        gb = x.nbytes / 1e9
        out = {"n": n, "batch": batch, "GB": round(gb, 3), "one_thread_ms": round(dt * 1e3, 2), "one_thread_payload_GBps": round(gb / dt, 2),
               "GFLOPs": round(5 * n * np.log2(n) * batch / dt / 1e9, 1)}
        if gb <= 2.5:
            xs = [x.copy() for _ in range(4)]
            for xx in xs[1:]:
                fft.process(xx)  # warm the extra contexts
            t0 = time.perf_counter()
            th = [threading.Thread(target=fft.process, args=(xx,)) for xx in xs]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt4 = time.perf_counter() - t0
            out.update({"four_threads_ms": round(dt4 * 1e3, 2), "four_threads_payload_GBps": round(4 * gb / dt4, 2),
                        "speedup_over_serial": round(4 * dt / dt4, 2)})
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
