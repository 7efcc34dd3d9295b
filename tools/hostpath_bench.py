import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, rustfft_amd
pl = rustfft_amd.FftPlanner(np.complex64)
for n, batch in ((1024, 1 << 17), (1 << 20, 128), (1009, 100000)):
    x = (np.random.default_rng(0).uniform(0, 1, n * batch) + 0j).astype(np.complex64)
    fft = pl.plan_fft_forward(n)
    fft.process(x)
    t0 = time.perf_counter()
    for _ in range(3):
        fft.process(x)
    dt = (time.perf_counter() - t0) / 3
    gb = x.nbytes / 1e9
    print(f"n={n} batch={batch} {gb:.2f} GB: {dt*1e3:.1f} ms  -> {gb/dt:.1f} GB/s one-way equivalent, {5*n*np.log2(n)*batch/dt/1e9:.1f} GFLOP/s")
