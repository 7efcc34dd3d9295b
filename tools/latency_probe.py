import sys, time
sys.path.insert(0, "/root/repo")
t0 = time.perf_counter()
import numpy as np, torch
import rustfft_amd
t1 = time.perf_counter()
pl = rustfft_amd.FftPlanner(np.complex64)
t2 = time.perf_counter()
fft = pl.plan_fft_forward(1024)
t3 = time.perf_counter()
x = torch.zeros(1024 * 4, dtype=torch.complex64, device="cuda")
torch.cuda.synchronize()
t4 = time.perf_counter()
fft.process(x); torch.cuda.synchronize()
t5 = time.perf_counter()
fft.process(x); torch.cuda.synchronize()
t6 = time.perf_counter()
print(f"import {t1-t0:.2f}s planner {t2-t1:.3f}s plan(1024) {t3-t2:.3f}s first process {t5-t4:.3f}s second {1e6*(t6-t5):.0f}us")
for n in (1009, 5000, 1 << 20, 10007, 100003):
    a = time.perf_counter(); f = pl.plan_fft_forward(n); b = time.perf_counter()
    y = torch.zeros(n * 2, dtype=torch.complex64, device="cuda"); torch.cuda.synchronize()
    c = time.perf_counter(); f.process(y); torch.cuda.synchronize(); d = time.perf_counter()
    f.process(y); torch.cuda.synchronize(); e = time.perf_counter()
    print(f"n={n}: plan {1e3*(b-a):.1f} ms, first process {1e3*(d-c):.1f} ms, second {1e6*(e-d):.0f} us  {f.describe()[:50]}")
