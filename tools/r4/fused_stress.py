#!/usr/bin/env python3
"""Repeatability of the fused two-pass kernel under load: K forward transforms of the same input, every result compared BIT
FOR BIT with the first one (a stale read of the ring shows up as a difference long before it breaks a tolerance), the error
word checked after every call, and the first result compared with the two-launch plan over the whole batch."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rustfft_amd
from rustfft_amd import _native
log2n, batch, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
libname = sys.argv[4] if len(sys.argv) > 4 else "libmi355fft_tuning_min.so"
lib = _native.load(os.path.join(ROOT, "rustfft_amd", "lib", libname))
n = 1 << log2n
ref = rustfft_amd.FftPlannerHip(np.complex64, lib=lib).plan_fft_forward(n)
os.environ["MI355FFT_FUSE"] = sys.argv[5] if len(sys.argv) > 5 else "7"
fus = rustfft_amd.FftPlannerHip(np.complex64, lib=lib).plan_fft_forward(n)
del os.environ["MI355FFT_FUSE"]
x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
torch.view_as_real(x).uniform_(-1, 1)
y0 = x.clone(); ref.process(y0)
first = None; bad = 0; status = 0; maxd = 0.0
side = torch.cuda.Stream()
for r in range(reps):
    y = x.clone()
    if r % 2:  # uneven load: a copy kernel hammering HBM on another stream while the fused launch runs
        with torch.cuda.stream(side):
            z = x.clone()
    fus.process(y)
    status |= fus.fused_status()
    if first is None:
        first = y
        maxd = float((torch.view_as_real(y) - torch.view_as_real(y0)).abs().max().item())
    elif not torch.equal(torch.view_as_real(y), torch.view_as_real(first)):
        bad += 1
torch.cuda.synchronize()
print(json.dumps({"n": n, "batch": batch, "reps": reps, "mismatching_runs": bad, "fused_status": status, "max_abs_diff_vs_two_launch": maxd, "plan": fus.describe()}))
