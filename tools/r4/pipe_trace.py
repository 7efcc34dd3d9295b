#!/usr/bin/env python3
"""Runs a few forward transforms of one length through the tuning-min library (environment decides the pipeline mode);
meant to be run under `rocprofv3 --kernel-trace --stats`, so that the per-kernel average durations of chunk launches can be
compared with those of full-batch launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rustfft_amd
from rustfft_amd import _native
log2n, batch, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = _native.load(os.path.join(ROOT, "rustfft_amd", "lib", "libmi355fft_tuning_min.so"))
pl = rustfft_amd.FftPlannerHip(np.complex64, lib=lib)
f = pl.plan_fft_forward(1 << log2n)
x = torch.empty(batch << log2n, dtype=torch.complex64, device="cuda")
torch.view_as_real(x).uniform_(-1, 1)
x.mul_(1e-20)
for _ in range(reps):
    f.process(x)
torch.cuda.synchronize()
print(f.describe())
