#!/usr/bin/env python3
"""Per-call duration of consecutive fused / two-launch forward transforms right after plan creation and buffer allocation: does
the fused launch need a warm-up (tools/sweep.py measured it 8 % slower than tools/ab.py)?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rustfft_amd
log2n, gib, calls = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
n = 1 << log2n
batch = int(gib * 2**30) // (n * 8)
for fused in (1, 0, 1):
    fft = rustfft_amd.FftPlanner(np.complex64).plan_fft_forward(n)
    fft.set_fused(fused)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    x.mul_(1e-30)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(calls):
        fft.process(x)
        ev[i + 1].record()
        if i % 6 == 5:
            x.mul_(1e-18)  # keep the unnormalised data finite (outside no event pair: it lands in the NEXT interval, marked below)
    torch.cuda.synchronize()
    ms = [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(calls)]
    print(json.dumps({"log2n": log2n, "batch": batch, "fused": bool(fft.is_fused()), "per_call_ms": ms, "note": "every 7th interval (index 6, 12, ...) includes a rescaling pass"}))
    del x, fft
