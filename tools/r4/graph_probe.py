#!/usr/bin/env python3
"""Which plans replay correctly from a HIP graph?  For each length: forward captured after a warm-up call, replayed, compared with the direct call."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import rustfft_amd

for n, batch, fused in ((4096, 64, -1), (8192, 64, -1), (16384, 64, -1), (32768, 32, -1), (1 << 16, 96, -1), (1 << 17, 96, -1), (1 << 19, 48, -1), (1 << 20, 40, -1), (1 << 20, 40, 0), (1 << 22, 8, -1), (1 << 23, 8, -1), (1 << 23, 8, 0)):
    fwd = rustfft_amd.FftPlanner(np.complex64).plan_fft_forward(n)
    if fused >= 0:
        fwd.set_fused(fused)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    want = x.clone()
    fwd.process(want)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    buf = x.clone()
    with torch.cuda.stream(s):
        tmp = x.clone()
        fwd.process(tmp)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    err = None
    try:
        with torch.cuda.graph(g, stream=s):
            fwd.process(buf)
        g.replay()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        err = str(e)[:200]
    same = bool(torch.equal(torch.view_as_real(buf), torch.view_as_real(want)))
    untouched = bool(torch.equal(torch.view_as_real(buf), torch.view_as_real(x)))
    lib = rustfft_amd._native.load()
    print(json.dumps({"n": n, "batch": batch, "plan": fwd.describe(), "graph_ok": same, "untouched": untouched, "error": err, "last_error": lib.mi355fft_last_error().decode()}), flush=True)
