#!/usr/bin/env python3
"""Latencies a caller sees outside the transforms themselves (VERDICT r3 ask 7): library load, planner construction, plan
creation (p50 / p99 / max over every length 1 .. 1000 and a few large ones; tables are built on the host and uploaded), and the
FIRST process() call of a plan against the second (the first launch of a kernel loads its translation unit's code object --
decompressing it when the library was built with --offload-compress).  One JSON object."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t1 = time.perf_counter()
import rustfft_amd  # noqa: E402
from rustfft_amd import _native  # noqa: E402

lib = _native.load(sys.argv[1]) if len(sys.argv) > 1 else _native.load()
t2 = time.perf_counter()
pl = rustfft_amd.FftPlannerHip(np.complex64, lib=lib)
t3 = time.perf_counter()
create, first, second = [], [], []
for n in list(range(1, 1001)) + [1009, 4099, 1 << 13, 1 << 16, 1 << 20, 1 << 22, 10007, 12289, 100003]:
    a = time.perf_counter()
    f = pl.plan_fft_forward(n)
    b = time.perf_counter()
    x = torch.zeros(n * 4, dtype=torch.complex64, device="cuda")
    torch.cuda.synchronize()
    c = time.perf_counter()
    f.process(x)
    torch.cuda.synchronize()
    d = time.perf_counter()
    f.process(x)
    torch.cuda.synchronize()
    e = time.perf_counter()
    create.append((b - a) * 1e3)
    first.append((d - c) * 1e3)
    second.append((e - d) * 1e3)


def q(v, p):
    s = sorted(v)
    return round(s[min(len(s) - 1, int(p * len(s)))], 3)


print(json.dumps({"library": os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "libmi355fft.so", "library_bytes": os.path.getsize(sys.argv[1] if len(sys.argv) > 1 else _native.LIB_PATH),
                  "torch_cuda_init_s": round(t1 - t0, 2), "library_load_s": round(t2 - t1, 3), "planner_s": round(t3 - t2, 3), "lengths": len(create),
                  "plan_create_ms": {"p50": q(create, 0.5), "p99": q(create, 0.99), "max": round(max(create), 3), "sum": round(sum(create), 1)},
                  "first_process_ms": {"p50": q(first, 0.5), "p99": q(first, 0.99), "max": round(max(first), 3), "sum": round(sum(first), 1)},
                  "second_process_ms": {"p50": q(second, 0.5), "p99": q(second, 0.99), "max": round(max(second), 3)},
                  "what": "Complex<f32>, forward, 4 rows, device-resident; first call of a plan includes the code-object load of its kernels' translation unit"}))
