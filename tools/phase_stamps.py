#!/usr/bin/env python3
"""Phase timeline of the column-tile kernels from in-kernel timestamps (tuning probe builds only: DevExec::stamp,
ABL bit 12; the library must export mi355fft_debug_read_stamps).  Wave 0 of every workgroup records s_memtime at kernel
entry (0), when its first loads have landed (1), when the last sub-pass starts (2) and when its stores are acknowledged (3),
plus HW_ID / XCC_ID.  Prints per-phase medians and the per-CU occupancy picture of the LAST kernel of the plan.
usage: python tools/phase_stamps.py --lib libmi355fft_exp.so --log2n 22 --batch 64 --variant 50"""
import argparse
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--variant", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    import torch

    import rustfft_amd
    from rustfft_amd import _native

    lib = _native.load(os.path.join(ROOT, "rustfft_amd", "lib", args.lib))
    os.environ["MI355FFT_VARIANT"] = str(args.variant)
    os.environ["MI355FFT_STAMPS"] = "1"
    n = 1 << args.log2n
    planner = rustfft_amd.FftPlannerHip(np.complex64, lib=lib)
    fft = planner.plan_fft_forward(n)
    x = torch.randn(args.batch * n, dtype=torch.complex64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        fft.process_outofplace_with_scratch(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fft.process_outofplace_with_scratch(x, y)
    e1.record()
    torch.cuda.synchronize()
    nwg = 65536
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    lib.mi355fft_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = lib.mi355fft_debug_read_stamps(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    s = buf.reshape(nwg, 8)
    used = s[:, 0] != 0
    s = s[used]
    t = s[:, :4].astype(np.int64)
    base = t[:, 0].min()
    t -= base
    span = int(t[:, 3].max())
    ms_pair = e0.elapsed_time(e1)
    med = lambda a: float(np.median(a))
    hw = s[:, 4].astype(np.int64)
    xcc = s[:, 5].astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    out = {"plan": fft.describe(), "workgroups_recorded": int(used.sum()), "span_ticks_last_kernel": span, "pair_ms_all_kernels": ms_pair,
           "median_ticks": {"load(0->1)": med(t[:, 1] - t[:, 0]), "compute_to_last_subpass(1->2)": med(t[:, 2] - t[:, 1]),
                            "last_subpass+stores(2->3)": med(t[:, 3] - t[:, 2]), "total(0->3)": med(t[:, 3] - t[:, 0])},
           "p10_p90_total": [float(np.percentile(t[:, 3] - t[:, 0], 10)), float(np.percentile(t[:, 3] - t[:, 0], 90))],
           "distinct_cu_keys": int(len(set(key.tolist())))}
    # per-CU: fraction of the span in which NO resident workgroup is between stamps 0..1 or 2..3 (i.e. nobody is moving data)
    idle_frac, both_mem = [], []
    for k in list(set(key.tolist()))[:64]:
        rows = t[key == k]
        ev = []
        for r in rows:
            ev.append((r[0], +1)); ev.append((r[1], -1)); ev.append((r[2], +1)); ev.append((r[3], -1))
        ev.sort()
        cur, last, idle, busy2 = 0, ev[0][0], 0, 0
        for tt, d in ev:
            if cur == 0:
                idle += tt - last
            if cur >= 2:
                busy2 += tt - last
            cur += d
            last = tt
        tot = ev[-1][0] - ev[0][0]
        idle_frac.append(idle / max(tot, 1))
        both_mem.append(busy2 / max(tot, 1))
    out["per_cu_frac_time_no_workgroup_in_a_memory_phase"] = statistics.median(idle_frac)
    out["per_cu_frac_time_two_workgroups_in_memory_phases"] = statistics.median(both_mem)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
