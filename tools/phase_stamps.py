#!/usr/bin/env python3
"""Phase timeline of the column-tile kernels from in-kernel timestamps (tuning PROBE builds only: DevExec::stamp, ABL bit 12;
the library must export mi355fft_debug_read_stamps -- see profiles/r3/stamp_probe.patch).  EVERY wave of the first 2048
workgroups records s_memtime at: 0 kernel entry, 1 its first loads landed, 2 + 5P sub-pass P's arithmetic done,
3..6 + 5P the four barriers of exchange P (after scatter re / gather re / scatter im / gather im), 13 its stores acknowledged.
Prints, for the LAST kernel of the plan: the median time a wave spends between consecutive stamps, and for every stamp that
follows a barrier the median skew between the first and the last wave of a workgroup reaching the PREVIOUS stamp (i.e. how
long the early waves waited at that barrier).
usage: python tools/phase_stamps.py --lib libmi355fft_exp.so --log2n 22 --batch 64 --variant 50"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--variant", type=int, default=50)
    ap.add_argument("--waves", type=int, default=8, help="waves per workgroup of the last kernel")
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
    nwg, slots = 2048, 16
    buf = np.zeros(nwg * 16 * slots, dtype=np.uint64)
    lib.mi355fft_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = lib.mi355fft_debug_read_stamps(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    s = buf.reshape(nwg, 16, slots)[:, :args.waves, :].astype(np.int64)
    ids = [i for i in range(14) if (s[:, :, i] != 0).all()]
    t = s[:, :, ids]
    t = t - t[:, :, 0].min(axis=1)[:, None, None]  # per workgroup: relative to its first wave's entry
    names = {0: "entry", 1: "loads_landed", 13: "stores_acked"}
    for q in range(3):
        names[2 + 5 * q] = f"sp{q}_arith_done"
        for j, nm in enumerate(("scatter_re", "gather_re", "scatter_im", "gather_im")):
            names[3 + 5 * q + j] = f"x{q}_barrier_after_{nm}"
    med = lambda a: float(np.median(a))
    steps = {}
    for i in range(len(ids) - 1):
        d = t[:, :, i + 1] - t[:, :, i]
        skew_prev = t[:, :, i].max(axis=1) - t[:, :, i].min(axis=1)
        steps[f"{names[ids[i]]} -> {names[ids[i + 1]]}"] = {"median_wave": med(d), "fastest_wave": med(d.min(axis=1)), "slowest_wave": med(d.max(axis=1)),
                                                           "skew_of_waves_at_start": med(skew_prev)}
    total = t[:, :, -1].max(axis=1) - t[:, :, 0].min(axis=1)
    out = {"plan": fft.describe(), "stamp_ids": ids, "steps_ticks": steps, "median_workgroup_total": med(total)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
