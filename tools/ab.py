#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants on ONE box in ONE process (boxes differ by several per cent, and so do
back-to-back processes on one box: only interleaved rounds separate a 2 % effect from that noise).

Arms are `lib[:VAR=value[,VAR=value...]]`: `lib` is "default" (rustfft_amd/lib/libmi355fft.so), "min" (libmi355fft_tuning_min.so, `make tuning-min`: power-of-two kernels only) or "tuning"
(libmi355fft_tuning.so, `make -C rustfft_amd/csrc tuning`); the MI355FFT_* variables are set while the arm's plans are
created (tuning builds read them at plan creation, the shipped build reads none).
Example: python tools/ab.py --log2n 20 --batch 1024 default tuning:MI355FFT_VARIANT=10
Prints one JSON line per arm: median / min milliseconds of a forward + inverse pair and per-kernel medians."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("arms", nargs="+")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--oop", action="store_true", help="out-of-place calls x -> y (no plan-owned workspace: every arm touches the same two buffers)")
    ap.add_argument("--dist", default="pm1", choices=["pm1", "bench", "zero"], help="input distribution: U[-1,1) | bench.py's U[0,10) * 2^-100 | zeros")
    ap.add_argument("--instances", type=int, default=1, help="plans per arm, each with its own workspace / ring allocations (identical plans run up to 3.6 %% apart depending on which device allocation holds their workspace: the median over instances separates an arm's effect from that lottery)")
    ap.add_argument("--fwd-only", action="store_true", help="time forward transforms only (2 per iteration) instead of forward + inverse pairs")
    ap.add_argument("--check-all", action="store_true", help="compare every arm's forward output over the WHOLE batch with arm 0's (max abs difference)")
    ap.add_argument("--shift-mib", type=float, default=0, help="allocate this many MiB first (moves the buffers' relative addresses)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import rustfft_amd
    from rustfft_amd import _native

    n = args.n or (1 << args.log2n)
    dt, tdt, esz = (np.complex64, torch.complex64, 8) if args.dtype == "f32" else (np.complex128, torch.complex128, 16)
    libs = {}
    arms = []
    for spec in args.arms:
        name, _, envs = spec.partition(":")
        names = {"default": "libmi355fft.so", "tuning": "libmi355fft_tuning.so", "min": "libmi355fft_tuning_min.so"}
        path = os.path.join(ROOT, "rustfft_amd", "lib", names.get(name, name))  # any other name: a library file in rustfft_amd/lib (an alternative build)
        if path not in libs:
            libs[path] = _native.load(path)
        kv = dict(e.split("=") for e in envs.split(",") if e)
        fused = kv.pop("FUSED", None)  # not an environment variable: mi355fft_plan_set_fused on the arm's plans (works with the shipped library)
        os.environ.update(kv)
        inst = []
        for _ in range(args.instances):
            planner = rustfft_amd.FftPlannerHip(dt, lib=libs[path])  # a planner caches one plan per (len, direction): one planner per instance
            inst.append((planner.plan_fft_forward(n), planner.plan_fft_inverse(n)))
            if fused is not None:
                for f in inst[-1]:
                    f.set_fused(int(fused))
        for k in kv:
            del os.environ[k]
        arms.append({"spec": spec, "fwd": inst[0][0], "inv": inst[0][1], "inst": inst, "pair_ms": [], "inst_ms": [[] for _ in inst], "kernel_ms": []})
    pad = torch.empty(int(args.shift_mib * (1 << 20)), dtype=torch.uint8, device="cuda") if args.shift_mib else None
    x = torch.empty(args.batch * n, dtype=tdt, device="cuda")

    def refill():
        if args.dist == "pm1":
            torch.view_as_real(x).uniform_(-1.0, 1.0)
        elif args.dist == "bench":
            torch.view_as_real(x).uniform_(0.0, 10.0)
            x.mul_(2.0 ** -100)
        else:
            x.zero_()

    y2 = torch.empty_like(x) if args.oop else None
    torch.view_as_real(x).uniform_(-1.0, 1.0)
    x0 = x[:n].cpu().numpy()
    want = np.fft.fft(x0.astype(np.complex128))
    yref = None
    for a in arms:  # correctness of row 0 + warm-up
        y = x.clone()
        a["fwd"].process(y)
        torch.cuda.synchronize()
        a["rel_l2"] = float(np.linalg.norm(y[:n].cpu().numpy() - want) / np.linalg.norm(want))
        if args.check_all:
            if yref is None:
                yref = y.clone()
                a["max_abs_diff_vs_arm0"] = 0.0
            else:
                a["max_abs_diff_vs_arm0"] = float((torch.view_as_real(y) - torch.view_as_real(yref)).abs().max().item())
        a["inv"].process(y)
        a["fused_status"] = a["fwd"].fused_status() | a["inv"].fused_status()
        del y
    del yref
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for a in arms:  # warm every further instance (first call: workspace / ring allocation)
        for fwd, inv in a["inst"][1:]:
            fwd.process(x)
            inv.process(x)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for a in arms:
            for ii, (fwd, inv) in enumerate(a["inst"]):
                refill()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    if args.oop:
                        fwd.process_outofplace_with_scratch(x, y2)
                        inv.process_outofplace_with_scratch(y2, x)
                    elif args.fwd_only:
                        fwd.process(x)
                        fwd.process(x)
                    else:
                        fwd.process(x)
                        inv.process(x)
                e1.record()
                torch.cuda.synchronize()
                a["pair_ms"].append(e0.elapsed_time(e1) / args.iters)
                a["inst_ms"][ii].append(a["pair_ms"][-1])
            refill()
            a["kernel_ms"].append(a["fwd"].profile_kernels(x, reps=2))
    alg = args.batch * 2 * n * esz
    for a in arms:
        km = [statistics.median(r[i] for r in a["kernel_ms"]) for i in range(len(a["kernel_ms"][0]))]
        print(json.dumps({"arm": a["spec"], "n": n, "batch": args.batch, "pair_ms_median": round(statistics.median(a["pair_ms"]), 4),
                          "pair_ms_min": round(min(a["pair_ms"]), 4), "instance_medians_ms": [round(statistics.median(v), 4) for v in a["inst_ms"]], "kernel_ms_median": [round(k, 4) for k in km],
                          "kernel_GBps": [round(alg / k / 1e6) for k in km], "rel_l2_row0": a["rel_l2"], "max_abs_diff_vs_arm0": a.get("max_abs_diff_vs_arm0"), "fused_status": a.get("fused_status"), "plan": a["fwd"].describe()}), flush=True)


if __name__ == "__main__":
    main()
