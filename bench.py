#!/usr/bin/env python3
"""bench.py — the hot-path benchmark (BASELINE.json metric: GFLOP/s at 5*N*log2(N) flops per transform,
plus the HBM roofline fraction of the dominant kernel).

Workload at N=1 GPU = BASELINE.json configs[1]: batched power-of-two N = 2^20 Complex<f32>, batch = 1024,
forward + inverse, device-resident (inputs in HBM when the timed region starts).  One "step" = one
forward and one inverse pass of Fft::process over the whole batch.  With --gpus G every rank owns its
own 1024-transform shard (weak scaling, no data-path collective; the batch dimension is the shard axis).

Methodology mirrors benches/bench_rustfft.rs:43-54 of the reference: plan once, allocate once, time
only the process calls.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def copy_ceiling_gbps(torch, nbytes=1 << 31):
    """Attainable HBM rate of this box for a plain device copy (read + write bytes / time): SURVEY.md section 8(d) asks for
    the roofline fraction against both the 8 TB/s spec and the measured copy bandwidth."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(n, log):
    """Oracle (C++ restatement of RustFFT's scalar path, kind="port") timed on this box's host cores:
    every core owns a slice of the sample batch and shares one plan (examples/concurrency.rs:9-30)."""
    import numpy as np

    from oracle import rustfft_oracle as oracle

    oracle.build()
    cores = os.cpu_count() or 1
    fwd, inv = oracle.plan(np.complex64, n, 0), oracle.plan(np.complex64, n, 1)
    rng = np.random.default_rng(1)
    one = (rng.uniform(0, 10, n) + 1j * rng.uniform(0, 10, n)).astype(np.complex64)
    t1 = fwd.time_batch(one.copy(), 1, 1, 1)
    target_s = 12.0
    per_core = max(1, int(target_s / (2 * max(t1, 1e-4))))
    batch = min(per_core * cores, 4096)
    buf = np.tile(one * np.float32(1e-30), batch)
    t = fwd.time_batch(buf, batch, 1, cores) + inv.time_batch(buf, batch, 1, cores)
    flops = 2 * batch * 5.0 * n * math.log2(n)
    log(f"cpu_baseline: {batch} transforms fwd+inv on {cores} threads in {t:.2f}s")
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": flops / t / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port", "cpu_model": model,
           "sample": f"N=2^{int(math.log2(n))} Complex<f32>, {batch} transforms forward+inverse, {cores} caller threads sharing one plan "
                     f"(scalar-path restatement compiled -O2 -ffp-contract=off, not the RustFFT binary; no rustc/cargo on this box)"}
    try:  # the same restatement compiled -O3 -march=native on this host
        tn = oracle.time_batch_native(np.complex64, n, 0, buf, batch, 1, cores)
        ti = oracle.time_batch_native(np.complex64, n, 1, buf, batch, 1, cores)
        if tn and ti:
            out["native_build"] = {"value": flops / (tn + ti) / 1e9, "unit": "GFLOP/s", "flags": "-O3 -march=native -ffp-contract=off"}
    except Exception as e:
        log(f"native oracle build unavailable: {e}")
    try:  # external CPU yardstick asked for by SURVEY.md section 8(d): pocketfft (scipy.fft) on every core, same workload shape
        import scipy.fft

        yb = min(batch, 512)
        ybuf = np.tile(one, yb).reshape(yb, n)
        scipy.fft.fft(ybuf[:8], axis=1, workers=cores)
        t0 = time.perf_counter()
        scipy.fft.ifft(scipy.fft.fft(ybuf, axis=1, workers=cores), axis=1, workers=cores)
        ty = time.perf_counter() - t0
        out["yardstick"] = {"what": f"scipy.fft (pocketfft) complex64, {yb} transforms forward+inverse, workers={cores}",
                            "value": 2 * yb * 5.0 * n * math.log2(n) / ty / 1e9, "unit": "GFLOP/s"}
    except Exception as e:
        log(f"scipy yardstick unavailable: {e}")
    return out


def side_config(args, rank, local_rank, world, dist, log):
    """BASELINE configs 3-5 (reported in profiles/, not the driver's default line): forward transforms through the
    immutable-input entry point (input stays pristine, so magnitudes do not grow step to step)."""
    import numpy as np
    import torch

    import rustfft_amd
    from rustfft_amd.sharding import reduce_max

    n, batch, dt, tdt, esz, name = {"c3": (1200, 65536, np.complex128, torch.complex128, 16, "f64"),
                                    "c4": (1009, 1 << 20, np.complex64, torch.complex64, 8, "f32"),
                                    "c5": (1 << 22, 1024, np.complex64, torch.complex64, 8, "f32")}[args.config]
    planner = rustfft_amd.FftPlanner(dt, device=local_rank)
    fft = planner.plan_fft_forward(n)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + rank)
    x = torch.empty(batch * n, dtype=tdt, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    y = torch.empty_like(x)
    for _ in range(args.warmup):
        fft.process_immutable_with_scratch(x, y)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fft.process_immutable_with_scratch(x, y)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = reduce_max(time.perf_counter() - t0, dist, device="cuda")
    flops = world * batch * 5.0 * n * math.log2(n)
    out = {"metric": f"GFLOP/s (5*N*log2N), batched Complex<{name}> FFT", "value": flops * args.steps / elapsed / 1e9, "unit": "GFLOP/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": name, "data": "synthetic",
           "config": {"workload": f"{args.config}: N={n} Complex<{name}>, batch={batch} per GPU, forward, out of place (immutable input), HBM-resident",
                      "plan": fft.describe()}}
    if rank == 0:
        kms = fft.profile_kernels(y, reps=args.steps)
        alg = batch * 2 * n * esz
        per_kernel = [{"kernel": nm, "ms": ms, "GBps": alg / (ms * 1e-3) / 1e9 if ms > 0 else None} for nm, ms in zip(fft.kernel_names(), kms)]
        timed = [r for r in per_kernel if r["GBps"]]
        if timed:
            dom = max(timed, key=lambda r: r["ms"])
            out["roofline"] = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
                               "traffic": None, "kernel": dom["kernel"], "algorithmic_bytes_per_launch": alg, "kernels": per_kernel}
            if world == 1 and not args.no_pmc:
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import pmc_traffic

                    for rk, rv in pmc_traffic.collect(["--config", args.config, "--steps", "1", "--warmup", "1"]).items():
                        if pmc_traffic.rocprof_name_matches(dom["kernel"], rk):
                            out["roofline"]["traffic"] = rv["traffic_bytes"]
                except Exception as e:
                    log(f"pmc traffic unavailable: {e}")
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


def side_line(torch, np, log, steps=3):
    """BASELINE configs 3, 4 and the one-GPU shard of config 5 in the SAME JSON line (key "side"), so that the driver's record
    carries them next to config 2: a few forward steps each through the immutable-input entry point, per-kernel HIP-event
    durations, the dominant kernel's fraction of 8 TB/s, and a Parseval check of the last output over the WHOLE batch.
    `value`, `config` and `roofline` of the line stay config 2's."""
    import rustfft_amd

    def energy(t, chunk=1 << 26):  # sum |t|^2 in f64, chunked: no batch-sized temporaries
        v = torch.view_as_real(t).reshape(-1)
        acc = 0.0
        for i in range(0, v.numel(), chunk):
            acc += float((v[i:i + chunk].double() ** 2).sum().item())
        return acc

    # schema 2 (round 5 on): ms_per_step / frac_of_8TBps of a side config are the IN-PLACE figures (Fft::process), the x -> y figures of rounds
    # 3 - 4 live under "immutable_input" -- same-named keys of BENCH_r03 / r04 are 5 - 10 % lower for that reason alone
    res = {"schema": 2}
    for key, n, batch, dt, tdt, esz, name in (("c3", 1200, 65536, np.complex128, torch.complex128, 16, "f64"),
                                             ("c4", 1009, 1 << 20, np.complex64, torch.complex64, 8, "f32"),
                                             # BASELINE.md's C4 row names 1019 as the complementary prime (1018 = 2 x 509: the reference plans it as Bluestein over
                                             # M = 2048, src/plan.rs:636-665; so does this library): the weakest BASELINE-named workload rides in the driver's line
                                             ("c4_bluestein", 1019, 1 << 20, np.complex64, torch.complex64, 8, "f32"),
                                             # round 6: a length with a large prime factor through the LDS stage machine (64 x 37: the reference's
                                             # MixedRadix over Rader, one kernel from a run-time program) -- the family that left whole-length Bluestein
                                             ("tree_2368", 2368, 1 << 19, np.complex64, torch.complex64, 8, "f32"),
                                             ("c5_shard", 1 << 22, 1024, np.complex64, torch.complex64, 8, "f32")):
        try:
            fft = rustfft_amd.FftPlanner(dt).plan_fft_forward(n)
            g = torch.Generator(device="cuda")
            g.manual_seed(0x52555354 + n)
            x = torch.empty(batch * n, dtype=tdt, device="cuda")
            torch.view_as_real(x).uniform_(-1.0, 1.0, generator=g)
            y = torch.empty_like(x)
            for _ in range(3):  # warm-up: the first launches of a kernel in a process run 5 - 30 % slow (rocprof min / max of the same launch)
                fft.process_immutable_with_scratch(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = steps if key == "c5_shard" else 4 * steps  # (30 ms per step at config 5's shard, 0.5 - 4.5 ms at configs 3 and 4)
            e0.record()
            for _ in range(reps):
                fft.process_immutable_with_scratch(x, y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            ex, ey = energy(x), energy(y)
            ms_immutable = ms
            # IN PLACE -- `Fft::process`, the reference's own benchmark mode (benches/bench_rustfft.rs:43-54 times process_with_scratch on one
            # buffer) and config 2's: forward transforms back to back on y, scaled down so that `reps` unnormalised passes stay finite (a pass
            # grows the values by sqrt(N) on average).  Reads and writes hit the same DRAM pages and the footprint halves: a one-kernel plan runs
            # 5 - 10 % faster this way than x -> y (config 4: 4.08 against 3.70 TB/s), which is why both are reported -- `ms_per_step` /
            # `frac_of_8TBps` are the in-place figures, `immutable_input` the x -> y ones earlier rounds quoted.
            y.copy_(x)
            y.mul_(2.0 ** (-100 if esz == 8 else -600))
            for _ in range(2):
                fft.process(y)
            torch.cuda.synchronize()
            y.copy_(x)
            y.mul_(2.0 ** (-100 if esz == 8 else -600))
            reps_ip = min(reps, 12)
            e0.record()
            for _ in range(reps_ip):
                fft.process(y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps_ip
            finite = bool(torch.isfinite(torch.view_as_real(y)).all().item())
            y.copy_(x)
            kms = fft.profile_kernels(y, reps=max(steps, 5))
            alg = batch * 2 * n * esz
            bracketed = None
            if len(kms) == 1:
                # a one-kernel plan: the step IS the launch -- its average duration over the back-to-back launches timed above (HIP events on the
                # launch stream around the launches; this is what rocprofv3's per-kernel average agrees with), ALWAYS that number; the mean of
                # individually bracketed launches (each timed from an idle chip) is reported beside it, not min()'d with it
                bracketed = kms[0]
                kms = [ms]
            per_kernel = [{"kernel": nm, "ms": k, "GBps": alg / (k * 1e-3) / 1e9} for nm, k in zip(fft.kernel_names(), kms) if k > 0]
            if bracketed is not None and per_kernel:
                per_kernel[0]["ms_individually_bracketed"] = bracketed
            dom = max(per_kernel, key=lambda r: r["ms"])
            res[key] = {"workload": f"N={n} Complex<{name}>, batch={batch}, forward, in place (Fft::process), HBM-resident", "plan": fft.describe(), "steps": steps,
                        "ms_per_step": ms, "GFLOPs": batch * 5.0 * n * math.log2(n) / (ms * 1e-3) / 1e9, "dominant_kernel": dom["kernel"],
                        "dominant_GBps": dom["GBps"], "frac_of_8TBps": dom["GBps"] / HBM_PEAK_GBS, "kernels": per_kernel,
                        "transform_frac_of_8TBps": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "in_place_values_finite": finite,
                        "immutable_input": {"ms_per_step": ms_immutable, "transform_GBps": alg / (ms_immutable * 1e-3) / 1e9,
                                            "transform_frac_of_8TBps": alg / (ms_immutable * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "what": "the same plan x -> y through process_immutable_with_scratch, back to back (what rounds 3 - 4 quoted as side.*)"},
                        "check": {"parseval_rel_err": abs(ey / n - ex) / ex, "what": "sum|X|^2 / N vs sum|x|^2 over the whole batch, x re/im ~ U[-1,1)"}}
            if res[key]["check"]["parseval_rel_err"] > (1e-4 if esz == 8 else 1e-10):
                res[key]["check"]["FAILED"] = True
            del x, y, fft
            torch.cuda.empty_cache()
        except Exception as e:  # side numbers never fail the line
            res[key] = {"error": str(e)}
            log(f"side config {key} failed: {e}")
    return res


def via_cabi(args):
    """`--via-cabi`: the multi-GPU path BEHIND THE BOUNDARY.  One process, `--gpus` devices, one mi355fft_multi_plan per
    direction (include/mi355fft.h): every device holds its shard of the rows resident in its HBM (weak scaling: --batch
    transforms per device), a step is one forward and one inverse mi355fft_multi_process_inplace_dev over all shards; the
    region is bracketed by mi355fft_multi_synchronize, so the clock sees the slowest device.  With --one-device every shard
    lives on device 0 (the control flow on a one-GPU box)."""
    import numpy as np
    import torch

    import rustfft_amd

    G, n, per = args.gpus, 1 << args.log2n, args.batch
    devices = [0] * G if args.one_device else list(range(G))
    planner = rustfft_amd.FftPlannerHipMulti(np.complex64, devices=devices)
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    batch = per * G
    shards = []
    for g, d in enumerate(devices):
        rows = fwd.shard_rows(batch, g)[1]
        with torch.cuda.device(d):
            gen = torch.Generator(device=f"cuda:{d}")
            gen.manual_seed(0x52555354 + g)
            t = torch.empty(rows * n, dtype=torch.complex64, device=f"cuda:{d}")
            torch.view_as_real(t).uniform_(0.0, 10.0, generator=gen)
            t.mul_(2.0 ** -100)
            shards.append(t)
    keep = shards[0][:n].clone()
    # the unnormalised pairs stay finite without a renormalisation for 224 / log2 N pairs: warm-up and steps share that budget (at least one of each)
    budget = max(2, 224 // args.log2n)
    warmup = max(1, min(args.warmup, budget - 1))
    steps = max(1, min(args.steps, budget - warmup))
    for _ in range(warmup):
        fwd.process(shards)
        inv.process(shards)
    fwd.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd.process(shards)
        inv.process(shards)
    fwd.synchronize()
    elapsed = time.perf_counter() - t0
    scale = float(n) ** (warmup + steps)
    err = float(((shards[0][:n] / scale - keep).abs().max() / keep.abs().max()).item())
    finite = all(bool(torch.isfinite(torch.view_as_real(t)).all().item()) for t in shards)
    out = {"metric": "GFLOP/s (5*N*log2N), batched Complex<f32> FFT", "value": 2 * batch * 5.0 * n * math.log2(n) * steps / elapsed / 1e9, "unit": "GFLOP/s",
           "n_gpus": G, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"N=2^{args.log2n} Complex<f32>, batch={per} per GPU, forward+inverse per step, in place, HBM-resident shards",
                      "driver": "one process, mi355fft_multi_plan over devices %s (C ABI; no torch.distributed)" % devices, "plan": fwd.describe(), "finite": finite},
           "check": {"roundtrip_rel_max_err_row0": err, "what": "ifft(fft(x)) / N^steps vs x, first row of shard 0"}}
    if err > 1e-4 or not finite:
        out["check"]["FAILED"] = True
    print(json.dumps(out), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start one process per GPU (rank i on GPU i) with the torchrun
    environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and relay rank 0's JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"rank exit codes: {rcs}")


def selftest_spawn():
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    locals_ = [None] * world
    dist.all_gather_object(locals_, local)
    if rank == 0:
        print(json.dumps({"selftest": "spawn", "n_gpus": world, "rank_sum": int(t.item()), "local_ranks": locals_}), flush=True)
    dist.destroy_process_group()


def fill_blocks(torch, buf, batch, n, gen, rows_per_block=64):
    """U[0,10) re/im, generated in row blocks so that the same seeded stream can be regenerated block by block later."""
    rows = buf.view(batch, n)
    for r0 in range(0, batch, rows_per_block):
        torch.view_as_real(rows[r0:r0 + rows_per_block]).uniform_(0.0, 10.0, generator=gen)


def row_energy(torch, buf, batch, n):
    rows = buf.view(batch, n)
    return torch.cat([(torch.view_as_real(rows[r0:r0 + 64]).double() ** 2).sum(dim=(1, 2)) for r0 in range(0, batch, 64)])


def roundtrip_check(torch, fwd, inv, data, n, batch, seed_gen):
    """Untimed full-batch verification on fresh data: y = ifft(fft(x)) / N against x (max abs error, relative L2) and
    Parseval on the forward result, every row of the shard.  The input is regenerated block by block from the same seed
    for the comparison, so no second batch-sized buffer is needed."""
    fill_blocks(torch, data, batch, n, seed_gen())
    e_in = row_energy(torch, data, batch, n)
    fwd.process(data)
    parseval = ((row_energy(torch, data, batch, n) / n - e_in).abs() / e_in).max().item()
    inv.process(data)
    rows = data.view(batch, n)
    ref = torch.empty(64 * n, dtype=data.dtype, device=data.device)
    g = seed_gen()
    max_err, num, den = 0.0, 0.0, 0.0
    for r0 in range(0, batch, 64):
        cnt = min(64, batch - r0)
        blk = ref[: cnt * n].view(cnt, n)
        torch.view_as_real(blk).uniform_(0.0, 10.0, generator=g)
        d = rows[r0:r0 + cnt] * (1.0 / n) - blk
        max_err = max(max_err, d.abs().max().item())
        num += (d.abs().double() ** 2).sum().item()
        den += (blk.abs().double() ** 2).sum().item()
    return {"roundtrip_max_abs_err": max_err, "roundtrip_rel_l2": math.sqrt(num / den), "parseval_max_rel_err": parseval,
            "what": f"untimed ifft(fft(x))/N vs x on all {batch} rows, x re/im ~ U[0,10); Parseval on the forward result"}


def config5_nested(torch, planner, local_rank, rank, world, dist, steps, warmup, np, batch=1024):
    """BASELINE config 5 (N = 2^22, 1024 transforms per GPU = 8192 over 8 GPUs), forward, immutable input: measured after the
    headline config on the same ranks and reported inside the one JSON line (key "config5")."""
    from rustfft_amd.sharding import reduce_max

    n = 1 << 22
    fft = planner.plan_fft_forward(n)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + 500 + rank)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(0.0, 10.0, generator=g)
    y = torch.empty_like(x)
    for _ in range(warmup):
        fft.process_immutable_with_scratch(x, y)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fft.process_immutable_with_scratch(x, y)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = reduce_max(time.perf_counter() - t0, dist, device="cuda")
    # per-rank parity guard: Parseval on every row of the shard, MAX-reduced
    ex = torch.cat([(torch.view_as_real(x.view(batch, n)[r0:r0 + 32]).double() ** 2).sum(dim=(1, 2)) for r0 in range(0, batch, 32)])
    ey = torch.cat([(torch.view_as_real(y.view(batch, n)[r0:r0 + 32]).double() ** 2).sum(dim=(1, 2)) for r0 in range(0, batch, 32)]) / n
    err = reduce_max(((ey - ex).abs() / ex).max().item(), dist, device="cuda")
    kms = fft.profile_kernels(y, reps=2) if rank == 0 else []
    alg = batch * 2 * n * 8
    return {"workload": f"N=2^22 Complex<f32>, batch={batch} per GPU ({batch * world} total), forward, immutable input, HBM-resident",
            "value": world * batch * 5.0 * n * 22 * steps / elapsed / 1e9, "unit": "GFLOP/s", "ms_per_step": elapsed / steps * 1e3,
            "parseval_max_rel_err_over_ranks": err, "plan": fft.describe(),
            "kernels": [{"kernel": nm, "ms": ms, "GBps": alg / (ms * 1e-3) / 1e9} for nm, ms in zip(fft.kernel_names(), kms)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=8)  # (a fused plan reaches its steady rate after ~10 launches of a fresh process: profiles/r4/fused_warmup_*.jsonl; a step is two launches)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="transforms per GPU")
    ap.add_argument("--chunk", type=int, default=-1, help="transforms per workspace chunk (-1 = library default)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c2 (default, the metric's config) N=2^20 f32 x1024 fwd+inv; c3 N=1200 f64 x65536; "
                         "c4 N=1009 f32 x2^20; c5 N=2^22 f32 x1024 per GPU (8192 over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side configs (BASELINE configs 3, 4 and config 5's one-GPU shard) in the JSON line")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the one-GPU control-flow smoke test)")
    ap.add_argument("--one-device", action="store_true", help="smoke test of the N > 1 control flow on a one-GPU box: every rank uses cuda:0")
    ap.add_argument("--selftest-spawn", action="store_true", help="CPU self-test of the --gpus launcher (gloo): no GPU work")
    ap.add_argument("--edges", action="store_true", help="at --gpus > 1 also time the scatter / gather edges (batch originating on rank 0)")
    ap.add_argument("--no-config5", action="store_true", help="at --gpus > 1: skip the nested BASELINE config-5 measurement")
    ap.add_argument("--fused", type=int, default=-1, choices=[-1, 0, 1], help="fused two-pass launch: -1 the planner's measured choice (default), 0 never, 1 whenever compiled")
    ap.add_argument("--via-cabi", action="store_true", help="ONE process drives --gpus devices through the multi-device plan of the C ABI (mi355fft_multi_*): device-resident shards, no torch.distributed")
    args = ap.parse_args()

    if args.via_cabi:
        return via_cabi(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    if args.selftest_spawn:
        return selftest_spawn()

    import numpy as np
    import torch

    import rustfft_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)  # before the process group: RCCL binds the communicator to the current device
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    def log(msg):
        if rank == 0:
            print(msg, file=sys.stderr, flush=True)

    n, batch = 1 << args.log2n, args.batch
    if args.config != "c2":
        return side_config(args, rank, local_rank, world, dist, log)
    planner = rustfft_amd.FftPlanner(np.complex64, device=local_rank)
    fwd, inv = planner.plan_fft_forward(n), planner.plan_fft_inverse(n)
    if args.chunk >= 0:
        fwd.set_chunk_batch(args.chunk)
        inv.set_chunk_batch(args.chunk)
    if args.fused >= 0:
        fwd.set_fused(args.fused)
        inv.set_fused(args.fused)
    log(f"plan: {fwd.describe()}")

    # synthetic data, re/im ~ U[0,10) (tests/accuracy.rs:84-95), pre-scaled by 2^-100 so that the
    # unnormalised forward+inverse pair (x -> N x per step) stays finite in f32 for <= 10 steps
    g = torch.Generator(device="cuda")
    g.manual_seed(0x52555354 + rank)
    data = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(data).uniform_(0.0, 10.0, generator=g)
    scale0 = 2.0 ** -100
    data.mul_(scale0)
    # The transforms are unnormalised: every step multiplies the data by N.  Starting at 2^-100 the magnitudes stay finite
    # in f32 for `span` steps (11 at N = 2^20), which covers the default warmup + steps with NO extra kernel in the timed
    # region.  Longer runs renormalise between steps with the clock stopped (synchronised on both sides, counted in
    # `renorm_pauses`), so the timed region contains the K forward+inverse pairs and nothing else.
    span = max(1, 224 // args.log2n)
    done = 0

    def renorm_if_needed():
        nonlocal done
        if done and done % span == 0:
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(span):  # one in-range factor per step: n ** -span underflows to 0 when cast to f32
                data.mul_(1.0 / n)
            if not bool((data[:n].abs().max() > 0).item()):
                raise RuntimeError("renormalisation zeroed the data")
            torch.cuda.synchronize()
            return time.perf_counter() - t
        return 0.0

    def step():
        nonlocal done
        fwd.process(data)
        inv.process(data)
        done += 1

    for i in range(args.warmup):
        renorm_if_needed()
        step()
    # back to the starting magnitudes before the clock starts: the default K = 8 steps then fit the span without a pause
    for _ in range(((done - 1) % span) + 1 if done else 0):  # the steps since the last renormalisation (it runs in front of a step)
        data.mul_(1.0 / n)
    done = 0
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    paused, pauses = 0.0, 0
    for i in range(args.steps):
        dt = renorm_if_needed()
        paused += dt
        pauses += dt > 0
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0 - paused
    from rustfft_amd.sharding import reduce_max

    elapsed = reduce_max(elapsed, dist, device="cuda")  # MAX over ranks (RCCL all-reduce of one scalar)
    finite = bool(torch.isfinite(torch.view_as_real(data)).all().item())

    def seed_gen():
        gg = torch.Generator(device="cuda")
        gg.manual_seed(0x52555354 + 100 + rank)
        return gg

    check = roundtrip_check(torch, fwd, inv, data, n, batch, seed_gen)  # untimed, every row of this rank's shard
    for key in ("roundtrip_max_abs_err", "roundtrip_rel_l2", "parseval_max_rel_err"):
        check[key] = reduce_max(check[key], dist, device="cuda")  # worst rank

    flops_per_step = world * 2 * batch * 5.0 * n * math.log2(n)
    value = flops_per_step * args.steps / elapsed / 1e9

    out = {
        "metric": "GFLOP/s (5*N*log2N), batched Complex<f32> FFT", "value": value, "unit": "GFLOP/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"N=2^{args.log2n} Complex<f32>, batch={batch} per GPU, forward+inverse per step, in place, HBM-resident",
                   "plan": fwd.describe(), "finite": finite, "renorm_pauses": pauses},
        "check": check,
    }
    if check["roundtrip_rel_l2"] > 5e-6 or check["parseval_max_rel_err"] > 1e-4 or not finite:
        out["check"]["FAILED"] = True
    if world > 1 and args.edges:
        # SURVEY section 8(e) edges, timed apart from the compute path: a batch that lives on rank 0 goes out with grouped
        # send / recv, every rank transforms its rows, the rows come back (64 transforms per rank keep the root's buffer small)
        from rustfft_amd import sharding

        eb = 64 * world
        full = None
        if rank == 0:
            full = torch.empty(eb * n, dtype=torch.complex64, device="cuda")
            fill_blocks(torch, full, eb, n, seed_gen())
        _, t = sharding.process_from_root(fwd, full, n, eb, dist, root=0, device="cuda", dtype=torch.complex64, sync=torch.cuda.synchronize)
        gib = eb * n * 8 / 2**30
        out["edges"] = {"workload": f"{eb} transforms of 2^{args.log2n} on rank 0 ({gib:.1f} GiB) -> {world} ranks -> rank 0",
                        **{k: reduce_max(v, dist, device="cuda") for k, v in t.items()},
                        "scatter_GBps": gib * 2**30 * (world - 1) / world / max(t["scatter_s"], 1e-9) / 1e9}
        del full
    if world > 1 and not args.no_config5:
        del data
        torch.cuda.empty_cache()
        data = None
        out["config5"] = config5_nested(torch, planner, local_rank, rank, world, dist, max(2, args.steps // 2), 1, np, batch=batch)
    if rank == 0:
        # per-kernel durations with HIP events on the launch stream, same buffers, same step count
        torch.cuda.synchronize()
        if data is None:
            data = torch.zeros(batch * n, dtype=torch.complex64, device="cuda")
        ms_f = fwd.profile_kernels(data, reps=args.steps)  # (profiling runs the passes as separate launches, event-bracketed)
        ms_i = inv.profile_kernels(data, reps=args.steps)
        names = fwd.kernel_names()
        alg_bytes = batch * 2 * n * 8  # one compulsory read + one write of the data per launch
        per_kernel = []
        for k, nm in enumerate(names):
            ms = 0.5 * (ms_f[k] + ms_i[k])
            per_kernel.append({"kernel": nm, "ms": ms, "GBps": alg_bytes / (ms * 1e-3) / 1e9})
        dom = max(per_kernel, key=lambda r: r["ms"])
        two_launch_frac = (batch * 2 * n * 8) / (sum(r["ms"] for r in per_kernel) * 1e-3) / 1e9 / HBM_PEAK_GBS
        if fwd.is_fused() and len(names) == 2:
            # The timed region ran ONE launch per direction (both passes fused, the intermediate through a cache-resident ring):
            # its duration from HIP events on the launch stream (torch's current stream is the stream the library launches on).
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fms = []
            for plan in (fwd, inv):
                plan.process(data)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.steps):
                    plan.process(data)
                e1.record()
                torch.cuda.synchronize()
                fms.append(e0.elapsed_time(e1) / args.steps)
            fused_ms = 0.5 * (fms[0] + fms[1])
            status = fwd.fused_status() | inv.fused_status()
            gbps = alg_bytes / (fused_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k2fused[" + " | ".join(names) + "]", "ms": fused_ms, "algorithmic_bytes_per_launch": alg_bytes,
                               "launches_per_transform": 1, "fused_error_word": status,
                               "what": "ONE launch does the whole transform: algorithmic bytes = one read + one write of the batch (SURVEY section 8(d)); the "
                                       "intermediate crosses the CU <-> memory fabric twice more, out of and into a ring that stays in the Infinity Cache.  "
                                       "`traffic` (FETCH_SIZE / WRITE_SIZE) counts the L2 <-> fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM "
                                       "section), i.e. all four crossings: compare it with 2 x algorithmic_bytes_per_launch",
                               "per_pass_equivalent": {"GBps": 2 * gbps, "frac": 2 * gbps / HBM_PEAK_GBS,
                                                       "what": "both passes' algorithmic bytes over the launch: the figure comparable with the per-kernel "
                                                               "fractions of a two-launch plan (earlier rounds' `frac`)"},
                               "two_launch_plan": {"kernels": per_kernel, "dominant_frac": dom["GBps"] / HBM_PEAK_GBS, "transform_algorithmic_frac": two_launch_frac,
                                                   "what": "the same two passes as separate launches through a full-size HBM workspace (event-bracketed)"},
                               "transform_algorithmic_frac": gbps / HBM_PEAK_GBS}
            if status:
                out["check"]["FAILED"] = True
            dom = {"kernel": "k2f_kernel", "GBps": gbps}
        else:
            out["roofline"] = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": None, "kernel": dom["kernel"],
                               "algorithmic_bytes_per_launch": alg_bytes,
                               "kernels": per_kernel,
                               "transform_algorithmic_frac": two_launch_frac}
            if fwd.is_fused():  # a three-pass plan: the timed region ran passes 0 + 1 as one launch; the figures here are the passes alone
                out["roofline"]["note"] = "three-pass plan whose first two passes run fused in the timed region; per-kernel figures: each pass as its own launch"
        try:
            import ctypes

            from rustfft_amd import _native

            g = ctypes.c_double(0.0)
            if _native.load().mi355fft_measure_copy_ceiling(2 << 30, ctypes.byref(g)) != 0:
                raise RuntimeError("mi355fft_measure_copy_ceiling failed")
            out["roofline"]["copy_ceiling"] = {"GBps": g.value, "frac_of_copy": dom["GBps"] / g.value,
                                               "what": "fastest plain device copy of 2 GiB on this box (one float4 per thread, read + write bytes; the guide's 6.29 TB/s)",
                                               "torch_d2d_GBps": copy_ceiling_gbps(torch)}
        except Exception as e:
            log(f"copy ceiling unavailable: {e}")
        if world == 1 and not args.no_pmc:
            # HBM bytes per launch of the dominant kernel from PMC counters (two extra short rocprofv3 runs)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_traffic

                pmc = pmc_traffic.collect(["--steps", "1", "--warmup", "1", "--log2n", str(args.log2n), "--batch", str(batch)])
                for rk, rv in pmc.items():
                    if ("k2f_kernel" in rk) if dom["kernel"] == "k2f_kernel" else pmc_traffic.rocprof_name_matches(dom["kernel"], rk):
                        out["roofline"]["traffic"] = rv["traffic_bytes"]
                        out["roofline"]["traffic_detail"] = {"fetch_bytes_x2_corrected": rv["fetch_bytes"], "write_bytes": rv["write_bytes"],
                                                             "algorithmic_bytes": alg_bytes, "unit": "bytes per launch"}
            except Exception as e:
                log(f"pmc traffic unavailable: {e}")
        if world == 1 and not args.no_side:
            del data
            data = None
            torch.cuda.empty_cache()
            out["side"] = side_line(torch, np, log)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, log)
            except Exception as e:  # the baseline is a reported side number; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
