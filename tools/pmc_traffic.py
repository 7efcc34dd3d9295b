#!/usr/bin/env python3
"""HBM traffic of the bench's kernels from rocprofv3 PMC counters, collected the way
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
(they do not fit one pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2), counters in KiB, and on gfx950
FETCH_SIZE reports exactly half the bytes of a wide coalesced stream, so it is doubled.  WRITE_SIZE is taken
as reported (the guide marks it uncalibrated).

Prints {"<kernel name>": {"fetch_bytes": .., "write_bytes": .., "traffic_bytes": .., "dispatches": ..}}.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(bench_args, timeout=600):
    if not shutil.which("rocprofv3"):
        raise RuntimeError("rocprofv3 not found")
    res = collections.defaultdict(dict)
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mi355pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pmc", "--no-side"] + list(bench_args)
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = collections.defaultdict(list)
        for f in files:
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter and "mi355::" in r.get("Kernel_Name", ""):
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k][counter] = sum(v) / len(v) * 1024.0  # KiB -> bytes, mean per dispatch
            res[k]["dispatches"] = len(v)
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch = 2.0 * v["FETCH_SIZE"]  # gfx950 correction (guide §HBM)
            out[k] = {"fetch_bytes": fetch, "write_bytes": v["WRITE_SIZE"], "traffic_bytes": fetch + v["WRITE_SIZE"],
                      "dispatches": v["dispatches"]}
    return out


def rocprof_name_matches(entry_name, rocprof_name):
    """'k2later<1024, 32, 8, 8, 16>xF16' <-> 'void mi355::k2_kernel<float, mi355::Sched<1024, 32, 8, 8, 16>, 16, false, true>(...)'"""
    kind, rest = entry_name.split("<", 1)
    args, tail = rest.rsplit(">xF", 1)
    f = ""
    for ch in tail:  # leading digits = sequences / columns per workgroup; suffixes (v3, m2, abl13) follow
        if not ch.isdigit():
            break
        f += ch
    if f"Sched<{args}>, {f}," not in rocprof_name:
        return False
    if kind == "k2first":
        return "k2_kernel" in rocprof_name and f"Sched<{args}>, {f}, true" in rocprof_name
    if kind == "k2later":
        return "k2_kernel" in rocprof_name and f"Sched<{args}>, {f}, false" in rocprof_name
    return {"k1": "k1_kernel", "rader": "rader_kernel", "bluestein": "bluestein_kernel"}.get(kind, kind) in rocprof_name


if __name__ == "__main__":
    print(json.dumps(collect(sys.argv[1:]), indent=1))
