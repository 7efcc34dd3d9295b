#!/usr/bin/env python3
"""Issue/stall breakdown of the bench's kernels from rocprofv3 SQ counters (two --pmc passes, 8 SQ slots each;
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: python tools/pmc_sq.py [bench.py args...]  |  python tools/pmc_sq.py --sweep [tools/sweep.py args...]
Prints one JSON object per kernel: counters averaged per dispatch + the derived fractions of SQ_WAVE_CYCLES."""
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
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
     "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
     "SQ_INSTS_SALU", "SQ_WAVES"],
]


def main():
    env = dict(os.environ, TMPDIR="/tmp")
    res = collections.defaultdict(dict)
    for counters in PASSES:
        d = tempfile.mkdtemp(prefix="mi355sq_", dir="/tmp")
        if len(sys.argv) > 1 and sys.argv[1] == "--sweep":  # python tools/pmc_sq.py --sweep --dtype f32 --sizes 1019
            target = [sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "--reps", "2"] + sys.argv[2:]
        else:
            target = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pmc", "--no-side"] + sys.argv[1:]
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--"] + target
        r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "mi355::" in row.get("Kernel_Name", ""):
                    acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
        shutil.rmtree(d, ignore_errors=True)
    for k, v in res.items():
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        out = {"kernel": k[:150], **{c: round(x, 1) for c, x in v.items()}}
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in v:
                out["frac_" + c] = round(v[c] / wc, 3)
        if v.get("SQ_LDS_IDX_ACTIVE"):
            # array cycles beyond the conflict-free 2 per access over all array cycles: 1 - 1/k for a k-way conflict; counts write
            # conflicts that cost no time and the 16-lane groups of ds_read2_b64 (profiles/r4/README.md) -- occupancy, not time
            out["lds_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 3)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
