#!/usr/bin/env python3
"""What SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE count on gfx950: tools/ldsbench's labelled access patterns (conflict-free,
2-way, 4-way, 8-way strides; the column-tile exchange patterns) run under the two counters, one dispatch per pattern, next to
the cycles per wave-instruction the probe itself measures."""
import csv, glob, json, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "tools", "ldsbench", "ldsbench")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, exe + ".hip"])
waves = sys.argv[1] if len(sys.argv) > 1 else "8"
d = tempfile.mkdtemp(prefix="ldsprobe_", dir="/tmp")
cmd = ["rocprofv3", "--pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", exe, waves, "brief"]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
lines = [l for l in r.stdout.splitlines() if l.startswith("ds_")]
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "probe" in row.get("Kernel_Name", ""):  # (the copies of the address table are dispatches too)
            rows.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
shutil.rmtree(d, ignore_errors=True)
for (did, c), l in zip(sorted(rows.items()), lines):
    parts = l.split()
    out = {"op": parts[0], "pattern": " ".join(parts[1:-2]), "off1_or_stride": int(parts[-2]), "cycles_per_wave_instr": float(parts[-1]), **{k: v for k, v in c.items()}}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out["conflict_over_idx_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    if c.get("SQ_INSTS_LDS"):
        out["idx_active_per_instr"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0) / c["SQ_INSTS_LDS"], 2)
        out["conflict_per_instr"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_INSTS_LDS"], 2)
    print(json.dumps(out))
if len(lines) != len(rows):
    print(json.dumps({"warning": "dispatch / pattern count mismatch", "patterns": len(lines), "dispatches": len(rows)}))
