#!/usr/bin/env python3
"""Round 6: summary of tools/r6_lsm_calib.sh -- the LDS stage machine (the reference's tree, whatever its program costs) against forced
Bluestein, grouped by length range and number of stages.  python tools/r6_lsm_calib_report.py <dir with lsm_calib_f32.jsonl / _f64.jsonl>"""
import collections
import json
import re
import statistics as st
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6"
for p in ("f32", "f64"):
    rows = [json.loads(l) for l in open(f"{d}/lsm_calib_{p}.jsonl")]
    lsm = [r for r in rows if r["plan_b"].startswith("lsm")]
    print(p, len(rows), "rows,", len(lsm), "through the stage machine; largest relative L2 difference to the Bluestein plan", max(r["rel_l2_b_vs_a"] for r in lsm))
    g = collections.defaultdict(list)
    for r in lsm:
        nt, stg, F = map(int, re.search(r">x(\d+)t(\d+)sF(\d+)", r["plan_b"]).groups())
        rng = "<=4096" if r["n"] <= 4096 else ("<=8192" if r["n"] <= 8192 else ">8192")
        g[(rng, stg)].append((r["b_over_a"], r["b_TBps"]))
    for k in sorted(g):
        v = [x[0] for x in g[k]]
        print(f"  {k[0]:7s} stages {k[1]:2d}: n={len(v):3d} median x{st.median(v):.2f} min x{min(v):.2f} max x{max(v):.2f} wins(>3%) {sum(1 for x in v if x > 1.03):3d}  median TB/s {st.median(x[1] for x in g[k]):.2f}")
