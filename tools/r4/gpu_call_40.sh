#!/bin/bash
# the shipped library with the per-length no-SLP units against the one before (libmi355fft_prev.so) on the 489 moved lengths; then the GPU suite
set -u
O=gpurun_out/r4_40; mkdir -p $O
timeout 900 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes-file tools/r4/smooth_noslp_all_lengths.txt --dtype f32 --gib 0.5 > $O/ab_final_noslp_smooth_f32.jsonl 2> $O/err1.txt
python - $O/ab_final_noslp_smooth_f32.jsonl <<'PY'
import json,sys,statistics as st
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
v=[r["b_over_a"] for r in rows]
print(len(rows),"moved lengths: ratio median",round(st.median(v),3),"min",round(min(v),3),"max",round(max(v),3),"below 1.0:",sum(1 for x in v if x<1.0),"worst check",max((r.get("rel_l2_b_vs_a") or 0) for r in rows))
PY
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
