#!/bin/bash
# the library with the no-SLP units against the one before (libmi355fft_prev.so): every prime <= 4096 f32, then the GPU suite
set -u
O=gpurun_out/r4_33; mkdir -p $O
timeout 900 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --set primes --dtype f32 --gib 0.5 > $O/ab_final_noslp_primes_f32.jsonl 2> $O/err1.txt
timeout 300 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes 4099,6007,8191,10007,12007,16001 --dtype f32 --gib 1 > $O/ab_final_noslp_big_f32.jsonl 2> $O/err2.txt
python - $O <<'PY'
import json,sys,glob,statistics as st,re
for f in sorted(glob.glob(sys.argv[1]+"/ab_final_*.jsonl")):
    rows=[json.loads(l) for l in open(f) if l.startswith("{")]
    by={}
    for r in rows:
        k=re.match(r"[a-z0-9_]+", r["plan_a"]).group(0)
        by.setdefault(k,[]).append(r)
    print(f.split("/")[-1], len(rows), "rows; worst check", max((r.get("rel_l2_b_vs_a") or 0) for r in rows))
    for k,v in sorted(by.items()): print("   ", k, len(v), "ratio median", round(st.median(r["b_over_a"] for r in v),3), "min", round(min(r["b_over_a"] for r in v),3), "max", round(max(r["b_over_a"] for r in v),3), "TB/s median prev", round(st.median(r["a_TBps"] for r in v),3), "new", round(st.median(r["b_TBps"] for r in v),3))
PY
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
