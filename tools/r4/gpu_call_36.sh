#!/bin/bash
# every Complex<f32> rows-loop Rader body (MODE 2 / 4) as MODE 3 (no next-row prefetch, 128-VGPR cap: four waves per SIMD) in the no-SLP units
# (RADER_ALT=6 build, libmi355fft_alt6.so) against the shipped choice, one process, every Rader prime
set -u
O=gpurun_out/r4_36; mkdir -p $O
timeout 900 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt6.so --all --check --sizes $(cat tools/r4/rader_primes_f32.txt) --dtype f32 --gib 0.5 > $O/rader_mode3_noslp_ab_f32.jsonl 2> $O/err1.txt
python - $O/rader_mode3_noslp_ab_f32.jsonl <<'PY'
import json,sys,statistics as st
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
ch=[r for r in rows if r["plan_a"]!=r["plan_b"]]
print(len(rows),"primes;",len(ch),"changed bodies; ratio median",round(st.median(r["b_over_a"] for r in ch),3),"min",round(min(r["b_over_a"] for r in ch),3),"max",round(max(r["b_over_a"] for r in ch),3))
win=[(r["n"],r["b_over_a"]) for r in ch if r["b_over_a"]>=1.03]
print(len(win),"win >= 3%:",win)
same=[r["b_over_a"] for r in rows if r["plan_a"]==r["plan_b"]]
print("unchanged bodies noise: median",round(st.median(same),3),"min",round(min(same),3),"max",round(max(same),3))
print("family median TB/s: shipped",st.median(r["a_TBps"] for r in rows),"pick-best",st.median(max(r["a_TBps"],r["b_TBps"]) for r in rows))
PY
tail -n 2 $O/err1.txt
