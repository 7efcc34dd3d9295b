#!/bin/bash
# the whole library compiled without the SLP vectoriser (libmi355fft_noslp.so) against the shipped one, every family, one process per set
set -u
O=gpurun_out/r4_32; mkdir -p $O
A="--a libmi355fft.so --b libmi355fft_noslp.so --all --check"
S3=34,51,68,85,119,136,187,272,323,391,493,544,589,608,667,713,899,961,1088,1292,1472,1615,1856,1984,2057,2108,2176,2431,2584,2976,3179,3553,3944,4048
BIG=4875,5000,6006,8192,10000,10007,12289,16807,20449,25000,32768,41959,44100,45056,65536,100000,131072,1000000,1048576,2097152,4194304,8388608
timeout 900 python tools/ab_lengths.py $A --set primes --dtype f32 --gib 0.5 > $O/ab_noslp_primes_f32.jsonl 2> $O/err1.txt
timeout 900 python tools/ab_lengths.py $A --set primes --dtype f64 --gib 0.5 > $O/ab_noslp_primes_f64.jsonl 2> $O/err2.txt
timeout 900 python tools/ab_lengths.py $A --set smooth13 --dtype f32 --gib 0.5 > $O/ab_noslp_smooth_f32.jsonl 2> $O/err3.txt
timeout 900 python tools/ab_lengths.py $A --set smooth13 --dtype f64 --gib 0.5 > $O/ab_noslp_smooth_f64.jsonl 2> $O/err4.txt
timeout 600 python tools/ab_lengths.py $A --sizes $S3 --dtype f32 --gib 0.5 > $O/ab_noslp_primeradix_f32.jsonl 2> $O/err5.txt
timeout 600 python tools/ab_lengths.py $A --sizes $S3 --dtype f64 --gib 0.5 > $O/ab_noslp_primeradix_f64.jsonl 2> $O/err6.txt
timeout 600 python tools/ab_lengths.py $A --sizes $BIG --dtype f32 --gib 2 > $O/ab_noslp_big_f32.jsonl 2> $O/err7.txt
timeout 600 python tools/ab_lengths.py $A --sizes $BIG --dtype f64 --gib 2 > $O/ab_noslp_big_f64.jsonl 2> $O/err8.txt
python - $O <<'PY'
import json,sys,glob,statistics as st,re
for f in sorted(glob.glob(sys.argv[1]+"/ab_noslp_*.jsonl")):
    rows=[json.loads(l) for l in open(f) if l.startswith("{")]
    by={}
    for r in rows:
        k=re.match(r"[a-z0-9_]+", r["plan_a"].replace("fused{","")).group(0)
        by.setdefault(k,[]).append(r["b_over_a"])
    print(f.split("/")[-1], len(rows), "rows; worst check", max((r.get("rel_l2_b_vs_a") or 0) for r in rows))
    for k,v in sorted(by.items()): print("   ", k, len(v), "median", round(st.median(v),3), "min", round(min(v),3), "max", round(max(v),3))
PY
tail -n 2 $O/err*.txt | grep -v "amdgpu.ids\|^$\|==>"
