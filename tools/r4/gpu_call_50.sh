#!/bin/bash
# Bluestein ladder with the 9 * 2^k and 15 * 2^k inner lengths (tuning-min pair: every prime <= 4096 runs as Bluestein there except 1009), two runs
set -u
O=gpurun_out/r4_50; mkdir -p $O
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft_tuning_min.so --b libmi355fft_tuning_min_ladder.so --check --set primes --dtype f32 --gib 0.5 > $O/ab_ladder915_f32_rep$rep.jsonl 2> $O/err_$rep.txt
done
python - $O <<'PY'
import json,sys,statistics as st,re,collections
r=[{},{}]; pb={}
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_ladder915_f32_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l); r[rep-1][d["n"]]=d["b_over_a"]; pb[d["n"]]=d["plan_b"]
by=collections.defaultdict(list)
for n in r[0]:
    if n in r[1]:
        m=re.match(r"bluestein<(\d+),", pb[n]); by[int(m.group(1)) if m else 0].append((min(r[0][n],r[1][n]),max(r[0][n],r[1][n])))
for M,v in sorted(by.items()): print("new M",M,len(v),"primes: median lo",round(st.median(x[0] for x in v),3),"hi",round(st.median(x[1] for x in v),3),"min",round(min(x[0] for x in v),3))
PY
tail -n 2 $O/err_1.txt
