#!/bin/bash
# prime tile heights (k2r) and the Rader-fused general passes (k2gr), Complex<f32>, without the SLP vectoriser (libmi355fft_alt10.so): prime-tile composites and large Rader primes, two runs
set -u
O=gpurun_out/r4_45; mkdir -p $O
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt10.so --all --check --sizes-file tools/r4/prime_tile_rader_large_lengths.txt --dtype f32 --gib 1 > $O/ab_noslp_k2r_k2gr_f32_rep$rep.jsonl 2> $O/err_$rep.txt
done
python - $O <<'PY'
import json,sys,statistics as st,re,collections
r=[{},{}]; plan={}
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_noslp_k2r_k2gr_f32_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l); r[rep-1][d["n"]]=d["b_over_a"]; plan[d["n"]]=d["plan_a"]
by=collections.defaultdict(list)
for n in r[0]:
    if n in r[1]:
        k="rader_large" if plan[n].startswith("rader_large") else ("k2r" if "k2r" in plan[n] else re.match(r"[a-z0-9_]+",plan[n]).group(0))
        by[k].append((min(r[0][n],r[1][n]),max(r[0][n],r[1][n])))
for k,v in by.items(): print(k,len(v),"median lo",round(st.median(x[0] for x in v),3),"median hi",round(st.median(x[1] for x in v),3),">=+2% both",sum(1 for x in v if x[0]>=1.02),"<=-2% both",sum(1 for x in v if x[1]<=0.98))
PY
tail -n 2 $O/err_1.txt
