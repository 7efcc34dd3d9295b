#!/bin/bash
# the general column-tile passes (k2g / k2gr / k2r units), Complex<f32>, without the SLP vectoriser (libmi355fft_alt8.so) against the shipped library: 200 multi-pass lengths, two runs
set -u
O=gpurun_out/r4_41; mkdir -p $O
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt8.so --all --check --sizes-file tools/r4/general_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_noslp_general_f32_rep$rep.jsonl 2> $O/err_$rep.txt
done
python - $O <<'PY'
import json,sys,statistics as st,re,collections
r=[{},{}]; plan={}
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_noslp_general_f32_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l); r[rep-1][d["n"]]=d["b_over_a"]; plan[d["n"]]=d["plan_a"]
both={n:min(r[0][n],r[1][n]) for n in r[0] if n in r[1]}
hi={n:max(r[0][n],r[1][n]) for n in r[0] if n in r[1]}
print(len(both),"lengths; medians",round(st.median(r[0].values()),3),round(st.median(r[1].values()),3),"; >=+2% both:",sum(1 for v in both.values() if v>=1.02),"; <=-2% both:",sum(1 for v in hi.values() if v<=0.98),"; max",round(max(both.values()),3),"min",round(min(hi.values()),3))
by=collections.defaultdict(list)
for n,v in both.items():
    k=re.match(r"[a-z0-9_]+", plan[n]).group(0)+" x%d"%(plan[n].count("->")+1)
    by[k].append(v)
for k,v in sorted(by.items()): print("  ",k,len(v),"median(min of runs)",round(st.median(v),3))
PY
