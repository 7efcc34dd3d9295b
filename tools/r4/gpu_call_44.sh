#!/bin/bash
set -u
O=gpurun_out/r4_44; mkdir -p $O
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes-file tools/r4/general_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_final2_noslp_general_f32_rep$rep.jsonl 2> $O/err_$rep.txt
done
python - $O <<'PY'
import json,sys,statistics as st
r=[{},{}]; plan={}
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_final2_noslp_general_f32_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l)
            if "k2g" in d["plan_a"]: r[rep-1][d["n"]]=d["b_over_a"]; plan[d["n"]]=d["plan_a"]
both={n:min(r[0][n],r[1][n]) for n in r[0] if n in r[1]}; hi={n:max(r[0][n],r[1][n]) for n in r[0] if n in r[1]}
print(len(both),"general plans: medians",round(st.median(r[0].values()),3),round(st.median(r[1].values()),3),"means",round(st.mean(r[0].values()),3),round(st.mean(r[1].values()),3),">=+2% both",sum(1 for v in both.values() if v>=1.02),"<=-2% both",sum(1 for v in hi.values() if v<=0.98))
print([(n,round(hi[n],3),plan[n]) for n in hi if hi[n]<=0.98])
PY
