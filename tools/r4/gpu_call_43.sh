#!/bin/bash
# the shipped library with the 89 general tile heights moved into no-SLP units against the one before: whole plans over the 200 sampled lengths (2 runs), per-kernel sums, GPU suite
set -u
O=gpurun_out/r4_43; mkdir -p $O
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes-file tools/r4/general_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_final_noslp_general_f32_rep$rep.jsonl 2> $O/err_$rep.txt
done
timeout 900 python tools/r4/k2g_kernel_ab.py libmi355fft_prev.so libmi355fft.so 700 1 > $O/k2g_kernel_ab_final.jsonl 2> $O/err3.txt
python - $O <<'PY'
import json,sys,statistics as st
r=[{},{}]
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_final_noslp_general_f32_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l)
            if "k2g" in d["plan_a"]: r[rep-1][d["n"]]=d["b_over_a"]
both={n:min(r[0][n],r[1][n]) for n in r[0] if n in r[1]}; hi={n:max(r[0][n],r[1][n]) for n in r[0] if n in r[1]}
print(len(both),"general plans: medians",round(st.median(r[0].values()),3),round(st.median(r[1].values()),3),">=+2% both",sum(1 for v in both.values() if v>=1.02),"<=-2% both",sum(1 for v in hi.values() if v<=0.98))
a=b=0.0
for l in open(f"{sys.argv[1]}/k2g_kernel_ab_final.jsonl"):
    d=json.loads(l); a+=d["ms_a"]; b+=d["ms_b"]
print("per-kernel sums: prev",round(a,2),"ms, new",round(b,2),"ms, ratio",round(a/b,3))
PY
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
