#!/bin/bash
# round 5, call 8: non-temporal loads (snt16) / loads + stores (snt48) in EVERY compiled whole-row schedule against the shipped library, both
# precisions, one interleaved process per run
set -u
O=gpurun_out/r5_09; mkdir -p $O
for v in 16 48; do for t in f32 f64; do
timeout 600 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_snt$v.so --all --sizes-file tools/r5/compiled_whole_row_gt512_$t.txt --dtype $t --gib 0.5 > $O/ab_smooth_nt${v}_$t.jsonl 2> $O/err_${v}_$t.txt; echo rc $?
done; done
python - $O <<'PY'
import json,sys,statistics as st
O=sys.argv[1]
for v in (16,48):
    for t in ("f32","f64"):
        r=[json.loads(l) for l in open(f"{O}/ab_smooth_nt{v}_{t}.jsonl") if l.startswith("{")]
        k=[d for d in r if d["plan_a"].startswith("k1<")]
        x=[d["b_over_a"] for d in k]
        print(v,t,len(r),len(k),"median",st.median(x),">=+3%",sum(1 for q in x if q>=1.03),"<=-3%",sum(1 for q in x if q<=0.97),"max",max(x),"min",min(x))
PY
