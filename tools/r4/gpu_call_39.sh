#!/bin/bash
# the compiled single-kernel schedules (13-smooth <= 4096, prime-radix <= 4096, 7-smooth <= 32768), Complex<f32>, without the SLP vectoriser
# (libmi355fft_alt7.so: only those units differ) against the shipped library, every length, one process per family, twice (order / noise check)
set -u
O=gpurun_out/r4_39; mkdir -p $O
for fam in smooth smooth3 smooth2; do
for rep in 1 2; do
timeout 900 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt7.so --all --check --sizes-file tools/r4/${fam}_f32_lengths.txt --dtype f32 --gib 0.5 > $O/ab_noslp_${fam}_f32_rep$rep.jsonl 2> $O/err_${fam}_$rep.txt
done
done
python - $O <<'PY'
import json,sys,glob,statistics as st
for fam in ("smooth","smooth3","smooth2"):
    r1={}; r2={}
    for rep,dst in ((1,r1),(2,r2)):
        for l in open(f"{sys.argv[1]}/ab_noslp_{fam}_f32_rep{rep}.jsonl"):
            if l.startswith("{"):
                d=json.loads(l); dst[d["n"]]=d["b_over_a"]
    both=[(n,min(r1[n],r2[n])) for n in r1 if n in r2]
    win=[n for n,v in both if v>=1.03]
    print(fam, len(both), "lengths; median", round(st.median(r1.values()),3), round(st.median(r2.values()),3), "; >= +3 % in BOTH runs:", len(win), "; max", round(max(v for _,v in both),3), "min of run1", round(min(r1.values()),3))
PY
