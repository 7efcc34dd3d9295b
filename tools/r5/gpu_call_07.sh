#!/bin/bash
# round 5, call 7: non-temporal row loads in every compiled Rader rows loop (library with the ten Rader units rebuilt), all primes <= 4096, both
# precisions, two runs; config 3's kernel with non-temporal loads + stores confirmed
set -u
O=gpurun_out/r5_07; mkdir -p $O
for rep in 1 2; do
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_rnt.so --all --check --set primes --dtype f32 --gib 1 > $O/ab_rader_nt_f32_rep$rep.jsonl 2> $O/err_f32_$rep.txt
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_rnt.so --all --check --set primes --dtype f64 --gib 1 > $O/ab_rader_nt_f64_rep$rep.jsonl 2> $O/err_f64_$rep.txt
done
timeout 120 python tools/ab.py --n 1200 --dtype f64 --batch 65536 --rounds 11 --fwd-only min min:MI355FFT_VARIANT=50 min min:MI355FFT_VARIANT=50 min:MI355FFT_VARIANT=51 > $O/ab_c3_nt_confirm.jsonl 2>> $O/ab.err
python - $O <<'PY'
import json,sys,statistics as st
O=sys.argv[1]
for tag in ("f32","f64"):
    r=[{json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/ab_rader_nt_{tag}_rep{k}.jsonl") if l.startswith("{")} for k in (1,2)]
    both=[n for n in r[0] if n in r[1]]
    ra=[n for n in both if r[0][n]["plan_a"].startswith("rader")]
    bs=[n for n in both if not r[0][n]["plan_a"].startswith("rader")]
    for name,S in (("rader",ra),("other (same kernels: noise)",bs)):
        if not S: continue
        print(tag,name,len(S),"median ratio",st.median(r[0][n]["b_over_a"] for n in S),st.median(r[1][n]["b_over_a"] for n in S),">=+2% both",sum(1 for n in S if min(r[0][n]["b_over_a"],r[1][n]["b_over_a"])>=1.02),"<=-2% both",sum(1 for n in S if max(r[0][n]["b_over_a"],r[1][n]["b_over_a"])<=0.98), "median TB/s a",st.median(r[0][n]["a_TBps"] for n in S),"b",st.median(r[0][n]["b_TBps"] for n in S))
for l in open(f"{O}/ab_c3_nt_confirm.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d["arm"], d["pair_ms_median"], d.get("kernel_GBps"), d["rel_l2_row0"])
PY
