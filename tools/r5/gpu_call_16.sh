#!/bin/bash
# round 5, call 16: the non-rows-loop Rader bodies (MODE 1 / 5) with their sub-pass factors fetched one exchange ahead (probe library) against the
# shipped library, every prime <= 4096, both precisions, two runs
set -u
O=gpurun_out/r5_16; mkdir -p $O
for rep in 1 2; do
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_rpf.so --all --check --set primes --dtype f32 --gib 1 > $O/ab_rader_pf_f32_rep$rep.jsonl 2> $O/err_f32_$rep.txt
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_rpf.so --all --check --set primes --dtype f64 --gib 1 > $O/ab_rader_pf_f64_rep$rep.jsonl 2> $O/err_f64_$rep.txt
done
python - $O <<'PY'
import json,sys,statistics as st
O=sys.argv[1]
for tag in ("f32","f64"):
    r=[{json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/ab_rader_pf_{tag}_rep{k}.jsonl") if l.startswith("{")} for k in (1,2)]
    both=[n for n in r[0] if n in r[1]]
    for mode in ("m1","m5","m2","m3","m4"):
        S=[n for n in both if r[0][n]["plan_a"].endswith(mode)]
        if S: print(tag,mode,len(S),"median",round(st.median(r[0][n]["b_over_a"] for n in S),3),round(st.median(r[1][n]["b_over_a"] for n in S),3),">=+2% both",sum(1 for n in S if min(r[0][n]["b_over_a"],r[1][n]["b_over_a"])>=1.02),"<=-2% both",sum(1 for n in S if max(r[0][n]["b_over_a"],r[1][n]["b_over_a"])<=0.98),"max rel", max(r[0][n]["rel_l2_b_vs_a"] for n in S))
PY
