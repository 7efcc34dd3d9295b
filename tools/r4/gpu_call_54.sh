#!/bin/bash
# the 36 large 7-smooth lengths that run the 32-values-per-thread rule (BIG32_WINS, measured in round 2 with the SLP vectoriser) against the 16-values rule, both in today's units
set -u
O=gpurun_out/r4_54; mkdir -p $O
for rep in 1 2; do
timeout 600 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt11.so --check --sizes-file tools/r4/big32_lengths.txt --dtype f32 --gib 0.5 > $O/ab_big32_rule_rep$rep.jsonl 2> $O/err_$rep.txt
done
python - $O <<'PY'
import json,sys
r=[{},{}]; pb={}
for rep in (1,2):
    for l in open(f"{sys.argv[1]}/ab_big32_rule_rep{rep}.jsonl"):
        if l.startswith("{"):
            d=json.loads(l); r[rep-1][d["n"]]=d["b_over_a"]; pb[d["n"]]=(d["plan_a"],d["plan_b"])
for n in sorted(r[0]):
    if n in r[1]: print(n, round(r[0][n],3), round(r[1][n],3), pb[n][0][:34], "|", pb[n][1][:34])
PY
