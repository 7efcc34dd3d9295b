#!/bin/bash
# after the planner rules of call 25: f32 2^18 default (256 x 1024 fused) against the old plan, f64 2^19 fused through a 16-column split later tile,
# f64 2^24 and f32 2^24 / 2^23 three-pass defaults, f64 2^15 default; every arm checked over the whole batch against arm 0
set -u
O=gpurun_out/r4_26; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 18 --batch 2048 --instances 3 --fwd-only --check-all min:MI355FFT_R0=512,FUSED=0 min > $O/ab_default_2p18.jsonl 2> $O/err_18.txt
run --log2n 19 --batch 512 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused_f64_2p19.jsonl 2> $O/err_19_64.txt
run --log2n 24 --batch 16 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min > $O/ab_fused3_f64_2p24.jsonl 2> $O/err_24_64.txt
run --log2n 23 --batch 64 --instances 3 --fwd-only --check-all min:FUSED=0 min > $O/ab_default_2p23.jsonl 2> $O/err_23.txt
run --log2n 22 --batch 64 --dtype f64 --instances 2 --fwd-only --check-all min:FUSED=0 min > $O/ab_default_f64_2p22.jsonl 2> $O/err_22_64.txt
run --log2n 15 --batch 8192 --dtype f64 --instances 3 --check-all min:FUSED=0 min > $O/ab_default_f64_2p15_pairs.jsonl 2> $O/err_15_64.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','pair_ms_min','instance_medians_ms','plan','max_abs_diff_vs_arm0','rel_l2_row0','fused_status')})
PY
done
tail -n 3 $O/err_*.txt
