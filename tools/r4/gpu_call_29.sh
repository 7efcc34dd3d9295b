#!/bin/bash
# Complex<f64> 2^21 (standard order, the 1024-row later tile as 16 columns on 1024 threads) and 2^22 (both 2048-row tiles, 1024 threads) fused
set -u
O=gpurun_out/r4_29; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 21 --batch 128 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused_f64_2p21.jsonl 2> $O/err_21_64.txt
run --log2n 22 --batch 64 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused_f64_2p22.jsonl 2> $O/err_22_64.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','fused_status')})
PY
done
tail -n 3 $O/err_*.txt
