#!/bin/bash
# 2^21 as 1024 x 2048 with the later pass on narrow 512-thread tiles inside the fused launch (f32: 8 columns, f64: 4 columns)
set -u
O=gpurun_out/r4_28; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 21 --batch 256 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=1024,FUSED=0 min:MI355FFT_R0=1024,FUSED=1 > $O/ab_fused_rev_2p21.jsonl 2> $O/err_21.txt
run --log2n 21 --batch 128 --dtype f64 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=1024,FUSED=0 min:MI355FFT_R0=1024,FUSED=1 > $O/ab_fused_rev_f64_2p21.jsonl 2> $O/err_21_64.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','fused_status')})
PY
done
tail -n 3 $O/err_*.txt
