#!/bin/bash
# three-pass plans with passes 0 + 1 fused over units (2^23, 2^24 f32; 2^23 f64), and three cheap pairings: f64 2^15, f32 2^18 as 1024 x 256 / 256 x 1024
set -u
O=gpurun_out/r4_25; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 23 --batch 64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused3_2p23.jsonl 2> $O/err_23.txt
run --log2n 24 --batch 32 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused3_2p24.jsonl 2> $O/err_24.txt
run --log2n 23 --batch 32 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused3_f64_2p23.jsonl 2> $O/err_23_64.txt
run --log2n 15 --batch 8192 --dtype f64 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 > $O/ab_fused_f64_2p15.jsonl 2> $O/err_15_64.txt
run --log2n 18 --batch 2048 --instances 3 --fwd-only --check-all min:FUSED=0 min:FUSED=1 min:MI355FFT_R0=1024,FUSED=0 min:MI355FFT_R0=1024,FUSED=1 min:MI355FFT_R0=256,FUSED=0 min:MI355FFT_R0=256,FUSED=1 > $O/ab_fused_2p18_splits.jsonl 2> $O/err_18.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','pair_ms_min','instance_medians_ms','plan','max_abs_diff_vs_arm0','rel_l2_row0','fused_status')})
PY
done
tail -3 $O/err_*.txt
