#!/bin/bash
set -u
O=gpurun_out/r4_51; mkdir -p $O
L=libmi355fft_tuning_min_ladder.so
timeout 300 python tools/ab.py --n 4093 --batch 32768 --instances 2 --fwd-only --check-all $L:MI355FFT_BLUESTEIN_M=0 $L:MI355FFT_VARIANT=4 $L:MI355FFT_VARIANT=5 > $O/ab_bs8192_variants.jsonl 2> $O/err1.txt
timeout 300 python tools/ab.py --n 3583 --batch 32768 --instances 2 --fwd-only --check-all $L $L:MI355FFT_VARIANT=4 $L:MI355FFT_VARIANT=5 > $O/ab_bs7168_variants.jsonl 2> $O/err2.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','plan','max_abs_diff_vs_arm0','kernel_GBps')})
PY
done
