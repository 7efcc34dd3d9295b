#!/bin/bash
# config 4: no-prefetch rows loops (MODE 3, four waves per SIMD) compiled without the SLP vectoriser -- rows per workgroup and schedules
set -u
O=gpurun_out/r4_35; mkdir -p $O
L=libmi355fft_tuning_min_ns.so
timeout 600 python tools/ab.py --n 1009 --batch 524288 --instances 2 --fwd-only --check-all $L:MI355FFT_VARIANT=64 $L:MI355FFT_VARIANT=65 $L:MI355FFT_VARIANT=66 $L:MI355FFT_VARIANT=67 $L:MI355FFT_VARIANT=68 $L:MI355FFT_VARIANT=69 $L:MI355FFT_VARIANT=70 $L:MI355FFT_VARIANT=71 $L > $O/ab_c4_mode3_variants.jsonl 2> $O/err.txt
python - $O/ab_c4_mode3_variants.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','kernel_GBps')})
PY
tail -n 3 $O/err.txt
