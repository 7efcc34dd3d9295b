#!/bin/bash
# config 4 (Rader 1009): the production rows loop with / without the SLP vectoriser, and without the next-row prefetch at four waves per SIMD (variant 64)
set -u
O=gpurun_out/r4_34; mkdir -p $O
timeout 600 python tools/ab.py --n 1009 --batch 524288 --instances 2 --fwd-only --check-all min libmi355fft_tuning_min_ns.so min:MI355FFT_VARIANT=64 libmi355fft_tuning_min_ns.so:MI355FFT_VARIANT=64 libmi355fft_tuning_min_ns.so:MI355FFT_VARIANT=4 libmi355fft_tuning_min_ns.so:MI355FFT_VARIANT=5 min:MI355FFT_VARIANT=5 > $O/ab_c4_noslp_variants.jsonl 2> $O/err.txt
python - $O/ab_c4_noslp_variants.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','kernel_GBps')})
PY
tail -n 3 $O/err.txt
