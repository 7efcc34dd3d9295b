#!/bin/bash
# the whole-row power-of-two kernels 2^13 .. 2^15 (Complex<f32>): schedule variants with the unit compiled without the SLP vectoriser (the choices date from SLP builds)
set -u
O=gpurun_out/r4_53; mkdir -p $O
L=libmi355fft_tuning_min_ns.so
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 13 --batch 65536 --instances 2 --fwd-only --check-all min $L $L:MI355FFT_VARIANT=4 $L:MI355FFT_VARIANT=5 $L:MI355FFT_VARIANT=7 $L:MI355FFT_VARIANT=8 > $O/ab_k1_noslp_2p13.jsonl 2> $O/err_13.txt
run --log2n 14 --batch 32768 --instances 2 --fwd-only --check-all min $L $L:MI355FFT_VARIANT=5 $L:MI355FFT_VARIANT=6 $L:MI355FFT_VARIANT=7 > $O/ab_k1_noslp_2p14.jsonl 2> $O/err_14.txt
run --log2n 15 --batch 16384 --instances 2 --fwd-only --check-all min $L $L:MI355FFT_VARIANT=5 $L:MI355FFT_VARIANT=6 min:MI355FFT_VARIANT=5 min:MI355FFT_VARIANT=6 > $O/ab_k1_noslp_2p15.jsonl 2> $O/err_15.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','plan','max_abs_diff_vs_arm0')})
PY
done
