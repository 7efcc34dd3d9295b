#!/bin/bash
# the fused kernels compiled without the SLP vectoriser (no scratch anywhere: 2^20 128 VGPRs + 20 B -> 96 VGPRs; the 2048-row first tile 72 B -> 0):
# do 2^22 and 2^21 in the standard order now win?  and the shipped ones: 2^16 .. 2^21
set -u
O=gpurun_out/r4_37; mkdir -p $O
L=libmi355fft_tuning_min_ns.so
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 22 --batch 128 --instances 3 --fwd-only --check-all min min:FUSED=1 $L:FUSED=1 > $O/ab_fused_noslp_2p22.jsonl 2> $O/err_22.txt
run --log2n 21 --batch 256 --instances 3 --fwd-only --check-all min $L $L:MI355FFT_R0=2048,FUSED=1 min:MI355FFT_R0=2048,FUSED=1 > $O/ab_fused_noslp_2p21.jsonl 2> $O/err_21.txt
for k in 16 18 19 20; do
run --log2n $k --batch $((1 << (29 - k))) --instances 3 --fwd-only --check-all min $L > $O/ab_fused_noslp_2p$k.jsonl 2> $O/err_$k.txt
done
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','fused_status')})
PY
done
