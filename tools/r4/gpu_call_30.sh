#!/bin/bash
# other splits of the fused two-pass lengths 2^16 ... 2^20 (Complex<f32>): arm 0 is the shipped default
set -u
O=gpurun_out/r4_30; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 16 --batch 8192 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=128,FUSED=1 min:MI355FFT_R0=512,FUSED=1 > $O/ab_fused_splits_2p16.jsonl 2> $O/err_16.txt
run --log2n 17 --batch 4096 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=128,FUSED=1 min:MI355FFT_R0=1024,FUSED=1 > $O/ab_fused_splits_2p17.jsonl 2> $O/err_17.txt
run --log2n 18 --batch 2048 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=128,FUSED=1 > $O/ab_fused_splits_2p18.jsonl 2> $O/err_18.txt
run --log2n 19 --batch 1024 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=256,FUSED=1 > $O/ab_fused_splits_2p19.jsonl 2> $O/err_19.txt
run --log2n 20 --batch 512 --instances 3 --fwd-only --check-all min min:MI355FFT_R0=512,FUSED=1 > $O/ab_fused_splits_2p20.jsonl 2> $O/err_20.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','fused_status')})
PY
done
tail -n 2 $O/err_*.txt | grep -v amdgpu.ids
