#!/bin/bash
# the 256 x 256 fused pair on 16-column tiles of 256 threads (tuning 23) against the shipped 32-column / 512-thread form: 2^16, 2^23, 2^24
set -u
O=gpurun_out/r4_49; mkdir -p $O
run() { timeout 300 python tools/ab.py "$@" ; }
run --log2n 16 --batch 8192 --instances 3 --fwd-only --check-all min min:MI355FFT_FUSE_RING=113 > $O/ab_fused_f16tiles_2p16.jsonl 2> $O/err_16.txt
run --log2n 23 --batch 64 --instances 3 --fwd-only --check-all min min:MI355FFT_FUSE_RING=113 > $O/ab_fused_f16tiles_2p23.jsonl 2> $O/err_23.txt
run --log2n 24 --batch 32 --instances 3 --fwd-only --check-all min min:MI355FFT_FUSE_RING=113 > $O/ab_fused_f16tiles_2p24.jsonl 2> $O/err_24.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','max_abs_diff_vs_arm0','fused_status')})
PY
done
tail -n 2 $O/err_16.txt
