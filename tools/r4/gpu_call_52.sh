#!/bin/bash
set -u
O=gpurun_out/r4_52; mkdir -p $O
timeout 300 python tools/ab.py --n 7001 --batch 16384 --instances 2 --fwd-only --check-all min min:MI355FFT_VARIANT=4 min:MI355FFT_VARIANT=5 > $O/ab_bs14336_variants.jsonl 2> $O/err1.txt
timeout 300 python tools/ab.py --n 4093 --batch 32768 --instances 2 --fwd-only --check-all min:MI355FFT_VARIANT=6 min > $O/ab_bs8192_shipped.jsonl 2> $O/err2.txt
for f in $O/*.jsonl; do echo "== $f"; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','plan','max_abs_diff_vs_arm0','kernel_GBps','rel_l2_row0')})
PY
done
tail -n 2 $O/err1.txt
