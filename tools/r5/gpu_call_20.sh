#!/bin/bash
# round 5, call 20: the whole-row power-of-two kernels with their non-staged sub-pass factors fetched one exchange ahead (tuning 53)
set -u
O=gpurun_out/r5_20; mkdir -p $O
for k in 10 11 12 13 14 15; do
  b=$(( (1<<29) >> k ))
  timeout 120 python tools/ab.py --log2n $k --batch $b --rounds 9 min min:MI355FFT_VARIANT=53 min min:MI355FFT_VARIANT=53 > $O/ab_k1_pf_2p$k.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_k1_pf_2p*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"][-10:], d["pair_ms_median"], d.get("kernel_GBps"), d["plan"][:50])
PY
