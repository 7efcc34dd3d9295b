#!/bin/bash
# round 5, call 15: Complex<f64> Bluestein bodies with prefetched / staged sub-pass factors (tuning 60 .. 63); then the whole GPU suite on the
# final library
set -u
O=gpurun_out/r5_15; mkdir -p $O
for n in 1019 1279 1523 1789 2039 2557 3067 3581 4091; do
  b=$(( (1<<26) / n ))
  timeout 120 python tools/ab.py --n $n --dtype f64 --batch $b --rounds 9 --fwd-only min min:MI355FFT_VARIANT=60 min:MI355FFT_VARIANT=61 min:MI355FFT_VARIANT=62 min:MI355FFT_VARIANT=63 min > $O/ab_bs_f64_$n.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_bs_f64_*.jsonl"), key=lambda s:int(s.split("_")[-1].split(".")[0])):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"][-10:], d["pair_ms_median"], d.get("kernel_GBps"), d["plan"][:56], "%.2e"%d["rel_l2_row0"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
