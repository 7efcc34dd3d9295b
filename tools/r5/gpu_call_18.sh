#!/bin/bash
# round 5, call 18: Bluestein bodies on MORE threads per row (8 values per thread: tuning 81 with the shipped staging, 82 fetch-ahead only, 83 five sub-passes)
set -u
O=gpurun_out/r5_18; mkdir -p $O
for n in 1279 1523 1789 2557 3067 3581 4091; do
  b=$(( (1<<27) / n ))
  timeout 120 python tools/ab.py --n $n --batch $b --rounds 9 --fwd-only min min:MI355FFT_VARIANT=81 min:MI355FFT_VARIANT=82 min:MI355FFT_VARIANT=83 min > $O/ab_bs_threads_$n.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_bs_threads_*.jsonl"), key=lambda s:int(s.split("_")[-1].split(".")[0])):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"][-10:], d["pair_ms_median"], d.get("kernel_GBps"), d["plan"][:60], "%.2e"%d["rel_l2_row0"])
PY
tail -2 $O/ab.err
