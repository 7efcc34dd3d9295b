#!/bin/bash
# round 5, call 14: Bluestein bodies (tuning-min build, the per-inner-length staging / prefetch choice of call 13 now the default): + the spectrum
# multiplier fetched ahead (70), + the output chirp too (71), the chirp alone (72)
set -u
O=gpurun_out/r5_14; mkdir -p $O
for n in 509 613 761 1019 1279 1523 1789 2039 2557 3067 3581 4091; do
  b=$(( (1<<27) / n ))
  timeout 120 python tools/ab.py --n $n --batch $b --rounds 9 --fwd-only min min:MI355FFT_VARIANT=70 min:MI355FFT_VARIANT=71 min:MI355FFT_VARIANT=72 min > $O/ab_bs_pre_$n.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_bs_pre_*.jsonl"), key=lambda s:int(s.split("_")[-1].split(".")[0])):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"][-10:], d["pair_ms_median"], d.get("kernel_GBps"), d["plan"][:56], "%.2e"%d["rel_l2_row0"])
PY
tail -2 $O/ab.err
