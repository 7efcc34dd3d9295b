#!/bin/bash
# Round 4, call 6: (a) persistent workgroups (tuning 60) and two-columns-per-lane tiles (52 / 53) for the one-workgroup-per-CU
# 2048-row later tile, 2^22; (b) the same at 2^20 for comparison; (c) bit-for-bit repeatability of the fused kernel under load.
set -u
O=gpurun_out/r4_06; mkdir -p $O
run() { name=$1; shift; timeout 150 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-40s pair %.3f ms %s kernels %s GB/s %s rel %.2e" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["kernel_ms_median"], d["kernel_GBps"], d["rel_l2_row0"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_2048_persist --log2n 22 --batch 256 --rounds 4 --instances 2 min min:MI355FFT_VARIANT=60 min:MI355FFT_VARIANT=52 min:MI355FFT_VARIANT=53
run ab_1024_persist --log2n 20 --batch 1024 --rounds 4 --instances 2 min min:MI355FFT_VARIANT=60
for k in 16 19 20; do
  b=$(( (1 << 29) >> k ))
  timeout 120 python tools/r4/fused_stress.py $k $b 40 > $O/stress_2p$k.json 2> $O/stress_2p$k.err; echo "stress 2^$k rc=$?"; cat $O/stress_2p$k.json; tail -2 $O/stress_2p$k.err | cut -c1-200
done
