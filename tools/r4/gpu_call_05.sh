#!/bin/bash
# Round 4, call 5: the fused two-pass kernel with the write-through ring (default form) against two launches, three plan
# instances per arm (allocation lottery), every two-pass size; tickets and lag variants at 2^20.
set -u
O=gpurun_out/r4_05; mkdir -p $O
run() { name=$1; shift; timeout 150 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-62s pair %.3f ms %s rel %.2e diff %s status %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_fused_2p20 --log2n 20 --batch 1024 --rounds 4 --instances 3 --check-all min min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=5,MI355FFT_FUSE_LAG=4,MI355FFT_FUSE_SLOTS=8 min:MI355FFT_FUSE=5,MI355FFT_FUSE_LAG=8,MI355FFT_FUSE_SLOTS=16 min:MI355FFT_FUSE=5,MI355FFT_FUSE_RING=103
for k in 16 17 18 19 21 22; do
  b=$(( (1 << 30) >> k ))
  run ab_fused_2p$k --log2n $k --batch $b --rounds 4 --instances 2 --check-all min min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7
done
