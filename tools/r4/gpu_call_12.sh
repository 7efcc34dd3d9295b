#!/bin/bash
# Round 4, call 12: lag / ring variations for the f32 lengths where the default fused launch did not win (tuning-min library).
set -u
O=gpurun_out/r4_12; mkdir -p $O
run() { name=$1; shift; timeout 150 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-62s pair %.3f ms %s rel %.2e diff %s status %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_fused_lag_2p17 --log2n 17 --batch 8192 --rounds 4 --instances 2 --check-all min min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=24,MI355FFT_FUSE_SLOTS=48 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=10,MI355FFT_FUSE_SLOTS=24
run ab_fused_lag_2p18 --log2n 18 --batch 4096 --rounds 4 --instances 2 --check-all min min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=14,MI355FFT_FUSE_SLOTS=28 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=6,MI355FFT_FUSE_SLOTS=14
run ab_fused_lag_2p21 --log2n 21 --batch 512 --rounds 4 --instances 2 --check-all min min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=2,MI355FFT_FUSE_SLOTS=4 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=4,MI355FFT_FUSE_SLOTS=7
run ab_fused_lag_2p20 --log2n 20 --batch 1024 --rounds 4 --instances 2 --check-all min min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=8,MI355FFT_FUSE_SLOTS=16 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=3,MI355FFT_FUSE_SLOTS=8 min:MI355FFT_FUSE=7,MI355FFT_FUSE_LAG=12,MI355FFT_FUSE_SLOTS=24
