#!/bin/bash
# Round 4, call 4: what each element of the fused kernel's cross-workgroup protocol costs (2^20 x 1024).
set -u
O=gpurun_out/r4_04; mkdir -p $O
run() { name=$1; shift; timeout 120 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-70s pair %.3f ms  rel %.2e  diff %s  status %s" % (d["arm"], d["pair_ms_median"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_proto_2p20 --log2n 20 --batch 1024 --rounds 5 --check-all min min:MI355FFT_FUSE=4 min:MI355FFT_FUSE=5,MI355FFT_FUSE_PROBE=12 min:MI355FFT_FUSE=5,MI355FFT_FUSE_PROBE=8 min:MI355FFT_FUSE=5,MI355FFT_FUSE_PROBE=4 min:MI355FFT_FUSE=5,MI355FFT_FUSE_RING=1 min:MI355FFT_FUSE=5,MI355FFT_FUSE_RING=1,MI355FFT_FUSE_PROBE=8 min:MI355FFT_FUSE=5,MI355FFT_FUSE_RING=2,MI355FFT_FUSE_PROBE=4 min:MI355FFT_FUSE=5,MI355FFT_FUSE_RING=3 min:MI355FFT_FUSE=7,MI355FFT_FUSE_RING=3
