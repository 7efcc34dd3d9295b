#!/bin/bash
# Round 4, call 21: (a) final library (compressed, f64 fused on): the whole -m gpu suite; (b) fused 2^20 with the second pass as
# two columns per lane (tuning 21) against the shipped fused form, three instances.
set -u
O=gpurun_out/r4_21; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-250
run() { name=$1; shift; timeout 200 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-52s pair %.3f ms %s rel %.2e diff %s st %s %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"], d["plan"][:50]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_fused_pair2_2p20 --log2n 20 --batch 1024 --rounds 4 --instances 3 --check-all min:MI355FFT_FUSE=8 min min:MI355FFT_FUSE_RING=111
