#!/bin/bash
# Round 4, call 20: Complex<f64> fused launch with ONE 16-byte write-through store per element (tuning-min library).
set -u
O=gpurun_out/r4_20; mkdir -p $O
run() { name=$1; shift; timeout 150 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-30s pair %.3f ms %s rel %.2e diff %s status %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
for k in 16 17 18 20; do
  b=$(( (1 << 29) >> k ))
  run ab_fused_f64_16B_2p$k --dtype f64 --log2n $k --batch $b --rounds 4 --instances 2 --check-all min:FUSED=0 min:FUSED=1
done
