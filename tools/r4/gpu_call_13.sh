#!/bin/bash
# Round 4, call 13 (shipped library, new default lag / ring): fused vs two launches at every two-pass length, three plan
# instances per arm; latencies; the power-of-two sweep at 4 GiB of rows.
set -u
O=gpurun_out/r4_13; mkdir -p $O
run() { name=$1; shift; timeout 200 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-30s pair %.3f ms %s rel %.2e diff %s status %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["rel_l2_row0"], d["max_abs_diff_vs_arm0"], d["fused_status"]))
PY
tail -2 $O/$name.err | cut -c1-200; }
for k in 16 17 18 19 20 21 22; do
  b=$(( (1 << 30) >> k ))
  run ab_fused_final_2p$k --log2n $k --batch $b --rounds 4 --instances 3 --check-all default:FUSED=0 default:FUSED=1
done
timeout 300 python tools/latency_probe.py > $O/latency_default.json 2> $O/latency.err; echo "latency rc=$?"; cut -c1-700 $O/latency_default.json
timeout 600 python tools/sweep.py --bytes 4 --check > $O/sweep_pow2_f32_4GiB.jsonl 2> $O/sweep.err; echo "sweep rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/r4_13/sweep_pow2_f32_4GiB.jsonl"):
    d=json.loads(l); print(d["log2n"], d["gflops"], d["alg_GBps"], d["kernel_GBps"], d.get("fused"), d.get("fused_per_pass_equivalent_GBps"), "%.1e" % d["rel_l2"])
PY
