#!/bin/bash
# Round 4, call 18: the shipped library after the planner changes (2^17 reversed + fused, pair tile for the 2048-row later pass,
# waste-aware splits): the whole -m gpu suite, bench.py without PMC / CPU baseline, the power-of-two sweep.
set -u
O=gpurun_out/r4_18; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-250
timeout 400 python bench.py --no-pmc --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_18/bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "check", {k:v for k,v in d["check"].items() if k!="what"})
r=d["roofline"]; print("fused ms", r.get("ms"), "frac", r["frac"], "per_pass", r.get("per_pass_equivalent",{}).get("frac"), "two_launch", r.get("two_launch_plan",{}).get("dominant_frac"), "copy", r.get("copy_ceiling",{}).get("GBps"))
for k,v in d.get("side",{}).items(): print(k, v.get("GFLOPs"), v.get("frac_of_8TBps"), v.get("dominant_kernel"))
PY
timeout 600 python tools/sweep.py --bytes 4 --check > $O/sweep_pow2_f32_4GiB.jsonl 2> $O/sweep.err; echo "sweep rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/r4_18/sweep_pow2_f32_4GiB.jsonl"):
    d=json.loads(l); print(d["log2n"], d["gflops"], "ms", d["ms"], d["kernel_GBps"], d.get("fused"), d.get("fused_per_pass_equivalent_GBps"), "%.1e" % d["rel_l2"], d["plan"][:70])
PY
