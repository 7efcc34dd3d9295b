#!/bin/bash
# Round 4, call 7: the production library -- the whole -m gpu suite (round-4 tests included), smoke(), bench.py (no PMC, no CPU baseline).
set -u
O=gpurun_out/r4_07; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log | cut -c1-200
timeout 400 python bench.py --no-pmc --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_07/bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "check", d["check"])
r=d["roofline"]; print({k:r[k] for k in r if k not in ("what","two_launch_plan","per_pass_equivalent","copy_ceiling")})
print("per_pass", r.get("per_pass_equivalent",{}).get("frac"), "two_launch", r.get("two_launch_plan",{}).get("dominant_frac"), "copy", r.get("copy_ceiling"))
for k,v in d.get("side",{}).items(): print(k, v.get("GFLOPs"), v.get("frac_of_8TBps"), v.get("dominant_kernel"), v.get("check"))
PY
tail -3 $O/bench.err | cut -c1-200
