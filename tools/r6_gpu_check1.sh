#!/bin/bash
# round 6: the new GPU tests (stage machine, give-up visibility, repeatability), the calibration sweep on the final planner, one bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stage_machine or giveup or repeatability" > gpurun_out/r6/pytest_new.log 2>&1
tail -n 6 gpurun_out/r6/pytest_new.log
bash tools/r6_lsm_calib.sh
timeout 600 python bench.py > gpurun_out/r6/bench_mid.json 2> gpurun_out/r6/bench_mid.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/bench_mid.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"])
for k, v in d["side"].items():
    if isinstance(v, dict): print(k, v.get("plan", v)[:80] if isinstance(v.get("plan"), str) else v, v.get("ms_per_step"), v.get("transform_frac_of_8TBps"), v.get("check"))
PY
