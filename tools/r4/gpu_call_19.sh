#!/bin/bash
# Round 4, call 19: `python bench.py` exactly as the driver runs it (PMC traffic passes, side configs, CPU baseline), and the
# same command under rocprofv3 --kernel-trace --stats.
set -u
O=$PWD/gpurun_out/r4_19; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_19/bench_default.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"])
r=d["roofline"]; print({k:r[k] for k in ("achieved","frac","traffic","ms","kernel","fused_error_word") if k in r}); print(r.get("traffic_detail")); print(r.get("per_pass_equivalent")); print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
for k,v in d.get("side",{}).items(): print(k, v.get("GFLOPs"), v.get("frac_of_8TBps"), v.get("dominant_kernel"))
PY
tail -3 $O/bench_default.err | cut -c1-200
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --no-side > $O/bench_under_rocprof.json 2>/dev/null )
f=$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-230
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "via_cabi or random_lengths" 2>&1 | tail -3
