#!/bin/bash
# round 5, call 27: the tier (8192, 16384] of the prime-radix whole-row kernels (kernels_smooth5_*) against the Bluestein plans they replace (every
# fourth length, same library, host-planner entry point), then the parity tests that run every new length, and the smoke run
set -u
O=gpurun_out/r5_27; mkdir -p $O
python - <<'PY'
for dt in ("f32","f64"):
    v=open(f"tools/r5/smooth5b_{dt}_lengths.txt").read().strip().split(",")
    open(f"/tmp/s5b_{dt}.txt","w").write(",".join(v[::4]))
PY
timeout 100 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --sizes-file /tmp/s5b_f32.txt --dtype f32 > $O/ab_smooth5_16384_vs_bluestein_f32.jsonl 2> $O/ab.err
timeout 100 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --sizes-file /tmp/s5b_f64.txt --dtype f64 > $O/ab_smooth5_16384_vs_bluestein_f64.jsonl 2>> $O/ab.err
python - $O <<'PY'
import json,sys,statistics
o=sys.argv[1]
for dt in ("f32","f64"):
    r=[json.loads(l) for l in open(f"{o}/ab_smooth5_16384_vs_bluestein_{dt}.jsonl") if l.startswith("{")]
    if r:
        v=[d["b_over_a"] for d in r]
        print(dt,len(r),"median x",statistics.median(v),"min",min(v),"max",max(v),"TB/s a",statistics.median(d["a_TBps"] for d in r),"b",statistics.median(d["b_TBps"] for d in r),"min b",min(d["b_TBps"] for d in r),"worst rel l2",max(d["rel_l2_b_vs_a"] for d in r), "losers", [d["n"] for d in r if d["b_over_a"]<1.03][:20])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "single_kernel_above_4096 or runtime_scheduled" > $O/pytest_gpu_subset.log 2>&1
tail -2 $O/pytest_gpu_subset.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; tail -1 $O/smoke_final.log
