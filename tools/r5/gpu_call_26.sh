#!/bin/bash
# round 5, call 26: the whole-row kernels of the lengths with a prime factor 17 .. 31 above the smooth3 limits (kernels_smooth5_*: f32 (4096, 8192],
# f64 (2048, 8192]) against the one-kernel Bluestein that served them (same library, host-planner entry point), 0.5 GiB of rows, results compared;
# then the GPU suite, the smoke run and the bench line on this library
set -u
O=gpurun_out/r5_26; mkdir -p $O
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --sizes-file tools/r5/smooth5_f32_lengths.txt --dtype f32 > $O/ab_smooth5_vs_bluestein_f32.jsonl 2> $O/ab.err
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --sizes-file tools/r5/smooth5_f64_lengths.txt --dtype f64 > $O/ab_smooth5_vs_bluestein_f64.jsonl 2>> $O/ab.err
python - $O <<'PY'
import json,sys,statistics
o=sys.argv[1]
for dt in ("f32","f64"):
    r=[json.loads(l) for l in open(f"{o}/ab_smooth5_vs_bluestein_{dt}.jsonl") if l.startswith("{")]
    if r:
        v=[d["b_over_a"] for d in r]
        print(dt,len(r),"median x",statistics.median(v),"min",min(v),"max",max(v),"TB/s a",statistics.median(d["a_TBps"] for d in r),"b",statistics.median(d["b_TBps"] for d in r),"min b",min(d["b_TBps"] for d in r),"worst rel l2",max(d["rel_l2_b_vs_a"] for d in r), "losers", [d["n"] for d in r if d["b_over_a"]<1.03][:20])
PY
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; tail -1 $O/smoke_final.log
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench.stderr
python -c "
import json
d=json.loads(open('$O/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], {k:(v.get('frac_of_8TBps') if isinstance(v,dict) else v) for k,v in d.get('side',{}).items()})"
