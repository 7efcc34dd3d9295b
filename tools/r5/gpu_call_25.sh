#!/bin/bash
# round 5, call 25: the library with 89 / 77 more primes on compiled Rader bodies: GPU suite, bench line, prime sweeps, smoke
set -u
O=gpurun_out/r5_25; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log; grep "31-smooth Rader" $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; tail -1 $O/smoke_final.log
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench.stderr
timeout 400 python tools/prime_sweep.py > $O/primes_le_4096_f32.json 2>/dev/null
timeout 400 python tools/prime_sweep.py --dtype f64 > $O/primes_le_4096_f64.json 2>/dev/null
python - $O <<'PY'
import json,sys
o=sys.argv[1]
d=json.loads(open(o+"/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], {k:(v.get("frac_of_8TBps") if isinstance(v,dict) else v) for k,v in d.get("side",{}).items()})
for dt in ("f32","f64"):
    try:
        j=json.loads(open(f"{o}/primes_le_4096_{dt}.json").read().strip().splitlines()[-1]); print(dt, j["summary"])
    except Exception as e: print(dt, "parse", e)

PY
