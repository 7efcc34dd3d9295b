#!/bin/bash
# round 5, call 22: Rader rows loop with non-temporal row loads for p = 1009 (mode 9) against the previous library
# (rustfft_amd/lib/libmi355fft_prev.so = the build of commit f41c202), the GPU suite, the bench line
set -u
O=gpurun_out/r5_22; mkdir -p $O
timeout 300 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes 1009,1013,997,1021,2017 --dtype f32 --gib 1 > $O/ab_rader_mode9_1GiB.jsonl 2> $O/ab.err
timeout 300 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --all --check --sizes 1009 --dtype f32 --gib 7.89 > $O/ab_rader_mode9_c4batch.jsonl 2>> $O/ab.err
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.stderr
python - $O <<'PY'
import json,sys
o=sys.argv[1]
for f in ("ab_rader_mode9_1GiB.jsonl","ab_rader_mode9_c4batch.jsonl"):
    for l in open(o+"/"+f):
        if l.startswith("{"): print(l.strip()[:300])
d=json.loads(open(o+"/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], {k:(v.get("frac_of_8TBps") if isinstance(v,dict) else v) for k,v in d.get("side",{}).items()})
PY
