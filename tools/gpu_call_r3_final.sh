#!/bin/bash
# Round 3, final validation on the GPU box: the whole -m gpu suite and the prime sweeps (the driver takes the bench line).
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
timeout 60 python tools/prime_sweep.py > $OUT/primes_le_4096_f32.json 2>/dev/null
timeout 60 python tools/prime_sweep.py --dtype f64 > $OUT/primes_le_4096_f64.json 2>/dev/null
python -c "
import json
for t in ('f32','f64'):
    try: print(t, json.load(open('$OUT/primes_le_4096_%s.json'%t))['summary']['rader'])
    except Exception as e: print(t, 'missing', e)
"
