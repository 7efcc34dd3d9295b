#!/bin/bash
# Round 3, very last GPU seconds: every compiled Rader prime + the Rader family tests on the final library, then the f32 / f64 prime sweeps.
set -u
OUT=gpurun_out/r3e
mkdir -p $OUT
timeout 40 python -m pytest tests/test_gpu_parity.py -q -x -k "every_compiled_rader_prime or test_rader" 2>&1 | tail -4 > $OUT/pytest_rader.log
cat $OUT/pytest_rader.log
timeout 20 python tools/prime_sweep.py > $OUT/primes_le_4096_f32.json 2>/dev/null
timeout 20 python tools/prime_sweep.py --dtype f64 > $OUT/primes_le_4096_f64.json 2>/dev/null
python -c "
import json
for t in ('f32','f64'):
    try: print(t, json.load(open('$OUT/primes_le_4096_%s.json'%t))['summary']['rader'])
    except Exception as e: print(t, 'missing', e)
"
