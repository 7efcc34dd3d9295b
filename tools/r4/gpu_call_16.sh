#!/bin/bash
set -u
for reps in 5 20; do for i in 1 2; do timeout 100 python tools/sweep.py --sizes 1048576,1048576,1048576 --bytes 4 --fused 1 --reps $reps 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('reps $reps run $i', d['batch'], 'ms', d['ms'], 'gflops', d['gflops'], d['kernel_ms'], d['fused'])"; done; done
