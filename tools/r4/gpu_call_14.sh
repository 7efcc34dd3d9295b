#!/bin/bash
set -u
O=gpurun_out/r4_14; mkdir -p $O
echo "--- sweep 2^20: 4 GiB and 8 GiB, fused -1 / 0 / 1"
for b in 4 8; do for f in 0 1; do timeout 100 python tools/sweep.py --sizes 1048576 --bytes $b --fused $f --reps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('bytes $b fused $f', d['batch'], 'ms', d['ms'], 'gflops', d['gflops'], d['kernel_ms'], d['fused'])"; done; done
echo "--- ab at batch 512 and 1024, forward+inverse"
for b in 512 1024; do timeout 100 python tools/ab.py --log2n 20 --batch $b --rounds 4 --instances 2 default:FUSED=0 default:FUSED=1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('batch $b', d['arm'], d['pair_ms_median'], d['instance_medians_ms'])"; done
