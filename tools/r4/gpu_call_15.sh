#!/bin/bash
set -u
for b in 512 1024; do for mode in "" "--fwd-only"; do timeout 100 python tools/ab.py --log2n 20 --batch $b --rounds 4 --instances 2 $mode default:FUSED=0 default:FUSED=1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('batch $b $mode', d['arm'], d['pair_ms_median'], d['instance_medians_ms'])"; done; done
