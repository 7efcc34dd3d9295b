#!/bin/bash
set -u
O=gpurun_out/r5_05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('frac_of_8TBps'), v.get('ms_per_step')) for k,v in d['side'].items()})"
timeout 600 python tools/sweep.py --dtype f32 --min 10 --max 24 --bytes 4 --check > $O/sweep_pow2_f32_4GiB.jsonl 2>/dev/null; python -c "
import json
for l in open('$O/sweep_pow2_f32_4GiB.jsonl'):
    d=json.loads(l); print(d['log2n'], d['ms'], d['alg_GBps'], round(d['alg_GBps']/8000,3), d['kernel_GBps'], d.get('kernel_ms_individually_bracketed'))"
