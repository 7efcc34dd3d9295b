mkdir -p gpurun_out/r3
O=gpurun_out/r3/ab_ws_placement.jsonl
: > $O
timeout 300 python tools/ab.py --rounds 5 --log2n 20 --batch 1024 default default default default default default 2>/dev/null | grep '^{' | cut -c1-1000 >> $O
timeout 300 python tools/ab.py --rounds 5 --log2n 22 --batch 256 default default default default 2>/dev/null | grep '^{' | cut -c1-1000 >> $O
python3 - <<'PY'
import json
for l in open('gpurun_out/r3/ab_ws_placement.jsonl'):
    d=json.loads(l); print(d['n'], d['arm'][-26:], d['pair_ms_median'], d['kernel_GBps'])
PY
python bench.py --no-cpu-baseline --no-side > gpurun_out/r3/bench_ws_placement.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/r3/bench_ws_placement.json').read()); print(d['value'], d['roofline']['frac'], [k['GBps'] for k in d['roofline']['kernels']])"
