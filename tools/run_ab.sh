mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py -x -q -k "large_primes or host_supplied or prime" 2>&1 | tail -4 | tee gpurun_out/r3/pytest_gpu_rader_large.log
python tools/algo_compare.py --rader --sizes 4481,4621,5281,6301,7681,8191,8641,12289,17011,25601,40961,65537,114689,786433 > gpurun_out/r3/rader_large_ab_f32.jsonl 2>gpurun_out/r3/rader_large_ab_f32.err
python tools/algo_compare.py --rader --dtype f64 --sizes 4481,7681,8641,12289,40961,65537 > gpurun_out/r3/rader_large_ab_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json
for fn in ('gpurun_out/r3/rader_large_ab_f32.jsonl','gpurun_out/r3/rader_large_ab_f64.jsonl'):
    for l in open(fn):
        d=json.loads(l); print(d['n'], d['dtype'], {k:(d[k].get('TBps'), d[k].get('plan','')[:40], '%.1e'%d[k].get('rel_l2',0)) for k in ('auto','bluestein','rader') if k in d})
PY
python bench.py --no-cpu-baseline > gpurun_out/r3/bench_side.json 2> gpurun_out/r3/bench_side.stderr
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_side.json').read())
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d.get('side',{}).items(): print(k, {a:v.get(a) for a in ('ms_per_step','GFLOPs','dominant_kernel','frac_of_8TBps','check','error')})
PY
