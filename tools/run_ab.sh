mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3/pytest_gpu_k2r.log
python tools/algo_compare.py --mixed --sizes 1517,3599,4087,8384,8633,10403,10763,65231,158381 > gpurun_out/r3/prime_tiles_ab_f32.jsonl 2>gpurun_out/r3/prime_tiles_ab_f32.err
python tools/algo_compare.py --mixed --dtype f64 --sizes 1517,8633,10403,65231 > gpurun_out/r3/prime_tiles_ab_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json
for fn in ('gpurun_out/r3/prime_tiles_ab_f32.jsonl','gpurun_out/r3/prime_tiles_ab_f64.jsonl'):
    for l in open(fn):
        d=json.loads(l); print(d['n'], d['dtype'], {k:(d[k].get('TBps'), d[k].get('plan','')[:44], '%.1e'%d[k].get('rel_l2',0)) for k in ('auto','bluestein','mixed') if k in d})
PY
