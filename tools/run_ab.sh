mkdir -p gpurun_out/r2
python tools/prime_sweep.py > gpurun_out/r2/primes2.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/primes2.json')); print(json.dumps(d['summary']))
r=[x for x in d['primes'] if x[2]=='rader']; print(sorted(r,key=lambda x:x[1])[:8]); print(sorted(r,key=lambda x:-x[1])[:8])
PY
python bench.py --config c4 --no-pmc --no-cpu-baseline 2>/dev/null | cut -c1-900
