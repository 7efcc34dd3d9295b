mkdir -p gpurun_out/r3
timeout 200 python tools/offset_scan.py --log2n 20 --batch 1024 --reps 3 > gpurun_out/r3/offset_scan_2p20.jsonl 2>/dev/null
python3 - <<'PY'
import json
for l in open('gpurun_out/r3/offset_scan_2p20.jsonl'):
    d=json.loads(l)
    if 'offset' in d: print(d['offset'], d['delta_mod_64MiB'], d['pair_ms'], d['GBps_per_pass'])
    else: print(d)
PY
