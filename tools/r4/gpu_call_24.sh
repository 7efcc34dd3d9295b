#!/bin/bash
set -u
O=gpurun_out/r4_24; mkdir -p $O
for k in 16 17 19; do
timeout 200 python tools/r4/fused_warmup_probe.py $k 4 30 > $O/fused_warmup_2p${k}_4GiB.jsonl 2>/dev/null
python - $O/fused_warmup_2p${k}_4GiB.jsonl <<'PY'
import json,sys,statistics
for l in open(sys.argv[1]):
    d=json.loads(l); v=[x for i,x in enumerate(d["per_call_ms"]) if i%6!=0 or i==0]
    v=[x for i,x in enumerate(d["per_call_ms"]) if (i%6)!=0]
    print(d["log2n"], "fused" if d["fused"] else "two  ", "first5", d["per_call_ms"][:5], "steady median", statistics.median(v[12:]))
PY
done
