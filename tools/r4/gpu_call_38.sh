#!/bin/bash
# config 4's body (rows loop MODE 3) without the SLP vectoriser: compiler-paired scalars against explicit natural-pair arithmetic (-DMI355_PK); and a few Bluestein lengths the same way
set -u
O=gpurun_out/r4_38; mkdir -p $O
A=libmi355fft_tuning_min_ns.so; B=libmi355fft_tuning_min_pk.so
timeout 600 python tools/ab.py --n 1009 --batch 524288 --instances 3 --fwd-only --check-all $A $B $A:MI355FFT_VARIANT=65 $B:MI355FFT_VARIANT=65 > $O/ab_c4_pk.jsonl 2> $O/err.txt
python - $O/ab_c4_pk.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('arm','pair_ms_median','instance_medians_ms','plan','max_abs_diff_vs_arm0','kernel_GBps')})
PY
timeout 600 python tools/ab_lengths.py --a $A --b $B --all --check --sizes 719,1019,1531,2039,2557,3067,3583,4093 --dtype f32 --gib 1 2>/dev/null | cut -c1-120
