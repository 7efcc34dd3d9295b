#!/bin/bash
set -u
O=gpurun_out/r4_10; mkdir -p $O
export TMPDIR=/tmp
for w in 8 16; do
timeout 200 python tools/r4/lds_counter_probe.py $w > $O/lds_counter_probe_${w}waves.jsonl 2> $O/lds_probe.err; echo "lds probe $w rc=$?"
python - $O/lds_counter_probe_${w}waves.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if "op" in d: print("%-14s %-16s %3d  cyc/instr %5.2f  idx_active/instr %5.2f  conflict/instr %5.2f  ratio %s" % (d["op"], d["pattern"], d["off1_or_stride"], d["cycles_per_wave_instr"], d.get("idx_active_per_instr",-1), d.get("conflict_per_instr",-1), d.get("conflict_over_idx_active")))
    else: print(d)
PY
done
