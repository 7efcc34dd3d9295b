#!/bin/bash
# round 5, call 10: the library with the per-length non-temporal choice against the one before (every compiled whole-row length above 512, both
# precisions, results compared), the whole GPU suite, the bench line
set -u
O=gpurun_out/r5_10; mkdir -p $O
for t in f32 f64; do
timeout 600 python tools/ab_lengths.py --a libmi355fft_prev.so --b libmi355fft.so --check --sizes-file tools/r5/compiled_whole_row_gt512_$t.txt --dtype $t --gib 0.5 > $O/ab_nt_choice_$t.jsonl 2> $O/err_$t.txt; echo rc $?
done
python - $O <<'PY'
import json,sys,statistics as st
O=sys.argv[1]
for t in ("f32","f64"):
    r=[json.loads(l) for l in open(f"{O}/ab_nt_choice_{t}.jsonl") if l.startswith("{")]
    x=[d["b_over_a"] for d in r]
    print(t,len(r),"changed plans; median",st.median(x),">=+2%",sum(1 for q in x if q>=1.02),"<=-2%",sum(1 for q in x if q<=0.98),"max rel diff",max(d["rel_l2_b_vs_a"] for d in r), [(d["n"],d["b_over_a"]) for d in r if d["b_over_a"]<0.97][:15])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac']); [print(k, v.get('frac_of_8TBps'), v.get('transform_frac_of_8TBps'), v.get('ms_per_step'), v.get('immutable_input',{}).get('transform_frac_of_8TBps'), v.get('in_place_values_finite'), v.get('error')) for k,v in d['side'].items()]"
