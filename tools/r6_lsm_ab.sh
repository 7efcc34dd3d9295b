#!/bin/bash
# round 6: AUTO (the LDS stage machine where the planner takes it) against the forced Bluestein plan, one process per precision
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
TAG=${1:-x}
S=${2:-74,148,185,222,296,370,592,629,703,1110,1369,1517,2368,3034,3774,1283,3067,4218,8144}
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f32_$TAG.jsonl 2> gpurun_out/r6/lsm_ab_f32.err
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --dtype f64 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f64_$TAG.jsonl 2> gpurun_out/r6/lsm_ab_f64.err
for p in f32 f64; do echo $p; python -c "
import sys, json
for l in open('gpurun_out/r6/lsm_vs_bluestein_${p}_$TAG.jsonl'):
    d = json.loads(l); print(d['n'], d['a_TBps'], d['b_TBps'], d['b_over_a'], '%.1e' % d['rel_l2_b_vs_a'], d['plan_b'][:100])
"; done
tail -3 gpurun_out/r6/lsm_ab_f32.err gpurun_out/r6/lsm_ab_f64.err
