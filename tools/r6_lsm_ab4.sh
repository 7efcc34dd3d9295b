#!/bin/bash
# round 6: (1) one-wave workgroups with and without the workgroup barrier, (2) the stage machine's kernels with and without the SLP vectoriser
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
S=74,122,246,370,592,678,710,938,1110,1351,1582,1834,2368,2892,3297,4070,5661,8144,9990,12321
run() { timeout 400 python tools/ab_lengths.py "$@"; }
run --a libmi355fft_wb.so --b libmi355fft.so --check --all --gib 0.25 --dtype f32 --sizes $S > gpurun_out/r6/lsm_barrier_ab_f32.jsonl 2> gpurun_out/r6/lsm_ab.err
run --a libmi355fft.so --b libmi355fft_ns.so --check --all --gib 0.25 --dtype f32 --sizes $S > gpurun_out/r6/lsm_noslp_ab_f32.jsonl 2>> gpurun_out/r6/lsm_ab.err
run --a libmi355fft.so --b libmi355fft_ns.so --check --all --gib 0.25 --dtype f64 --sizes $S > gpurun_out/r6/lsm_noslp_ab_f64.jsonl 2>> gpurun_out/r6/lsm_ab.err
for f in lsm_barrier_ab_f32 lsm_noslp_ab_f32 lsm_noslp_ab_f64; do echo $f; python -c "
import sys, json
for l in open('gpurun_out/r6/${f}.jsonl'):
    d = json.loads(l); print(d['n'], d['a_TBps'], d['b_TBps'], d['b_over_a'], '%.1e' % d['rel_l2_b_vs_a'], d['plan_b'][-16:])
"; done
tail -n 3 gpurun_out/r6/lsm_ab.err
