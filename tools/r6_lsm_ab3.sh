#!/bin/bash
# round 6: the LDS stage machine -- (1) AUTO against forced Bluestein, (2) plain item order against the bank-aware order, (3) 142 VGPRs against a
# 128-VGPR (four waves per SIMD) build of the Complex<f32> kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
TAG=${1:-x}
S=${2:-74,148,296,592,629,703,1110,1369,1517,2368,3034,3774,4070,4218,5661,1283,3067}
run() { timeout 300 python tools/ab_lengths.py "$@"; }
run --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --dtype f32 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f32_$TAG.jsonl 2> gpurun_out/r6/lsm_ab.err
run --a libmi355fft_alt.so --b libmi355fft.so --check --all --gib 0.25 --dtype f32 --sizes $S > gpurun_out/r6/lsm_order_ab_f32_$TAG.jsonl 2>> gpurun_out/r6/lsm_ab.err
run --a libmi355fft.so --b libmi355fft_w4.so --check --all --gib 0.25 --dtype f32 --sizes $S > gpurun_out/r6/lsm_w4_ab_f32_$TAG.jsonl 2>> gpurun_out/r6/lsm_ab.err
run --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --dtype f64 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f64_$TAG.jsonl 2>> gpurun_out/r6/lsm_ab.err
run --a libmi355fft_alt.so --b libmi355fft.so --check --all --gib 0.25 --dtype f64 --sizes $S > gpurun_out/r6/lsm_order_ab_f64_$TAG.jsonl 2>> gpurun_out/r6/lsm_ab.err
for f in lsm_vs_bluestein_f32 lsm_order_ab_f32 lsm_w4_ab_f32 lsm_vs_bluestein_f64 lsm_order_ab_f64; do echo $f; python -c "
import sys, json
for l in open('gpurun_out/r6/${f}_$TAG.jsonl'):
    d = json.loads(l); print(d['n'], d['a_TBps'], d['b_TBps'], d['b_over_a'], '%.1e' % d['rel_l2_b_vs_a'], d['plan_b'][:100])
"; done
tail -n 3 gpurun_out/r6/lsm_ab.err
