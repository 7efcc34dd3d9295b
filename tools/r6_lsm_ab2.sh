#!/bin/bash
# round 6: (1) AUTO against forced Bluestein, (2) AUTO against the build that prefers the single-stage prime radices (libmi355fft_alt.so), f32 + f64
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
TAG=${1:-x}
S=${2:-74,148,296,592,629,703,1110,1369,1517,2368,3034,3599,3774,2183,1739,4070,4218,5661,7992}
for p in f32 f64; do
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --dtype $p --sizes $S > gpurun_out/r6/lsm_vs_bluestein_${p}_$TAG.jsonl 2> gpurun_out/r6/lsm_ab_$p.err
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_alt.so --check --all --gib 0.25 --dtype $p --sizes $S > gpurun_out/r6/lsm_auto_vs_primes_${p}_$TAG.jsonl 2>> gpurun_out/r6/lsm_ab_$p.err
done
for f in lsm_vs_bluestein_f32 lsm_auto_vs_primes_f32 lsm_vs_bluestein_f64 lsm_auto_vs_primes_f64; do echo $f; python -c "
import sys, json
for l in open('gpurun_out/r6/${f}_$TAG.jsonl'):
    d = json.loads(l); print(d['n'], d['a_TBps'], d['b_TBps'], d['b_over_a'], '%.1e' % d['rel_l2_b_vs_a'], d['plan_b'][:100])
"; done
tail -n 3 gpurun_out/r6/lsm_ab_f32.err gpurun_out/r6/lsm_ab_f64.err
