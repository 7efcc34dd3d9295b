#!/bin/bash
# round 6: calibration of the planner's choice -- the reference's tree (the LDS stage machine, whatever its program costs) against forced Bluestein
# on a random sample of lengths with a prime factor above 31 (260 up to 4096, 140 in (4096, 16384]), both precisions
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
for p in f32 f64; do
timeout 1200 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --b-algo tree --check --all --gib 0.25 --dtype $p --sizes-file tools/r6_lsm_calib_sizes.txt > gpurun_out/r6/lsm_calib_$p.jsonl 2> gpurun_out/r6/lsm_calib_$p.err
tail -n 2 gpurun_out/r6/lsm_calib_$p.err; wc -l gpurun_out/r6/lsm_calib_$p.jsonl
done
