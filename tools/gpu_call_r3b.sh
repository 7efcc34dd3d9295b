#!/bin/bash
# Round 3, late: (1) the new host-planner recipe tests on the device, (2) side-by-side Rader bodies (batched row loads) against the
# shipped rows-loop bodies for every Rader prime, one process per precision (tools/ab_lengths.py; the alternative build is
# RADER_ALT=3 of tools/gen_rader_kernels.py), (3) config 4's prime at its full batch.
set -u
OUT=gpurun_out/r3b
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "recipe or host_planner_options" > $OUT/pytest_recipe.log 2>&1
tail -3 $OUT/pytest_recipe.log
python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f32 --gib 1 --check > $OUT/rader_mode1_back_f32.jsonl 2> $OUT/ab_f32.err
python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f64 --gib 1 --check > $OUT/rader_mode1_back_f64.jsonl 2> $OUT/ab_f64.err
python tools/ab_lengths.py --b libmi355fft_alt.so --sizes 1009 --dtype f32 --gib 7.875 --check > $OUT/rader_1009_full_f32.jsonl 2>> $OUT/ab_f32.err
wc -l $OUT/*.jsonl
