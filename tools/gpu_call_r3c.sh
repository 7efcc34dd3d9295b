#!/bin/bash
# Round 3, late: the side-by-side Rader bodies with the register hand-over (rader_body MODE 5; alternative build RADER_ALT=5 of
# tools/gen_rader_kernels.py) against the shipped MODE 1 bodies, every Rader prime, one process per precision; 1009 at its full batch.
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f32 --gib 1 --check > $OUT/rader_mode5_ab_f32.jsonl 2> $OUT/ab_f32.err
python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f64 --gib 1 --check > $OUT/rader_mode5_ab_f64.jsonl 2> $OUT/ab_f64.err
python tools/ab_lengths.py --b libmi355fft_alt.so --sizes 1009 --dtype f32 --gib 7.875 --check > $OUT/rader_1009_full_f32.jsonl 2>> $OUT/ab_f32.err
wc -l $OUT/*.jsonl; tail -2 $OUT/*.err
