#!/bin/bash
# Round 3, last GPU seconds: MODE 5 against MODE 1 once more, now with the branch-free slot select in both arms, for the side-by-side
# primes that still run MODE 1 (identical plans are skipped by tools/ab_lengths.py).
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
timeout 30 python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f32 --gib 1 --check > $OUT/rader_mode5_ab2_f32.jsonl 2> $OUT/ab_f32.err
timeout 30 python tools/ab_lengths.py --b libmi355fft_alt.so --set primes --dtype f64 --gib 1 --check > $OUT/rader_mode5_ab2_f64.jsonl 2> $OUT/ab_f64.err
wc -l $OUT/*.jsonl
