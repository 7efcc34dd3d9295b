#!/bin/bash
# Round 3, last GPU seconds: config 4's prime (1009 f32, 2^20 rows) -- the shipped rows loop against the rows loop with the register
# hand-over (rader_rows_body HO, MODE 6: tuning variants 6 / 61 / 62 / 63 of the tuning-min build), one interleaved process.
set -u
mkdir -p gpurun_out/r3g
timeout 40 python tools/ab.py --n 1009 --batch 1048576 --rounds 5 min min:MI355FFT_VARIANT=6 min:MI355FFT_VARIANT=61 min:MI355FFT_VARIANT=62 min:MI355FFT_VARIANT=63 > gpurun_out/r3g/ab_rader1009_mode6.jsonl 2> gpurun_out/r3g/ab.err
cut -c1-330 gpurun_out/r3g/ab_rader1009_mode6.jsonl; tail -3 gpurun_out/r3g/ab.err
