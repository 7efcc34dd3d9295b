#!/bin/bash
# Round 4, call 3: the fused two-pass kernel (launch.h k2f_kernel) against two launches.  FUSE=4: no dependency protocol
# (timing bound, results undefined), 5: counters + fences, 7: + tickets.  Each arm under its own timeout: a hang cannot eat the box.
set -u
O=gpurun_out/r4_03; mkdir -p $O
run() { name=$1; shift; timeout 90 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; cut -c1-330 $O/$name.jsonl; tail -2 $O/$name.err | cut -c1-200; }
run ab_fuse_2p20 --log2n 20 --batch 1024 --rounds 5 --check-all min min:MI355FFT_FUSE=4 min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=5,MI355FFT_FUSE_LAG=8,MI355FFT_FUSE_SLOTS=16 min:MI355FFT_FUSE=5,MI355FFT_FUSE_LAG=3,MI355FFT_FUSE_SLOTS=8
run ab_fuse_2p16 --log2n 16 --batch 16384 --rounds 5 --check-all min min:MI355FFT_FUSE=4 min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7
run ab_fuse_2p18 --log2n 18 --batch 4096 --rounds 5 --check-all min min:MI355FFT_FUSE=4 min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7
run ab_fuse_2p22 --log2n 22 --batch 256 --rounds 5 --check-all min min:MI355FFT_FUSE=4 min:MI355FFT_FUSE=5 min:MI355FFT_FUSE=7
