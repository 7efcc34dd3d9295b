#!/bin/bash
set -u
O=gpurun_out/r4_42; mkdir -p $O
timeout 900 python tools/r4/k2g_kernel_ab.py libmi355fft.so libmi355fft_alt9.so 700 1 > $O/k2g_kernel_ab_rep1.jsonl 2> $O/err1.txt
timeout 900 python tools/r4/k2g_kernel_ab.py libmi355fft.so libmi355fft_alt9.so 700 1 > $O/k2g_kernel_ab_rep2.jsonl 2> $O/err2.txt
wc -l $O/*.jsonl; tail -n 2 $O/err1.txt
