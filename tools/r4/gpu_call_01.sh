#!/bin/bash
# Round 4, call 1: the chunk pipeline (plan.cpp execute_pipelined) against the full-size workspace, real kernels, interleaved A/B.
set -u
O=gpurun_out/r4_01; mkdir -p $O
timeout 120 python tools/ab.py --log2n 20 --batch 1024 --rounds 5 --check-all min min:MI355FFT_PIPE=1 min:MI355FFT_PIPE=2 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32,MI355FFT_PIPE_SLOTS=4 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=128 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=16,MI355FFT_PIPE_SLOTS=4 > $O/ab_pipe_2p20.jsonl 2> $O/ab_pipe_2p20.err
cut -c1-260 $O/ab_pipe_2p20.jsonl; tail -3 $O/ab_pipe_2p20.err
timeout 120 python tools/ab.py --log2n 22 --batch 256 --rounds 5 --check-all min min:MI355FFT_PIPE=1 min:MI355FFT_PIPE=2 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32,MI355FFT_PIPE_SLOTS=4 > $O/ab_pipe_2p22.jsonl 2> $O/ab_pipe_2p22.err
cut -c1-260 $O/ab_pipe_2p22.jsonl; tail -3 $O/ab_pipe_2p22.err
timeout 120 python tools/ab.py --log2n 24 --batch 64 --rounds 5 --check-all min min:MI355FFT_PIPE=1 min:MI355FFT_PIPE=2 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32 > $O/ab_pipe_2p24.jsonl 2> $O/ab_pipe_2p24.err
cut -c1-260 $O/ab_pipe_2p24.jsonl; tail -3 $O/ab_pipe_2p24.err
timeout 100 python tools/ab.py --log2n 16 --batch 16384 --rounds 5 --check-all min min:MI355FFT_PIPE=1 min:MI355FFT_PIPE=2 min:MI355FFT_PIPE=2,MI355FFT_PIPE_MIB=32 > $O/ab_pipe_2p16.jsonl 2> $O/ab_pipe_2p16.err
cut -c1-260 $O/ab_pipe_2p16.jsonl; tail -3 $O/ab_pipe_2p16.err
