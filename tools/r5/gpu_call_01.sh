#!/bin/bash
# round 5, call 1: skeletons (2048-row tile in column halves; whole-row kernels with LDS-DMA), the round's new -m gpu tests (soak of the 300
# lengths of the two failed round-4 sweeps, fused give-up on the device, fused launches beside a second process), the literal repro of the
# round-4 failure with the new error messages, a short bench
set -u
O=gpurun_out/r5_01; mkdir -p $O
timeout 200 tools/membench/skel3 > $O/skel3.txt 2>&1
timeout 200 tools/membench/skel4 > $O/skel4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "soak or giveup or second_process" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 420 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --all --check --sizes-file tools/r4/general_f32_lengths.txt --dtype f32 --gib 1 > $O/repro_general.jsonl 2> $O/repro_general.err; echo "repro rc $?"; tail -2 $O/repro_general.err; wc -l $O/repro_general.jsonl
timeout 300 python bench.py --steps 10 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
