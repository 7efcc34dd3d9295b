#!/bin/bash
# Round 4, call 9: (a) what the SQ LDS counters count (ldsbench patterns under the counters); (b) SQ counters of the fused
# config-2 kernel and of config 4's Rader kernel; (c) bench.py --via-cabi (one process, multi-device plan) on this one GPU.
set -u
O=gpurun_out/r4_09; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/r4/lds_counter_probe.py 8 > $O/lds_counter_probe_8waves.jsonl 2> $O/lds_probe.err; echo "lds probe rc=$?"; cut -c1-260 $O/lds_counter_probe_8waves.jsonl | head -40
timeout 300 python tools/pmc_sq.py --steps 2 --warmup 1 > $O/sq_counters_c2_fused.jsonl 2> $O/sq_c2.err; echo "sq c2 rc=$?"; cut -c1-400 $O/sq_counters_c2_fused.jsonl
timeout 300 python tools/pmc_sq.py --config c4 --steps 2 --warmup 1 > $O/sq_counters_c4.jsonl 2> $O/sq_c4.err; echo "sq c4 rc=$?"; cut -c1-400 $O/sq_counters_c4.jsonl
timeout 200 python bench.py --via-cabi --gpus 2 --one-device --batch 256 > $O/bench_via_cabi_2shards_one_gpu.json 2> $O/via.err; echo "via-cabi rc=$?"; cut -c1-500 $O/bench_via_cabi_2shards_one_gpu.json; tail -2 $O/via.err
timeout 200 python bench.py --via-cabi --gpus 1 --batch 1024 > $O/bench_via_cabi_1.json 2>> $O/via.err; echo "via-cabi-1 rc=$?"; cut -c1-300 $O/bench_via_cabi_1.json
