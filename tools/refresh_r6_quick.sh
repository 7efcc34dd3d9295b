#!/bin/bash
# Round 6, after a planner-only change (which plan a length gets; no kernel changed): the evidence that depends on the planner, on the final library
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_r6
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.stderr
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $ROOT/bench.py --no-pmc --no-cpu-baseline --no-side > $OUT/bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lsm -o lsm --output-format csv -- python $ROOT/tools/sweep.py --dtype f32 --sizes 592,2368,4070,5661 --bytes 2 > $OUT/sweep_lsm_f32_under_rocprof.jsonl 2>/dev/null )
cp $(find /tmp/prof_lsm -name "*kernel_stats.csv" | head -1) $OUT/lsm_kernel_stats.csv 2>/dev/null
NP2=3,7,17,74,77,100,127,251,289,360,592,719,899,1000,1001,1009,1019,1200,1201,1517,2003,2310,2368,3000,4070,4093,4099,4875,5000,5082,5661,6006,8144,8633,9990,10000,10007,10403,12289,12321,16206,19683,20449,25000,41959,44100,45056,65231,65537,100000,100003,158381,216569,417623,1000000,1000003,1536000,7340032
python tools/sweep.py --dtype f32 --sizes $NP2 --check > $OUT/sweep_np2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --sizes $NP2 --check > $OUT/sweep_np2_f64.jsonl 2>/dev/null
python tools/fuzz_gpu.py 400 61 > $OUT/fuzz_gpu_host.log 2>&1
python tools/fuzz_gpu.py 400 62 device > $OUT/fuzz_gpu_device.log 2>&1
FUZZ_ROUND3=1 python tools/fuzz_gpu.py 200 63 device > $OUT/fuzz_gpu_round3_device.log 2>&1
FUZZ_ROUND6=1 python tools/fuzz_gpu.py 400 64 device > $OUT/fuzz_gpu_round6_device.log 2>&1
FUZZ_ROUND6=1 python tools/fuzz_gpu.py 200 65 > $OUT/fuzz_gpu_round6_host.log 2>&1
python tools/r6_full_occupancy_parity.py > $OUT/full_occupancy_parity.jsonl 2> $OUT/full_occupancy_parity.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1
python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
timeout 900 python tools/r6_lsm_all_lengths.py $OUT/lsm_all_lengths.json > $OUT/lsm_all_lengths.log 2>&1
tail -n 3 $OUT/pytest_gpu.log $OUT/fuzz_gpu_*.log $OUT/smoke_final.log $OUT/lsm_all_lengths.log | cut -c1-200
tail -n 1 $OUT/full_occupancy_parity.jsonl
