#!/bin/bash
# Runs on the GPU box (through gpurun): regenerates every artifact kept under profiles/<round>/ into gpurun_out/profiles/.
# Usage: bash tools/refresh_profiles.sh   (then copy gpurun_out/profiles/* into profiles/rN/)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the bench line exactly as the driver runs it (roofline with PMC traffic + cpu_baseline)
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.stderr
# 2. rocprofv3 kernel trace + stats of the same command (PMC and CPU legs off: they would only add their own processes)
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $ROOT/bench.py --no-pmc --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
# 3. the other BASELINE configs
for c in c3 c4 c5; do python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null; done  # roofline.traffic from PMC included
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 --output-format csv -- python $ROOT/bench.py --config c4 --no-pmc --no-cpu-baseline > /dev/null 2>&1 )
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $OUT/bench_c4_kernel_stats.csv 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $ROOT/bench.py --config c3 --no-pmc --no-cpu-baseline > /dev/null 2>&1 )
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $OUT/bench_c3_kernel_stats.csv 2>/dev/null
# 4. sweeps
python tools/sweep.py --dtype f32 --min 1 --max 24 > $OUT/sweep_pow2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --min 1 --max 23 > $OUT/sweep_pow2_f64.jsonl 2>/dev/null
NP2=3,7,17,77,100,127,251,360,719,1000,1001,1009,1019,1200,2310,3000,4093,4099,5000,10000,10007,19683,25000,44100,65537,100000,100003,1000000,1000003,1536000,7340032
python tools/sweep.py --dtype f32 --sizes $NP2 > $OUT/sweep_np2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --sizes $NP2 > $OUT/sweep_np2_f64.jsonl 2>/dev/null
# 5. issue / stall breakdown (SQ counters, two passes each)
( cd /tmp && python $ROOT/tools/pmc_sq.py --steps 2 --warmup 1 > $OUT/sq_counters_c2.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --config c3 --steps 2 --warmup 1 > $OUT/sq_counters_c3.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --config c4 --steps 2 --warmup 1 > $OUT/sq_counters_c4.jsonl 2>/dev/null )
# 6. the copy ceiling of this box
if [ ! -x tools/membench/membench ]; then hipcc --offload-arch=gfx950 -O3 tools/membench/membench.hip -o tools/membench/membench 2>/dev/null; fi
if [ -x tools/membench/membench ]; then tools/membench/membench > $OUT/membench.txt 2>&1; fi
ls -la $OUT
