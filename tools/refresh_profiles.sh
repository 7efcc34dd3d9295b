#!/bin/bash
# Runs on the GPU box (through gpurun): regenerates every artifact kept under profiles/<round>/ into gpurun_out/profiles/.
# Usage: bash tools/refresh_profiles.sh   (then copy gpurun_out/profiles/* into profiles/rN/)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
# 0. what the box is
( lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; cargo --version 2>&1; rustc --version 2>&1; rocminfo | grep -E "gfx950|Compute Unit" | head -4 ) > $OUT/toolchain_probe.txt 2>&1
# 1. the bench line exactly as the driver runs it (roofline with PMC traffic + cpu_baseline)
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.stderr
# 2. rocprofv3 kernel trace + stats of the same command (PMC and CPU legs off: they would only add their own processes)
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $ROOT/bench.py --no-pmc --no-cpu-baseline --no-side > $OUT/bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
# 3. the other BASELINE configs
for c in c3 c4 c5; do python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null; done  # roofline.traffic from PMC included
for c in c3 c4 c5; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c --output-format csv -- python $ROOT/bench.py --config $c --no-pmc --no-cpu-baseline > /dev/null 2>&1 )
  cp $(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1) $OUT/bench_${c}_kernel_stats.csv 2>/dev/null
done
# 4. sweeps (row 0 of every size checked against numpy complex128)
python tools/sweep.py --dtype f32 --min 1 --max 24 --check > $OUT/sweep_pow2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --min 1 --max 23 --check > $OUT/sweep_pow2_f64.jsonl 2>/dev/null
NP2=3,7,17,77,100,127,251,289,360,719,899,1000,1001,1009,1019,1200,1201,2003,2310,3000,4093,4099,5000,10000,10007,19683,20449,25000,44100,45056,65537,100000,100003,1000000,1000003,1536000,7340032
python tools/sweep.py --dtype f32 --sizes $NP2 --check > $OUT/sweep_np2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --sizes $NP2 --check > $OUT/sweep_np2_f64.jsonl 2>/dev/null
# 5. every prime <= 4096 (the planner's Rader / prime-radix / Bluestein split): AUTO against forced Bluestein, a sample
python tools/algo_compare.py --dtype f32 --sizes 17,31,73,127,257,401,541,761,1009,1019,1201,1453,2003,2311,3001,4001,4051,4093 > $OUT/primes_auto_vs_bluestein_f32.jsonl 2>/dev/null
python tools/prime_sweep.py > $OUT/primes_le_4096_f32.json 2>/dev/null
python tools/prime_sweep.py --dtype f64 > $OUT/primes_le_4096_f64.json 2>/dev/null
python tools/bs_ladder.py > $OUT/bluestein_ladder.jsonl 2>/dev/null
# 6. issue / stall breakdown (SQ counters, two passes each)
( cd /tmp && python $ROOT/tools/pmc_sq.py --steps 2 --warmup 1 > $OUT/sq_counters_c2.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --config c4 --steps 2 --warmup 1 > $OUT/sq_counters_c4.jsonl 2>/dev/null )
# 7. the data-movement ceilings of this box: copy calibration, fabric / Infinity Cache probes, tile skeletons
for t in membench/membench mallbench/mallbench membench/skel; do
  if [ -x tools/$t ]; then tools/$t > $OUT/$(basename $t).txt 2>&1; fi
done
# 8. interleaved A/B of the tile-order knob and the pair-fused variant (tuning build), same box, same buffers
if [ -f rustfft_amd/lib/libmi355fft_tuning_min.so ]; then
  python tools/ab.py --oop --log2n 20 --batch 1024 default min:MI355FFT_DBG=2 min:MI355FFT_XP=1 min:MI355FFT_XP=4 min:MI355FFT_VARIANT=12 2>/dev/null | grep arm > $OUT/ab_tile_order_2p20.jsonl
fi
ls -la $OUT
