#!/bin/bash
# Round-6 refresh of the artifacts kept under profiles/r6/ (runs on the GPU box through gpurun; outputs in gpurun_out/profiles_r6/).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_r6
mkdir -p $OUT
export TMPDIR=/tmp
( lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; cargo --version 2>&1; rustc --version 2>&1; go version 2>&1; rocminfo | grep -E "gfx950|Compute Unit" | head -4; ls -la rustfft_amd/lib/libmi355fft.so ) > $OUT/toolchain_probe.txt 2>&1
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.stderr
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $ROOT/bench.py --no-pmc --no-cpu-baseline --no-side > $OUT/bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
for c in c4 c5; do
  python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c --output-format csv -- python $ROOT/bench.py --config $c --no-pmc --no-cpu-baseline > /dev/null 2>&1 )
  cp $(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1) $OUT/bench_${c}_kernel_stats.csv 2>/dev/null
done
# the stage machine under rocprof: one length per block size, kernel durations for the roofline rows of DESIGN.md
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lsm -o lsm --output-format csv -- python $ROOT/tools/sweep.py --dtype f32 --sizes 592,2368,4070,9990 --bytes 2 > $OUT/sweep_lsm_f32_under_rocprof.jsonl 2>/dev/null )
cp $(find /tmp/prof_lsm -name "*kernel_stats.csv" | head -1) $OUT/lsm_kernel_stats.csv 2>/dev/null
python tools/sweep.py --dtype f32 --min 10 --max 24 --bytes 4 --check > $OUT/sweep_pow2_f32_4GiB.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --min 10 --max 24 --bytes 4 --check > $OUT/sweep_pow2_f64_4GiB.jsonl 2>/dev/null
NP2=3,7,17,74,77,100,127,251,289,360,592,719,899,1000,1001,1009,1019,1200,1201,1517,2003,2310,2368,3000,4070,4093,4099,4875,5000,5082,6006,8144,8633,9990,10000,10007,10403,12289,12321,16206,19683,20449,25000,41959,44100,45056,65231,65537,100000,100003,158381,216569,417623,1000000,1000003,1536000,7340032
python tools/sweep.py --dtype f32 --sizes $NP2 --check > $OUT/sweep_np2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --sizes $NP2 --check > $OUT/sweep_np2_f64.jsonl 2>/dev/null
python tools/prime_sweep.py > $OUT/primes_le_4096_f32.json 2>/dev/null
python tools/prime_sweep.py --dtype f64 > $OUT/primes_le_4096_f64.json 2>/dev/null
# fuzz on the FINAL library: host slices, device-resident, the round-3 and round-6 plan families
python tools/fuzz_gpu.py 400 61 > $OUT/fuzz_gpu_host.log 2>&1
python tools/fuzz_gpu.py 400 62 device > $OUT/fuzz_gpu_device.log 2>&1
FUZZ_ROUND3=1 python tools/fuzz_gpu.py 200 63 device > $OUT/fuzz_gpu_round3_device.log 2>&1
FUZZ_ROUND6=1 python tools/fuzz_gpu.py 400 64 device > $OUT/fuzz_gpu_round6_device.log 2>&1
FUZZ_ROUND6=1 python tools/fuzz_gpu.py 200 65 > $OUT/fuzz_gpu_round6_host.log 2>&1
python tools/r6_full_occupancy_parity.py > $OUT/full_occupancy_parity.jsonl 2> $OUT/full_occupancy_parity.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1
python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
( cd /tmp && python $ROOT/tools/pmc_sq.py --config c4 > $OUT/sq_counters_c4.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --sweep --dtype f32 --sizes 1019 > $OUT/sq_counters_bluestein_2048.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --sweep --dtype f32 --sizes 3067 > $OUT/sq_counters_bluestein_6144.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --sweep --dtype f32 --sizes 2368 > $OUT/sq_counters_lsm_2368.jsonl 2>/dev/null )
( cd /tmp && python $ROOT/tools/pmc_sq.py --sweep --dtype f32 --sizes 592 > $OUT/sq_counters_lsm_592.jsonl 2>/dev/null )
ls -la $OUT
tail -n 3 $OUT/pytest_gpu.log $OUT/fuzz_gpu_*.log $OUT/smoke_final.log
tail -n 1 $OUT/full_occupancy_parity.jsonl
