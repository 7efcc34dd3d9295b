#!/bin/bash
# Round-5 refresh of the artifacts kept under profiles/r5/ (runs on the GPU box through gpurun; outputs in gpurun_out/profiles_r5/).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_r5
mkdir -p $OUT
export TMPDIR=/tmp
( lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; cargo --version 2>&1; rustc --version 2>&1; rocminfo | grep -E "gfx950|Compute Unit" | head -4; ls -la rustfft_amd/lib/libmi355fft.so ) > $OUT/toolchain_probe.txt 2>&1
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.stderr
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $ROOT/bench.py --no-pmc --no-cpu-baseline --no-side > $OUT/bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
python bench.py --fused 0 --no-pmc --no-cpu-baseline --no-side > $OUT/bench_two_launch.json 2>/dev/null
for c in c4 c5; do
  python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c --output-format csv -- python $ROOT/bench.py --config $c --no-pmc --no-cpu-baseline > /dev/null 2>&1 )
  cp $(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1) $OUT/bench_${c}_kernel_stats.csv 2>/dev/null
done
python tools/sweep.py --dtype f32 --min 10 --max 24 --bytes 4 --check > $OUT/sweep_pow2_f32_4GiB.jsonl 2>/dev/null
python tools/sweep.py --dtype f32 --min 16 --max 24 --bytes 4 --fused 0 > $OUT/sweep_pow2_f32_4GiB_two_launch.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --min 10 --max 24 --bytes 4 --check > $OUT/sweep_pow2_f64_4GiB.jsonl 2>/dev/null
NP2=3,7,17,77,100,127,251,289,360,719,899,1000,1001,1009,1019,1200,1201,1517,2003,2310,3000,4093,4099,4875,5000,5082,6006,8633,10000,10007,10403,12289,19683,20449,25000,41959,44100,45056,65231,65537,100000,100003,158381,216569,417623,1000000,1000003,1536000,7340032
python tools/sweep.py --dtype f32 --sizes $NP2 --check > $OUT/sweep_np2_f32.jsonl 2>/dev/null
python tools/sweep.py --dtype f64 --sizes $NP2 --check > $OUT/sweep_np2_f64.jsonl 2>/dev/null
python tools/prime_sweep.py > $OUT/primes_le_4096_f32.json 2>/dev/null
python tools/prime_sweep.py --dtype f64 > $OUT/primes_le_4096_f64.json 2>/dev/null
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --all --sizes-file tools/r5/smooth13_4096_20000.txt --dtype f32 --gib 1 > $OUT/abs_smooth13_4096_20000_f32.jsonl 2>/dev/null
python bench.py --gpus 2 --one-device --dist-backend gloo --steps 2 --warmup 1 --batch 64 --no-pmc --no-cpu-baseline > $OUT/bench_2rank_one_gpu_smoke.json 2>/dev/null
python bench.py --via-cabi --gpus 2 --one-device --batch 512 > $OUT/bench_via_cabi_2shards_one_gpu.json 2>/dev/null
python tools/fuzz_gpu.py > $OUT/fuzz_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1
python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
python tools/pmc_sq.py --config c4 > $OUT/sq_counters_c4.jsonl 2>/dev/null
python tools/pmc_sq.py --sweep --dtype f32 --sizes 3067,4091 > $OUT/sq_counters_bluestein_6144_8192.jsonl 2>/dev/null
ls -la $OUT
