#!/bin/bash
# Round 4, call 2: per-kernel durations of chunk launches (ring in the Infinity Cache) vs full-batch launches, rocprofv3 kernel trace.
set -u
O=$PWD/gpurun_out/r4_02; mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  ( cd /tmp && env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O/$name -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/r4/pipe_trace.py 20 1024 3 > $O/$name.log 2>&1 )
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1)
  echo "== $name"; head -6 "$f" | cut -c1-200
}
run full MI355FFT_PIPE=0
run pipe1_64 MI355FFT_PIPE=1 MI355FFT_PIPE_MIB=64
run pipe1_128 MI355FFT_PIPE=1 MI355FFT_PIPE_MIB=128
run pipe1_32 MI355FFT_PIPE=1 MI355FFT_PIPE_MIB=32
run pipe1_256 MI355FFT_PIPE=1 MI355FFT_PIPE_MIB=256
run pipe1_1024 MI355FFT_PIPE=1 MI355FFT_PIPE_MIB=1024
find $O -name '*.db' -delete; find $O -name '*trace.csv' -size +2M -delete
