#!/bin/bash
# alternative library for the A/B of the new whole-row kernels (kernels_smooth4_f32_*): the same objects, those eight units compiled
# WITHOUT the SLP vectoriser -> rustfft_amd/lib/libmi355fft_s4ns.so
set -eu
cd "$(dirname "$0")/../../rustfft_amd/csrc"
mkdir -p build_s4ns
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-value --offload-compress -fno-slp-vectorize"
for i in 0 1 2 3 4 5 6 7; do
  ( /opt/rocm/bin/hipcc $FLAGS -c kernels_smooth4_f32_$i.hip -o build_s4ns/kernels_smooth4_f32_$i.o ) &
done
wait
OBJS=$(ls build/*.o | grep -v "kernels_smooth4_f32_[0-7].o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmi355fft_s4ns.so $OBJS build_s4ns/kernels_smooth4_f32_*.o
ls -la ../lib/libmi355fft_s4ns.so
