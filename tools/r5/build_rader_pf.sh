#!/bin/bash
# alternative library for the A/B of non-temporal row loads in the Rader rows loops (kernels.h rader_rows_body, -DMI355_RADER_PF): the shipped
# objects with the ten Rader units recompiled -> rustfft_amd/lib/libmi355fft_rpf.so
set -eu
cd "$(dirname "$0")/../../rustfft_amd/csrc"
mkdir -p build_rpf
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-value --offload-compress -DMI355_RADER_PF"
for u in rader_f32_0 rader_f32_1 rader_f32_2 rader_f32_3 rader_f64_0 rader_f64_1 rader_f64_2 rader_f64_3; do
  ( /opt/rocm/bin/hipcc $FLAGS -c kernels_$u.hip -o build_rpf/kernels_$u.o ) &
done
for u in rader_f32_ns0 rader_f32_ns1; do
  ( /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -c kernels_$u.hip -o build_rpf/kernels_$u.o ) &
done
wait
OBJS=$(ls build/*.o | grep -v "kernels_rader_f")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmi355fft_rpf.so $OBJS build_rpf/*.o
ls -la ../lib/libmi355fft_rpf.so
