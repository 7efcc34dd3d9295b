#!/bin/bash
# alternative library for the round-5 re-measurement of EVERY prime <= 4096 with a 31-smooth p - 1 as a compiled Rader body (round 2 kept 6 / 13
# of 99; the side-by-side bodies and the compiler flags have changed since): RADER_X31=all python tools/gen_rader_kernels.py  # RADER_X31_SLP=1 / RADER_X31_M5=1 and OUT=libmi355fft_x31b.so from the environment: the variants, the six changed
# Rader units recompiled over the shipped objects -> rustfft_amd/lib/libmi355fft_x31.so; the generated sources are restored afterwards
set -eu
cd "$(dirname "$0")/../.."
RADER_X31=all python tools/gen_rader_kernels.py
cd rustfft_amd/csrc
mkdir -p build_x31
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-value --offload-compress"
for u in rader_f64_0 rader_f64_1 rader_f64_2 rader_f64_3 rader_f32_0 rader_f32_1 rader_f32_2 rader_f32_3; do
  ( /opt/rocm/bin/hipcc $FLAGS -c kernels_$u.hip -o build_x31/kernels_$u.o ) &
done
for u in rader_f32_ns0 rader_f32_ns1; do
  ( /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -c kernels_$u.hip -o build_x31/kernels_$u.o ) &
done
wait
OBJS=$(ls build/*.o | grep -v "kernels_rader_f")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/${OUT:-libmi355fft_x31.so} $OBJS build_x31/*.o
ls -la ../lib/${OUT:-libmi355fft_x31.so}
cd ../..
python tools/gen_rader_kernels.py
git status --short rustfft_amd/csrc
