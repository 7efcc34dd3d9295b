#!/bin/bash
# the VALU-bound prime paths (Rader 1009, one-kernel Bluestein bodies) compiled WITHOUT the SLP vectoriser (-fno-slp-vectorize):
# the vectoriser pairs re / im of DIFFERENT values into v_pk_* operations and pays for it in v_mov_b32 (330 of the 1056 VALU instructions of rader<1008>)
set -u
O=gpurun_out/r4_31; mkdir -p $O
S=47,67,131,257,401,503,719,809,1009,1019,1283,1531,2039,2557,3067,3583,4093,4099,6007,8191
timeout 600 python tools/ab_lengths.py --a libmi355fft_tuning_min.so --b libmi355fft_tuning_min_noslp.so --sizes $S --dtype f32 --gib 1 --check --all > $O/ab_noslp_f32.jsonl 2> $O/err_f32.txt
timeout 600 python tools/ab_lengths.py --a libmi355fft_tuning_min.so --b libmi355fft_tuning_min_noslp.so --sizes $S --dtype f64 --gib 1 --check --all > $O/ab_noslp_f64.jsonl 2> $O/err_f64.txt
timeout 300 python tools/ab_lengths.py --a libmi355fft_tuning_min.so --b libmi355fft_tuning_min_noslp.so --sizes 1024,4096,32768,65536,1048576,4194304 --dtype f32 --gib 2 --check --all > $O/ab_noslp_pow2_f32.jsonl 2> $O/err_p2.txt
for f in $O/*.jsonl; do echo "== $f"; cut -c1-330 $f; done
tail -n 3 $O/err_*.txt | grep -v amdgpu.ids
