#!/bin/bash
# round 5, call 2: the new -m gpu tests; the round-4 failure reproduced with the tool as it was (no workspace trim) to read the new message;
# absolute rates of the prime-tile / large-Rader list on the shipped library; non-temporal accesses re-measured on the real whole-row kernels
# and on the 2048-row tiles (tuning-min build, one interleaved process per size)
set -u
O=gpurun_out/r5_02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "soak or giveup or second_process" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --all --no-trim --sizes-file tools/r4/general_f32_lengths.txt --dtype f32 --gib 1 > $O/repro_notrim.jsonl 2> $O/repro_notrim.err; echo "no-trim repro rc $?"; tail -1 $O/repro_notrim.err; wc -l $O/repro_notrim.jsonl
timeout 300 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --all --check --sizes-file tools/r4/prime_tile_rader_large_lengths.txt --dtype f32 --gib 1 > $O/abs_prime_tile_rader_large.jsonl 2> $O/abs_prime.err; echo "prime list rc $?"; wc -l $O/abs_prime_tile_rader_large.jsonl
for k in 10 11 12 13 14 15; do
  b=$(( (1<<29) >> k ))
  timeout 120 python tools/ab.py --log2n $k --batch $b --rounds 7 min min:MI355FFT_VARIANT=50 min:MI355FFT_VARIANT=51 min:MI355FFT_VARIANT=52 > $O/ab_k1_nt_2p$k.jsonl 2>> $O/ab.err
done
timeout 200 python tools/ab.py --log2n 22 --batch 256 --rounds 7 --fwd-only min min:MI355FFT_VARIANT=70 min:MI355FFT_VARIANT=71 min:MI355FFT_VARIANT=72 > $O/ab_2048_nt_2p22.jsonl 2>> $O/ab.err
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_*nt_2p*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"], d["pair_ms_median"], d.get("kernel_GBps"), d["plan"][:80], d.get("rel_l2_row0"))
PY
