#!/bin/bash
# round 5, call 6: non-temporal LOADS on config 3's kernel (1200, f64), config 4's Rader rows loop (probe build), the f64 whole-row kernels
set -u
O=gpurun_out/r5_06; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_process" > $O/pytest_second.log 2>&1; tail -3 $O/pytest_second.log
timeout 120 python tools/ab.py --n 1200 --dtype f64 --batch 65536 --rounds 11 --fwd-only min min:MI355FFT_VARIANT=51 min:MI355FFT_VARIANT=50 min min:MI355FFT_VARIANT=51 > $O/ab_c3_nt.jsonl 2>> $O/ab.err
timeout 120 python tools/ab.py --n 1009 --batch 1048576 --rounds 9 --fwd-only min libmi355fft_tuning_min_nt.so min libmi355fft_tuning_min_nt.so > $O/ab_c4_nt.jsonl 2>> $O/ab.err
for k in 10 11 12 13 14; do
  b=$(( (1<<28) >> k ))
  timeout 120 python tools/ab.py --log2n $k --dtype f64 --batch $b --rounds 9 --fwd-only min min:MI355FFT_VARIANT=51 min min:MI355FFT_VARIANT=51 > $O/ab_k1_f64_ntload_2p$k.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/ab_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"], d["pair_ms_median"], d["pair_ms_min"], d.get("kernel_GBps"), d["plan"][:60])
PY
tail -3 $O/ab.err
