mkdir -p gpurun_out/r3
O=gpurun_out/r3/ab_pair_fused_twl.jsonl
: > $O
L=libmi355fft_exp.so
timeout 300 python tools/ab.py --rounds 5 --log2n 20 --batch 1024 --oop $L $L:MI355FFT_VARIANT=60 $L:MI355FFT_VARIANT=61 $L $L:MI355FFT_VARIANT=60 $L:MI355FFT_VARIANT=61 2>/dev/null | grep '^{' | cut -c1-1000 >> $O
python3 - <<'PY'
import json
for l in open('gpurun_out/r3/ab_pair_fused_twl.jsonl'):
    d=json.loads(l); print(d['n'], d['arm'][-26:], d['pair_ms_median'], d['kernel_GBps'], '%.1e'%d['rel_l2_row0'], d['plan'][-40:])
PY
