mkdir -p gpurun_out/r3
O=gpurun_out/r3/ab_pk.jsonl
: > $O
L=libmi355fft_exp.so
P=libmi355fft_exp_pk.so
run() { python tools/ab.py "$@" 2>&1 | grep '^{' | cut -c1-420 >> $O; }
run --log2n 20 --batch 1024 --oop $L:MI355FFT_VARIANT=20 $P:MI355FFT_VARIANT=20 $L:MI355FFT_VARIANT=40 $P:MI355FFT_VARIANT=40
run --log2n 22 --batch 256 --oop $L $P
run --log2n 12 --batch 131072 --oop $L $P
run --log2n 14 --batch 32768 --oop $L $P
python3 - <<'PY'
import json
for l in open('gpurun_out/r3/ab_pk.jsonl'):
    d=json.loads(l); print(d['n'], d['arm'][-34:], d['pair_ms_median'], d['kernel_GBps'], '%.2e'%d['rel_l2_row0'], d['plan'][-100:])
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3/pytest_gpu_twl.log
python bench.py > gpurun_out/r3/bench_twl.json 2> gpurun_out/r3/bench_twl.stderr
tail -3 gpurun_out/r3/pytest_gpu_twl.log; cut -c1-1500 gpurun_out/r3/bench_twl.json
