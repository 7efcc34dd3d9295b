mkdir -p gpurun_out/r2
for d in pm1 bench zero; do python tools/ab.py --log2n 20 --batch 1024 --dist $d default 2>&1 | grep arm | cut -c1-230; done
python tools/ab.py --log2n 20 --batch 1024 --iters 8 --rounds 3 --dist bench default 2>&1 | grep arm | cut -c1-230
python tools/ab.py --log2n 20 --batch 1024 --shift-mib 33.5 default 2>&1 | grep arm | cut -c1-230
python bench.py --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'][:12], round(k['GBps'])) for k in d['roofline']['kernels']])"
python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'][:12], round(k['GBps'])) for k in d['roofline']['kernels']])"
