mkdir -p gpurun_out/r2
python tools/ab.py --oop --log2n 20 --batch 1024 min min:MI355FFT_VARIANT=12 min 2>&1 | grep arm | tee gpurun_out/r2/ab6.jsonl | cut -c1-300
python tools/ab.py --oop --log2n 18 --batch 4096 min min:MI355FFT_VARIANT=12 min 2>&1 | grep arm | tee -a gpurun_out/r2/ab6.jsonl | cut -c1-300
python tools/ab.py --oop --log2n 19 --batch 2048 min min:MI355FFT_VARIANT=12 2>&1 | grep arm | tee -a gpurun_out/r2/ab6.jsonl | cut -c1-300
