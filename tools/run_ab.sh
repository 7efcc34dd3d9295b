mkdir -p gpurun_out/r2
python tools/ab.py --oop --n 1009 --batch 1048576 --rounds 5 min min:MI355FFT_VARIANT=30 min:MI355FFT_VARIANT=31 min:MI355FFT_VARIANT=32 min:MI355FFT_VARIANT=33 min:MI355FFT_VARIANT=34 min:MI355FFT_VARIANT=35 min:MI355FFT_VARIANT=36 2>&1 | grep arm | cut -c1-260 | tee gpurun_out/r2/ab8.jsonl
