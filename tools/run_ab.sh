mkdir -p gpurun_out/r2
python tools/ab.py --oop --dtype f64 --log2n 21 --batch 256 min min:MI355FFT_MAXR=1024 2>&1 | grep arm | cut -c1-330 | tee gpurun_out/r2/ab10.jsonl
python tools/ab.py --oop --dtype f64 --log2n 22 --batch 128 min min:MI355FFT_MAXR=1024 2>&1 | grep arm | cut -c1-330 | tee -a gpurun_out/r2/ab10.jsonl
python tools/ab.py --oop --dtype f64 --log2n 20 --batch 512 min min:MI355FFT_VARIANT=39 2>&1 | grep arm | cut -c1-330 | tee -a gpurun_out/r2/ab10.jsonl
