mkdir -p gpurun_out/r2
python tools/ab.py --log2n 20 --batch 1024 default tuning:MI355FFT_VARIANT=1 tuning:MI355FFT_VARIANT=3 2>&1 | grep arm | tee gpurun_out/r2/ab4.jsonl | cut -c1-330
python tools/ab.py --log2n 18 --batch 4096 default tuning:MI355FFT_VARIANT=9 2>&1 | grep arm | tee -a gpurun_out/r2/ab4.jsonl | cut -c1-330
python tools/ab.py --log2n 21 --batch 512 default tuning:MI355FFT_VARIANT=2 2>&1 | grep arm | tee -a gpurun_out/r2/ab4.jsonl | cut -c1-330
python tools/ab.py --log2n 20 --batch 512 --dtype f64 default 2>&1 | grep arm | tee -a gpurun_out/r2/ab4.jsonl | cut -c1-330
python tools/ab.py --log2n 19 --batch 1024 --dtype f64 default 2>&1 | grep arm | tee -a gpurun_out/r2/ab4.jsonl | cut -c1-330
