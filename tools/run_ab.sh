mkdir -p gpurun_out/r2
python tools/sweep.py --dtype f32 --sizes 4099,5759,7919,8191,10007,10403,12289,16381 --check --bytes 1 2>/dev/null | cut -c1-330 | tee gpurun_out/r2/bss_f32.jsonl
python tools/sweep.py --dtype f64 --sizes 4099,5759,8191 --check --bytes 1 2>/dev/null | cut -c1-330 | tee gpurun_out/r2/bss_f64.jsonl
python -m pytest tests/test_gpu_parity.py -q -k repeatability 2>&1 | tail -3
