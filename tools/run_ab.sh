mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py -x -q -k "host_slices or error_paths or api" 2>&1 | tail -4
python tools/hostpath_bench.py 2>&1 | tee gpurun_out/r3/hostpath_bench.jsonl
