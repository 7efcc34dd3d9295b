mkdir -p gpurun_out/r2
python tools/sweep.py --dtype f32 --sizes 7919,19,31,127,251,509,719,1019,1531,2039,3079,4093 --check --bytes 1 2>/dev/null | cut -c1-250 | tee gpurun_out/r2/bs_regs_f32.jsonl
python tools/sweep.py --dtype f64 --sizes 19,127,719,1019,2039,4093 --check --bytes 1 2>/dev/null | cut -c1-250 | tee gpurun_out/r2/bs_regs_f64.jsonl
