mkdir -p gpurun_out/r2
P=17,31,73,101,127,193,257,331,401,541,641,761,881,1009,1201,1301,1453,1621,2003,2311,2521,2731,3001,3301,3511,3851,4001,4051
python tools/algo_compare.py --sizes $P --dtype f32 > gpurun_out/r2/rader_vs_bs_f32.jsonl 2>gpurun_out/r2/rader_err.txt; cut -c1-330 gpurun_out/r2/rader_vs_bs_f32.jsonl; tail -3 gpurun_out/r2/rader_err.txt
python tools/algo_compare.py --sizes 127,331,1009,2003,3001,4001 --dtype f64 > gpurun_out/r2/rader_vs_bs_f64.jsonl 2>/dev/null; cut -c1-330 gpurun_out/r2/rader_vs_bs_f64.jsonl
R=17,34,68,119,289,391,437,527,620,899,961,992,1023,1088,1734,2465,3553,4352,6448
python tools/algo_compare.py --sizes $R --dtype f32 > gpurun_out/r2/pr_vs_bs_f32.jsonl 2>/dev/null; cut -c1-330 gpurun_out/r2/pr_vs_bs_f32.jsonl
python tools/algo_compare.py --sizes 17,289,527,992,1088,4352 --dtype f64 > gpurun_out/r2/pr_vs_bs_f64.jsonl 2>/dev/null; cut -c1-330 gpurun_out/r2/pr_vs_bs_f64.jsonl
