mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r3/pytest_gpu_smooth_twl.log
python tools/sweep.py --dtype f32 --sizes 100,360,1000,1200,1500,2000,2310,3000,3553,4000,6000,10000,30000 --check 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['n'], round(r['gflops']), [round(x) for x in r['kernel_GBps']], '%.1e'%r['rel_l2'], r['plan'][:50])"
