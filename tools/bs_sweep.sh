for v in ${VARS:-0 1 2 3 4}; do
  for dt in f32 f64; do
  MI355FFT_VARIANT=$v timeout 300 python tools/sweep.py --dtype $dt --sizes ${SIZES:-127,509,719,1019,2039} 2>&1 | grep '"n"' | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$dt v$v', r['n'], r['gflops'], r['kernel_GBps'], r['plan'])"
  done
done
