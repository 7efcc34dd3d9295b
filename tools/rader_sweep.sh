for v in ${VARS32:-0 5 16 17 18 19 20}; do
  MI355FFT_VARIANT=$v timeout 120 python tools/sweep.py --dtype f32 --sizes 1009 --bytes 2 2>&1 | grep '"n"' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('f32 v$v', r['plan'], r['ms'], r['gflops'], r['alg_GBps'])"
done
for v in ${VARS64:-0 3 16 18}; do
  MI355FFT_VARIANT=$v timeout 120 python tools/sweep.py --dtype f64 --sizes 1009 --bytes 2 2>&1 | grep '"n"' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('f64 v$v', r['plan'], r['ms'], r['gflops'], r['alg_GBps'])"
done
