mkdir -p gpurun_out/r3
python tools/ab_lengths.py --all --a libmi355fft.so --b libmi355fft_exp.so --sizes 47,59,83,107,167,179,227,263,347,383,467,503,587,719,839,887,983,1019,1187,1283,1367,1439,1523,1619,1823,1907,2027,2063,2207,2459,2579,2819,2903,2999,3119,3203,3467,3623,3803,3947,4079,4093 > gpurun_out/r3/ab_bluestein_input_batched_f32.jsonl 2>/dev/null
python tools/ab_lengths.py --all --dtype f64 --a libmi355fft.so --b libmi355fft_exp.so --sizes 59,167,263,503,719,1019,1283,1523,2027,2579,3119,4079 > gpurun_out/r3/ab_bluestein_input_batched_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json,statistics
for fn in ('ab_bluestein_input_batched_f32','ab_bluestein_input_batched_f64'):
    rows=[json.loads(l) for l in open('gpurun_out/r3/%s.jsonl'%fn) if l.startswith('{')]
    r=[x['b_over_a'] for x in rows]
    print(fn, 'n=%d median %.3f min %.3f max %.3f'%(len(r), statistics.median(r), min(r), max(r)))
    print('   ', ' '.join('%d:%.2f(%.2f)'%(x['n'],x['b_over_a'],x['b_TBps']) for x in rows))
PY
