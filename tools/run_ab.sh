mkdir -p gpurun_out/r3
python tools/ab_lengths.py --all --a libmi355fft_prev.so --b libmi355fft.so --sizes 4200,5000,6561,8748,10000,15625,20449,30000,44100,45056,65000,100000,177147,362880,500000,1000000,1536000,3000000,7340032,10007,100003,12289,65537 > gpurun_out/r3/ab_k2g_fence_f32.jsonl 2>/dev/null
python tools/ab_lengths.py --all --dtype f64 --a libmi355fft_prev.so --b libmi355fft.so --sizes 5000,10000,20449,44100,100000,362880,1000000,1536000,10007,65537 > gpurun_out/r3/ab_k2g_fence_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json,statistics
for fn in ('ab_k2g_fence_f32','ab_k2g_fence_f64'):
    rows=[json.loads(l) for l in open('gpurun_out/r3/%s.jsonl'%fn) if l.startswith('{')]
    r=[x['b_over_a'] for x in rows]
    print(fn, 'n=%d median %.3f min %.3f max %.3f'%(len(r), statistics.median(r), min(r), max(r)))
    print('   ', ' '.join('%d:%.2f(%.2f)'%(x['n'],x['b_over_a'],x['b_TBps']) for x in rows))
PY
