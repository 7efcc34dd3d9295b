mkdir -p gpurun_out/r3
PR=$(python3 -c "print(','.join(str(p) for p in range(17,1300) if all(p%q for q in range(2,int(p**0.5)+1))))")
python tools/ab_lengths.py --all --a libmi355fft_prev.so --b libmi355fft.so --sizes $PR > gpurun_out/r3/ab_batched_loads_primes_f32.jsonl 2>/dev/null
python tools/ab_lengths.py --all --dtype f64 --a libmi355fft_prev.so --b libmi355fft.so --sizes 37,41,53,61,73,97,101,113,127,151,181,193,241,257,281,313,337,401,433,449,521,541,577,601,641,673,769 > gpurun_out/r3/ab_batched_loads_primes_f64.jsonl 2>/dev/null
python tools/ab_lengths.py --all --a libmi355fft_prev.so --b libmi355fft.so --sizes 8384,8633,10403,10763,65231,158381 > gpurun_out/r3/ab_batched_loads_k2r_f32.jsonl 2>/dev/null
python tools/ab_lengths.py --all --dtype f64 --a libmi355fft_prev.so --b libmi355fft.so --sizes 8633,10403,65231 > gpurun_out/r3/ab_batched_loads_k2r_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json,statistics
for fn in ('ab_batched_loads_primes_f32','ab_batched_loads_primes_f64','ab_batched_loads_k2r_f32','ab_batched_loads_k2r_f64'):
    rows=[json.loads(l) for l in open('gpurun_out/r3/%s.jsonl'%fn) if l.startswith('{')]
    if not rows: print(fn,'EMPTY'); continue
    rad=[x for x in rows if 'rader' in x['plan_b'] or 'k2r' in x['plan_b']]
    oth=[x for x in rows if x not in rad]
    for nm,rr in (('rader/k2r',rad),('other',oth)):
        if not rr: continue
        r=[x['b_over_a'] for x in rr]
        print(fn, nm, 'n=%d median %.3f min %.3f max %.3f'%(len(r), statistics.median(r), min(r), max(r)))
    print('   ', ' '.join('%d:%.2f(%.2f)'%(x['n'],x['b_over_a'],x['b_TBps']) for x in rad[:80]))
PY
