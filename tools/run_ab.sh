mkdir -p gpurun_out/r3
python tools/ab_lengths.py --all --a libmi355fft.so --b libmi355fft_bst.so --sizes-file tools/bluestein_sample_f32.txt > gpurun_out/r3/ab_bluestein_tw1_f32.jsonl 2>/dev/null
python tools/ab_lengths.py --all --dtype f64 --a libmi355fft.so --b libmi355fft_bst.so --sizes-file tools/bluestein_sample_f64.txt > gpurun_out/r3/ab_bluestein_tw1_f64.jsonl 2>/dev/null
python3 - <<'PY'
import json,statistics,re,collections
for fn in ('ab_bluestein_tw1_f32','ab_bluestein_tw1_f64'):
    rows=[json.loads(l) for l in open('gpurun_out/r3/%s.jsonl'%fn) if l.startswith('{')]
    r=[x['b_over_a'] for x in rows]
    print(fn, 'n=%d median %.3f min %.3f max %.3f'%(len(r), statistics.median(r), min(r), max(r)))
    byM=collections.defaultdict(list)
    for x in rows:
        m=re.match(r'bluestein<(\d+),', x['plan_a'])
        if m: byM[int(m.group(1))].append(x['b_over_a'])
    print('   ', ' '.join('%d:%.2f(%d)'%(M, statistics.median(v), len(v)) for M,v in sorted(byM.items())))
PY
