#!/bin/bash
# round 5, call 24: the new Rader bodies of the 31-smooth primes: f32 with / without the SLP vectoriser (x31 vs x31b), plain side-by-side body
# against the register hand-over (x31 vs x31c, f32 and f64); one process, interleaved, two runs each
set -u
O=gpurun_out/r5_24; mkdir -p $O
for run in 1 2; do
  if [ $SECONDS -gt 300 ]; then break; fi
  timeout 240 python tools/ab_lengths.py --a libmi355fft_x31.so --b libmi355fft_x31b.so --all --check --sizes-file tools/r5/primes_x31.txt --dtype f32 > $O/ab_x31_slp_f32_run$run.jsonl 2>> $O/ab.err
  timeout 240 python tools/ab_lengths.py --a libmi355fft_x31.so --b libmi355fft_x31c.so --check --sizes-file tools/r5/primes_x31.txt --dtype f32 > $O/ab_x31_m5_f32_run$run.jsonl 2>> $O/ab.err
  timeout 240 python tools/ab_lengths.py --a libmi355fft_x31.so --b libmi355fft_x31c.so --check --sizes-file tools/r5/primes_x31.txt --dtype f64 > $O/ab_x31_m5_f64_run$run.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,statistics,glob
o=sys.argv[1]
for tag in ("slp_f32","m5_f32","m5_f64"):
    r={}
    for fn in glob.glob(f"{o}/ab_x31_{tag}_run*.jsonl"):
        for l in open(fn):
            if l.startswith("{"):
                d=json.loads(l)
                if "b_over_a" in d: r.setdefault(d["n"],[]).append(d["b_over_a"])
    v=[min(x) for x in r.values()]
    print(tag, len(v), "median", statistics.median(v) if v else None, "wins>3%", sum(1 for x in v if x>1.03))
PY
