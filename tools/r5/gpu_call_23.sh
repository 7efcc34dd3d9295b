#!/bin/bash
# round 5, call 23: every prime <= 4096 with a 31-smooth (not 13-smooth) p - 1 through a compiled Rader body (libmi355fft_x31.so, tools/r5/build_x31.sh)
# against the shipped library (the one-kernel Bluestein for all but 6 / 13 of them), one process, interleaved, 0.5 GiB of rows, two runs
set -u
O=gpurun_out/r5_23; mkdir -p $O
for run in 1 2; do
  if [ $SECONDS -gt 300 ]; then break; fi
  timeout 240 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_x31.so --check --sizes-file tools/r5/primes_x31.txt --dtype f32 > $O/ab_x31_f32_run$run.jsonl 2>> $O/ab.err
  timeout 240 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_x31.so --check --sizes-file tools/r5/primes_x31.txt --dtype f64 > $O/ab_x31_f64_run$run.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,statistics
o=sys.argv[1]
for dt in ("f32","f64"):
    r={}
    import glob
    for fn in glob.glob(f"{o}/ab_x31_{dt}_run*.jsonl"):
        for l in open(fn):
            if l.startswith("{"):
                d=json.loads(l)
                if "b_over_a" in d: r.setdefault(d["n"],[]).append(d["b_over_a"])
    v=[min(x) for x in r.values()]
    print(dt, len(v), "median", statistics.median(v) if v else None, "wins>3%", sum(1 for x in v if x>1.03), "max rel_l2 see files")
PY
tail -3 $O/ab.err
