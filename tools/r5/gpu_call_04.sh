#!/bin/bash
# round 5, call 4: the 13-smooth whole-row kernels (sorted into SLP / no-SLP units) against the two-pass plans of the previous library (the
# 365 / 173 new lengths; the 492 13-smooth lengths in (4096, 20000) the round-4 review names), non-temporal loads confirmed, the whole GPU suite
set -u
O=gpurun_out/r5_04; mkdir -p $O
timeout 600 python tools/ab_lengths.py --a libmi355fft_r4k.so --b libmi355fft.so --check --sizes-file tools/r5/smooth4_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_smooth4_vs_twopass_f32.jsonl 2> $O/err1.txt; echo rc $?
timeout 400 python tools/ab_lengths.py --a libmi355fft_r4k.so --b libmi355fft.so --check --sizes-file tools/r5/smooth4_f64_lengths.txt --dtype f64 --gib 1 > $O/ab_smooth4_vs_twopass_f64.jsonl 2> $O/err2.txt; echo rc $?
timeout 600 python tools/ab_lengths.py --a libmi355fft_r4k.so --b libmi355fft.so --all --check --sizes-file tools/r5/smooth13_4096_20000.txt --dtype f32 --gib 1 > $O/ab_smooth13_4096_20000_f32.jsonl 2> $O/err3.txt; echo rc $?
for k in 10 13 14; do
  b=$(( (1<<29) >> k ))
  timeout 120 python tools/ab.py --log2n $k --batch $b --rounds 11 min min:MI355FFT_VARIANT=51 min min:MI355FFT_VARIANT=51 > $O/ab_k1_ntload_confirm_2p$k.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,statistics as st,glob
O=sys.argv[1]
for f in ("ab_smooth4_vs_twopass_f32","ab_smooth4_vs_twopass_f64","ab_smooth13_4096_20000_f32"):
    r=[json.loads(l) for l in open(f"{O}/{f}.jsonl") if l.startswith("{")]
    if r:
        v=[d["b_over_a"] for d in r]; print(f, len(r), "median", st.median(v), "min", min(v), "max", max(v), "a median TB/s", st.median(d["a_TBps"] for d in r), "b", st.median(d["b_TBps"] for d in r), "losers", [(d["n"],d["b_over_a"]) for d in r if d["b_over_a"]<0.98][:30], "max rel", max(d["rel_l2_b_vs_a"] for d in r))
for f in sorted(glob.glob(O+"/ab_k1_ntload_confirm_2p*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"], d["pair_ms_median"], d.get("kernel_GBps"))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
