#!/bin/bash
# round 5, call 3: the new -m gpu tests on the library with the 13-smooth whole-row kernels; those kernels against the two-pass plans they
# replace (old library), with and without the SLP vectoriser (two runs), Complex<f64> as well; non-temporal LOADS on the whole-row kernels confirmed
set -u
O=gpurun_out/r5_03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "soak or giveup or second_process" > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log
timeout 600 python tools/ab_lengths.py --a libmi355fft_r4k.so --b libmi355fft.so --check --sizes-file tools/r5/smooth4_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_smooth4_vs_twopass_f32.jsonl 2> $O/err1.txt; echo rc $?
timeout 400 python tools/ab_lengths.py --a libmi355fft_r4k.so --b libmi355fft.so --check --sizes-file tools/r5/smooth4_f64_lengths.txt --dtype f64 --gib 1 > $O/ab_smooth4_vs_twopass_f64.jsonl 2> $O/err2.txt; echo rc $?
for rep in 1 2; do
timeout 600 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_s4ns.so --all --check --sizes-file tools/r5/smooth4_f32_lengths.txt --dtype f32 --gib 1 > $O/ab_smooth4_noslp_f32_rep$rep.jsonl 2> $O/err3_$rep.txt; echo rc $?
done
for k in 10 13 14; do
  b=$(( (1<<29) >> k ))
  timeout 120 python tools/ab.py --log2n $k --batch $b --rounds 11 min min:MI355FFT_VARIANT=51 min min:MI355FFT_VARIANT=51 > $O/ab_k1_ntload_confirm_2p$k.jsonl 2>> $O/ab.err
done
python - $O <<'PY'
import json,sys,statistics as st,glob
O=sys.argv[1]
for f in ("ab_smooth4_vs_twopass_f32","ab_smooth4_vs_twopass_f64"):
    r=[json.loads(l) for l in open(f"{O}/{f}.jsonl") if l.startswith("{")]
    if r:
        v=[d["b_over_a"] for d in r]; print(f, len(r), "median", st.median(v), "min", min(v), "max", max(v), "losers", [(d["n"],d["b_over_a"]) for d in r if d["b_over_a"]<1.0][:20], "max rel", max(d["rel_l2_b_vs_a"] for d in r))
r=[{json.loads(l)["n"]:json.loads(l)["b_over_a"] for l in open(f"{O}/ab_smooth4_noslp_f32_rep{k}.jsonl") if l.startswith("{")} for k in (1,2)]
both=[n for n in r[0] if n in r[1]]
print("noslp: n", len(both), "median", st.median(r[0].values()), st.median(r[1].values()), "faster>=2% both", sum(1 for n in both if min(r[0][n],r[1][n])>=1.02), "slower<=-2% both", sum(1 for n in both if max(r[0][n],r[1][n])<=0.98))
for f in sorted(glob.glob(O+"/ab_k1_ntload_confirm_2p*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["arm"], d["pair_ms_median"], d.get("kernel_GBps"))
PY
