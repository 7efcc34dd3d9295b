#!/bin/bash
# Round 4, call 17: (a) the waste-aware split of the general two-pass plans against the balanced one (two builds, one process);
# (b) the --offload-compress build: latencies and results against the plain build; (c) fused pairings: reversed pass order at
# 2^17 / 2^19, 2^22 on 1024-thread workgroups; (d) the two-columns-per-lane LATER pass of the 2048-row tile alone.
set -u
O=gpurun_out/r4_17; mkdir -p $O
timeout 400 python tools/ab_lengths.py --a libmi355fft_bal.so --b libmi355fft.so --sizes-file tools/r4/split_changed_lengths.txt --check > $O/ab_split_waste_f32.jsonl 2> $O/split.err; echo "split f32 rc=$?"
timeout 300 python tools/ab_lengths.py --a libmi355fft_bal.so --b libmi355fft.so --sizes 4225,4290,4550,5005,5265,6435,8085,9009,12005,15015,17325 --dtype f64 --check > $O/ab_split_waste_f64.jsonl 2>> $O/split.err; echo "split f64 rc=$?"
python - <<'PY'
import json,statistics
for f in ("f32","f64"):
    rows=[json.loads(l) for l in open("gpurun_out/r4_17/ab_split_waste_%s.jsonl"%f)]
    r=[x["b_over_a"] for x in rows]
    print(f, len(rows), "median b/a", statistics.median(r), "min", min(r), "max", max(r), "wins>3%", sum(v>1.03 for v in r), "losses>3%", sum(v<0.97 for v in r), "max rel", max(x["rel_l2_b_vs_a"] for x in rows))
    for x in rows[:6]: print("   ", x["n"], x["a_TBps"], x["b_TBps"], x["plan_a"][:40], "|", x["plan_b"][:40])
PY
timeout 200 python tools/latency_probe.py rustfft_amd/lib/libmi355fft_z.so > $O/latency_compressed.json 2> $O/lat.err; echo "latency z rc=$?"; cut -c1-600 $O/latency_compressed.json
timeout 200 python tools/latency_probe.py > $O/latency_plain.json 2>> $O/lat.err; cut -c1-600 $O/latency_plain.json
timeout 200 python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft_z.so --sizes 1009,1200,4099,65536,1048576,4225,12289 --check --all > $O/ab_compressed_vs_plain.jsonl 2>> $O/lat.err; cut -c1-200 $O/ab_compressed_vs_plain.jsonl
run() { name=$1; shift; timeout 200 python tools/ab.py "$@" > $O/$name.jsonl 2> $O/$name.err; echo "== $name rc=$?"; python - $O/$name.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-52s pair %.3f ms %s kern %s rel %.2e st %s %s" % (d["arm"], d["pair_ms_median"], d["instance_medians_ms"], d["kernel_ms_median"], d["rel_l2_row0"], d["fused_status"], d["plan"][:60]))
PY
tail -2 $O/$name.err | cut -c1-200; }
run ab_fused_rev_2p17 --log2n 17 --batch 8192 --rounds 4 --instances 2 min:MI355FFT_FUSE=8 min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=8,MI355FFT_ORDER=1 min:MI355FFT_FUSE=7,MI355FFT_ORDER=1
run ab_fused_rev_2p19 --log2n 19 --batch 2048 --rounds 4 --instances 2 min:MI355FFT_FUSE=8 min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=8,MI355FFT_ORDER=1 min:MI355FFT_FUSE=7,MI355FFT_ORDER=1
run ab_fused_1024thr_2p22 --log2n 22 --batch 256 --rounds 4 --instances 2 min:MI355FFT_FUSE=8 min:MI355FFT_FUSE=7 min:MI355FFT_FUSE=7,MI355FFT_FUSE_RING=110 min:MI355FFT_FUSE=8,MI355FFT_VARIANT=54
