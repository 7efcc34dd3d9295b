# scratch entry point for `gpurun -- 'bash tools/run_ab.sh'` during kernel A/B work: edit, run, copy what matters from
# gpurun_out/ into profiles/rN/.  The A/B drivers themselves: tools/ab.py (variants of one length), tools/ab_lengths.py (two builds
# over a set of lengths), tools/algo_compare.py (AUTO vs forced recipe families), tools/phase_stamps.py (probe builds).
mkdir -p gpurun_out/r3
python tools/ab.py --log2n 20 --batch 1024 default 2>/dev/null | grep '^{' | cut -c1-400
