#!/bin/bash
# round 6: first device run of the LDS stage machine -- parity against numpy on a handful of lengths, then AUTO (lsm) against the forced Bluestein plan
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
python - <<'PY' > gpurun_out/r6/lsm_sanity.txt 2>&1
import numpy as np, torch, rustfft_amd
for dt in (np.complex64, np.complex128):
    pl = rustfft_amd.FftPlanner(dt)
    for n in (74, 148, 167, 592, 629, 1369, 1517, 2368, 1019, 3034, 4218, 8144, 7919, 16206):
        for d in (0, 1):
            fft = pl.plan_fft(n, d)
            rng = np.random.default_rng(n)
            b = 300
            x = (rng.uniform(0, 10, n * b) + 1j * rng.uniform(0, 10, n * b)).astype(dt)
            t = torch.from_numpy(x).cuda()
            fft.process(t)
            torch.cuda.synchronize()
            got = t.cpu().numpy().reshape(b, n)
            xx = x.reshape(b, n).astype(np.complex128)
            want = np.fft.fft(xx, axis=1) if d == 0 else np.fft.ifft(xx, axis=1) * n
            rel = np.linalg.norm(got - want) / np.linalg.norm(want)
            print(n, np.dtype(dt).name, d, f"{rel:.3e}", fft.describe(), flush=True)
            assert rel < (3e-6 if dt == np.complex64 else 1e-14)
print("sanity ok")
PY
tail -3 gpurun_out/r6/lsm_sanity.txt
S=74,148,185,222,296,370,592,629,703,1110,1369,1517,2368,3034,3774,167,347,1019,1283,2027,3067,4093,4218,5328,8144,7919,9472,12321,16206,10007
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f32_first.jsonl 2> gpurun_out/r6/lsm_ab_f32.err
python tools/ab_lengths.py --a libmi355fft.so --b libmi355fft.so --a-algo bluestein --check --all --gib 0.25 --dtype f64 --sizes $S > gpurun_out/r6/lsm_vs_bluestein_f64_first.jsonl 2> gpurun_out/r6/lsm_ab_f64.err
cat gpurun_out/r6/lsm_vs_bluestein_f32_first.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['a_TBps'], d['b_TBps'], d['b_over_a'], d['rel_l2_b_vs_a'], d['plan_b'][:90])
"
