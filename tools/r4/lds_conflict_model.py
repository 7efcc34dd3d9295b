#!/usr/bin/env python3
"""Bank-conflict model of the column-tile exchanges (engine.h lds_scatter / lds_gather, linear padded layout of the
32-values-per-thread tiles) under the rules the counter probe confirmed on gfx950 (tools/r4/lds_counter_probe.py): a 64-lane
ds_*_b32 access is served in two groups of 32 lanes over 32 banks of 4 bytes; a k-way conflict in a group costs k array cycles
(SQ_LDS_IDX_ACTIVE) of which k - 1 are counted in SQ_LDS_BANK_CONFLICT; identical addresses broadcast.  Prints, per exchange
half of a schedule, the ways of every wave-instruction and the resulting conflict / idx-active ratio, to compare with the
kernel's measured 0.41 (profiles/r3/sq_counters_c2.jsonl)."""
import collections, json, sys


def phys(i):
    return i + i // 32


def pitch_for(n, mod):
    p = phys(n - 1) + 1
    while p % 32 != mod % 32:
        p += 1
    return p


def ways(addrs):
    """array cycles of one 64-lane b32 access: sum over the two 32-lane groups of the worst bank's distinct addresses"""
    tot = 0
    for g in (addrs[:32], addrs[32:]):
        banks = collections.defaultdict(set)
        for a in g:
            banks[a % 32].add(a)
        tot += max(len(v) for v in banks.values())
    return tot  # conflict-free = 2


def analyse(N, TPF, R, F):
    NP = len(R)
    stride = [1]
    for r in R[:-1]:
        stride.append(stride[-1] * r)
    pitch = pitch_for(N, 32 // F if F < 32 else 1)
    out = {"sched": f"<{N},{TPF},{','.join(map(str, R))}>xF{F}", "pitch": pitch, "halves": []}
    tot_c = tot_a = 0
    for p in range(NP - 1):
        # scatter of sub-pass p
        r, st, nb = R[p], stride[p], N // R[p]
        bpt = nb // TPF
        for kind in ("scatter", "gather"):
            if kind == "gather":
                r, nb = R[p + 1], N // R[p + 1]
                bpt = nb // TPF
            hist = collections.Counter()
            for wave in range(F * TPF // 64):
                for m in range(bpt):
                    for k in range(r):
                        addrs = []
                        for lane in range(64):
                            tid = wave * 64 + lane
                            f, u = tid % F, tid // F
                            b = u + m * TPF
                            if kind == "scatter":
                                base = (b // st) * (st * R[p]) + (b % st)
                                i = base + k * st
                            else:
                                i = b + k * nb
                            addrs.append(f * pitch + phys(i))
                        hist[ways(addrs)] += 1
            n_instr = sum(hist.values())
            act = sum(w * c for w, c in hist.items())
            conf = act - 2 * n_instr
            out["halves"].append({"exchange": p, "half": kind, "array_cycles_per_instr_histogram": dict(sorted(hist.items())), "conflict_over_idx_active": round(conf / act, 3)})
            # split exchange: each half runs twice (real and imaginary plane), same addresses
            tot_c += 2 * conf
            tot_a += 2 * act
    out["conflict_over_idx_active_all_exchanges"] = round(tot_c / tot_a, 3)
    return out


if __name__ == "__main__":
    for N, TPF, R, F in ((1024, 32, [8, 8, 16], 16), (2048, 64, [8, 16, 16], 16), (2048, 64, [8, 16, 16], 8), (512, 16, [8, 8, 8], 32)):
        print(json.dumps(analyse(N, TPF, R, F)))


def scan(N, TPF, R, F):
    """the same for every pitch residue mod 32 (the layout's free parameter): is there one without the first-scatter conflict?"""
    global pitch_for
    orig = pitch_for
    res = []
    for mod in range(32):
        pitch_for = lambda n, _m, mod=mod: next(p for p in range(phys(n - 1) + 1, phys(n - 1) + 40) if p % 32 == mod)
        a = analyse(N, TPF, R, F)
        res.append((a["conflict_over_idx_active_all_exchanges"], mod, [h["conflict_over_idx_active"] for h in a["halves"]]))
    pitch_for = orig
    return sorted(res)[:6]
