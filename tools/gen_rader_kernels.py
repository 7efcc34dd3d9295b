#!/usr/bin/env python3
"""Generates rustfft_amd/csrc/kernels_rader_{f32,f64}_{0..3}.hip: a compiled Rader body (kernels.h rader_body /
rader_rows_body) for every prime 17 <= p <= 4096 whose p - 1 is 13-smooth -- the primes the reference plans as
RadersAlgorithm (src/plan.rs:636-665: every prime factor of p - 1 small) get their own instantiation, like 1009 has had
since round 1.  Schedule of the inner length p - 1: tools/gen_smooth_kernels.py (radices <= 16, fewest sub-passes).

Body per prime:
  rows loop (MODE 2 / 4 f32, MODE 3 f64): one workgroup pushes eight rows through one LDS buffer, every per-thread table in
      registers -- when the inner schedule has at least 64 threads per row and the tables fit the register budget (MODE 4:
      up to 256 VGPRs, two waves per SIMD; measured 1.97 TB/s for p = 4051 against 1.44 for its MODE 1 neighbours);
  MODE 1: F rows side by side, scatter on load, one LDS buffer -- the smaller primes;
  MODE 0: staged rows -- when the row pitch leaves no spare slot for x[0] / X[0].
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_smooth_kernels as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NFILES = 4
# Measured choices (profiles/r2/rader_ab1_*.json: every prime through the default choice and through "rows loop wherever it
# can be instantiated", same box, 1 GiB of rows; a prime is listed when the alternative ran > 5 % faster):
# f64 rows-loop bodies with the default schedule (the first five compile without scratch, the others spill 8 .. 100 bytes per
# lane and still win; 1459 and 2801 compiled without scratch but lost 11 - 18 % in the one-process confirmation and are MODE 1 again)
F64_ROWS = {727, 811, 991, 1201, 1297, 1373, 1621, 1783, 1801, 1951, 2081, 2251, 2593, 2663, 3169, 3457, 3697, 4051}
# primes below ~800 whose default schedule has fewer than 64 threads per row: schedule_wide + the rows loop (1.1 - 2.3x)
WIDE_ROWS = ({(32, p) for p in (193, 257, 271, 281, 331, 337, 353, 397, 401, 421, 433, 449, 463, 487, 491, 541, 577, 601, 617, 631, 641,
                                661, 673, 769)} |
             {(64, p) for p in (193, 211, 241, 257, 281, 331, 337, 397, 401, 421, 433, 449, 463, 487, 491, 541, 577, 601, 641, 769)})
SKIP = {(64, 1009)}  # (prec, p): hand-tuned instantiation in kernels_np2_f64.hip
# config 4's prime, Complex<f32>: the rows loop WITHOUT the next-row prefetch (MODE 3, 124 VGPRs: four waves per SIMD) compiled without the
# SLP vectoriser: 4.85 -> 4.35 ms for 2^19 rows (+11 %; with the vectoriser MODE 3 spills at the 128-VGPR cap: 5.88;
# profiles/r4/ab_c4_noslp_variants.jsonl, ab_c4_mode3_variants.jsonl: 16 or 32 rows per workgroup the same, other schedules spill)
FORCE = {(32, 1009): (8, 3, [14, 9, 8], 126)}  # (mode 9 = mode 3 + non-temporal row loads, round 5: +1.3 % at config 4's batch, -2 % at 1 GiB: not shipped)


def is_prime(n):
    return n > 1 and all(n % d for d in range(2, int(n**0.5) + 1))


def layout(n, rad, tpf):
    """(pitch, XS, emax) exactly as engine.h SchedImpl computes them."""
    pow2 = all(r & (r - 1) == 0 for r in rad) and n >= 32
    bpt = [math.ceil((n // r) / tpf) for r in rad]
    emax = max(r * b for r, b in zip(rad, bpt))
    paddiv = rad[0] if (not pow2 and len(rad) > 1 and rad[0] % 2 == 0) else 0

    def phys(i):
        if pow2 and emax > 16:
            return i + i // 32
        if paddiv:
            return i + i // paddiv
        return i

    pitch = phys(n - 1) + 1
    while pitch % 32 != 1:
        pitch += 1
    xs = max(phys(n - 1) + 1, n + 1)  # kernels.h: the x[0] / X[0] slot lies past the exchange span AND past output index p - 1
    twreg = sum(b * (r - 1) for r, b in list(zip(rad, bpt))[1:])
    return pitch, xs, emax, twreg


def schedule_wide(n, e=12):
    """A three-sub-pass schedule with radices <= e and at least 64 threads per row (one butterfly per thread in the widest
    sub-pass): what lets a prime below ~700 run the rows loop.  None if n has no such factorisation."""
    best = None
    for rad in g.factorizations(n):
        if max(rad) > e or len(rad) > 3:
            continue
        t = max(n // r for r in rad)
        if t < 64:
            continue
        util = sum((n // r) / t for r in rad) / len(rad) * t / (math.ceil(t / 64) * 64)
        if best is None or util > best[0]:
            best = (util, sorted(rad, reverse=True), t)
    return (best[1], best[2]) if best else None


# primes with >= 64 threads per row that run faster with one butterfly per thread and the smallest radices (profiles/r2/rader_ab2_*.json,
# same method: 1.1 - 2.2x)
WIDE2_ROWS = ({(32, p) for p in (701, 727, 757, 881, 883, 1297)} |
              {(64, p) for p in (881, 883, 937, 1051, 1153, 1171, 1249, 1321, 1373, 1409, 1471, 1801)})
# primes whose p - 1 has a prime factor 17 .. 31 (schedule31: prime-radix sub-passes): 99 of them <= 4096; as Rader bodies most
# lose to the one-kernel Bluestein (median 0.9x), these win by 7 - 59 % (profiles/r2/rader_ab3_*.json) -- mostly where
# Bluestein has to pad 2p - 1 up to 5120 or 6144
EXTRA31 = ({(32, p) for p in (137, 647, 683, 2089, 2143, 2857)} |
           {(64, p) for p in (613, 2053, 2129, 2281, 2393, 2437, 2531, 2843, 3469, 3571, 3673, 3877, 3911)})
X31_SLP = os.environ.get("RADER_X31_SLP") == "1"  # ... with the SLP vectoriser (the standard units)
X31_M5 = os.environ.get("RADER_X31_M5") == "1"  # ... with the register hand-over (MODE 5)
X31_ALL = os.environ.get("RADER_X31") == "all"  # experiment (round 5): EVERY prime <= 4096 with a 31-smooth p - 1 as a Rader body (f32: in the no-SLP units)
if X31_ALL:
    _s13 = set(g.smooth(4096, [2, 3, 5, 7, 11, 13]))
    _s31 = set(g.smooth(4096, [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31])) - _s13
    EXTRA31 = EXTRA31 | {(pr, p) for pr in (32, 64) for p in range(17, 4097) if (p - 1) in _s31 and all(p % q for q in range(2, int(p ** 0.5) + 1))}
# Round 5: all 99 re-measured (RADER_X31=all build against the shipped library, one process, two runs, profiles/r5/ab_x31_*.jsonl) -- the
# side-by-side bodies batch their row loads since round 3 and the Complex<f32> ones compile without the SLP vectoriser: 90 of 93 / 75 of 86
# now beat the one-kernel Bluestein (median +35 % / +31 %, up to +83 % / +114 %); with the register hand-over (MODE 5, RADER_X31_M5=1) where
# that measured > 3 % faster and, for six f32 primes, WITH the vectoriser (RADER_X31_SLP=1: median -10 %, these +8 .. 19 %) the lists below
# ship (89 / 77 primes; 929, 1217, 3041 -- no side-by-side layout fits, MODE 0 -- lose 28 - 51 % and stay with Bluestein)
EXTRA31_R5 = ({(32, p) for p in (47, 59, 103, 139, 191, 229, 233, 239, 277, 307, 311, 349, 373, 409, 419, 443, 457, 461, 523, 571, 599, 613, 691, 761, 829, 919, 953, 967, 1013, 1021, 1103, 1117, 1123, 1151, 1277, 1289, 1303, 1327, 1361, 1381, 1427, 1429, 1451, 1483, 1531, 1567, 1597, 1613, 1657, 1667, 1741, 1861, 1871, 1901, 1933, 1973, 2053, 2129, 2281, 2347, 2357, 2381, 2393, 2437, 2531, 2551, 2729, 2791, 2843, 2851, 2927, 3037, 3061, 3079, 3163, 3191, 3221, 3307, 3313, 3469, 3571, 3673, 3727, 3877, 3907, 3911, 4003, 4049, 4093)} |
              {(64, p) for p in (47, 59, 103, 137, 139, 191, 229, 239, 277, 307, 311, 349, 373, 409, 419, 443, 457, 523, 571, 599, 647, 691, 761, 829, 919, 953, 967, 1013, 1021, 1103, 1117, 1123, 1151, 1277, 1289, 1327, 1361, 1381, 1427, 1429, 1451, 1483, 1531, 1567, 1597, 1613, 1657, 1667, 1741, 1861, 1871, 1901, 1933, 1973, 2089, 2143, 2347, 2357, 2381, 2551, 2729, 2791, 2851, 2857, 2927, 3037, 3061, 3079, 3163, 3191, 3221, 3307, 3313, 3727, 3907, 4049, 4093)})
X31_SLP_F32 = {59, 523, 1117, 1451, 1741}  # the new Complex<f32> bodies that keep the SLP vectoriser (the standard units)
EXTRA31_R2 = set(EXTRA31)
EXTRA31 = EXTRA31 | EXTRA31_R5
# (the sweep listed 14 / 17; the Bluestein bodies were then rescheduled -- 1280 and 5120 run 22 % faster -- and the one-process
# confirmation, profiles/r2/rader_choices_confirm_*.jsonl, kept the 6 / 13 that still win by > 3 %)
# f32 bodies whose tables spill at MODE 2's 168 VGPRs and run 5 - 23 % faster at MODE 4's 256 (profiles/r2/rader_ab4_*.json; 1301
# is the one spiller that loses 24 % and stays; the same run: the rows loop for the EXTRA31 primes loses 5 - 60 %)
MODE4_F32 = {991, 1453, 2179, 2917, 2971, 4051}
F_TARGET = int(os.environ.get("RADER_F_TARGET", "256"))  # experiment knob: threads per workgroup of the rows-side-by-side bodies
# (prec, p) -> rows per workgroup of the rows-side-by-side bodies (MODE 1) where half the rule's count ran > 4 % faster in a one-process
# interleaved A/B of every prime (tools/ab_lengths.py --set primes, profiles/r2/rader_mode1_rows_ab_*.jsonl; twice the count loses 13 - 20 %)
MODE1_ROWS = {(32, 37): 42, (32, 41): 32, (32, 43): 42, (32, 53): 32, (32, 61): 32, (32, 67): 21, (32, 73): 21, (32, 79): 18, (32, 113): 16, (32, 127): 9, (32, 197): 9, (32, 379): 3, (32, 521): 3, (32, 547): 3, (32, 677): 2, (64, 37): 42, (64, 41): 32, (64, 43): 42, (64, 53): 32, (64, 61): 32, (64, 67): 21, (64, 71): 25, (64, 73): 21, (64, 79): 18, (64, 97): 16, (64, 109): 10, (64, 113): 16, (64, 163): 7, (64, 197): 9, (64, 199): 7, (64, 271): 4, (64, 521): 3, (64, 547): 3, (64, 617): 2, (64, 661): 2}
# (prec, p) -> back from the rows loop to rows side by side: the primes where the side-by-side body with batched row loads beat the
# rows loop by > 4 % in a one-process A/B of every Rader prime (RADER_ALT=3 build; profiles/r3/rader_mode1_back_*.jsonl)
MODE1_BACK = ({(32, p) for p in (193, 271, 281, 331, 337, 353, 397, 463, 751, 859, 1093, 1601, 1621, 1951, 2003, 2029, 2113, 2647, 2801, 2917, 2971, 3001, 3511, 3851, 4001, 4051, 4057)} |
              {(64, p) for p in (193, 211, 241, 331, 337, 397, 401, 433, 463, 487, 491, 541, 577, 641, 727, 769, 811, 1051, 1153, 1171, 1249, 1297, 1321, 1373, 1471, 1621, 1783, 1801, 1951, 2251, 2593, 2663, 3169, 3697)})
ALT = os.environ.get("RADER_ALT") == "1"  # experiment 1: the rows loop wherever it can be instantiated (A/B against the default choice)
ALT2 = os.environ.get("RADER_ALT") == "2"  # experiment 2: one butterfly per thread, smallest radices, for the primes with >= 64 threads per row
ALT3 = os.environ.get("RADER_ALT") == "3"  # experiment 3 (round 3, after the batched row loads made the side-by-side bodies 30 - 80 % faster): rows side by side wherever the layout allows


def choose(p, prec):
    n = p - 1
    if (prec, p) in FORCE:
        f, mode, rad, tpf = FORCE[(prec, p)]
        return (f, mode, list(rad), tpf)
    if (ALT3 or (prec, p) in MODE1_BACK) and (prec, p) not in EXTRA31:
        rad, tpf = g.schedule(n)
        pitch, xs, emax, twreg = layout(n, rad, tpf)
        esz = 8 if prec == 32 else 16
        f = max(1, min(F_TARGET // tpf, (60 * 1024) // (pitch * esz)))
        if xs < pitch and p <= pitch and pitch * esz * f <= 64 * 1024:
            return (MODE1_ROWS.get((prec, p), f), 1, rad, tpf)
    if (prec, p) in EXTRA31:
        rad, tpf = g.schedule31(n)
        pitch, xs, emax, twreg = layout(n, rad, tpf)
        esz = 8 if prec == 32 else 16
        f = max(1, min(256 // tpf, (60 * 1024) // (pitch * esz)))
        if xs < pitch and p <= pitch:
            return (f, 1, rad, tpf)
        return (max(1, min(256 // tpf, (60 * 1024) // ((pitch + p) * esz))), 0, rad, tpf)
    rad, tpf = g.schedule(n)
    wide = (prec, p) in WIDE_ROWS
    if (ALT or wide) and tpf < 64 and schedule_wide(n):
        rad, tpf = schedule_wide(n)
    wide2 = (prec, p) in WIDE2_ROWS
    if (ALT2 or wide2) and tpf >= 64:
        for e in (12, 13, 14, 15, 16):
            w = schedule_wide(n, e)
            if w and w[1] <= 512:
                rad, tpf = w
                break
    pitch, xs, emax, twreg = layout(n, rad, tpf)
    esz = 8 if prec == 32 else 16
    nl = math.ceil(p / tpf)
    nreg = 3 * emax + twreg + 2 * nl
    # (the rows loop sizes its one row buffer itself -- kernels.h RaderRows::SLOTS -- so the pitch rounding does not matter here)
    if prec == 32 and tpf >= 64 and nreg <= 80:
        return (8, 4 if p in MODE4_F32 else 2, rad, tpf)
    # f64 (MODE 3, no prefetch): 256 VGPRs at two waves per SIMD hold the per-thread tables of few schedules -- the ones listed
    # compile without scratch (hipcc -Rpass-analysis=kernel-resource-usage over all 68 candidates with >= 64 threads per row;
    # the others spill 8 .. 220 bytes per lane, mostly in the radix-11 / 13 / 15 butterflies, and stay MODE 1)
    if prec == 64 and (p in F64_ROWS or wide or wide2 or (ALT and tpf >= 64) or (ALT2 and tpf >= 64 and (rad, tpf) != g.schedule(n))):
        return (8, 3, rad, tpf)
    if prec == 32 and tpf >= 64 and nreg <= 118:  # tables up to 256 VGPRs, two waves per SIMD
        return (8, 4, rad, tpf)
    f = max(1, min(F_TARGET // tpf, (60 * 1024) // (pitch * esz)))
    if (prec, p) in MODE1_ROWS:
        f = MODE1_ROWS[(prec, p)]
    if xs < pitch and p <= pitch:
        return (f, 1, rad, tpf)
    f = max(1, min(256 // tpf, (60 * 1024) // ((pitch + p) * esz)))
    return (f, 0, rad, tpf)


# Complex<f32> rows loops that run 3 - 14 % faster WITHOUT the next-row prefetch (MODE 3: 128-VGPR cap, four waves per SIMD) once they are
# compiled without the SLP vectoriser (RADER_ALT=6 build against the shipped choice, every Rader prime in one process,
# profiles/r4/rader_mode3_noslp_ab_f32.jsonl: the other 60 rows loops spill at that cap and lose, median -30 %)
MODE3_F32 = {401, 421, 433, 487, 541, 601, 641, 769, 811, 881, 883, 991, 1297}
ALT6 = os.environ.get("RADER_ALT") == "6"  # experiment 6 (round 4): every Complex<f32> rows loop (MODE 2 / 4) as MODE 3 in the no-SLP units
ALT5 = os.environ.get("RADER_ALT") == "5"  # experiment 5: every side-by-side body with the register hand-over (MODE 5)
# (prec, p) -> MODE 5 instead of MODE 1 where the hand-over measured > 4 % faster in a one-process A/B of every side-by-side body
# (RADER_ALT=5 build; profiles/r3/rader_mode5_ab_*.jsonl: f32 median +3.5 %, -36 .. +22 %; f64 median +2.5 %, -33 .. +50 %; a second
# pass over the primes still on MODE 1, rader_mode5_ab2_*.jsonl, added 2 f32 / 13 f64 more at +4 .. 9 %)
MODE5 = ({(32, p) for p in (37, 41, 43, 53, 67, 71, 101, 109, 127, 131, 137, 151, 199, 281, 331, 521, 677, 751, 859, 2003, 2143, 2647, 2801, 2857, 2917, 2971, 3001, 3851, 4001, 4057)} |
         {(64, p) for p in (41, 43, 53, 71, 113, 127, 131, 193, 197, 211, 251, 331, 337, 379, 397, 463, 487, 491, 521, 541, 547, 631, 641, 701, 751, 769, 911, 1249, 1321, 1601, 1621, 1801, 1951, 2003, 2113, 2251, 2281, 2311, 2377, 2549, 2647, 2689, 2731, 2801, 2861, 2917, 2971, 3001, 3121, 3169, 3251, 3329, 3389, 3529, 3631, 3697, 3851, 4001)})
# round 5, the 31-smooth primes (profiles/r5/ab_x31_m5_*.jsonl: > 3 % in both runs; family median +2.5 % / -0.8 %)
MODE5 = MODE5 | {(32, p) for p in (103, 139, 229, 239, 311, 349, 443, 571, 691, 829, 953, 967, 1013, 1021, 1123, 1151, 1289, 1361, 1381, 1429, 1871, 1933, 2053, 2129, 2347, 2437, 2531, 2927, 3037, 3061, 3191, 3221, 3469, 3571, 3727, 3877, 4049)} | {(64, p) for p in (137, 191, 229, 277, 311, 409, 457, 571, 761, 1103, 1361, 1597, 2089, 2143, 2357, 2551, 2851, 3163, 3313)}


# Complex<f32> bodies that run >= 3 % faster compiled WITHOUT the SLP vectoriser (one-process A/B of two builds over every prime,
# profiles/r4/ab_noslp_primes_f32.jsonl: +3 ... +29 %, mostly radix-11 rows loops; the 256-VGPR rows loops lose 12 - 21 % without it and
# the family median is -1 %): they go into their own translation units, which the Makefile compiles with -fno-slp-vectorize.
NOSLP_F32 = {1009, 89, 353, 463, 617, 631, 661, 673, 701, 727, 757, 859, 881, 991, 2029, 2143, 2179, 2269, 2647, 2801, 2857, 3511, 3851, 4057}
NS_FILES = 2


def main():
    s13 = set(g.smooth(4096, [2, 3, 5, 7, 11, 13]))
    primes13 = [p for p in range(17, 4097) if is_prime(p) and (p - 1) in s13]
    for tag, ty, prec in (("f32", "float", 32), ("f64", "double", 64)):
        modes = {}
        primes = sorted([p for p in primes13 if (prec, p) not in SKIP] + [p for (pr, p) in EXTRA31 if pr == prec])
        alt6 = {p for p in primes if (ALT6 or p in MODE3_F32) and prec == 32 and choose(p, prec)[1] in (2, 4)}
        noslp = [p for p in primes if prec == 32 and (p in NOSLP_F32 or p in alt6 or (not X31_SLP and (32, p) in EXTRA31 and (32, p) not in EXTRA31_R2 and p not in X31_SLP_F32))]
        primes = [p for p in primes if p not in noslp]
        units = [(str(ci), primes[ci::NFILES], "") for ci in range(NFILES)]
        if noslp:
            units += [(f"ns{ci}", noslp[ci::NS_FILES], " (the bodies that win without the SLP vectoriser: compiled with -fno-slp-vectorize)") for ci in range(NS_FILES)]
        for name, plist, note in units:
            lines = []
            for p in plist:
                f, mode, rad, tpf = choose(p, prec)
                if mode == 1 and (ALT5 or (prec, p) in MODE5 or (X31_M5 and (prec, p) in EXTRA31 and (prec, p) not in EXTRA31_R2)) and len(rad) >= 2:
                    mode = 5
                if p in alt6:
                    mode = 3
                modes[mode] = modes.get(mode, 0) + 1
                lines.append(f"    MI_RADER({ty}, {prec}, {f}, {mode}, {p - 1}, {tpf}, {', '.join(map(str, rad))});  // p = {p}")
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_rader_{tag}_{name}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_rader_kernels.py — do not edit.  Compiled Rader bodies for the primes <= 4096 whose p - 1 is\n"
                         f"// 13-smooth, and the few with a factor 17 .. 31 that beat Bluestein (unit {name}){note}, Complex<{ty}>.\n"
                         + ("#define MI355_PK_CMUL 1\n" if prec == 32 else "") +
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_rader_{tag}_{name}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(tag, len(primes) + len(noslp), "primes; bodies by mode:", modes)


if __name__ == "__main__":
    main()
