#!/usr/bin/env python3
"""Generates rustfft_amd/csrc/kernels_smooth_{f32,f64}_{0..7}.hip: one compiled K1 schedule for every 13-smooth length
in [3, 4096] that is not a power of two (476 lengths).  Compiled schedules run at the HBM ceiling (N = 1200: 4.5 TB/s)
where the run-time scheduled kernel needs ~150 VGPRs and reaches 1.0 - 1.8 TB/s, so the common mixed-radix lengths —
the reference's RadixN territory, src/plan.rs:508-607 — get their own instantiation, like the reference's planner gives
them their own recipe.

Schedule choice per length: radices <= 16 (any composite the in-register butterfly template supports), fewest
sub-passes first, then the best lane utilisation under the 16-values-per-thread register budget.
"""
import itertools
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Sub-pass 1's twiddle table staged in LDS (engine.h TWL; kernel name suffix "t1") -- a per-length measured choice, round 3:
# one-process interleaved A/B of EVERY compiled length, shipped build vs a build with the table staged everywhere
# (profiles/r3/ab_smooth_twl1_all_{f32,f64}.jsonl): f32 median +11 % (1162 of 1251 lengths win by > 2 %, 50 do not win and are
# listed as exceptions), f64 median -1 % (129 lengths win by >= 4 % and are listed).
try:
    import json as _json

    _TWL = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smooth_twl_choice.json")))
except OSError:
    _TWL = {"f32_exclude": [], "f64_include": []}
_TWL32_EXCLUDE, _TWL64_INCLUDE = set(_TWL["f32_exclude"]), set(_TWL["f64_include"])


# Non-temporal row accesses (kernels.h k1_body ABL bits 16 / 32; kernel name suffix "n" = loads, "nn" = loads + stores) -- a per-length measured
# choice, round 5: every compiled length above 512, shipped build against builds with the hint everywhere (tools/r5/build_smooth_nt.sh), two
# one-process runs (profiles/r5/ab_smooth_nt{16,48}_{f32,f64}_rep{1,2}.jsonl).  The median over all lengths is 1.00 (loads) / 0.93 (both) --
# short rows lose 20 %, long rows gain -- so only the lengths that gain >= 2.5 % (loads) / >= 3 % (both, and more than loads alone) in BOTH runs
# are listed: f32 404 + 27 lengths (median +3.9 % / +5.7 %), f64 159 + 61 (+3.9 % / +5.9 %).
try:
    _NT = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smooth_nt_choice.json")))
except OSError:
    _NT = {}
_NT_LOADS = {32: set(_NT.get("f32_loads", [])), 64: set(_NT.get("f64_loads", []))}
_NT_BOTH = {32: set(_NT.get("f32_both", [])), 64: set(_NT.get("f64_both", []))}


def k1_line(ty, prec, f, split, n, tpf, rad):
    staged = len(rad) >= 2 and ((prec == 32 and n not in _TWL32_EXCLUDE) or (prec == 64 and n in _TWL64_INCLUDE))
    args = f"{n}, {tpf}, {', '.join(map(str, rad))}"
    nt, suf = (48, "nn") if n in _NT_BOTH[prec] else (16, "n") if n in _NT_LOADS[prec] else (0, "")
    if staged:
        return f'    MI_K1X({ty}, {prec}, {f}, {split}, {1024 | nt}, "t1{suf}", {args});'
    if nt:
        return f'    MI_K1X({ty}, {prec}, {f}, {split}, {nt}, "{suf}", {args});'
    return f"    MI_K1({ty}, {prec}, {f}, {split}, {args});"

RADICES = [16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2]
EMAX = 16
NFILES = 8


def smooth(limit, primes):
    s = {1}
    for p in primes:
        s2 = set()
        for v in s:
            x = v
            while x <= limit:
                s2.add(x)
                x *= p
        s = s2
    return sorted(s)


def factorizations(n, max_r=16, start=0):
    if n == 1:
        yield []
        return
    for i in range(start, len(RADICES)):
        r = RADICES[i]
        if n % r == 0:
            for rest in factorizations(n // r, max_r, i):
                yield [r] + rest


def schedule(n, emax=EMAX, max_passes=5):
    best = None
    for rad in factorizations(n):
        if len(rad) > max_passes:
            continue
        tpf = max(math.ceil((n // r) / (emax // r)) for r in rad)
        # lane utilisation: useful butterfly slots / allotted slots, averaged over the sub-passes
        util = sum((n // r) / (tpf * math.ceil((n // r) / tpf)) for r in rad) / len(rad)
        waves = math.ceil(tpf / 64) * 64
        util *= tpf / waves if tpf >= 64 else 1.0
        key = (len(rad), -util)
        if best is None or key < best[0]:
            best = (key, rad, tpf)
    _, rad, tpf = best
    # order: the largest radix first (the untwiddled sub-pass), the rest descending
    rad = sorted(rad, reverse=True)
    return rad, tpf


def schedule_small_radix(n):
    """Experiment / measured alternative: as many sub-passes as schedule(n), the smallest largest-radix, one butterfly per
    thread in the widest sub-pass (what the Rader family's A/B favoured for the VALU-heavy radices 11 .. 16)."""
    rad0, tpf0 = schedule(n)
    best = None
    for rad in factorizations(n):
        if len(rad) != len(rad0):
            continue
        tpf = max(n // r for r in rad)
        if tpf > 512:
            continue
        util = sum((n // r) / tpf for r in rad) / len(rad) * (tpf / (math.ceil(tpf / 64) * 64) if tpf >= 64 else 1.0)
        key = (max(rad), -util)
        if best is None or key < best[0]:
            best = (key, sorted(rad, reverse=True), tpf)
    return (best[1], best[2]) if best else (rad0, tpf0)


SCHED_ALT = os.environ.get("SMOOTH_SCHED_ALT") == "1"
ROWS31_TARGET = int(os.environ.get("SMOOTH_ROWS31_TARGET", "256"))  # experiment knob for the prime-radix schedules (main_primes)
ROWS_TARGET = int(os.environ.get("SMOOTH_ROWS_TARGET", "256"))  # experiment knob: threads per workgroup the row count aims at


# rows per workgroup that measured > 4 % faster than the 256-thread rule on MI355X (profiles/r2/smooth_rows_ab_*.json: every
# length with the 128-, 256- and 512-thread rule, same box, 1 GiB of rows; no single rule wins, so the winners are listed)
_CHOICES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smooth_rows_choice.json")))
ROWS_CHOICE = _CHOICES["rows"]


# lengths that take schedule_small_radix: 19 f32 / 27 f64 of the 295 / 263 lengths where it differs ran > 4 % faster with it
# (profiles/r2/smooth_sched_ab_*.json; the median over all of them is 0.91x / 0.99x -- the default rule stays the rule)
SCHED_CHOICE = {k: {str(n) for n in v} for k, v in _CHOICES["small_radix"].items()}


def rows_per_wg(n, tpf, esz):
    tag = "f32" if esz == 8 else "f64"
    if ROWS_TARGET == 256 and not SCHED_ALT and str(n) in ROWS_CHOICE[tag] and str(n) not in SCHED_CHOICE[tag]:
        return ROWS_CHOICE["f32" if esz == 8 else "f64"][str(n)]
    f = max(1, round(ROWS_TARGET / tpf))
    pitch = n + n // 8 + 2
    while f > 1 and (f * pitch * esz > 60 * 1024 or f * tpf > max(512, ROWS_TARGET)):
        f -= 1
    return f


RADICES_BIG = [32, 30, 28, 27, 25, 24, 21, 20, 18, 16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2]


# the same with the multiples of 11 and 13 (round 5: the 13-smooth lengths above 16384; a radix-26 butterfly is one in-register step 2 x 13)
RADICES_BIG13 = sorted(set(RADICES_BIG + [26, 22, 13, 11]), reverse=True)


def factorizations_big(n, start=0, radices=RADICES_BIG):
    if n == 1:
        yield []
        return
    for i in range(start, len(radices)):
        r = radices[i]
        if n % r == 0:
            for rest in factorizations_big(n // r, i, radices):
                yield [r] + rest


def big_schedule32(n, radices=RADICES_BIG):
    """f32, 32 values per thread, radices up to 32: fewest sub-passes (each one is an LDS round trip and two barriers)."""
    best = None
    for rad in factorizations_big(n, 0, radices):
        if len(rad) > 4:
            continue
        tpf = max(math.ceil((n // r) / (32 // r)) for r in rad)
        if tpf > 1024:
            continue
        util = sum((n // r) / (tpf * math.ceil((n // r) / tpf)) for r in rad) / len(rad)
        waves = math.ceil(tpf / 64) * 64
        util *= tpf / waves
        key = (len(rad), -util)
        if best is None or key < best[0]:
            best = (key, sorted(rad, reverse=True), tpf)
    return (best[1], best[2]) if best else None


# f32 lengths for which the radix-32 rule measured > 20 % faster on MI355X than the 16-values-per-thread rule (A/B over all
# 150 lengths, 1 GiB of rows each; the other 78 lengths are up to 37 % slower with it: odd thread counts, radices 25 - 30)
BIG32_WINS = {4375, 4608, 4725, 5103, 5292, 5400, 5760, 5832, 6125, 6300, 7203, 7776, 7938, 8000, 8400, 8575, 9072, 9261, 9720,
              10584, 11200, 11760, 11907, 12000, 12500, 12544, 12800, 13230, 13824, 14112, 14406, 15309, 15435, 15552,
              15625, 16000}


BIG32_OFF = os.environ.get("SMOOTH_BIG32_OFF") == "1"  # experiment: the 16-values-per-thread rule for every length <= 16384


def big_schedule(n, emaxes=(16, 32)):
    if 32 in emaxes and n in BIG32_WINS and not BIG32_OFF:
        r = big_schedule32(n)
        if r:
            return r
    """(4096, 16384]: one workgroup per transform, up to 1024 threads; 16 values per thread, 32 where that needs more threads
    (f32 only: 32 double-precision values per thread spill)."""
    for emax in emaxes:
        try:
            rad, tpf = schedule(n, emax, 6)
        except TypeError:
            continue
        if tpf <= 1024:
            return rad, tpf
    return None


# Complex<f32> lengths whose kernel runs >= 2 % faster in BOTH of two one-process A/B runs when compiled WITHOUT the SLP vectoriser
# (profiles/r4/ab_noslp_{smooth,smooth3,smooth2}_f32_rep{1,2}.jsonl: the vectoriser pairs re / im parts of different values into packed
# operations and pays in register moves -- 62 of 476 13-smooth lengths, 280 of 563 prime-radix lengths, 147 of 223 large 7-smooth ones, median
# gain 5 / 8 / 10 %, up to 76 %).  They go into their own translation units ("ns"), which the Makefile compiles with -fno-slp-vectorize.
_NOSLP = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smooth_noslp_choice.json")))
NS_UNITS = {"smooth": 1, "smooth2": 3, "smooth3": 7, "smooth4": 10}


def units_of(family, tag, sizes, nfiles):
    """[(unit name, lengths)]: the numbered units hold what stays with the vectoriser, the "ns" units (Complex<f32> only) the rest."""
    ns = sorted(set(_NOSLP.get(family, [])) & set(sizes)) if tag == "f32" else []
    keep = [x for x in sizes if x not in set(ns)]
    out = [(str(ci), keep[ci::nfiles]) for ci in range(nfiles)]
    if ns:
        out += [(f"ns{ci}", ns[ci::NS_UNITS[family]]) for ci in range(NS_UNITS[family])]
    return out


def main_big():
    """kernels_smooth2_*: 7-smooth lengths in (4096, 16384] as single split-exchange kernels (one HBM pass instead of two)."""
    nfiles = 4
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        emaxes = (16, 32) if prec == 32 else (16,)
        sizes = [x for x in smooth(16384, [2, 3, 5, 7]) if x > 4096 and (x & (x - 1)) and big_schedule(x, emaxes)]
        if prec == 32:  # f32 rows up to 32768 points still fit one workgroup's LDS through the split exchange (<= 132 KB)
            sizes += [x for x in smooth(32768, [2, 3, 5, 7]) if x > 16384 and (x & (x - 1)) and big_schedule32(x)]
        for ci, chunk in units_of("smooth2", tag, sizes, nfiles):
            lines = []
            for n in chunk:
                rad, tpf = big_schedule(n, emaxes) if n <= 16384 else big_schedule32(n)
                lines.append(k1_line(ty, prec, 1, "true", n, tpf, rad))
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_smooth2_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_smooth_kernels.py — do not edit.  Single-kernel schedules (split exchange) for the 7-smooth\n"
                         f"// lengths in (4096, 16384] (unit {ci}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_smooth2_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(len(sizes), "lengths in (4096, 16384] (f32: 32768],", tag)


def big13_sizes(emaxes=(16,)):
    """13-smooth lengths in (4096, 16384] with a factor 11 or 13 (the 7-smooth ones are main_big's).  Complex<f64>: the 173 of the 264 that
    run on at most 1024 threads with 16 values per thread (32 double-precision values per thread spill)."""
    s7 = set(smooth(16384, [2, 3, 5, 7]))
    return [x for x in smooth(16384, [2, 3, 5, 7, 11, 13]) if x > 4096 and x not in s7 and big_schedule(x, emaxes)]


def big13_sizes32():
    """Complex<f32> only: the 13-smooth lengths in (16384, 32768] with a factor 11 or 13 that have a schedule of at most four sub-passes on at
    most 1024 threads with 32 values per thread (101 of 198), minus the ones that measured slower than their two-pass plan (BIG13_32_LOSERS)."""
    s7 = set(smooth(32768, [2, 3, 5, 7]))
    return [x for x in smooth(32768, [2, 3, 5, 7, 11, 13]) if x > 16384 and x not in s7 and x not in BIG13_32_LOSERS and big_schedule32(x, RADICES_BIG13)]


BIG13_32_LOSERS = set()


def main_big13():
    """kernels_smooth4_*: the 13-smooth lengths in (4096, 16384] that have a factor 11 or 13 as single split-exchange kernels (round 5).
    Until then they ran as two general column-tile passes (k2g_body) -- four HBM crossings per transform and partially filled 64-column
    tiles: 2.0 - 2.5 TB/s against 4 - 5 TB/s for their 7-smooth neighbours, which had whole-row kernels (gpurun_out r5_01, the reference
    plans both kinds the same way: RadixN / MixedRadix over butterflies, src/plan.rs:508-607).  16 values per thread (the radix-11 / 13
    butterflies: butterflies.h), up to 1024 threads."""
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        emaxes = (16, 32) if prec == 32 else (16,)
        sizes = big13_sizes(emaxes) + (big13_sizes32() if prec == 32 else [])
        # Complex<f32>: 302 of the 365 lengths measured >= 2 % faster in both of two runs without the SLP vectoriser (median +11 %:
        # 3.14 -> 3.86 TB/s; profiles/r5/ab_smooth4_noslp_f32_rep*.jsonl) and sit in the ten "ns" units, the rest in two
        nfiles = 2 if prec == 32 else 8
        for ci, chunk in units_of("smooth4", tag, sizes, nfiles):
            lines = []
            for n in chunk:
                rad, tpf = big_schedule(n, emaxes) if n <= 16384 else big_schedule32(n, RADICES_BIG13)
                lines.append(k1_line(ty, prec, 1, "true", n, tpf, rad))
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_smooth4_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_smooth_kernels.py — do not edit.  Single-kernel schedules (split exchange) for the 13-smooth\n"
                         f"// lengths in (4096, 16384] with a factor 11 or 13 (unit {ci}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_smooth4_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(len(sizes), "13-smooth lengths with a factor 11 / 13 in (4096, 16384] (f32: 32768],", tag)


BIG_PRIMES = [31, 29, 23, 19, 17]


def schedule31(n, emax=32, max_passes=5):
    """Lengths with a prime factor 17 .. 31 (the reference's Butterfly17 .. Butterfly31, src/algorithm/butterflies.rs:1582-6241):
    the prime radices join the set, 32 values per thread so that one 31-point butterfly fits a thread."""
    global RADICES
    saved = RADICES
    RADICES = BIG_PRIMES + saved
    try:
        return schedule(n, emax, max_passes)
    finally:
        RADICES = saved


def main_primes(limits=(("f32", "float", 32, 8, 4096, 14), ("f64", "double", 64, 16, 2048, 8))):
    """kernels_smooth3_*: compiled schedules for the 31-smooth lengths <= limit that have a prime factor in 17 .. 31
    (measured on MI355X: 4.0 - 5.3 TB/s against 1.2 - 2.0 through Bluestein and 0.5 - 1.0 through the run-time scheduled
    kernel; f32 up to 4096, f64 up to 2048 -- the build time is what bounds the f64 set: ~1.7 s of hipcc per kernel)."""
    for tag, ty, prec, esz, limit, nfiles in limits:
        s13 = set(smooth(limit, [2, 3, 5, 7, 11, 13]))
        sizes = [x for x in smooth(limit, [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31]) if x not in s13]
        for ci, chunk in units_of("smooth3", tag, sizes, nfiles):
            lines = []
            for n in chunk:
                rad, tpf = schedule31(n)
                f = max(1, min(ROWS31_TARGET // tpf, (48 * 1024) // ((n + n // 8 + 2) * esz)))  # <= 256 threads: the prime butterflies want > 128 VGPRs
                # half the rows where a one-process interleaved A/B of every length measured > 4 % (profiles/r2/prime_radix_rows_ab_*.jsonl:
                # 113 of 389 changed f32 lengths, 36 of 89 f64; the median over all of them is 0.99 / 1.02, so the rule stays)
                if ROWS31_TARGET == 256 and str(n) in _CHOICES.get("rows31", {}).get(tag, {}):
                    f = _CHOICES["rows31"][tag][str(n)]
                lines.append(k1_line(ty, prec, f, "false", n, tpf, rad))
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_smooth3_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_smooth_kernels.py — do not edit.  Compiled K1 schedules for the lengths <= {limit} with a prime\n"
                         f"// factor 17 .. 31 (in-register prime butterflies; unit {ci}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_smooth3_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(len(sizes), "lengths with a factor 17 .. 31, <=", limit, tag)


RADICES_BIG31 = sorted(set(RADICES_BIG13 + BIG_PRIMES), reverse=True)
S5_LIMIT = int(os.environ.get("S5_LIMIT", "16384"))  # (8192 in the first cut: profiles/r5/ab_smooth5_vs_bluestein_*; the tier above it added after that measurement)
S5_FILES = 8


def big31_sizes(prec):
    """Round 5, late: the lengths with a prime factor 17 .. 31 ABOVE main_primes' limits -- Complex<f32> in (4096, 8192] (split exchange, 32 values
    per thread, at most four sub-passes on at most 1024 threads), Complex<f64> in (2048, 4096] (the plain exchange, schedule31 as below 2048) and
    in (4096, 8192] (split exchange).  They ran through Bluestein (inner length 5120 .. 16384: 0.15 - 0.2 of 8 TB/s)."""
    s13 = set(smooth(S5_LIMIT, [2, 3, 5, 7, 11, 13]))
    s31 = [x for x in smooth(S5_LIMIT, [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31]) if x not in s13]
    if prec == 32:
        return [x for x in s31 if x > 4096 and big_schedule32(x, RADICES_BIG31)]
    out = []
    for x in s31:
        if x <= 2048:
            continue
        r = schedule31(x, 32, 6)
        if r and r[1] <= 1024:
            out.append(x)
    return out


def main_primes_big():
    """kernels_smooth5_*: see big31_sizes.  Complex<f32>: every unit is compiled without the SLP vectoriser (the prime-radix and the 32-value
    kernels are the ones that lose most with it: smooth3 / smooth4 measurements)."""
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        sizes = big31_sizes(prec)
        for ci in range(S5_FILES):
            lines = []
            for n in sizes[ci::S5_FILES]:
                if prec == 32:
                    rad, tpf = big_schedule32(n, RADICES_BIG31)
                else:
                    rad, tpf = schedule31(n, 32, 6)
                lines.append(k1_line(ty, prec, 1, "true" if n > 4096 else "false", n, tpf, rad))
            unit = f"ns{ci}" if prec == 32 else str(ci)
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_smooth5_{tag}_{unit}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_smooth_kernels.py — do not edit.  Single-kernel schedules for the lengths with a prime factor 17 .. 31\n"
                         f"// in (4096, 8192] (Complex<f64>: (2048, 8192]; split exchange above 4096; unit {unit}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_smooth5_{tag}_{unit}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(len(sizes), "lengths with a factor 17 .. 31 above the smooth3 limits, <=", S5_LIMIT, tag)


def main():
    main_big()
    main_big13()
    main_primes()
    main_primes_big()
    sizes = [x for x in smooth(4096, [2, 3, 5, 7, 11, 13]) if x > 2 and (x & (x - 1)) and x != 1200]
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        for ci, chunk in units_of("smooth", tag, sizes, NFILES):
            lines = []
            for n in chunk:
                rad, tpf = schedule(n)
                if SCHED_ALT or str(n) in SCHED_CHOICE[tag]:
                    rad, tpf = schedule_small_radix(n)
                f = rows_per_wg(n, tpf, esz)
                lines.append(k1_line(ty, prec, f, "false", n, tpf, rad))
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_smooth_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_smooth_kernels.py — do not edit.  Compiled K1 schedules for the 13-smooth lengths in\n"
                         f"// (16, 4096] (unit {ci}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_smooth_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
    print(len(sizes), "lengths per precision")


if __name__ == "__main__":
    main()
