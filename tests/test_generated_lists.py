"""The generated kernel lists in rustfft_amd/csrc/ are what their generators produce today: a choice list edited in
tools/gen_rader_kernels.py without regenerating the .hip files (or the other way round) would ship kernels the measurements
behind the lists do not describe."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_rader_lists_match_the_generator():
    import gen_rader_kernels as gen

    assert not (gen.ALT or gen.ALT2 or gen.ALT3 or gen.ALT5 or gen.ALT6 or gen.X31_ALL or gen.X31_SLP or gen.X31_M5), "RADER_ALT / RADER_X31* must not be set while testing"
    s13 = set(gen.g.smooth(4096, [2, 3, 5, 7, 11, 13]))
    primes13 = [p for p in range(17, 4097) if gen.is_prime(p) and (p - 1) in s13]
    for tag, ty, prec in (("f32", "float", 32), ("f64", "double", 64)):
        primes = sorted([p for p in primes13 if (prec, p) not in gen.SKIP] + [p for (pr, p) in gen.EXTRA31 if pr == prec])
        have = {}
        units = [str(ci) for ci in range(gen.NFILES)] + ([f"ns{ci}" for ci in range(gen.NS_FILES)] if prec == 32 else [])
        for unit in units:
            text = open(os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_rader_{tag}_{unit}.hip")).read()
            for m in re.finditer(r"MI_RADER\((\w+), (\d+), (\d+), (\d+), ([\d, ]+)\);\s*// p = (\d+)", text):
                assert m.group(1) == ty and int(m.group(2)) == prec
                assert int(m.group(6)) not in have, "a prime has one body"
                have[int(m.group(6))] = (int(m.group(3)), int(m.group(4)), [int(v) for v in m.group(5).split(",")])
                # the bodies measured faster without the SLP vectoriser sit in the units the Makefile compiles with -fno-slp-vectorize
                # (round 5: so do the 31-smooth primes taken from Bluestein, but for the few that measured faster with the vectoriser)
                q = int(m.group(6))
                x31_noslp = (32, q) in gen.EXTRA31_R5 and (32, q) not in gen.EXTRA31_R2 and q not in gen.X31_SLP_F32
                assert (prec == 32 and (q in gen.NOSLP_F32 or q in gen.MODE3_F32 or x31_noslp)) == unit.startswith("ns"), (tag, unit, m.group(6))
        assert sorted(have) == primes, (tag, sorted(set(primes) ^ set(have)))
        mk = open(os.path.join(ROOT, "rustfft_amd", "csrc", "Makefile")).read()
        for unit in units:
            assert f"kernels_rader_{tag}_{unit}.o" in mk, unit
            assert (f"kernels_rader_{tag}_{unit}" in mk.split("NOSLP :=")[1].split("\n")[0]) == unit.startswith("ns")
        for p in primes:
            f, mode, rad, tpf = gen.choose(p, prec)
            if mode == 1 and (prec, p) in gen.MODE5 and len(rad) >= 2:
                mode = 5
            if prec == 32 and p in gen.MODE3_F32 and mode in (2, 4):
                mode = 3
            assert have[p] == (f, mode, [p - 1, tpf] + list(rad)), (tag, p, have[p], (f, mode, rad, tpf))
    # the measured lists only name primes that have a body, and a prime is in one list of a kind at most
    for prec in (32, 64):
        for name in ("MODE1_BACK", "MODE5"):
            for (pr, p) in getattr(gen, name):
                assert gen.is_prime(p) and 17 <= p <= 4096, (name, p)


def test_smooth_units_match_the_generator_and_the_noslp_choice():
    """The compiled single-kernel schedules (tools/gen_smooth_kernels.py): every length sits in exactly one unit, the Complex<f32> lengths
    of tools/smooth_noslp_choice.json (measured faster without the SLP vectoriser) in the "ns" units and nowhere else, and the Makefile
    compiles exactly those units with -fno-slp-vectorize."""
    import json

    import gen_smooth_kernels as gs

    choice = json.load(open(os.path.join(ROOT, "tools", "smooth_noslp_choice.json")))
    mk = open(os.path.join(ROOT, "rustfft_amd", "csrc", "Makefile")).read()
    noslp_line = mk.split("NOSLP :=")[1].split("\n")[0].split()
    s13 = set(gs.smooth(4096, [2, 3, 5, 7, 11, 13]))
    want = {
        "smooth": [x for x in gs.smooth(4096, [2, 3, 5, 7, 11, 13]) if x > 2 and (x & (x - 1)) and x != 1200],
        "smooth3": [x for x in gs.smooth(4096, [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31]) if x not in s13],
        "smooth2": ([x for x in gs.smooth(16384, [2, 3, 5, 7]) if x > 4096 and (x & (x - 1)) and gs.big_schedule(x, (16, 32))] +
                    [x for x in gs.smooth(32768, [2, 3, 5, 7]) if x > 16384 and (x & (x - 1)) and gs.big_schedule32(x)]),
    }
    want["smooth4"] = gs.big13_sizes((16, 32)) + gs.big13_sizes32()  # round 5: the 13-smooth lengths in (4096, 32768] with a factor 11 / 13
    want["smooth5"] = gs.big31_sizes(32)  # round 5: the lengths in (4096, 16384] with a prime factor 17 .. 31
    assert len(want["smooth5"]) == 351 + 526
    csrc = os.path.join(ROOT, "rustfft_amd", "csrc")
    for fam, sizes in want.items():
        seen = {}
        for fn in sorted(os.listdir(csrc)):
            m = re.fullmatch(rf"kernels_{fam}_f32_(\w+)\.hip", fn)
            if not m:
                continue
            unit = m.group(1)
            assert f"kernels_{fam}_f32_{unit}.o" in mk and (f"kernels_{fam}_f32_{unit}" in noslp_line) == unit.startswith("ns"), (fam, unit)
            for km in re.finditer(r"MI_K1X?\(float, 32, \d+, (?:true|false), (?:\d+, \"\w*\", )?(\d+),", open(os.path.join(csrc, fn)).read()):
                n = int(km.group(1))
                assert n not in seen, (fam, n)
                seen[n] = unit
        assert sorted(seen) == sorted(sizes), (fam, sorted(set(seen) ^ set(sizes))[:10])
        ns = {n for n, u in seen.items() if u.startswith("ns")}
        if fam == "smooth5":  # all of them: the prime-radix and 32-value kernels are the ones the vectoriser costs most (no per-length choice taken)
            assert ns == set(sizes)
            continue
        assert ns == set(choice.get(fam, [])) & set(sizes), (fam, sorted(ns ^ (set(choice.get(fam, [])) & set(sizes)))[:10])
    # Complex<f64> smooth5 (the f32 loop above reads the f32 units only): every length once, the plain exchange up to 4096, split above
    seen = {}
    for fn in sorted(os.listdir(csrc)):
        if re.fullmatch(r"kernels_smooth5_f64_\d+\.hip", fn):
            assert fn.replace(".hip", ".o") in mk and fn.replace(".hip", "") not in noslp_line
            for km in re.finditer(r"MI_K1X?\(double, 64, 1, (true|false), (?:\d+, \"\w*\", )?(\d+),", open(os.path.join(csrc, fn)).read()):
                n = int(km.group(2))
                assert n not in seen and (km.group(1) == "true") == (n > 4096), n
                seen[n] = fn
    assert sorted(seen) == gs.big31_sizes(64) and len(seen) == 576 + 526


def test_general_tile_units_match_the_noslp_choice():
    """tools/gen_k2g_kernels.py: every Complex<f32> tile height sits in exactly one k2g unit, the heights of tools/k2g_noslp_choice.json in
    the "ns" units (compiled with -fno-slp-vectorize by the Makefile) and nowhere else."""
    import json

    import gen_k2g_kernels as gk

    choice = set(json.load(open(os.path.join(ROOT, "tools", "k2g_noslp_choice.json")))["f32"])
    mk = open(os.path.join(ROOT, "rustfft_amd", "csrc", "Makefile")).read()
    noslp_line = mk.split("NOSLP :=")[1].split("\n")[0].split()
    csrc = os.path.join(ROOT, "rustfft_amd", "csrc")
    seen = {}
    for fn in sorted(os.listdir(csrc)):
        m = re.fullmatch(r"kernels_k2g_f32_(\w+)\.hip", fn)
        if not m:
            continue
        unit = m.group(1)
        assert f"kernels_k2g_f32_{unit}.o" in mk and (f"kernels_k2g_f32_{unit}" in noslp_line) == unit.startswith("ns"), unit
        for km in re.finditer(r"MI_K2GT\(float, 32, \d+, (\d+),", open(os.path.join(csrc, fn)).read()):
            h = int(km.group(1))
            assert h not in seen, h
            seen[h] = unit
    want = [x for x in gk.g.smooth(640, [2, 3, 5, 7, 11, 13]) if x >= 25]
    assert sorted(seen) == want
    assert {h for h, u in seen.items() if u.startswith("ns")} == choice & set(want) == gk.NOSLP_F32 & set(want)


def test_non_temporal_choice_is_what_the_generated_units_carry():
    """tools/smooth_nt_choice.json (round 5: per-length measured choice of non-temporal row loads / loads + stores in the compiled whole-row
    schedules): exactly the listed lengths carry the ABL bits 16 / 48 and the name suffix "n" / "nn" in the generated units."""
    import json

    choice = json.load(open(os.path.join(ROOT, "tools", "smooth_nt_choice.json")))
    csrc = os.path.join(ROOT, "rustfft_amd", "csrc")
    for tag, ty in (("f32", "float"), ("f64", "double")):
        loads, both = set(), set()
        for fn in sorted(os.listdir(csrc)):
            if not re.fullmatch(rf"kernels_smooth\d?_{tag}_\w+\.hip", fn):
                continue
            for m in re.finditer(rf'MI_K1X\({ty}, \d+, \d+, (?:true|false), (\d+), "(\w*)", (\d+),', open(os.path.join(csrc, fn)).read()):
                abl, suf, n = int(m.group(1)), m.group(2), int(m.group(3))
                assert (abl & 48) in (0, 16, 48) and suf.endswith("nn") == ((abl & 48) == 48) and (suf.endswith("n") and not suf.endswith("nn")) == ((abl & 48) == 16), (fn, n, abl, suf)
                if (abl & 48) == 16:
                    loads.add(n)
                elif (abl & 48) == 48:
                    both.add(n)
        assert loads == set(choice[tag + "_loads"]) and both == set(choice[tag + "_both"]), (tag, sorted(loads ^ set(choice[tag + "_loads"]))[:10], sorted(both ^ set(choice[tag + "_both"]))[:10])
