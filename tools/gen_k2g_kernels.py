#!/usr/bin/env python3
"""Generates rustfft_amd/csrc/kernels_k2g_{f32,f64}_{0..7}.hip: large-N pass kernels (k2g_body) for every 13-smooth tile
height R in [25, 640], so that 13-smooth lengths above one workgroup are transformed in two or three passes at HBM speed
instead of through Bluestein (the GPU form of the reference's generic MixedRadix, src/algorithm/mixed_radix.rs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_smooth_kernels as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NFILES = 8
# tile heights that also get the fused multi-kernel Bluestein passes (chirp on load, spectrum multiply / chirp on store):
# a roughly geometric ladder (ratio ~1.12), so that products of two or three of them land within a few per cent above any
# 2n - 1
FUSED = [64, 72, 80, 90, 100, 112, 128, 144, 160, 180, 200, 224, 256, 288, 320, 360, 400, 448, 512, 576, 640]


# tall tiles through the split exchange (f64: two passes for 7-smooth lengths up to ~3.7 million instead of three)
TALL = [720, 768, 800, 864, 900, 960, 1000, 1024, 1080, 1125, 1152, 1200, 1250, 1280, 1350, 1440, 1500, 1536, 1600, 1800, 1920, 2048]


def tall_schedule(n, emax, tpfmax, maxp):
    import math

    best = None
    for rad in g.factorizations_big(n):
        if len(rad) > maxp or any(r > emax for r in rad):
            continue
        tpf = max(math.ceil((n // r) / (emax // r)) for r in rad)
        if tpf > tpfmax:
            continue
        # the first sub-pass carries the inter-pass twiddle recurrence (one table look-up pair per butterfly, log2 R live
        # powers): a radix near 8 keeps both the look-ups and the live registers low, as in the power-of-two 1024-row tile
        firsts = [r for r in rad if 4 <= r <= 10]
        if not firsts or len(rad) < 3:
            continue
        first = min(firsts, key=lambda r: abs(r - 8))
        util = sum((n // r) / (tpf * math.ceil((n // r) / tpf)) for r in rad) / len(rad)
        key = (len(rad), max(rad) > 16, abs(first - 8), max(rad), -util)  # radices above 16 pull 31 table twiddles per butterfly into registers
        if best is None or key < best[0]:
            rest = sorted(rad)
            rest.remove(first)
            best = (key, [first] + rest, tpf)
    return best[1], best[2]


# Complex<f32> tile heights whose first + later kernels, summed over every sampled plan that uses them, run >= 2 % faster in BOTH of two
# runs when compiled WITHOUT the SLP vectoriser (tools/r4/k2g_kernel_ab.py: per-kernel times of 640 two- and three-pass lengths in two
# builds, profiles/r4/k2g_kernel_ab_rep{1,2}.jsonl: 89 of 166 heights, +2 ... +36 %, one height loses): their own units ("ns"), which the
# Makefile compiles with -fno-slp-vectorize.
import json as _json
NOSLP_F32 = set(_json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "k2g_noslp_choice.json")))["f32"])
NS_FILES = 4


def main():
    # round 2: heights with the factors 11 and 13 as well, so that EVERY 13-smooth length above one workgroup runs in two to four
    # column-tile passes (before: lengths with 11 or 13 above ~5100 went through the fused Bluestein at 7x the algorithmic traffic)
    rs = [x for x in g.smooth(640, [2, 3, 5, 7, 11, 13]) if x >= 25]
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        keep = [x for x in rs if not (prec == 32 and x in NOSLP_F32)]
        ns = [x for x in rs if prec == 32 and x in NOSLP_F32]
        units = [(str(i), keep[i::NFILES]) for i in range(NFILES)] + ([(f"ns{i}", ns[i::NS_FILES]) for i in range(NS_FILES)] if ns else [])
        for ci, chunk in units:
            lines = []
            for r in chunk:
                rad, tpf = g.schedule(r)
                f = 128 // esz  # at least 128-byte row segments
                lds = lambda ff: ff * (r + r // rad[0] + 34) * esz  # Sched::pitch upper bound
                while f > 4 and (f * tpf > 1024 or lds(f) > 78 * 1024):
                    f //= 2
                while f * tpf < 256 and f < 128 and lds(2 * f) <= 78 * 1024:  # short tiles: more columns per workgroup
                    f *= 2
                # f32: every sub-pass twiddle table staged in LDS (engine.h TWL, suffix "t"): interleaved A/B of 11 plans over these tiles
                # +1 .. +28 %, median +6 % (profiles/r3/ab_k2g_twl_f32.jsonl); f64: -6 .. +7 %, kept as it was
                macro = "MI_K2GT" if prec == 32 else "MI_K2G"
                lines.append(f"    {macro}({ty}, {prec}, {f}, {r}, {tpf}, {', '.join(map(str, rad))});")
                if r in FUSED:
                    lines.append(f"    MI_K2GF({ty}, {prec}, {f}, {r}, {tpf}, {', '.join(map(str, rad))});")
            # f64 only: 8 columns (128-byte segments) x <= 128 threads x 16 values.  The f32 form (16 columns x <= 64 threads x 32
            # values) does not fit 128 VGPRs with the general body (100 - 300 B of spills): measured -14 .. +13 % against three
            # passes of short tiles, so f32 keeps three passes; f64 gains 14 - 26 % (10^6: 5.5 -> 6.9 TFLOP/s).
            for r in (TALL[int(ci)::NFILES] if prec == 64 else []):
                rad, tpf = tall_schedule(r, 16, 128, 4)
                lines.append(f"    MI_K2GS({ty}, {prec}, 8, {r}, {tpf}, {', '.join(map(str, rad))});")
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_k2g_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_k2g_kernels.py — do not edit.  Large-N pass kernels for the 13-smooth tile heights in\n"
                         f"// [25, 640] (unit {ci}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_k2g_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        # Rader-fused forms of every tile (multi-kernel Rader for primes beyond one workgroup, k2g_body FUSE 4 / 5 / 6): the inner
        # length p - 1 has ONE factorisation into tile heights, so every height needs its gather / multiply / scatter pass
        NR = 4
        for ci in range(NR):
            lines = []
            for r in rs[ci::NR]:
                rad, tpf = g.schedule(r)
                f = 128 // esz
                lds = lambda ff: ff * (r + r // rad[0] + 34) * esz
                while f > 4 and (f * tpf > 1024 or lds(f) > 78 * 1024):
                    f //= 2
                while f * tpf < 256 and f < 128 and lds(2 * f) <= 78 * 1024:
                    f *= 2
                lines.append(f"    MI_K2GR({ty}, {prec}, {f}, {r}, {tpf}, {', '.join(map(str, rad))});")
            path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_k2gr_{tag}_{ci}.hip")
            with open(path, "w") as fh:
                fh.write(f"// GENERATED by tools/gen_k2g_kernels.py — do not edit.  Rader-fused large-N pass kernels (gather on the first load,\n"
                         f"// spectrum multiply / scatter on the last store) for the 13-smooth tile heights in [25, 640] (part {ci + 1} of {NR}), Complex<{ty}>.\n"
                         '#include "launch.h"\nnamespace mi355 {\n'
                         f"void register_k2gr_{tag}_{ci}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
    # prime tile heights: Rader inside the tile (kernels.h k2r_body) for every prime P in (31, 640] whose P - 1 is 31-smooth -- the
    # composite lengths whose prime factors all exceed 31 (37 x 41, 101 x 103, ...) then run as column-tile passes like every
    # other composite instead of through Bluestein (src/algorithm/mixed_radix.rs:53-158, src/plan.rs:474-506)
    import gen_rader_kernels as gr

    s13 = set(g.smooth(640, [2, 3, 5, 7, 11, 13]))
    s31 = set(g.smooth(640, [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31]))
    for tag, ty, prec, esz in (("f32", "float", 32, 8), ("f64", "double", 64, 16)):
        lines, done = [], []
        for P in range(37, 641):
            if not gr.is_prime(P) or (P - 1) not in s31:
                continue
            rad, tpf = g.schedule(P - 1) if (P - 1) in s13 else g.schedule31(P - 1)
            pitch, xs, emax, twreg = gr.layout(P - 1, rad, tpf)
            if not (xs < pitch and P <= pitch):
                continue
            f = 128 // esz
            while f > 128 // esz // 2 and (f * tpf > 1024 or f * pitch * esz > 64 * 1024):
                f //= 2
            if f * tpf > 1024 or f * pitch * esz > 64 * 1024:
                continue
            while f * tpf < 192 and f < 128 and 2 * f * pitch * esz <= 48 * 1024:  # short tiles: more columns per workgroup
                f *= 2
            lines.append(f"    MI_K2R({ty}, {prec}, {f}, {P - 1}, {tpf}, {', '.join(map(str, rad))});  // P = {P}")
            done.append(P)
        path = os.path.join(ROOT, "rustfft_amd", "csrc", f"kernels_k2r_{tag}.hip")
        with open(path, "w") as fh:
            fh.write(f"// GENERATED by tools/gen_k2g_kernels.py — do not edit.  Column-tile passes of PRIME tile heights (Rader inside the tile,\n"
                     f"// kernels.h k2r_body) for the primes in (31, 640] whose P - 1 is 31-smooth, Complex<{ty}>.\n"
                     '#include "launch.h"\nnamespace mi355 {\n'
                     f"void register_k2r_{tag}(std::vector<KernelEntry>& reg) {{\n" + "\n".join(lines) + "\n}\n}  // namespace mi355\n")
        print(tag, len(done), "prime tile heights:", done)
    print(len(rs), "tile heights per precision:", rs)


if __name__ == "__main__":
    main()
