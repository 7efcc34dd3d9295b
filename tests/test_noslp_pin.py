"""The per-unit `-fno-slp-vectorize` choice (rustfft_amd/csrc/Makefile NOSLP; profiles/r4/ab_noslp_*.jsonl) rests on what this compiler's SLP
vectoriser does to the VALU-heavy Complex<f32> bodies: it pairs the re / im parts of DIFFERENT values into v_pk_*_f32 operations and pays
for the pairing in register moves (a quarter to a third of the VALU instructions).  A compiler upgrade can silently change that.  This test
recompiles three sampled kernels of no-SLP units both ways (device code only, a few seconds each) and fails when the instruction counts no
longer favour the side the Makefile chose -- the moment to re-run the A/B sweeps (tools/ab_lengths.py) and re-sort the units.
CPU-only: hipcc cross-compiles gfx950 without a GPU; skipped where hipcc is absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rustfft_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (unit of the Makefile's NOSLP list, what to instantiate) -- one kernel each: the 8192-point Bluestein body (+41 % when it was re-scheduled
# without the vectoriser), config 4's Rader rows loop (p = 1009), and the 1024-row column tile that dominates the fused 2^20 launch
PROBES = {
    "kernels_bs_f32": ("#define MI355_PK_CMUL 1\n", "MI_BS(float, 32, 1, 8192, 512, 8, 8, 8, 16);"),
    "kernels_rader_f32_ns1": ("", "MI_RADER(float, 32, 8, 3, 1008, 126, 14, 9, 8);"),
    "kernels_k2f_f32": ("", 'using S1024 = Sched<1024, 32, 8, 8, 16>; MI_K2F(1, float, 32, "a", 16, true, 128, S1024, "b", 16, true, 128, S1024);'),
}


def _counts(asm):
    valu = len(re.findall(r"^\s+v_", asm, flags=re.M))
    movs = len(re.findall(r"^\s+v_mov_b32", asm, flags=re.M)) + len(re.findall(r"^\s+v_pk_mov_b32", asm, flags=re.M))
    scratch = max([int(m) for m in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", asm)] or [0])
    vgprs = max([int(m) for m in re.findall(r"\.vgpr_count:\s*(\d+)", asm)] or [0])
    return valu, movs, scratch, vgprs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
@pytest.mark.parametrize("unit", sorted(PROBES))
def test_the_slp_vectoriser_still_costs_these_bodies_their_register_moves(unit, tmp_path):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert unit in mk.split("NOSLP :=")[1].split("\n")[0].split(), f"{unit} left the Makefile's NOSLP list: update this test's sample"
    pre, body = PROBES[unit]
    src = tmp_path / "probe.hip"
    src.write_text(pre + '#include "launch.h"\n#include "kernel_lists.h"\nnamespace mi355 { void register_probe(std::vector<KernelEntry>& reg) { ' + body + " } }\n")
    res = {}
    for tag, flag in (("slp", []), ("noslp", ["-fno-slp-vectorize"])):
        out = tmp_path / f"{tag}.s"
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-Wno-unused-value", "--cuda-device-only", "-S"] + flag + ["-o", str(out), str(src)])
        res[tag] = _counts(out.read_text())
    (valu_s, mov_s, scr_s, vg_s), (valu_n, mov_n, scr_n, vg_n) = res["slp"], res["noslp"]
    print(unit, "with the vectoriser: VALU", valu_s, "moves", mov_s, "scratch", scr_s, "VGPRs", vg_s, "| without: VALU", valu_n, "moves", mov_n, "scratch", scr_n, "VGPRs", vg_n)
    # the chosen side (without) must still issue fewer VALU instructions, fewer register moves, and no more scratch
    assert valu_n < valu_s and mov_n < mov_s and scr_n <= scr_s, (unit, res)
    if unit == "kernels_k2f_f32":
        # the fused column tiles: what the vectoriser costs here is the 128-VGPR cap (a spill to scratch, or the registers of a fourth wave per SIMD)
        assert scr_n < scr_s or vg_n < vg_s, f"{unit}: neither scratch ({scr_s} -> {scr_n}) nor VGPRs ({vg_s} -> {vg_n}) favour the no-SLP build any more"
    else:
        assert mov_s >= 0.15 * valu_s, f"{unit}: the vectoriser's register moves are down to {mov_s} of {valu_s} VALU instructions -- re-measure whether this unit still belongs to NOSLP"
