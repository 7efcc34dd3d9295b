#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (VGPR/scratch/occupancy per kernel)."""
import re
import subprocess
import sys

log = open(sys.argv[1]).read()
blocks = re.split(r"remark: Function Name: ", log)[1:]
seen = set()
for b in blocks:
    name = b.split()[0]
    if name in seen:
        continue
    seen.add(name)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = d.replace("mi355::", "").replace("void ", "")
    d = re.sub(r"Sched<([^>]*)>", lambda m: "S<" + m.group(1).replace(" ", "") + ">", d)
    print("%-70s vgpr=%3d agpr=%3d scratch=%4d occ=%d sgpr=%3d lds=%6d" % (
        d[:70], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs"),
        g(r"LDS Size \[bytes/block\]")))
