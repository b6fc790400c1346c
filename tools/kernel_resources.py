#!/usr/bin/env python
"""Print VGPR / spill / LDS / occupancy of every les_strip_kernel instantiation (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localexpstereo_amd import build

cmd = [build._hipcc()] + build.HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, "les_hip.hip"), "-o", "/tmp/les_res.so"]
txt = subprocess.run(cmd, capture_output=True, text=True, cwd=build.CSRC).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(name[:90], "VGPR", g("VGPRs"), "spill", g("VGPRs Spill"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "LDS", g(r"LDS Size \[bytes/block\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"))
