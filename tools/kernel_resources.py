#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one csrc/*.hip unit (hipcc -Rpass-analysis).

    python tools/kernel_resources.py conv_mfma_split.hip [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "multi_view_active_learning_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
rows = []
for b in blocks:
    mangled = b.split(" ")[0]
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    g = lambda k: int(re.search(re.escape(k) + r": (\d+)", b).group(1))
    rows.append((name.split("(")[0], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]")))
print("%-90s %5s %5s %7s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "occ"))
for n, v, a, s, o in sorted(rows):
    if flt in n:
        print("%-90s %5d %5d %7d %4d" % (n[-90:], v, a, s, o))
