#!/usr/bin/env python
"""compact table from `hipcc -Rpass-analysis=kernel-resource-usage` output: name, VGPR, AGPR, spills, occupancy"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", txt)[1:]
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    if flt and flt not in name:
        continue
    print("%-70s V%4s A%4s S%3s spillV%3s occ%2s" % (name[:70], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
