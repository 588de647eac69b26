#!/usr/bin/env python3
"""registers, spills and scratch of the kernels in the last build (hm-opencl_amd/csrc/build/resource_usage.txt, written by the Makefile with
-Rpass-analysis=kernel-resource-usage): tools/resource_usage.py [substring of the kernel name]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "hm-opencl_amd/csrc/build/resource_usage.txt")).read()
want = sys.argv[1] if len(sys.argv) > 1 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = subprocess.run(["c++filt", b.split(" ")[0]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void hmme::", "")
    if want not in name:
        continue
    g = lambda k: re.search(k + r": (\S+)", b).group(1)
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name:38s} SGPRs {g('TotalSGPRs'):>3} (spilled {g('SGPRs Spill'):>2})  VGPRs {g(' VGPRs'):>3} (spilled {g('VGPRs Spill'):>2})  scratch {scratch:>3} B/lane  waves/SIMD {occ}")
