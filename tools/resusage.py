#!/usr/bin/env python3
"""Compile spc_hip.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, SGPRs, spills, scratch, occupancy, LDS).  usage: tools/resusage.py [filter-substring] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
       "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "sp_coupler_amd/csrc/spc_hip.hip"), "-o", "/tmp/resusage.so",
       "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        print("%-62s VGPR %3d AGPR %3d SGPR %3d spillS %3d spillV %3d scratch %4d occ %2d LDS %6d" % (
            name[:62], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs", -1)), r.get("SGPRs Spill", 0),
            r.get("VGPRs Spill", 0), r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", -1),
            r.get("LDS Size [bytes/block]", 0)))
