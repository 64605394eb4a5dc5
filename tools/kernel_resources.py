#!/usr/bin/env python3
"""kernel_resources.py -- VGPRs / SGPRs / scratch / occupancy of every kernel of one translation unit, from
hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

    python3 tools/kernel_resources.py aligntools/c_amd/csrc/at_k16_g16b.hip [-DAT_BITS16=8]
"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Iinclude", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-6s %-6s %-8s %-5s %-7s %-7s %s" % ("VGPR", "SGPR", "scratch", "occ", "sspill", "vspill", "kernel"))
for c in rows:
    print("%-6s %-6s %-8s %-5s %-7s %-7s %s" % (c.get("VGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"),
                                              c.get("SGPRs Spill"), c.get("VGPRs Spill"), c["name"].replace("void at::", "").replace("(at::Sweep16Args)", "")))
