#!/usr/bin/env python3
"""Registers, LDS, scratch and occupancy of every kernel of the offload, from hipcc's resource-usage remarks.
usage: tools/kres.py [-Dname=value ...] [substring]"""
import re
import subprocess
import sys

defs = [a for a in sys.argv[1:] if a.startswith("-D")]
want = [a for a in sys.argv[1:] if not a.startswith("-D")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", *defs, "unet_amd/csrc/unet_offload.hip",
       "-o", "/tmp/kres.so", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.check_output(["c++filt", m.group(1)]).decode().strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("unetdev::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    if want and not any(w in k for w in want):
        continue
    print(f"{k:34s} VGPR {v.get('VGPRs'):>4s} SGPR {v.get('TotalSGPRs'):>4s} scratch {v.get('ScratchSize'):>4s} occ {v.get('Occupancy'):>2s} LDS {v.get('LDS Size'):>6s} spill {v.get('VGPRs Spill')}")
