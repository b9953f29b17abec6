#!/usr/bin/env python3
"""Summarise a `hipcc -Rpass-analysis=kernel-resource-usage` log: registers, scratch and occupancy per kernel.
   python scripts/kernel_regs.py build.log [name-filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = (r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
       r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)")
seen = set()
for name, v, a, sc, occ, lds in re.findall(pat, txt, re.S):
    if flt not in name or name in seen:
        continue
    seen.add(name)
    try:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dn = name
    print(f"{dn[:100]:100s} VGPR {v:>3s} AGPR {a:>3s} scratch {sc:>4s} occ {occ}")
