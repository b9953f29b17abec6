#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid size) of a rocprofv3 --pmc --kernel-trace csv output directory."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key[0], "grid", key[1], "n", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        m, g = cs["SQ_VALU_MFMA_BUSY_CYCLES"][2:], cs["GRBM_GUI_ACTIVE"][2:]
        print(f"    MFMA busy = {sum(m) / (1024 * sum(g) / 8):.3f}; kernel cycles = {sum(g) / len(g) / 8:.0f}")
