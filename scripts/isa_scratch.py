#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses sit relative to its MFMAs: python scripts/isa_scratch.py file.s mangled-name"""
import sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index(name + ":")
j = txt.index(".end_amdhsa_kernel", i)
body = txt[i:j].split("\n")
mf = [k for k, l in enumerate(body) if "v_mfma" in l]
sc = [k for k, l in enumerate(body) if "scratch_" in l]
print("lines", len(body), "mfma", len(mf), "first", mf[0], "last", mf[-1], "barriers", sum("s_barrier" in l for l in body))
print("scratch ops", len(sc), "inside the MFMA range:", sum(mf[0] < k < mf[-1] for k in sc))
for k in sc:
    print(k, body[k].strip())
