#!/usr/bin/env python3
"""lds_dma16_buf writes M0 and does not restore it (csrc/igemm.hip).  That is only safe in kernels where nothing else
touches M0: compile igemm.hip to ISA and check that every M0 access of the kernels that use buffer_load ... lds is one of
the helper's own `s_mov_b32 m0, sN`.   python scripts/check_m0.py   (about a minute; exits non-zero on a finding)"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "vae_gan_amd", "csrc", "igemm.hip")
out = os.path.join(tempfile.gettempdir(), "vg_igemm_check_m0.s")
subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                src, "-o", out], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
bad = 0
for m in re.finditer(r"^(_Z\w+):", txt, re.M):
    name = m.group(1)
    end = txt.find(".end_amdhsa_kernel", m.end())
    nxt = txt.find("\n_Z", m.end())
    if end < 0 or (0 <= nxt < end):
        continue
    body = txt[m.end():end]
    if not re.search(r"buffer_load_dwordx4 .* lds", body):
        continue
    uses = [l.strip() for l in body.split("\n") if re.search(r"\bm0\b", l)]
    writes = [u for u in uses if re.fullmatch(r"s_mov_b32 m0, s\d+", u)]
    saves = [u for u in uses if re.fullmatch(r"s_mov_b32 s\d+, m0", u)]
    other = [u for u in uses if u not in writes and u not in saves]
    if saves:      # kernels on the saving helpers (lds_dma16_s / lds_dma16_bufk: save, write, restore around every DMA)
        if len(writes) != 2 * len(saves):
            other += [f"{len(saves)} saves of M0 against {len(writes)} writes: not save / write / restore triples"]
        print(f"{name[:70]:70s} M0 saved and restored around {len(saves)} DMAs, other M0 accesses: {len(other)}")
    else:
        print(f"{name[:70]:70s} M0 writes by the helper: {len(writes)}, other M0 accesses: {len(other)}")
    for u in other:
        print("    ", u)
    bad += len(other)
sys.exit(1 if bad else 0)
