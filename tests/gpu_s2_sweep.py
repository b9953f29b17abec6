"""Manual sweep: odd stride-2 shapes (ragged tiles, non-power-of-two chunk counts that fall back, thin inputs) through the
conv primitive checks of gpu_ops_check.py.  python tests/gpu_s2_sweep.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import gpu_ops_check as G
from vae_gan_amd import lib as L
lib = L.load()
bad = 0
for case in [(3, 50, 54, 64, 128, 4, 2, 1), (1, 48, 48, 96, 128, 4, 2, 1), (2, 66, 130, 32, 256, 4, 2, 1), (1, 128, 96, 128, 384, 4, 2, 1),
             (1, 50, 48, 64, 32, 4, 2, 1), (2, 48, 62, 16, 64, 4, 2, 1), (1, 256, 256, 64, 128, 4, 2, 1), (5, 48, 48, 512, 128, 4, 2, 1)]:
    r = G.check_conv(*case, torch.bfloat16, verbose=False)
    name = lib.vg_last_conv_kernel().decode()
    ok = r["fwd"] <= 6e-3 and r["dgrad"] <= 6e-3 and r["wgrad"] <= 5e-3 and r["stat_sum"] <= 6e-3 and r["pad0"] == 0.0
    print(case, "fwd %.1e dgrad %.1e wgrad %.1e stat %.1e" % (r["fwd"], r["dgrad"], r["wgrad"], r["stat_sum"]), "OK" if ok else "FAIL")
    bad += not ok
print("failures", bad)
