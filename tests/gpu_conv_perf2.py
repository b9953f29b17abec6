"""A/B of the wide 3x3 / 2x2-tap layers: python tests/gpu_conv_perf2.py  (prints the kernel each launch used)."""
import ctypes as C
import os
import torch
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
from gpu_conv_perf import bench
from gpu_ops_check import make_desc

if __name__ == "__main__":
    B = 64
    dt = torch.bfloat16
    for c in [(B, 64, 64, 256, 256, 3, 1, 1), (B, 32, 32, 512, 512, 3, 1, 1), (B, 64, 64, 256, 512, 4, 2, 1), (B, 128, 128, 128, 256, 4, 2, 1)]:
        for _ in range(2):
            bench(*c, dt, iters=20)
    print(L.load().vg_last_conv_kernel())
