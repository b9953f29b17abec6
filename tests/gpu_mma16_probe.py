"""Timing probe: every conv / weight-gradient shape of the VQ step, normal library against an experiment build
(VG_LIB_EXP=1 -> libvg_hip_exp.so).  python tests/gpu_mma16_probe.py"""
import os
import torch
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):
    _v = os.environ["VG_LIB_EXP"]
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
from gpu_conv_perf import bench

if __name__ == "__main__":
    B = 64
    dt = torch.bfloat16
    for c in [(B, 256, 256, 64, 64, 3, 1, 1), (B, 128, 128, 128, 128, 3, 1, 1), (B, 64, 64, 256, 256, 3, 1, 1),
              (B, 32, 32, 512, 512, 3, 1, 1), (B, 256, 256, 64, 128, 4, 2, 1), (B, 128, 128, 128, 256, 4, 2, 1),
              (B, 64, 64, 256, 512, 4, 2, 1)]:
        for _ in range(2):
            bench(*c, dt, iters=20)
