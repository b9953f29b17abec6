"""A/B timing of experimental conv tile configurations (VG_TUNE env) at the thin / 64-channel layer shapes."""
import sys
import torch
from gpu_conv_perf import bench
B = 64
for c in [(B, 256, 256, 64, 64, 3, 1, 1), (B, 256, 256, 64, 128, 4, 2, 1), (B, 256, 256, 64, 3, 3, 1, 1),
          (B, 256, 256, 3, 64, 3, 1, 1), (B, 256, 256, 3, 64, 4, 2, 1), (B, 128, 128, 64, 128, 4, 2, 1)]:
    bench(*c, torch.bfloat16)
