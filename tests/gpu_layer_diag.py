"""Diagnostic (not collected): per-unit bf16-mode errors against the float64 reference and the bf16-storage reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conftest import load_golden, rel_l2
from layer_units import UNITS, sample_index
from vae_gan_amd import ops
from vae_gan_amd.testing import digest
import test_gpu_layer_grads as T

for mode in (torch.bfloat16, torch.float32):
    ops.set_compute_dtype(mode)
    for i, u in enumerate(UNITS):
        u, g, y, dx, grads = T._run_unit(ops, i)
        row = []
        for tag, t in (("y", y), ("dx", dx)):
            idx = sample_index(u.name + ":" + tag, t.numel())
            row.append(f"{tag}: {rel_l2(t.flatten()[idx], g[f'{u.name}:{tag}:sample']):.1e}/{rel_l2(t.flatten()[idx], g[f'{u.name}:b:{tag}:sample']):.1e}")
        for k, t in grads.items():
            if k.endswith("weight"):
                d = digest(k, t)
                r, rb = g[f"{u.name}:gd:{k}"], g[f"{u.name}:b:gd:{k}"]
                row.append(f"{k}: l2 {abs(d[0]-r[0])/r[0]:.1e}/{abs(d[0]-rb[0])/rb[0]:.1e} pr {abs(d[2]-r[2])/r[0]:.1e}/{abs(d[2]-rb[2])/rb[0]:.1e}")
        print(str(mode)[6:], u.name, " | ".join(row), flush=True)
