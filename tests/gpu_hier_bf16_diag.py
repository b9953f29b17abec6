"""How far the first-iteration bf16 losses of the hierarchical step move when only the summation ORDER inside the conv
kernels changes (older kernel generations, other tile shapes): python tests/gpu_hier_bf16_diag.py  (one setting per process:
VG_CONV_KERNEL / VG_IGEMM3_WIDE_TILE / VG_LIB_EXP in the environment)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(__file__))
from vae_gan_amd import lib as L
if os.environ.get("VG_LIB_EXP"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_exp.so")
from vae_gan_amd import ops as vg
from conftest import load_golden, as_t
from vae_gan_amd.hierarchical_vae_gan.config import Config
from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN
from vae_gan_amd.testing import fill_state_dict, hier_step_eps
from vae_gan_amd.train import build_optimizers, hier_train_step

vg.set_compute_dtype(torch.bfloat16)
g = load_golden("hier_step_128")
cfg = Config()
cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
model = HierarchicalVAEGAN(cfg)
model.load_state_dict(fill_state_dict(model.state_dict()))
model.cuda().train()
opt_g, opt_d, _, _ = build_optimizers(model, cfg)
eps = hier_step_eps(2, 128)
pair = [[e.cuda() for e in fwd] for fwd in eps[0]]
dl, gl = hier_train_step(model, as_t(g["images"]).cuda(), opt_g, opt_d, eps_lists=pair)
torch.cuda.synchronize()
out = []
for k, v in {"d_loss": dl["total"], "d_kl": dl["kl"], "g_loss": gl["total"], "recon": gl["recon"], "kl": gl["kl"]}.items():
    ref = float(g[f"it0_{k}"])
    out.append(f"{k} {100 * (float(v.detach()) - ref) / max(abs(ref), 1e-3):+.2f}%")
print({k: os.environ.get(k) for k in ("VG_CONV_KERNEL", "VG_IGEMM3_WIDE_TILE", "VG_LIB_EXP", "VG_ATTN_F32")}, " ".join(out), flush=True)
