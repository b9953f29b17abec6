import sys, torch
sys.path.insert(0, "tests")
from conftest import load_golden, as_t
import test_gpu_step as T
from vae_gan_amd import lib as L
from vae_gan_amd.train import vq_train_step
g = load_golden("vq_step_64")
lib = L.load()
for mode in (3, 3 | 128, 3 | 64, 3 | 32, 3 | 32 | 64 | 128):
    lib.vg_set_conv_kernel(mode)
    model, opt_g, opt_d = T._setup(torch.bfloat16)
    images = as_t(g["images"]).cuda()
    dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d)
    torch.cuda.synchronize()
    print(mode, {k: round(float(v.detach()), 5) for k, v in {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"]}.items()},
          "ref", {k: round(float(g[f"it0_{k}"]), 5) for k in ("d_loss", "g_loss", "recon", "vq")}, flush=True)
