"""Where the host time of a train step goes (cProfile over 10 VQ steps, no device sync inside):
PYTHONPATH=. python tests/gpu_host_profile.py [vq|hier|beta]"""
import cProfile
import pstats
import sys
import time
import torch
from vae_gan_amd import ops
from vae_gan_amd import train as T

arch = sys.argv[1] if len(sys.argv) > 1 else "vq"
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
if arch == "vq":
    from vae_gan_amd.vq_vae_gan.config import Config
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN as M
    cfg, step, batch = Config(), T.vq_train_step, 64
elif arch == "beta":
    from vae_gan_amd.beta_vae_gans.config import Config
    from vae_gan_amd.beta_vae_gans.model import BetaVAEGAN as M
    cfg, step, batch = Config(), T.beta_train_step, 32
    cfg.recon_target_policy = "area"
else:
    from vae_gan_amd.hierarchical_vae_gan.config import Config
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN as M
    cfg, step, batch = Config(), T.hier_train_step, 16
    cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
cfg.image_size, cfg.batch_size = 256, batch
model = M(cfg).cuda().train()
opt_g, opt_d, _, _ = T.build_optimizers(model, cfg)
images = torch.rand(batch, 3, 256, 256, device="cuda")
for _ in range(3):
    step(model, images, opt_g, opt_d, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(model, images, opt_g, opt_d, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{arch}: host {1e2 * (t1 - t0):.2f} ms/step, total {1e2 * (t2 - t0):.2f} ms/step", flush=True)
# the backward functions run on the autograd engine's own thread, where cProfile (per thread) does not see them: for the
# profile alone, make the engine call them on this thread
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(model, images, opt_g, opt_d, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
