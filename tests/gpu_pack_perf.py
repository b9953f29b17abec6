"""Timing of the batched weight re-pack on the generator's conv layers (VQ-VAE-GAN, bf16)."""
import torch
from vae_gan_amd import ops
from vae_gan_amd.train import build_optimizers
from vae_gan_amd.vq_vae_gan.config import Config
from vae_gan_amd.vq_vae_gan.model import VQVAEGAN

ops.set_compute_dtype(torch.bfloat16)
cfg = Config()
model = VQVAEGAN(cfg).cuda().train()
opt_g, opt_d, _, _ = build_optimizers(model, cfg)
x = torch.rand(2, 3, 64, 64, device="cuda")
from vae_gan_amd.train import vq_train_step
for _ in range(2):
    vq_train_step(model, x, opt_g, opt_d)
torch.cuda.synchronize()
for opt, name in ((opt_g, "G"), (opt_d, "D")):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.repack_all(opt._epoch)
    e1.record()
    torch.cuda.synchronize()
    print(f"repack {name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
