"""Host-side enqueue time of a train step vs its GPU time (is the step launch-bound?)."""
import time, torch
from vae_gan_amd import ops
from vae_gan_amd.train import build_optimizers, vq_train_step
from vae_gan_amd.vq_vae_gan.config import Config
from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
ops.set_compute_dtype(torch.bfloat16)
cfg = Config(); torch.manual_seed(0)
model = VQVAEGAN(cfg).cuda().train()
og, od, _, _ = build_optimizers(model, cfg)
x = torch.rand(64, 3, 256, 256, device="cuda")
for _ in range(3): vq_train_step(model, x, og, od)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): vq_train_step(model, x, og, od)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.1f} ms/step, total {1e3*(t2-t0)/10:.1f} ms/step")
