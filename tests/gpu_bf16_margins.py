"""The table of DESIGN.md section 2: every bf16 whole-step bound with its source (the reference's own bf16-storage
floor, 'noise_bf16:*' in the goldens) and the value measured on this tree.  python tests/gpu_bf16_margins.py"""
import torch

from conftest import load_golden, as_t
from vae_gan_amd import ops
from vae_gan_amd.testing import fill_state_dict, hier_step_eps, synthetic_images
from vae_gan_amd.train import build_optimizers, vq_train_step, hier_train_step, beta_train_step
from test_gpu_step import bf16_bound


def report(name, g, mine):
    for k, v in mine.items():
        ref = float(g[f"it0_{k}"])
        err = abs(float(v.detach()) - ref) / max(abs(ref), 1e-3)
        fl = sorted(float(x) for x in g[f"noise_bf16v:it0_{k}"])
        print(f"| {name} | {k} | {bf16_bound(g, k):.3g} | {fl[0]:.1e} / {fl[len(fl) // 2]:.1e} / {fl[-1]:.1e} | {err:.2e} |", flush=True)


def main():
    ops.set_compute_dtype(torch.bfloat16)
    print("| golden | scalar | bound = max(2e-2, 1.5 x largest sample) | reference with bf16 storage, 8 samples: min / median / max | engine, this tree |")
    print("|---|---|---|---|---|")
    for name in ("vq_step_64", "vq_step_256"):
        from vae_gan_amd.vq_vae_gan.config import Config
        from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
        g = load_golden(name)
        cfg = Config()
        model = VQVAEGAN(cfg)
        model.load_state_dict(fill_state_dict(model.state_dict()))
        model.cuda().train()
        og, od, _, _ = build_optimizers(model, cfg)
        dl, gl, _ = vq_train_step(model, as_t(g["images"]).cuda(), og, od)
        report(name, g, {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"]})
    from vae_gan_amd.hierarchical_vae_gan.config import Config
    from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN
    g = load_golden("hier_step_128")
    cfg = Config()
    cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
    model = HierarchicalVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    og, od, _, _ = build_optimizers(model, cfg)
    pair = [[e.cuda() for e in fwd] for fwd in hier_step_eps(2, 128)[0]]
    dl, gl = hier_train_step(model, as_t(g["images"]).cuda(), og, od, eps_lists=pair)
    report("hier_step_128", g, {"d_loss": dl["total"], "d_kl": dl["kl"], "g_loss": gl["total"], "recon": gl["recon"], "kl": gl["kl"]})
    from vae_gan_amd.beta_vae_gans.config import Config
    from vae_gan_amd.beta_vae_gans.model import BetaVAEGAN
    g = load_golden("beta_step_256")
    cfg = Config()
    cfg.recon_target_policy = "area"
    model = BetaVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    og, od, _, _ = build_optimizers(model, cfg)
    dl, gl = beta_train_step(model, synthetic_images(2, 256, seed=int(g["images_seed"])).cuda(), og, od, eps=as_t(g["eps0"]).cuda())
    report("beta_step_256", g, {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "kl": gl["kl"]})


if __name__ == "__main__":
    main()
