"""Diagnostic: which tensors differ between two runs of the same f32 train step (same weights, same batch)."""
import sys
import torch
from conftest import load_golden, as_t
from vae_gan_amd.testing import fill_state_dict
from vae_gan_amd import ops
from vae_gan_amd.train import build_optimizers, vq_train_step
from vae_gan_amd.vq_vae_gan.config import Config
from vae_gan_amd.vq_vae_gan.model import VQVAEGAN


def run(images, cfg):
    model = VQVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    grads = {}
    dl, gl, idx = vq_train_step(model, images, opt_g, opt_d, None)
    torch.cuda.synchronize()
    out = {"d_loss": dl["total"].detach().double().cpu(), "g_loss": gl["total"].detach().double().cpu(), "idx": idx.cpu(),
           "flat_g": opt_g.flat_g.clone().cpu(), "flat_d": opt_d.flat_g.clone().cpu()}
    for k, v in model.state_dict().items():
        out["sd:" + k] = v.detach().clone().cpu()
    names = [n for n, _ in model.named_parameters()]
    return out, model, opt_g, opt_d


def main():
    ops.set_compute_dtype(torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.bfloat16)
    g = load_golden("vq_step_64")
    cfg = Config()
    images = as_t(g["images"]).cuda()
    a, model, opt_g, opt_d = run(images, cfg)
    b, _, _, _ = run(images, cfg)
    nbad = 0
    for k in a:
        x, y = a[k], b[k]
        if not torch.equal(x, y):
            nbad += 1
            d = (x.double() - y.double()).abs().max().item()
            print(f"DIFF {k:60s} max|d| {d:.3e} rel {d / max(x.double().abs().max().item(), 1e-30):.3e}")
    # locate the flat gradient differences by parameter
    names = {id(p_): n for n, p_ in model.named_parameters()}
    for opt, key in ((opt_g, "flat_g"), (opt_d, "flat_d")):
        for p_, off in zip(opt.params, opt.offsets):
            n = p_.numel()
            x, y = a[key][off:off + n], b[key][off:off + n]
            if not torch.equal(x, y):
                print(f"   {key} {names.get(id(p_), '?'):45s} {str(tuple(p_.shape)):20s} rel {((x - y).norm() / x.norm().clamp_min(1e-30)).item():.3e} |g| {x.norm().item():.3e}")
    print("tensors that differ:", nbad)


main()
