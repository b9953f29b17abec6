"""GPU tier: the full VQ-VAE-GAN train step (vq_vae_gan/main.py:75-111 semantics) on the HIP engine
against the golden captured from the reference model + torch.optim.Adam (two iterations, B=2, 64x64)."""
import pytest
import torch

from conftest import load_golden, as_t, close_l2
from vae_gan_amd.testing import fill_state_dict, digest

pytestmark = pytest.mark.gpu


def _setup(dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
    from vae_gan_amd.vq_vae_gan.config import Config
    from vae_gan_amd.train import build_optimizers
    ops.set_compute_dtype(dtype)
    cfg = Config()
    model = VQVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.cuda().train()
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    return model, opt_g, opt_d


def test_vq_step_f32_matches_reference():
    from vae_gan_amd.train import vq_train_step
    g = load_golden("vq_step_64")
    model, opt_g, opt_d = _setup(torch.float32)
    images = as_t(g["images"]).cuda()
    for it in range(2):
        dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d, elide_d_wgrad=(it == 1))
        torch.cuda.synchronize()
        mine = {"d_loss": dl["total"], "d_recon": dl["recon"], "g_loss": gl["total"], "recon": gl["recon"],
                "vq": gl["vq"], "adv": gl["adv"]}
        for k, v in mine.items():
            ref = float(g[f"it{it}_{k}"])
            tol = max(1e-4, 4 * float(g[f"noise:it{it}_{k}"]))
            assert abs(float(v.detach()) - ref) <= tol * max(abs(ref), 1e-3), (it, k, float(v.detach()), ref, tol)
        match = (idx2.cpu() == as_t(g[f"it{it}_idx2"])).float().mean().item()
        assert match >= (1.0 if it == 0 else float(g["idxeq64:it1_idx2"]) - 0.02), (it, match)
    sd = model.state_dict()
    noise = max(float(g["noise:it1_recon_img"]), 1e-4)
    for key, ref in g.items():
        if key.startswith("sd:") and sd[key[3:]].dtype == torch.int64:
            assert int(sd[key[3:]]) == int(ref), key       # BN counters: generator +2/iter, discriminator +3/iter
        elif key.startswith("sdd:"):
            d = digest(key[4:], sd[key[4:]])
            assert abs(d[0] - ref[0]) <= max(1e-4, 4 * noise) * max(ref[0], 1e-6), key
    from vae_gan_amd import ops
    ops.set_compute_dtype(torch.bfloat16)


def test_vq_step_bf16_close_to_reference():
    """bf16 mode: first-iteration losses within 2e-2 of the fp32 reference (north_star bf16 tolerance).

    The discriminator's hinge loss gets 3e-2: at random init it is 2 + (mean D(fake) - mean D(real)) with logit means of
    ~1e-2, so the bound is one on the ABSOLUTE error of the logit means (0.06), and those carry bf16 storage through five
    conv + train-mode BatchNorm layers on a 4-image batch.  Measured: 1.9 % and 2.5 % below the reference with two kernel
    selections whose forward tensors differ only in the last bit of the BatchNorm sums (tests/gpu_dl_diag.py) - the value
    is chaotic at the 0.6 % level in bf16, and the reference's own modules run with bf16 storage sit 2.6e-2..5.1e-2 from
    fp64 on every unit with a LeakyReLU.  The tight bf16 checks are the per-layer ones
    (tests/test_gpu_layer_grads.py), held against a bf16-storage run of the reference."""
    from vae_gan_amd.train import vq_train_step
    g = load_golden("vq_step_64")
    model, opt_g, opt_d = _setup(torch.bfloat16)
    images = as_t(g["images"]).cuda()
    dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d)
    torch.cuda.synchronize()
    for k, v in {"d_loss": dl["total"], "g_loss": gl["total"], "recon": gl["recon"], "vq": gl["vq"]}.items():
        ref = float(g[f"it0_{k}"])
        assert abs(float(v.detach()) - ref) <= (3e-2 if k == "d_loss" else 2e-2) * max(abs(ref), 1e-3), (k, float(v.detach()), ref)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_shared_encoder_equals_literal_schedule():
    """The default step runs the encoder once per iteration and books both BatchNorm buffer updates
    (train.vq_train_step); it must land where the literal two-forward schedule of vq_vae_gan/main.py:78,99 lands.
    Compared after ONE iteration: from the second on, Adam's lr*sign(g) on rounding-noise gradients makes even
    the reference's own fp32 run irreproducible below 2e-2 (noise:it1_d_loss in the golden)."""
    from vae_gan_amd.train import vq_train_step
    from vae_gan_amd import ops
    g = load_golden("vq_step_64")
    images = as_t(g["images"]).cuda()
    runs = []
    for share in (True, False):
        model, opt_g, opt_d = _setup(torch.float32)
        dl, gl, idx2 = vq_train_step(model, images, opt_g, opt_d, share_encoder=share)
        out = (float(dl["total"].detach()), float(gl["total"].detach()), float(gl["vq"]), float(gl["recon"]), idx2.cpu())
        runs.append((out, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    (a, sda), (b, sdb) = runs
    for x, y in zip(a[:4], b[:4]):
        assert abs(x - y) <= 1e-5 * abs(y), (a[:4], b[:4])
    assert torch.equal(a[4], b[4])
    for k in sda:
        if sda[k].dtype == torch.int64:
            assert int(sda[k]) == int(sdb[k]), k
        elif "running_" in k or k.startswith("vector_quantizer"):
            close_l2(sda[k], sdb[k], rtol=1e-5, what=k)
        else:
            close_l2(sda[k], sdb[k], rtol=2e-3, what=k)
    ops.set_compute_dtype(torch.bfloat16)
