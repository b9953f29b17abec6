"""GPU tier: the rest of train(): eval-mode forward + PSNR against the CPU oracle, StepLR,
checkpoints in the reference's layout, resume."""
import math
import os

import pytest
import torch

from conftest import close_l2
from vae_gan_amd.testing import fill_state_dict, synthetic_images

pytestmark = pytest.mark.gpu


@pytest.fixture()
def vq():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from vae_gan_amd import ops
    from vae_gan_amd.vq_vae_gan.config import Config
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
    ops.set_compute_dtype(torch.float32)
    cfg = Config()
    model = VQVAEGAN(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict()))
    yield model.cuda(), cfg
    ops.set_compute_dtype(torch.bfloat16)


def test_eval_forward_and_psnr_match_oracle(vq):
    """Eval mode: running-statistics BatchNorm, quantiser with training=False (no EMA update), PSNR as
    beta_vae_gans/utils.py:41-55."""
    from oracle import vaegan_oracle as O
    from vae_gan_amd import loop
    model, cfg = vq
    images = synthetic_images(2, 64, seed=5)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    want, _, _, _, idx_ref = O.vq_forward(sd, images, O.VQ_DEFAULT_CFG, training=False, train_bn=False)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        recon, _, _, _, idx = model(images.cuda(), training=False)
    close_l2(recon, want, rtol=1e-4, what="eval recon")
    assert (idx.cpu() == idx_ref).float().mean().item() >= 0.999
    for k, v in model.state_dict().items():               # nothing is updated in eval mode
        assert torch.equal(v, before[k]), k
    a, b = images.clamp(0, 1).double(), want.clamp(0, 1).double()
    ref_psnr = float(sum(10 * math.log10(1.0 / float(((a[i] - b[i]) ** 2).mean())) for i in range(2)) / 2)
    mine = loop.evaluate(model, [images], "cuda")
    assert abs(mine - ref_psnr) <= 1e-3, (mine, ref_psnr)
    assert abs(loop.vq_bpp_estimate(idx, 512, 64) - 64 * 9 / (64 * 64 * 3)) < 1e-12


def test_train_loop_checkpoints_and_resume(vq, tmp_path):
    """Two epochs of loop.train(): StepLR stepped per epoch, latest.pth in the reference's layout (loadable with
    weights_only=True), best_model.pth a bare state_dict; a fully resumed run continues exactly."""
    from vae_gan_amd import loop, train as T
    from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
    model, cfg = vq
    cfg.epochs, cfg.save_interval, cfg.output_dir = 2, 1, str(tmp_path)
    data = [synthetic_images(2, 64, seed=s) for s in (1, 2)]
    loop.train(model, data, [data[0]], cfg)
    latest = os.path.join(str(tmp_path), "checkpoints", "latest.pth")
    ck = torch.load(latest, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_G_state_dict", "optimizer_D_state_dict",
                       "scheduler_G_state_dict", "scheduler_D_state_dict", "best_psnr"}
    assert ck["epoch"] == 1 and ck["scheduler_G_state_dict"]["last_epoch"] == 2
    assert set(ck["optimizer_G_state_dict"]) == {"state", "param_groups"}
    assert set(ck["optimizer_G_state_dict"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    best = torch.load(os.path.join(str(tmp_path), "checkpoints", "best_model.pth"), map_location="cpu", weights_only=True)
    assert set(best) == set(model.state_dict())
    # resume into a fresh model + optimizers and take one more step on both
    m2 = VQVAEGAN(cfg).cuda().train()
    og2, od2, sg2, sd2 = T.build_optimizers(m2, cfg)
    epoch, best_psnr = loop.load_checkpoint(latest, m2, og2, od2, sg2, sd2, full=True)
    assert epoch == 1 and best_psnr == ck["best_psnr"] and og2.step_count == 4
    m1 = VQVAEGAN(cfg).cuda().train()
    og1, od1, sg1, sd1 = T.build_optimizers(m1, cfg)
    loop.load_checkpoint(latest, m1, og1, od1, sg1, sd1, full=True)
    x = data[0].cuda()
    r1 = T.vq_train_step(m1, x, og1, od1)
    r2 = T.vq_train_step(m2, x, og2, od2)
    assert abs(float(r1[1]["total"].detach()) - float(r2[1]["total"].detach())) <= 1e-5 * abs(float(r1[1]["total"].detach()))
    # reference --resume semantics: model only
    m3 = VQVAEGAN(cfg).cuda()
    assert loop.load_checkpoint(latest, m3) == (1, ck["best_psnr"])
    for k, v in m3.state_dict().items():
        assert torch.equal(v.cpu(), ck["model_state_dict"][k]), k
